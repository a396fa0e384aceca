/*
 * include/spmv_host_c.h -- C facade over the C++17 host mirror
 * (spmv_amd/csrc/host: spmv::HipExecutor, L2GMap, Matrix<double>, cg).
 *
 * The C++ classes are what a LIBSPMV user programs against; this facade only
 * exists so that non-C++ harnesses (this repo's pytest suite and bench.py,
 * via ctypes) can drive exactly those classes.  One function per C++ call,
 * named after it; the parity tests therefore read like the reference's
 * tests/test_spmv.cpp: create executor -> create_matrix -> alloc x,y ->
 * col_map()->update(x) -> mult(x, y).
 *
 * Every function returns 0 on success and -1 after catching a C++ exception;
 * spmvh_last_error() then returns its what().  Device pointers are plain
 * void* / double*.
 */
#ifndef SPMV_HOST_C_H
#define SPMV_HOST_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spmvh_exec spmvh_exec;     /* shared_ptr<spmv::HipExecutor>     */
typedef struct spmvh_comm spmvh_comm;     /* shared_ptr<const spmv::Comm>      */
typedef struct spmvh_matrix spmvh_matrix; /* spmv::Matrix<double>*             */

const char* spmvh_last_error(void);

/* spmv::CommunicationModel, same values as spmv/mpi_utils.h:43-52 */
enum {
  SPMVH_P2P_BLOCKING = 0,
  SPMVH_P2P_NONBLOCKING = 1,
  SPMVH_COLLECTIVE_BLOCKING = 2,
  SPMVH_COLLECTIVE_NONBLOCKING = 3,
  SPMVH_ONESIDED_PUT_ACTIVE = 4,
  SPMVH_ONESIDED_PUT_PASSIVE = 5,
  SPMVH_SHMEM = 6,
  SPMVH_SHMEM_NODUP = 7
};

/* ---- executor: HipExecutor::create(device_id, HostExecutor::create()) ---- */
int spmvh_exec_create(int device_id, spmvh_exec** exec);
int spmvh_exec_destroy(spmvh_exec* exec);
int spmvh_exec_alloc(spmvh_exec* exec, size_t num_bytes, void** ptr);
int spmvh_exec_free(spmvh_exec* exec, void* ptr);
int spmvh_exec_memset(spmvh_exec* exec, void* ptr, int value, size_t num_bytes);
int spmvh_exec_copy(spmvh_exec* exec, void* dst, const void* src,
                    size_t num_bytes); /* device -> device */
int spmvh_exec_copy_from_host(spmvh_exec* exec, void* dst, const void* host_src,
                              size_t num_bytes);
int spmvh_exec_copy_to_host(spmvh_exec* exec, void* host_dst, const void* src,
                            size_t num_bytes);
int spmvh_exec_synchronize(spmvh_exec* exec);
int spmvh_exec_num_cus(spmvh_exec* exec, int* num_cus);
int spmvh_exec_device_type(spmvh_exec* exec, int* type); /* 1 cpu, 2 gpu */
/* the spmv_hip_ctx* behind the executor (for harness-side timing events) */
int spmvh_exec_context(spmvh_exec* exec, void** ctx);
/* HostExecutor has no compute path: creating a CSRMatrix on it must throw.
 * Returns 0 if it did (message in spmvh_last_error), 1 if it did not. */
int spmvh_host_executor_rejects_compute(void);

/* ---- communicators ---------------------------------------------------------- */
int spmvh_comm_self(spmvh_comm** comm);
int spmvh_rccl_unique_id(void* id_bytes /* 128 bytes */);
int spmvh_comm_rccl(spmvh_exec* exec, int nranks, int rank, const void* id_bytes,
                    spmvh_comm** comm);
/* transport supplied by the caller; allgather is mandatory (host memory,
 * recv = nranks * bytes_per_rank), must return 0 */
typedef int (*spmvh_allgather_fn)(void* user, const void* send, void* recv,
                                  size_t bytes_per_rank);
/* optional device transport (may be NULL: device calls then fail).  Semantics
 * of spmv::Comm::neighbor_exchange / allreduce_sum: DEVICE pointers, offsets
 * in elements, everything ordered on `stream`. */
typedef int (*spmvh_exchange_fn)(void* user, size_t elem_bytes,
                                 int num_neighbours, const int* neighbours,
                                 const void* send_buf,
                                 const int32_t* send_counts,
                                 const int32_t* send_offsets, void* recv_base,
                                 const int32_t* recv_counts,
                                 const int32_t* recv_offsets, void* stream);
typedef int (*spmvh_allreduce_fn)(void* user, double* device_inout,
                                  size_t count, void* stream);
int spmvh_comm_callback(int rank, int nranks, spmvh_allgather_fn allgather,
                        spmvh_exchange_fn exchange, spmvh_allreduce_fn allreduce,
                        void* user, spmvh_comm** comm);
/* RcclComm::info(): out = {nranks, rank (both as RCCL reports them), RCCL
 * version code, 1 if the reductions have a communicator of their own} */
int spmvh_comm_rccl_info(spmvh_comm* comm, int out[4], char* lib_path,
                         int lib_path_len);
int spmvh_comm_destroy(spmvh_comm* comm);
/* the deterministic peer reduction of the CG scalars (Comm::enable_peer_reduce:
 * collective; *ok = 0: some rank cannot reach some window, the transport's
 * all-reduce stays) and one reduction through it (Comm::reduce_sum) */
int spmvh_comm_enable_peer_reduce(spmvh_comm* comm, spmvh_exec* exec, int* ok);
/* Comm::ranks_share_a_process (collective on its first call): do two ranks of
 * the communicator live in ONE process (ranks as threads)?  Decided by a
 * per-process token (pid + a 64-bit nonce), not the pid alone */
int spmvh_comm_ranks_share_a_process(spmvh_comm* comm, int* shared);
int spmvh_comm_reduce_sum(spmvh_comm* comm, double* device_inout, int count,
                          void* stream);

/* ---- matrix: Matrix<double>::create_matrix / create_poisson3d --------------- */
/* rowptr has nrows_local + num_row_ghosts + 1 entries: the extra rows hold
 * contributions to the global rows `row_ghosts` owned by other ranks and are
 * shipped to them (Matrix.cpp:188-292). */
int spmvh_matrix_create(spmvh_comm* comm, spmvh_exec* exec,
                        const int32_t* rowptr, const int32_t* colind,
                        const double* values, int64_t nrows_local,
                        int64_t ncols_local, const int64_t* row_ghosts,
                        int64_t num_row_ghosts, const int64_t* col_ghosts,
                        int64_t num_col_ghosts, int symmetric, int cm,
                        spmvh_matrix** A);
int spmvh_matrix_create_poisson3d(spmvh_comm* comm, spmvh_exec* exec, int32_t n,
                                  int symmetric, int cm, spmvh_matrix** A);
/* seeded unstructured test matrix generated on the device (one rank, general
 * storage; spmv_hip_unstructured_fill_f64) */
int spmvh_matrix_create_unstructured(spmvh_comm* comm, spmvh_exec* exec,
                                     int64_t nrows, int per_row, int64_t band,
                                     int far_permille, uint64_t seed,
                                     spmvh_matrix** A);
/* seeded FEM-like test matrix generated on the device (one rank, general
 * storage; spmv_hip_fem_count / spmv_hip_fem_fill_f64) */
struct spmv_hip_fem_params; /* include/spmv_hip.h */
int spmvh_matrix_create_fem_like(spmvh_comm* comm, spmvh_exec* exec,
                                 const struct spmv_hip_fem_params* params,
                                 spmvh_matrix** A);
/* ... its strictly lower part + diagonal in symmetric storage
 * (spmv_hip_csr_lower_split_*) */
int spmvh_matrix_create_fem_like_sym(spmvh_comm* comm, spmvh_exec* exec,
                                     const struct spmv_hip_fem_params* params,
                                     spmvh_matrix** A);
/* The Poisson matrix on a 3-D block partition (the reference
 * partitions by row slabs only, read_petsc.cpp:20-37): px * py * pz boxes,
 * rank = ix + px (iy + py iz), rank-major global numbering (a box's points are
 * consecutive, x fastest), so each rank still owns one contiguous row range.
 * px * py * pz must equal the communicator's size. */
int spmvh_matrix_create_poisson3d_boxes(spmvh_comm* comm, spmvh_exec* exec,
                                        int32_t n, int px, int py, int pz,
                                        int symmetric, int cm, spmvh_matrix** A);
/* Its host half (no device, no communicator): rank `rank`'s rows in
 * spmvh_matrix_create's input form.  sizes = {rows, non-zeros, ghosts, global
 * row offset, box extents x y z}; the arrays are filled when non-NULL (call
 * once with NULLs for the sizes). */
int spmvh_poisson3d_box_rows(int32_t n, int px, int py, int pz, int rank,
                             int64_t sizes[7], int32_t* rowptr, int32_t* colind,
                             double* values, int64_t* col_ghosts);
int spmvh_matrix_destroy(spmvh_matrix* A);
int spmvh_matrix_rows(spmvh_matrix* A, int* rows);
int spmvh_matrix_cols(spmvh_matrix* A, int* cols);
int spmvh_matrix_non_zeros(spmvh_matrix* A, int64_t* nnz);
int spmvh_matrix_format_size(spmvh_matrix* A, size_t* bytes);
int spmvh_matrix_symmetric(spmvh_matrix* A, int* symmetric);
/* local / remote block sizes: out[0..5] = rows, cols, nnz of local then remote
 * (remote all zero when the matrix has a single block) */
int spmvh_matrix_blocks(spmvh_matrix* A, int64_t out[6]);
/* spmv_hip_csr_plan_get / _set on the local (remote = 0) or remote block's
 * plan: which form it took, what it cost, launch-shape knobs */
int spmvh_matrix_plan_get(spmvh_matrix* A, int remote, const char* key,
                          int* value);
/* Matrix::enable_mixed / use_mixed: fp32 copies of the blocks'
 * values (ok = 0: symmetric storage, nothing done); mult / mult_dot then stream
 * those while `on`. */
int spmvh_matrix_enable_mixed(spmvh_matrix* A, int* ok);
int spmvh_matrix_use_mixed(spmvh_matrix* A, int on);
/* CSRMatrix::release_csr on both blocks: the device copies of colind / values
 * of a block whose plan holds the matrix in its own format are freed
 * (spmv_hip_csr_plan_owns_matrix); *bytes_freed = what came back (0: nothing) */
int spmvh_matrix_release_csr(spmvh_matrix* A, int64_t* bytes_freed);
int spmvh_matrix_plan_set(spmvh_matrix* A, int remote, const char* key,
                          int value);
/* A.col_map()->update(x) ; A.mult(x, y) ; A.col_map()->update_finalise(x) */
int spmvh_matrix_update(spmvh_matrix* A, double* x);
int spmvh_matrix_update_finalise(spmvh_matrix* A, double* x);
int spmvh_matrix_mult(spmvh_matrix* A, double* x, double* y);

/* ---- fp32 instantiation: Matrix<float> (device_executor.h:88-99 carries float
 * visitors).  Same calls, float data. */
typedef struct spmvh_matrix_f32 spmvh_matrix_f32;
int spmvh_matrix_f32_create(spmvh_comm* comm, spmvh_exec* exec,
                            const int32_t* rowptr, const int32_t* colind,
                            const float* values, int64_t nrows_local,
                            int64_t ncols_local, const int64_t* row_ghosts,
                            int64_t num_row_ghosts, const int64_t* col_ghosts,
                            int64_t num_col_ghosts, int symmetric, int cm,
                            spmvh_matrix_f32** A);
int spmvh_matrix_f32_destroy(spmvh_matrix_f32* A);
int spmvh_matrix_f32_info(spmvh_matrix_f32* A, int* rows, int64_t* nnz,
                          int32_t* local_size, int32_t* num_ghosts);
int spmvh_matrix_f32_update(spmvh_matrix_f32* A, float* x);
int spmvh_matrix_f32_mult(spmvh_matrix_f32* A, float* x, float* y);

/* Host half of create_matrix only (Matrix<double>::split_rows): no device.
 * sizes[0..7] = local rows, cols, nnz, remote rows, cols, nnz, number of
 * (renumbered) ghost columns, nnz_full. */
typedef struct spmvh_split spmvh_split;
int spmvh_split_create(const int32_t* rowptr, const int32_t* colind,
                       const double* values, int64_t nrows_local,
                       int64_t ncols_local, int64_t global_row_offset,
                       int64_t global_col_offset, const int64_t* col_ghosts,
                       int64_t num_col_ghosts, int symmetric, int cm,
                       spmvh_split** split, int64_t sizes[8]);
/* which: 0 local, 1 remote.  Arrays sized from `sizes`; diagonal/ghosts may
 * be NULL. */
/* collective variant with ghost-row elimination (split_rows_distributed) */
int spmvh_split_create_dist(spmvh_comm* comm, const int32_t* rowptr,
                            const int32_t* colind, const double* values,
                            int64_t nrows_local, int64_t ncols_local,
                            const int64_t* row_ghosts, int64_t num_row_ghosts,
                            const int64_t* col_ghosts, int64_t num_col_ghosts,
                            int symmetric, int cm, spmvh_split** split,
                            int64_t sizes[8]);
int spmvh_split_get(spmvh_split* split, int which, int32_t* rowptr,
                    int32_t* colind, double* values);
int spmvh_split_extra(spmvh_split* split, double* diagonal, int64_t* ghosts);
int spmvh_split_destroy(spmvh_split* split);

/* ---- column map (L2GMap) inspection ------------------------------------------ */
int spmvh_l2g_sizes(spmvh_matrix* A, int32_t* local_size, int32_t* num_ghosts,
                    int64_t* global_size, int64_t* global_offset,
                    int* overlapping, int* num_neighbours, int* num_indices,
                    int* packs);
/* 1 = the halo of A.col_map() moves by peer stores into the neighbours'
 * windows (onesided_put_* models on several ranks; include/spmv_hip.h) */
int spmvh_l2g_onesided(spmvh_matrix* A, int* onesided);
int spmvh_l2g_ghosts(spmvh_matrix* A, int64_t* ghosts);
/* plan arrays: neighbours[nn], send_count[nn], recv_count[nn],
 * send_offset[nn+1], recv_offset[nn+1], indexbuf[num_indices] */
int spmvh_l2g_plan(spmvh_matrix* A, int32_t* neighbours, int32_t* send_count,
                   int32_t* recv_count, int32_t* send_offset,
                   int32_t* recv_offset, int32_t* indexbuf);
int spmvh_l2g_global_to_local(spmvh_matrix* A, int64_t global, int32_t* local);

/* Stand-alone L2GMap (no matrix), for plan tests:
 * L2GMap(comm, local_size, ghosts, exec-or-host, cm).  use_host_exec != 0
 * builds it on a HostExecutor (plan only; update() would throw). */
typedef struct spmvh_l2g spmvh_l2g;
int spmvh_l2g_create(spmvh_comm* comm, spmvh_exec* exec, int use_host_exec,
                     int64_t local_size, const int64_t* ghosts,
                     int64_t num_ghosts, int cm, spmvh_l2g** map);
int spmvh_l2g_destroy(spmvh_l2g* map);
int spmvh_l2g_map_sizes(spmvh_l2g* map, int* num_neighbours, int* num_indices,
                        int* packs);
int spmvh_l2g_map_plan(spmvh_l2g* map, int32_t* neighbours, int32_t* send_count,
                       int32_t* recv_count, int32_t* send_offset,
                       int32_t* recv_offset, int32_t* indexbuf);
int spmvh_l2g_map_update(spmvh_l2g* map, double* x);
/* L2GMap::reverse_update(vec) (L2GMap.h:103): ghost tail -> owners, added */
int spmvh_l2g_map_reverse_update(spmvh_l2g* map, double* x);
int spmvh_l2g_map_reverse_update_f32(spmvh_l2g* map, float* x);

/* ---- cg: spmv::cg(comm, exec, A, b, x, kmax, rtol) ---------------------------- */
/* rnorm_history (host, kmax+1 doubles) may be NULL; *num_its = returned k */
int spmvh_cg(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
             const double* b, double* x, int kmax, double rtol, int* num_its,
             double* rnorm_history);

/* ---- PETSc binary ingest (spmv/read_petsc.{h,cpp}) ------------------------------
 * read_petsc_binary_matrix / read_petsc_binary_vector of the mirror.  The
 * vector comes back as a device pointer the caller frees with
 * spmvh_exec_free. */
int spmvh_read_petsc_matrix(spmvh_comm* comm, spmvh_exec* exec,
                            const char* filename, int symmetric, int cm,
                            spmvh_matrix** A);
int spmvh_read_petsc_vector(spmvh_comm* comm, spmvh_exec* exec,
                            const char* filename, double** device_vec,
                            int64_t* nrows_local);
/* host-only parse of rank `rank` of `size` (no device): sizes[0..6] = global
 * rows, cols, nnz, row_begin, row_end, local nnz, number of ghost columns */
typedef struct spmvh_petsc_rows spmvh_petsc_rows;
int spmvh_petsc_rows_read(const char* filename, int rank, int size,
                          spmvh_petsc_rows** rows, int64_t sizes[7]);
int spmvh_petsc_rows_get(spmvh_petsc_rows* rows, int32_t* rowptr,
                         int32_t* colind, double* values, int64_t* col_ghosts);
int spmvh_petsc_rows_destroy(spmvh_petsc_rows* rows);

/* Same solve with the optional arguments of the C++ overload: a reusable
 * spmv::CgWorkspace (may be NULL) and per-iteration HIP-event timing of the
 * local-block SpMV kernel (time_spmv bit 0 -> *spmv_ms_total, *spmv_launches;
 * bit 2 switches CgOptions::consumer_reductions off). */
typedef struct spmvh_cg_workspace spmvh_cg_workspace;
/* cg with CgOptions::mixed (fp32 copy of the matrix values in the SpMV,
 * residual replacement every `replace_every` iterations, fp64 correction
 * solve if the true residual misses rtol).  out_stats = {spmv_ms_total,
 * spmv_launches, replacements, true ||b-Ax||/||r0|| after the mixed loop,
 * fp64 iterations of the correction solve, true relative residual at the
 * end}.  rnorm_history holds up to history_capacity entries. */
int spmvh_cg_mixed(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
                   const double* b, double* x, int kmax, double rtol,
                   int replace_every, int* num_its, double* rnorm_history,
                   int history_capacity, spmvh_cg_workspace* ws, int time_spmv,
                   double out_stats[6]);
int spmvh_cg_workspace_create(spmvh_exec* exec, spmvh_cg_workspace** ws);
int spmvh_cg_workspace_destroy(spmvh_cg_workspace* ws);
/* create the timing events of a time_spmv solve of up to `iterations` steps
 * ahead of it (keeps them out of a benchmark's timed region) */
int spmvh_cg_workspace_reserve_timing(spmvh_cg_workspace* ws, int iterations);
int spmvh_cg_ex(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
                const double* b, double* x, int kmax, double rtol, int* num_its,
                double* rnorm_history, spmvh_cg_workspace* ws, int time_spmv,
                double* spmv_ms_total, int* spmv_launches);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HOST_C_H */
