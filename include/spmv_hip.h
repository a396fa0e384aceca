/*
 * include/spmv_hip.h -- C ABI of libspmv_hip.so, the MI355X (gfx950) backend
 * for the LIBSPMV hot path: fp64/fp32 CSR and symmetric-CSR SpMV, the ghost
 * pack kernel, the CG loop's fused BLAS-1 kernels, device memory/stream
 * plumbing and the RCCL halo / all-reduce transport.
 *
 * This is the drop-in boundary.  Each entry point names the reference
 * interface (file:line relative to the LIBSPMV tree) it stands behind; a
 * `spmv::HipExecutor` (spmv_amd/csrc/host/hip_executor.h) binds exactly these
 * symbols, and INTEGRATION.md shows the same binding inside the reference.
 *
 * Conventions
 *   - Every function returns int: 0 = success, >0 = hipError_t,
 *     >=10000 = 10000 + ncclResult_t, <0 = SPMV_HIP_E*.  Nothing throws,
 *     nothing calls exit().  spmv_hip_error_string() decodes a code.
 *   - All `const T*` / `T*` data arguments are DEVICE pointers unless the
 *     parameter name starts with `host_`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the context's current
 *     stream, see spmv_hip_set_stream).  Nothing synchronises the host unless
 *     its name says so.
 *   - One context per GPU; a context is used by one host thread at a time
 *     (same rule as the reference's executors).
 */
#ifndef SPMV_HIP_H
#define SPMV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_HIP_ABI_VERSION 5

enum {
  SPMV_HIP_OK = 0,
  SPMV_HIP_EINVAL = -1,   /* bad argument (null handle, negative size, ...) */
  SPMV_HIP_ENOMEM = -2,   /* host allocation failed                         */
  SPMV_HIP_ENOTSUP = -3,  /* feature not available in this build           */
  SPMV_HIP_ERANGE = -4,   /* size exceeds a 32-bit index the format fixes  */
  SPMV_HIP_EPEER = -5     /* one-sided halo: a neighbour did not answer in
                             time (an earlier exchange of this window failed) */
};

typedef struct spmv_hip_ctx spmv_hip_ctx;           /* one GPU             */
typedef struct spmv_hip_csr_plan spmv_hip_csr_plan; /* CSRSpMV::_aux_data  */
typedef struct spmv_hip_cg_ws spmv_hip_cg_ws;       /* cg() work vectors   */
typedef struct spmv_hip_comm spmv_hip_comm;         /* RCCL communicator   */

int spmv_hip_abi_version(void);
const char* spmv_hip_error_string(int code);

/* ---- context / device queries ------------------------------------------
 * HipExecutor ctor/dtor, get_num_devices, get_num_cus, synchronize
 * (device_executor.h:82-85; cuda/cuda_executor.cpp:13-47). */
int spmv_hip_device_count(int* count);
int spmv_hip_ctx_create(int device_id, spmv_hip_ctx** ctx);
int spmv_hip_ctx_destroy(spmv_hip_ctx* ctx);
int spmv_hip_ctx_device(const spmv_hip_ctx* ctx, int* device_id);
int spmv_hip_num_cus(const spmv_hip_ctx* ctx, int* num_cus);
int spmv_hip_synchronize(spmv_hip_ctx* ctx); /* whole device */
/* tuning options of a context; EINVAL for an unknown key.
 *   "blas1_nt_min_elems": the CG vector kernels stream vectors of at least
 *   this many elements past the caches (non-temporal loads and stores);
 *   shorter vectors stay cached between kernels.  Default 2^24.
 *   "lat_min_nnz", "lx_min_nnz", "lx_max_x_bytes": from how many entries (up to
 *   how large an x) plans build the lattice / LX forms.
 *   "bake_general": 0 = plan_bake_values on a general plan always returns
 *   SPMV_HIP_ENOTSUP (the CSR-order kernels on the caller's values).
 *   "const_diagonals": 1 (default) = plan_bake_values looks whether every
 *   diagonal of a lattice matrix is constant, bit for bit (constant-coefficient
 *   stencils: the Poisson operators); then the plan keeps ONE number per
 *   diagonal and the per-row presence mask instead of a copy of the values, and
 *   the kernel multiplies by that number -- same products, same sums, same
 *   order, same bits.  0 = always keep (and stream) the values.
 *   "const_tile": lattice lines per lane of that kernel on 3-D lattices (1, 2
 *   or 4; default 4).
 *   "wdia_half": 0 = the wide diagonal form (a baked copy of a general matrix
 *   with <= 32 diagonals) keeps every diagonal even when it finds the matrix
 *   symmetric bit for bit; default 1 = then only the diagonals <= 0.
 *   "poisson_skew_ppm": the device generator below writes a NON-symmetric
 *   variant (lower neighbours -1 - s, upper -1 + s, s = value * 1e-6).
 *   "poisson_stencil": 7 (default) or 27 -- the generator writes the 27-point
 *   operator (all neighbours with |dx|,|dy|,|dz| <= 1; diagonal 26,
 *   off-diagonal -1); its row slabs must be whole planes.
 *   "csr_in_place": 1 = plans make NO copy of the index or value stream of a
 *   matrix without lattice structure -- neither the LX form's 16-bit offsets
 *   (2 B per entry) nor the sliced jagged arrays (10 B per entry): the caller's
 *   CSR arrays are streamed as they are, x windows staged in LDS where the
 *   columns of a row block form <= 8 windows (the XW kernel, 144 B of plan
 *   memory per 256 rows), gathered otherwise.  Default 0.
 *   "xw_min_nnz", "xw_min_x_bytes": from how many entries / how large an x on
 *   such a plan stages the windows; "xw_probe": 1 (default) = its first four
 *   launches -- each a complete product with the same bits -- time the XW
 *   kernel against the gather kernel and the plan keeps the faster one
 *   (csr_plan_get "xw_pick"); 0 = XW wherever its records exist. */
int spmv_hip_ctx_set_option(spmv_hip_ctx* ctx, const char* key, int64_t value);
/* ... and read back (release_csr, put_timeout_ms, the *_min_nnz thresholds,
 * xw_min_x_bytes); SPMV_HIP_EINVAL for a key without a getter */
int spmv_hip_ctx_get_option(const spmv_hip_ctx* ctx, const char* key, int64_t* value);
/*   "lx_min_nnz": csr_plan_create builds the LX form of a general matrix
 *   (LDS-staged x windows + 16-bit column offsets, 2 B per entry of extra
 *   device memory) from this many entries on.  Default 2^20; a huge value
 *   switches the form off.
 *   "lx_max_x_bytes": ... and only for matrices whose input vector
 *   (num_cols * 8 bytes) is at most this large.  Default 128 MiB (the form
 *   pays while x stays in the Infinity Cache). */

/* ---- streams / events ---------------------------------------------------
 * CudaExecutor::set/reset/get_cuda_stream (cuda/cuda_executor.h:72-76). */
int spmv_hip_stream_create(spmv_hip_ctx* ctx, void** stream);
/* high_priority != 0: the stream's kernels are dispatched ahead of other
 * ready work -- used for the halo stream so the small RCCL send/recv kernel is
 * placed before the CU-filling local SpMV that becomes ready at the same time */
int spmv_hip_stream_create_priority(spmv_hip_ctx* ctx, int high_priority,
                                    void** stream);
int spmv_hip_stream_destroy(spmv_hip_ctx* ctx, void* stream);
int spmv_hip_stream_synchronize(spmv_hip_ctx* ctx, void* stream);
int spmv_hip_set_stream(spmv_hip_ctx* ctx, void* stream); /* NULL = reset */
int spmv_hip_get_stream(const spmv_hip_ctx* ctx, void** stream);
int spmv_hip_event_create(spmv_hip_ctx* ctx, int timing, void** event);
int spmv_hip_event_destroy(spmv_hip_ctx* ctx, void* event);
int spmv_hip_event_record(spmv_hip_ctx* ctx, void* event, void* stream);
int spmv_hip_event_synchronize(spmv_hip_ctx* ctx, void* event);
int spmv_hip_stream_wait_event(spmv_hip_ctx* ctx, void* stream, void* event);
int spmv_hip_event_elapsed_ms(spmv_hip_ctx* ctx, void* start, void* stop,
                              float* ms);

/* ---- memory ---------------------------------------------------------------
 * DeviceExecutor::_alloc/_free/_memset/_copy/_copy_async/_copy_from/_copy_to
 * (device_executor.h:129-139). */
int spmv_hip_alloc(spmv_hip_ctx* ctx, size_t num_bytes, void** ptr);
int spmv_hip_free(spmv_hip_ctx* ctx, void* ptr);
int spmv_hip_host_alloc(spmv_hip_ctx* ctx, size_t num_bytes, void** host_ptr);
int spmv_hip_host_free(spmv_hip_ctx* ctx, void* host_ptr);
int spmv_hip_memset_async(spmv_hip_ctx* ctx, void* ptr, int value,
                          size_t num_bytes, void* stream);
int spmv_hip_copy_d2d_async(spmv_hip_ctx* ctx, void* dst, const void* src,
                            size_t num_bytes, void* stream);
int spmv_hip_copy_h2d_async(spmv_hip_ctx* ctx, void* dst,
                            const void* host_src, size_t num_bytes,
                            void* stream);
int spmv_hip_copy_d2h_async(spmv_hip_ctx* ctx, void* host_dst, const void* src,
                            size_t num_bytes, void* stream);
/* peer copy between two contexts (cuda/cuda_executor.cpp:82-94) */
int spmv_hip_copy_peer_async(spmv_hip_ctx* dst_ctx, void* dst,
                             spmv_hip_ctx* src_ctx, const void* src,
                             size_t num_bytes, void* stream);

/* ---- CSR SpMV -------------------------------------------------------------
 * CSRSpMV<T>::init / run / finalize (csr_kernels.h:26-78); reference
 * arithmetic csr_kernels.cpp:20-52.
 *
 * plan_create inspects the (device-resident) row pointer once and picks the
 * kernel and launch shape; it stores no copy of the matrix (plan_bake_values_*
 * below is the one, explicit exception).  `symmetric`
 * selects the strictly-lower + diagonal kernel.  rowptr/colind may be NULL
 * when num_non_zeros == 0 (csr_matrix.cpp:34).
 *
 * Plan creation also analyses the index arrays and, where the matrix allows,
 * bakes a compressed form of their CONTENT into the plan (all of them produce
 * the same bits as the plain kernels):
 *   lattice form   every block of 256 rows has its columns at row + one of
 *                  <= 8 constant offsets: no index stream at all, values by
 *                  LDS-DMA one row block ahead
 *   LX form        few contiguous column windows per row block: x staged in
 *                  LDS, 16-bit column offsets
 *   row list       mostly-empty blocks (SPMV_HIP_ALGO_ROWLIST)
 *   symmetric      the symmetric lattice form (<= 3 constant lower offsets) or
 *                  the transposed map (entries sorted by column): atomic-free
 * CONTRACT: a plan with such a form must be launched with the very rowptr /
 * colind pointers it was created with -- the kernels no longer read colind,
 * so other arrays of the same shape would silently compute with the old
 * structure; spmv returns SPMV_HIP_EINVAL instead.  `values`, `in`, `out` are
 * free to change from call to call.
 *
 * run computes out = alpha * A * in + beta * out.
 *   - beta == 0: out is write-only and never read.
 *   - general kernel, SPMV_HIP_ALGO_ROWBLOCK: each row is summed left to
 *     right in fp64 without FMA contraction, i.e. bit-identical to
 *     csr_kernels.cpp:41-51.
 *   - symmetric kernel (strictly lower block): the reference's sequential
 *     order (csr_kernels.cpp:26-40) seen from the row -- finalise
 *     fl(alpha*sum + beta*out), then add fl(fl(alpha*v)*in[r]) for the
 *     entries (r, i) of the row's column in ascending r -- with no atomics:
 *     bit-identical to the reference.  plan_set "sym_det" / "slat" = 0 (or a
 *     block that is not strictly lower) selects the older atomic kernels: out
 *     is scaled by beta, every term accumulated with hardware fp64 atomics,
 *     order of additions not deterministic.
 *   - `dot_partials` (optional, may be NULL; not for a diagonal-only
 *     symmetric block): fuses
 *     the CG dot product.  The kernel writes spmv_hip_dot_partials_len()
 *     doubles whose sum is sum_i in[i] * (alpha * (A in)_i), this block's own
 *     share (beta*out is not included, so the shares of a local and a remote
 *     block add up to in . (A in)); reduce with spmv_hip_reduce_partials_f64.
 *     The atomic symmetric kernels use the mirror identity: row i contributes
 *     in_i * alpha * (2 (d_i in_i + (L in)_i) - d_i in_i).
 */
enum {
  SPMV_HIP_ALGO_AUTO = 0,
  SPMV_HIP_ALGO_ROWBLOCK = 1, /* row blocks streamed through LDS, exact order */
  SPMV_HIP_ALGO_VECTOR = 2,   /* sub-wavefront per row, shuffle reduction     */
  SPMV_HIP_ALGO_SCALAR = 3,   /* one lane per row (short rows, reference)     */
  SPMV_HIP_ALGO_ROWLIST = 4   /* mostly-empty block: walk the compacted list of
                                 non-empty rows (built by plan_create), exact
                                 order; general kernel only                    */
};

int spmv_hip_csr_plan_create(spmv_hip_ctx* ctx, int32_t num_rows,
                             int32_t num_cols, int64_t num_non_zeros,
                             const int32_t* rowptr, const int32_t* colind,
                             int symmetric, int algo,
                             spmv_hip_csr_plan** plan);
int spmv_hip_csr_plan_destroy(spmv_hip_csr_plan* plan);
/* Optional: let the plan keep its OWN copy of the matrix values in the layout
 * its kernel streams best -- one array per lower offset plus the diagonal
 * ("symmetric diagonal form", a DIA fast path).  This is what
 * CSRSpMV::init does with the matrix it is given (csr_kernels.h:26-78; the
 * cuSPARSE descriptor of cuda/csr_kernels.cu binds the values there too).
 *   symmetric plan   in the symmetric lattice form (<= 3 constant lower
 *                    offsets); pass values and diagonal
 *   general plan     (diagonal = NULL) in the lattice form, square, with <= 3
 *                    distinct |col - row| > 0.  If the device check finds the
 *                    matrix SYMMETRIC entry for entry and bit for bit, the
 *                    plan keeps the lower half and the diagonal only (49
 *                    instead of 73 B per row of a 7-point matrix); if not,
 *                    ALL values by offset (the "full" form: the bytes of the
 *                    CSR values, no index stream, no row pointer).  Either
 *                    way the kernel sums each row in the general kernel's own
 *                    order: same bits as csr_kernels.cpp:41-51.
 *                    A general matrix on MORE diagonals (<= 32: 27-point
 *                    stencils) takes the "wide diagonal form": values by
 *                    offset + a 32-bit presence mask per row; only the
 *                    diagonals <= 0 when the matrix is symmetric bit for bit.
 *   constant diagonals  (either storage; ctx option "const_diagonals") when
 *                    every diagonal is constant, bit for bit, the plan keeps
 *                    no values at all: one number per diagonal + the mask.
 * plan_create and plan_bake_values wait for `stream` before they start (the
 * plan's clock counts the plan's work) and return after the plan's own kernels
 * have completed.
 * Anything else: SPMV_HIP_ENOTSUP, nothing changes.  Costs (offsets + 1) * 8 B
 * per row of device memory (constant diagonals: 1 or 4 B per row).  A launch that passes these very `values` (and
 * `diagonal`) pointers takes the diagonal-form kernel; a launch with other
 * pointers takes the CSR-order kernels as before.  CONTRACT: whoever rewrites
 * the baked arrays in place calls spmv_hip_csr_plan_values_changed (or bakes
 * again, or drops the copy: values = NULL).  A general plan that took neither
 * the lattice nor the LX form (ragged rows, more than 16 entries per row) keeps
 * its copy in the SLICED JAGGED order instead (spmv_sjds.hip): 64-row slices
 * as jagged diagonals + 16-bit column codes into an LDS-staged copy of x. */
int spmv_hip_csr_plan_bake_values_f64(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const double* values,
                                      const double* diagonal, void* stream);
int spmv_hip_csr_plan_bake_values_f32(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const float* values, const float* diagonal,
                                      void* stream);
/* The caller has REWRITTEN the baked `values` (and `diagonal`) arrays in place
 * -- a time step with new coefficients on the same sparsity, the common use of
 * a raw C ABI; the reference's CSRMatrix is immutable (csr_matrix.cpp:22-59)
 * and needs no such call.  Refreshes every copy the plan keeps, from the
 * pointers it was baked with, in `stream` order (the call returns after the
 * refresh has completed):
 *   sliced jagged form   the copy is rewritten in place, nothing is allocated;
 *   diagonal forms       the checks run again on the new values (constant
 *                        diagonals? still symmetric?) and the copy is rebuilt
 *                        in the form they allow; a matrix the forms no longer
 *                        hold falls back to the CSR-order kernels (no other
 *                        form is built inside this call: bake again for the
 *                        sliced jagged form);
 *   fp32 copies of the mixed SpMV likewise.
 * A plan without a baked copy: nothing to do.  SPMV_HIP_OK in all these cases
 * -- launches with the same pointers then return the NEW matrix's product.
 * Without this call they return the OLD one (the copy is the plan's own; the
 * sliced jagged form, which reads its long rows from the caller's arrays, a
 * mixture of the two): the contract of plan_bake_values.  plan_get "values_changed_us" = what the last
 * call cost. */
int spmv_hip_csr_plan_values_changed(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                     void* stream);
/* PLAN MEMORY (ABI 4).  A plan that took one of the value-baking forms holds a
 * complete copy of the matrix in its own format; the caller's `colind` and
 * `values` are then read by nothing but a fallback.  (The reference's CSRMatrix
 * owns exactly one copy of the matrix, spmv/csr_matrix.cpp:34-70; with a plan
 * on top the device would hold two.)
 *   plan_owns_matrix     *mask = the arrays the plan no longer needs to read:
 *                        bit 0 (1) = colind, bit 1 (2) = values.  Non-zero for a
 *                        general plan in a diagonal form (values by offset,
 *                        or no values at all where every diagonal is constant)
 *                        or in the sliced jagged form WITHOUT long rows (those
 *                        are streamed from the caller's arrays), and for
 *                        symmetric storage in the merged sliced jagged form
 *                        without long rows (its kernel reads the merged copy,
 *                        `rowptr` and `diagonal`); 0 otherwise (CSR-order
 *                        kernels, LX, XW, lattice kernels, the other symmetric
 *                        forms, an fp32 twin baked for the mixed SpMV).
 *                        `rowptr` is never given up (4 B per row).
 *   plan_release_matrix  the caller frees the arrays of `mask` (a subset of
 *                        what plan_owns_matrix reports, else SPMV_HIP_EINVAL).
 *                        From then on the plan COMPARES the `colind` / `values`
 *                        pointers of a launch with the ones it was built from
 *                        and never reads them, and refuses with SPMV_HIP_EINVAL
 *                        whatever would: plan_values_changed, plan_bake_values_*
 *                        (a drop included), plan_set of a key that selects
 *                        another kernel, a launch the baked form does not take
 *                        (other pointers, an x that is not 16-byte aligned).
 *                        A symmetric plan also drops what only those refused
 *                        paths read (transposed map, value positions: 16 B per
 *                        stored entry).  Irreversible for the plan's lifetime. */
int spmv_hip_csr_plan_owns_matrix(const spmv_hip_csr_plan* plan, int* mask);
int spmv_hip_csr_plan_release_matrix(spmv_hip_csr_plan* plan, int mask);
/* ... and the fp32 copy for spmv_hip_csr_spmv_f32f64 (mixed precision) on a
 * general fp64 plan whose values are baked: `values32` is the caller's fp32
 * copy of the CSR values (the same device check runs on its bits).  Launches
 * of spmv_f32f64 with this pointer then stream 17 instead of 33 B of matrix
 * data per row of a 7-point matrix.  values32 = NULL drops the copy. */
int spmv_hip_csr_plan_bake_values_f32f64(spmv_hip_ctx* ctx,
                                         spmv_hip_csr_plan* plan,
                                         const float* values32, void* stream);
int spmv_hip_csr_plan_algo(const spmv_hip_csr_plan* plan, int* algo);
/* The plane-walk order table the lattice kernels take (host arithmetic only, no
 * device: for inspection and tests).  Slot (round * L + step) * grid + w holds
 * the row block workgroup w computes at that step, -1 = none; planes are
 * `plane_rows` rows apart; segments = runs along the plane axis (0 = choose).
 * Call with table = NULL for the size. */
int spmv_hip_zwalk_table(int32_t num_rows, int64_t plane_rows, int grid,
                         int segments, int32_t* table, int64_t capacity,
                         int64_t* num_slots, int* segments_out);
/* Knobs (key/value; EINVAL for an unknown key or a value out of range):
 *   "algo" "lanes_per_row" "chunks" "nontemporal" "xcd_group" "blocks_per_cu"
 *   "nt_store"                           the plain general kernels
 *   "lx" "lx_chunks"                     LX form on/off (built plans only)
 *   "lat" "lat_blocks_per_cu" "lat_xcd_group" "lat_chain"   lattice form
 *                  (lat_chain: x handed from plane to plane in registers)
 *   "slat" "slat_blocks_per_cu"          symmetric lattice form
 *   "sdia" "sdia_chain" "sdia_nt"        symmetric diagonal form (baked plans):
 *                  on/off, plane chain on/off, non-temporal streams (bit mask)
 *   "sdia_tile" "sdia_tile_segments" "sdia_tile_blocks_per_cu"   constant
 *                  diagonals on a 3-D lattice: lattice lines per lane (1, 2,
 *                  4), its own plane-walk table, workgroups per CU
 *   "lxw" "lxw_blocks_per_cu"            LX form: the LDS-DMA kernel on/off
 *   "xw" "xw_probe"                      the same kernel on the caller's CSR
 *                  arrays (plans with XW records): on/off -- 1 asks for it by
 *                  name and ends the probe --; xw_probe 1 = let the next four
 *                  launches choose between it and the gather kernel again
 *   "wdia" "wdia_xcd_group" "wdia_blocks_per_cu" "wdia_zwalk"
 *   "wdia_zwalk_segments"                wide diagonal form (baked plans)
 *   "wdia_box" "wdia_box_segments" "wdia_box_blocks_per_cu"   constant
 *                  27-point box stencils: lattice lines per lane (0 = the
 *                  general kernel, 2, 4), its plane-walk table, workgroups/CU
 *   "zwalk" "zwalk_segments"             plane-walk row-block order of the
 *                  three lattice kernels (every workgroup walks a 256-row
 *                  column of the lattice from plane to plane): use the table;
 *                  (re)build it with that many runs along the plane axis (0 =
 *                  choose), whatever the size of the matrix
 *   "sym_det"                            transposed-map kernel (0: atomics)
 *   "sym_window" "sym_rows"              the atomic symmetric kernels */
int spmv_hip_csr_plan_set(spmv_hip_csr_plan* plan, const char* key, int value);
/* What the plan decided and what it cost: "algo"; "lat", "lx", "slat", "sdia",
 * "sym_det" (1 = that form is in use), "lat_blocks", "lx_blocks", "lx_staged";
 * "lattice_d1", "lattice_d2" (row distance of the next grid line / plane when
 * the matrix is a 3-D lattice, else 0), "zwalk", "zwalk_segments",
 * "zwalk_grid", "lat_chain", "sdia_chain", "sdia_nt", "sdia_offsets" (lower
 * offsets of the baked copy), "sdia_general" (baked from a general matrix: 1 = symmetric, half stored; 2 =
 * full form), "sdia_mixed" (fp32 copy), "sdia_const" (constant diagonals: no
 * values kept), "sdia_tile" (lines per lane in use), "sdia_tile_walk"; "lxw";
 * "xw", "xw_staged" (row blocks with staged windows), "xw_pick" (1 = the XW
 * kernel runs, 0 = the plan's probe found the gather kernel faster, -1 = the
 * probe's four launches are not all complete), "xw_probe_xw_us",
 * "xw_probe_gather_us" (what it measured);
 * "wdia", "wdia_offsets", "wdia_half", "wdia_const", "wdia_box", "wdia_mixed", "wdia_d2",
 * "wdia_zwalk", "wdia_zwalk_segments";
 * "blocks_per_cu", "nontemporal"; "plan_us" (wall time of plan creation, its
 * analysis kernels included), "plan_mem_us" (the part of it -- and of later
 * bakes -- spent inside hipMalloc / hipFree) and "plan_kib" (device memory the
 * plan owns). */
int spmv_hip_csr_plan_get(const spmv_hip_csr_plan* plan, const char* key,
                          int* value);

/* out = alpha * A * in + beta * out  (CSRSpMV<T>::run, csr_kernels.cpp:20-52),
 * every row summed left to right, mul and add rounded separately: bit-identical
 * to the reference loops on finite data, with ONE defined difference:
 *   beta == 0 means `out` is WRITE-ONLY -- the kernels store alpha*sum and never
 *   read out[i] (the reference computes alpha*sum + 0*out[i] on a
 *   buffer cg.cpp:40 never initialises, so a NaN/Inf there would poison it).
 *   Consequence for the sign of zero: where alpha*sum is -0.0 the reference's
 *   "+ 0*out[i]" turns it into +0.0 (for out[i] >= +0) while this library
 *   returns -0.0.  The two compare equal (np.array_equal, ==); only the sign
 *   bit differs, and only for rows whose sum is an exact negative zero.
 * dot_partials (may be NULL; general fp64 and symmetric fp64): the launch also
 * leaves spmv_hip_dot_partials_len() partial sums of in . (alpha * A * in),
 * the block's share of p.Ap in CG (cg.cpp:63). */
int spmv_hip_csr_spmv_f64(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                          int32_t num_rows, int32_t num_cols,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          const int32_t* colind, const double* values,
                          const double* diagonal, double alpha,
                          const double* in, double beta, double* out,
                          double* dot_partials, void* stream);
/* Mixed precision (device_executor.h:88-99 carries the float
 * visitors): `values` in fp32 -- half the matrix bytes -- x, y and every
 * product and sum in fp64.  General (non-symmetric) blocks; the same plan
 * serves both value types.  With fp32-representable values (the Poisson
 * matrix) the result equals spmv_hip_csr_spmv_f64 bit for bit. */
int spmv_hip_csr_spmv_f32f64(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                             int32_t num_rows, int32_t num_cols,
                             int64_t num_non_zeros, const int32_t* rowptr,
                             const int32_t* colind, const float* values,
                             double alpha, const double* in, double beta,
                             double* out, double* dot_partials, void* stream);
int spmv_hip_csr_spmv_f32(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                          int32_t num_rows, int32_t num_cols,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          const int32_t* colind, const float* values,
                          const float* diagonal, float alpha, const float* in,
                          float beta, float* out, void* stream);

/* ---- ghost pack -------------------------------------------------------------
 * DeviceExecutor::gather_ghosts_run (device_executor.h:123-126;
 * reference_executor.cpp:150-164): out[i] = in[indices[i]]. */
int spmv_hip_gather_f64(spmv_hip_ctx* ctx, int num_indices,
                        const int32_t* indices, const double* in, double* out,
                        void* stream);
int spmv_hip_gather_f32(spmv_hip_ctx* ctx, int num_indices,
                        const int32_t* indices, const float* in, float* out,
                        void* stream);

/* ---- reverse halo accumulate -------------------------------------------------
 * The owner-side loop of L2GMap::reverse_update (L2GMap.cpp:921-922,947-948):
 * out[indices[i]] += in[i].  The indices of ONE call must be distinct (one
 * neighbour's segment of the index buffer); the caller issues one call per
 * neighbour, in neighbour order, which reproduces the reference's ascending-i
 * accumulation order bit for bit. */
int spmv_hip_scatter_add_f64(spmv_hip_ctx* ctx, int num_indices,
                             const int32_t* indices, const double* in,
                             double* out, void* stream);
int spmv_hip_scatter_add_f32(spmv_hip_ctx* ctx, int num_indices,
                             const int32_t* indices, const float* in,
                             float* out, void* stream);

/* ---- CG building blocks -----------------------------------------------------
 * spmv::cg (cg.cpp:21-98).  All scalars live on the device (precedent:
 * cuda/cg.cuda.cu:73-84); the host never waits inside an iteration.
 *
 * Reductions are two-stage and deterministic: a kernel writes
 * spmv_hip_dot_partials_len() per-block partial sums, then
 * spmv_hip_reduce_partials_f64 adds them in index order into one double.
 */
int spmv_hip_dot_partials_len(const spmv_hip_ctx* ctx, int* len);
int spmv_hip_dot_partial_f64(spmv_hip_ctx* ctx, int64_t n, const double* x,
                             const double* y, double* partials, void* stream);
int spmv_hip_reduce_partials_f64(spmv_hip_ctx* ctx, const double* partials,
                                 double* result, void* stream);

/* Device-resident CG state: scalar history + flags.
 *   rr[k]  = ||r_k||^2 (k = 0..kmax), pAp[k] = p_k . A p_k (k = 1..kmax)
 *   done   = 1 once sqrt(rr[k]) / sqrt(rr[0]) < rtol; kstop = that k.
 * After `done` every cg_* kernel and every SpMV issued through
 * spmv_hip_cg_guard_* is a no-op, so the host may enqueue iterations ahead
 * without reading anything back (cg.cpp:80-81 semantics: x, r updated, p not).
 */
int spmv_hip_cg_ws_create(spmv_hip_ctx* ctx, int kmax, spmv_hip_cg_ws** ws);
int spmv_hip_cg_ws_destroy(spmv_hip_cg_ws* ws);
int spmv_hip_cg_ws_reset(spmv_hip_cg_ws* ws, double rtol, void* stream);
/* device addresses of the scalar slots (for the all-reduce) */
int spmv_hip_cg_ws_rr(spmv_hip_cg_ws* ws, int k, double** slot);
int spmv_hip_cg_ws_pAp(spmv_hip_cg_ws* ws, int k, double** slot);
int spmv_hip_cg_ws_partials(spmv_hip_cg_ws* ws, double** partials);
int spmv_hip_cg_ws_done_flag(spmv_hip_cg_ws* ws, const int32_t** done);
/* kmax the workspace was created with: its history holds kmax + 1 doubles */
int spmv_hip_cg_ws_capacity(const spmv_hip_cg_ws* ws, int* kmax);
/* copies {done, kstop} (2 x int32) and rr[0..kmax] to the host (async on
 * stream).  `host_rr_len` = doubles the caller's `host_rr` can take: fewer than
 * kmax + 1 (spmv_hip_cg_ws_capacity) -> SPMV_HIP_EINVAL, nothing is copied.
 * Either destination may be NULL (then it is skipped; host_rr_len ignored).
 * (ABI 2: the length argument is new -- ABI 1 wrote kmax + 1 doubles into
 * whatever it was given.) */
int spmv_hip_cg_ws_read_async(spmv_hip_cg_ws* ws, int32_t* host_done_kstop,
                              double* host_rr, size_t host_rr_len,
                              void* stream);

/* x += alpha p ; r -= alpha Ap ; partials of r.r   (cg.cpp:66-73)
 * alpha = (s*s)/pAp[k] with s = sqrt(rr[k-1]), evaluated on the device. */
int spmv_hip_cg_update_xr_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                              int64_t n, const double* p, const double* Ap,
                              double* x, double* r, void* stream);
/* The same updates regrouped so that p is read once per iteration (8 vector
 * passes instead of 9); per-element arithmetic identical:
 *   update_r : r -= alpha Ap ; partials of r.r            (cg.cpp:66,70,73)
 *   update_xp: x += alpha p ; stop test ; p = beta p + r   (cg.cpp:69,77-85)
 * x still takes the update of the converging iteration, p does not. */
int spmv_hip_cg_update_r_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                             int64_t n, const double* Ap, double* r,
                             void* stream);
int spmv_hip_cg_update_xp_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                              int64_t n, const double* r, double* x, double* p,
                              void* stream);
/* ---- consumer-side reductions (one rank) -----------------------------------------
 * The same two updates with the reducer launches folded into their prologues:
 * every workgroup adds the partials of the preceding producer itself, in the
 * reducers' order (bit-identical scalars), workgroup 0 stores pAp[k] / rr[k].
 *   update_r_cs : [p.Ap partials of the SpMV (+ pap_partials2, may be NULL)]
 *                 -> pAp[k]; stop test of iteration k-1; r -= alpha Ap;
 *                 partials of r.r into the workspace's second partial array
 *   update_xp_cs: [those r.r partials] -> rr[k]; x += alpha p; stop test;
 *                 p = beta p + r
 * For a single rank only: no all-reduce can be placed between producer and
 * consumer. */
int spmv_hip_cg_update_r_cs_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                                int64_t n, const double* Ap, double* r,
                                const double* pap_partials2, void* stream);
int spmv_hip_cg_update_xp_cs_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                                 int64_t n, const double* r, double* x,
                                 double* p, void* stream);
/* CG start (cg.cpp:39-50) in one pass: r = p = b, x = 0, partials of r.r in
 * the workspace (then spmv_hip_cg_reduce_rr(0) installs rr[0]) */
int spmv_hip_cg_init_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int64_t n,
                         const double* b, double* r, double* p, double* x,
                         void* stream);
/* Mixed-precision CG support: fp32 copy of a value array;
 * y += a x; residual replacement r = b - Ax (Ax given) with the partials of
 * r.r left in the workspace (then spmv_hip_cg_reduce_rr(k) installs rr[k]).
 * respect_done = 1 inside the loop (no-op once converged), 0 for the closing
 * check of the true residual. */
int spmv_hip_convert_f64_f32(spmv_hip_ctx* ctx, int64_t n, const double* in,
                             float* out, void* stream);
int spmv_hip_axpy_f64(spmv_hip_ctx* ctx, int64_t n, double a, const double* x,
                      double* y, void* stream);
int spmv_hip_cg_residual_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws,
                             int respect_done, int64_t n, const double* b,
                             const double* Ax, double* r, void* stream);
/* convergence test on rr[k] then p = beta p + r  (cg.cpp:77-85) */
int spmv_hip_cg_update_p_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                             int64_t n, const double* r, double* p,
                             void* stream);
/* partials -> rr[k] / pAp[k] (local part; all-reduce it afterwards) */
int spmv_hip_cg_reduce_rr(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                          void* stream);
int spmv_hip_cg_reduce_pAp(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                           void* stream);
/* same, adding a second partial array (the remote block's share of p.Ap,
 * spmv_hip_dot_partials_len() doubles) */
int spmv_hip_cg_reduce_pAp2(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                            const double* partials2, void* stream);
/* r.r partials for k = 0 (cg.cpp:47) */
int spmv_hip_cg_dot_rr_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int64_t n,
                           const double* r, void* stream);

/* ---- 3-D Poisson generator (not in the reference)
 * 7-point stencil on an n^3 grid, natural ordering, diag 6, off-diag -1.
 * Generates, directly in device memory, the CSR block of global rows
 * [row_begin,row_end) in the local column numbering create_matrix would
 * produce (Matrix.cpp:295-318): owned columns first, then ghost columns in
 * ascending global order.  `part` selects which entries are kept:
 */
enum {
  SPMV_HIP_PART_ALL = 0,        /* every entry (blocking models, Matrix.cpp:357) */
  SPMV_HIP_PART_LOCAL = 1,      /* owned columns only (Matrix.cpp:350-353)       */
  SPMV_HIP_PART_REMOTE = 2,     /* ghost columns only (Matrix.cpp:354-355)       */
  SPMV_HIP_PART_LOCAL_LOWER = 3 /* owned columns, global_row > global_col
                                   (Matrix.cpp:337-347); diagonal separate       */
};
int spmv_hip_poisson3d_count(spmv_hip_ctx* ctx, int32_t n, int64_t row_begin,
                             int64_t row_end, int part, int32_t* rowptr,
                             int64_t* host_nnz, void* stream);
int spmv_hip_poisson3d_fill_f64(spmv_hip_ctx* ctx, int32_t n,
                                int64_t row_begin, int64_t row_end, int part,
                                const int32_t* rowptr, int32_t* colind,
                                double* values, double* diagonal,
                                void* stream);
/* number of ghost columns below / above the owned range for this row block */
int spmv_hip_poisson3d_ghosts(int32_t n, int64_t row_begin, int64_t row_end,
                              int64_t* ghosts_below, int64_t* ghosts_above);
/* The same matrix on ONE BOX of a 3-D block partition: rows =
 * the box's points (first[], len[] per axis; x fastest) in the rank-major
 * numbering of Matrix::create_poisson3d_boxes; owned columns 0 .. box points,
 * ghost columns behind them = the points one step outside the faces that have
 * a neighbour box, face by face in the order -z -y -x +x +y +z (ascending
 * global id), inside a face in the order of its two in-face coordinates.
 * box_count with rowptr == NULL only reports the number of ghosts. */
int spmv_hip_poisson3d_box_count(spmv_hip_ctx* ctx, int32_t n,
                                 const int32_t first[3], const int32_t len[3],
                                 int part, int32_t* rowptr, int64_t* host_nnz,
                                 int64_t* num_ghosts, void* stream);
int spmv_hip_poisson3d_box_fill_f64(spmv_hip_ctx* ctx, int32_t n,
                                    const int32_t first[3], const int32_t len[3],
                                    int part, const int32_t* rowptr,
                                    int32_t* colind, double* values,
                                    double* diagonal, void* stream);
/* Seeded unstructured test matrix, generated on the device (not in the
 * reference: a synthetic input for measuring the general kernels on a matrix
 * WITHOUT lattice or narrow-band structure).  Square, `per_row` (1..32)
 * entries per row; each entry's column is, with probability far_permille/1000,
 * anywhere, otherwise within `band` columns of the diagonal; columns ascending
 * within a row (repeats possible), values uniform in [-1, 1).  rowptr takes
 * num_rows + 1, colind / values num_rows * per_row entries.  The numpy twin is
 * spmv_amd/poisson.py:unstructured_csr. */
int spmv_hip_unstructured_fill_f64(spmv_hip_ctx* ctx, int64_t num_rows,
                                   int per_row, int64_t band, int far_permille,
                                   uint64_t seed, int32_t* rowptr,
                                   int32_t* colind, double* values,
                                   void* stream);
/* Seeded FEM-like test matrix, generated on the device (not in the reference:
 * a synthetic input with the row-length distribution and column layout of an
 * unstructured 3-D mesh matrix in a bandwidth-reducing order -- what
 * read_petsc_binary_matrix, spmv/read_petsc.cpp:40-228, typically delivers).
 * Square; short rows of min_len..max_len entries (skewed to the short side) in
 * three clusters around row - layer, row, row + layer, each within +-jitter
 * columns; tail_permille / 1000 of the rows are LONG: tail_min..tail_max
 * entries, one per tail_stride columns around the row.  Columns strictly
 * ascending, the diagonal always present (= entries of the row + 1, the other
 * values uniform in [-1, 1)).  fem_count writes the row pointer (num_rows + 1)
 * and reports the entry count (SPMV_HIP_ERANGE beyond int32); fem_fill writes
 * colind / values.  The numpy twin is spmv_amd/poisson.py:fem_like_csr. */
typedef struct spmv_hip_fem_params {
  int64_t num_rows;
  int32_t min_len, max_len;
  int32_t layer, jitter;
  int32_t tail_permille, tail_min, tail_max, tail_stride;
  uint64_t seed;
} spmv_hip_fem_params;
int spmv_hip_fem_count(spmv_hip_ctx* ctx, const spmv_hip_fem_params* params,
                       int32_t* rowptr, int64_t* host_nnz, void* stream);
int spmv_hip_fem_fill_f64(spmv_hip_ctx* ctx, const spmv_hip_fem_params* params,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          int32_t* colind, double* values, void* stream);
/* Symmetric storage from a square general CSR block on the device, by the
 * reference's rule (spmv/Matrix.cpp:337-349: entries below the diagonal stay in
 * the block in their order, entries on the diagonal are summed into the
 * diagonal array, entries above it are dropped: the matrix is taken to be
 * symmetric).  lower_split_count writes the new row pointer (num_rows + 1) and
 * reports its entry count; lower_split_fill writes colind / values / diagonal.
 * (The reference splits on the host, inside create_matrix.) */
int spmv_hip_csr_lower_split_count(spmv_hip_ctx* ctx, int32_t num_rows,
                                   const int32_t* rowptr, const int32_t* colind,
                                   int32_t* lower_rowptr, int64_t* host_nnz,
                                   void* stream);
int spmv_hip_csr_lower_split_fill_f64(spmv_hip_ctx* ctx, int32_t num_rows,
                                      const int32_t* rowptr, const int32_t* colind,
                                      const double* values,
                                      const int32_t* lower_rowptr,
                                      int32_t* lower_colind, double* lower_values,
                                      double* diagonal, void* stream);
/* x_i = exp(-10 (5 (i/N - 1/2))^2), i = i_begin.. (demos/spmv.cpp:63-67) */
int spmv_hip_fill_gaussian_f64(spmv_hip_ctx* ctx, int64_t N, int64_t i_begin,
                               int64_t count, double* x, void* stream);
int spmv_hip_fill_const_f64(spmv_hip_ctx* ctx, int64_t count, double value,
                            double* x, void* stream);

/* ---- one-sided halo: peer stores into a neighbour's window -----------------
 * The reference's onesided_put_* models (MPI_Put into a window,
 * L2GMap.cpp:645-682).  Every rank owns a window = a staging buffer for its
 * ghost tail (stage_bytes, sized for 8-byte elements) + flag words; it is
 * exported as a HIP IPC handle (ranks in other processes) and by address (ranks
 * that are threads of this process).  A rank connects each neighbour once
 * (where its data goes in the neighbour's staging buffer, which slot it has in
 * the neighbour's flags, which segments it sends / receives; offsets and
 * counts in ELEMENTS), then every exchange is one kernel launch on `stream`:
 * signal "my segment is free", wait for the neighbour's, store the send
 * segment into the neighbour's window, raise its data flag, wait for mine, copy
 * the staging buffer into `ghost_tail`.  Waits are bounded (ctx option
 * "put_timeout_ms", 60 s by default): a neighbour that does not answer leaves
 * NaN in its ghost segment, and the context's next synchronisation
 * (spmv_hip_synchronize / stream_synchronize / event_synchronize) as well as
 * every later exchange of the window return SPMV_HIP_EPEER instead of hanging
 * or handing stale ghosts to the SpMV.
 * The window is FINE-GRAINED device memory where the runtime can export that
 * over IPC (put_fine_grained; peers on other devices store into it while the
 * owner's kernel polls); a coarse-grained window only connects peers on the
 * same device (put_connect: SPMV_HIP_ENOTSUP otherwise -- L2GMap then keeps
 * the two-sided exchange).  Validated on one device only (processes sharing a
 * GPU through IPC, and threads); see spmv_amd/csrc/hip/put.hip. */
#define SPMV_HIP_IPC_HANDLE_BYTES 64
#define SPMV_HIP_PUT_MAX_PEERS 16
typedef struct spmv_hip_put spmv_hip_put;
int spmv_hip_put_create(spmv_hip_ctx* ctx, size_t stage_bytes,
                        spmv_hip_put** put, void* ipc_handle,
                        uint64_t* raw_address, int64_t* process_id);
int spmv_hip_put_connect(spmv_hip_put* put, int k, const void* peer_ipc_handle,
                         uint64_t peer_raw_address, int64_t peer_process_id,
                         size_t peer_stage_bytes, int32_t dst_offset,
                         int32_t slot_at_peer, int32_t send_offset,
                         int32_t send_count, int32_t recv_offset,
                         int32_t recv_count, int peer_fine_grained);
int spmv_hip_put_fine_grained(const spmv_hip_put* put, int* fine_grained);
/* labels for the diagnosis below: my rank, and the rank behind neighbour slot
 * k (k = -1: my rank only) */
int spmv_hip_put_label(spmv_hip_put* put, int my_rank, int k, int peer_rank);
int spmv_hip_put_finish(spmv_hip_put* put);
int spmv_hip_put_exchange(spmv_hip_ctx* ctx, spmv_hip_put* put, size_t elem_bytes,
                          const void* send_buf, void* ghost_tail, void* stream);
int spmv_hip_put_status(const spmv_hip_put* put, int* failed);
int spmv_hip_put_destroy(spmv_hip_put* put);

/* After SPMV_HIP_EPEER: WHICH bounded wait gave up (ABI 5).  The first waiter
 * that times out -- a put kernel at its FREE or DATA flag, a reduction kernel
 * at a peer's slot -- records which wait it was, on which neighbour slot / rank,
 * the epoch it saw there and the epoch it wanted; this formats that record of
 * the context's first failed window into `buf` (empty string: none failed). */
int spmv_hip_peer_error_detail(const spmv_hip_ctx* ctx, char* buf, size_t len);

/* ---- deterministic peer reduction of the CG scalars -------------------------
 * The two MPI_Allreduce of cg() (cg.cpp:65,75; and :49) move ONE double each.
 * Instead of a ring all-reduce (two RCCL launches per iteration, a summation
 * order that is RCCL's business) every rank owns a small window, exported like
 * the put windows above; a reduction is one single-wave kernel per rank on
 * `stream`: store my value(s) and the epoch into my slot of EVERY rank's window
 * (system-scope release), wait for every rank's slot in mine, add the values in
 * RANK ORDER -- the same bits on every rank, run after run -- and write the sum
 * over `inout`.  count <= SPMV_HIP_REDUCE_MAX_COUNT doubles; nranks <=
 * SPMV_HIP_REDUCE_MAX_RANKS.  Slots are double-buffered by the epoch's parity
 * (a rank cannot be two reductions ahead of a peer: the one in between needs
 * that peer's value).  Waits are bounded like the put kernels' ("put_timeout_ms":
 * NaN in `inout`, SPMV_HIP_EPEER at the context's next synchronisation).  Every
 * rank connects every other rank (reduce_connect) before the first reduction.
 * Same memory rules as the put windows; validated on one device only (ranks as
 * threads: 2 / 3 / 8, and the 8-rank 512^3 rehearsal; ranks as processes over
 * IPC handles), with the two-sided halo models -- the pair with the one-sided
 * halo is not validated.  With ranks as THREADS of one process nothing that
 * waits for the whole device (hipMalloc, hipFree) may run between a rank's
 * reduction and its peers': size the CG workspace first. */
#define SPMV_HIP_REDUCE_MAX_RANKS 64
#define SPMV_HIP_REDUCE_MAX_COUNT 4
typedef struct spmv_hip_reduce spmv_hip_reduce;
int spmv_hip_reduce_create(spmv_hip_ctx* ctx, int nranks, int rank,
                           spmv_hip_reduce** reduce, void* ipc_handle,
                           uint64_t* raw_address, int64_t* process_id,
                           int* fine_grained);
int spmv_hip_reduce_connect(spmv_hip_reduce* reduce, int peer_rank,
                            const void* peer_ipc_handle, uint64_t peer_raw_address,
                            int64_t peer_process_id, int peer_fine_grained);
int spmv_hip_reduce_sum_f64(spmv_hip_ctx* ctx, spmv_hip_reduce* reduce,
                            double* inout, int count, void* stream);
int spmv_hip_reduce_destroy(spmv_hip_reduce* reduce);

/* ---- RCCL transport (L2GMap::update p2p models, L2GMap.cpp:564-642;
 *      MPI_Allreduce in cg.cpp:49,65,75) ---------------------------------- */
#define SPMV_HIP_UNIQUE_ID_BYTES 128
int spmv_hip_comm_unique_id(void* host_id_bytes);
int spmv_hip_comm_create(spmv_hip_ctx* ctx, int nranks, int rank,
                         const void* host_id_bytes, spmv_hip_comm** comm);
int spmv_hip_comm_destroy(spmv_hip_comm* comm);
int spmv_hip_comm_rank(const spmv_hip_comm* comm, int* rank, int* nranks);
/* What the transport really is: rank count and rank as RCCL reports them,
 * RCCL's version code, whether the reductions run on a communicator of their
 * own (split off the halo one), and the path of the RCCL library that was
 * loaded.  Any output pointer may be NULL. */
int spmv_hip_comm_info(const spmv_hip_comm* comm, int* nranks, int* rank,
                       int* rccl_version, int* separate_reduction_comm,
                       char* lib_path, int lib_path_len);
/* one grouped exchange: for every neighbour i, send send_counts[i] doubles
 * from send_buf + send_offsets[i] and receive recv_counts[i] doubles into
 * recv_base + recv_offsets[i] (offsets in elements). */
int spmv_hip_comm_neighbor_exchange_f64(spmv_hip_comm* comm, int num_neighbours,
                                        const int32_t* host_neighbours,
                                        const double* send_buf,
                                        const int32_t* host_send_counts,
                                        const int32_t* host_send_offsets,
                                        double* recv_base,
                                        const int32_t* host_recv_counts,
                                        const int32_t* host_recv_offsets,
                                        void* stream);
int spmv_hip_comm_neighbor_exchange_f32(spmv_hip_comm* comm, int num_neighbours,
                                        const int32_t* host_neighbours,
                                        const float* send_buf,
                                        const int32_t* host_send_counts,
                                        const int32_t* host_send_offsets,
                                        float* recv_base,
                                        const int32_t* host_recv_counts,
                                        const int32_t* host_recv_offsets,
                                        void* stream);
int spmv_hip_comm_allreduce_sum_f64(spmv_hip_comm* comm, double* inout,
                                    size_t count, void* stream);
/* byte-wise all-gather of small host-side setup data, staged through device
 * memory (plan construction only; L2GMap.cpp:353-354,387-388) */
int spmv_hip_comm_allgather_host(spmv_hip_comm* comm, const void* host_send,
                                 void* host_recv, size_t bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_HIP_H */
