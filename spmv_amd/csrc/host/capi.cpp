// C facade over the host mirror: see include/spmv_host_c.h.
#include "spmv_host_c.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "cg.h"
#include "comm.h"
#include "csr.h"
#include "executor.h"
#include "l2gmap.h"
#include "matrix.h"
#include "read_petsc.h"

using namespace spmv;

struct spmvh_exec {
  std::shared_ptr<HostExecutor> host;
  std::shared_ptr<HipExecutor> hip;
};
struct spmvh_comm {
  std::shared_ptr<const Comm> comm;
};
struct spmvh_matrix {
  std::unique_ptr<Matrix<double>> A;
};
struct spmvh_matrix_f32 {
  std::unique_ptr<Matrix<float>> A;
};
struct spmvh_l2g {
  std::unique_ptr<L2GMap> map;
};
struct spmvh_split {
  Matrix<double>::Split s;
};
struct spmvh_petsc_rows {
  PetscRows rows;
};
struct spmvh_cg_workspace {
  std::shared_ptr<HipExecutor> exec; // keeps the executor alive
  std::unique_ptr<CgWorkspace> ws;
};

namespace
{
thread_local std::string g_last_error;

template <typename F>
int guarded(F&& f)
{
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
  } catch (...) {
    g_last_error = "unknown C++ exception";
  }
  return -1;
}

void require(bool ok, const char* what)
{
  if (!ok)
    throw std::invalid_argument(what);
}

CommunicationModel to_cm(int cm)
{
  require(cm >= 0 && cm <= 7, "unknown CommunicationModel");
  return static_cast<CommunicationModel>(cm);
}

void copy_plan(const L2GMap& m, int32_t* neighbours, int32_t* send_count,
               int32_t* recv_count, int32_t* send_offset, int32_t* recv_offset,
               int32_t* indexbuf)
{
  const size_t nn = m.neighbours().size();
  for (size_t i = 0; i < nn; ++i) {
    neighbours[i] = m.neighbours()[i];
    send_count[i] = m.send_count()[i];
    recv_count[i] = m.recv_count()[i];
  }
  for (size_t i = 0; i <= nn; ++i) {
    send_offset[i] = m.send_offset()[i];
    recv_offset[i] = m.recv_offset()[i];
  }
  std::copy(m.indexbuf().begin(), m.indexbuf().end(), indexbuf);
}
} // namespace

extern "C" {

const char* spmvh_last_error(void) { return g_last_error.c_str(); }

// ---- executor -------------------------------------------------------------------
int spmvh_exec_create(int device_id, spmvh_exec** exec)
{
  return guarded([&] {
    require(exec != nullptr, "exec == NULL");
    auto e = std::make_unique<spmvh_exec>();
    e->host = HostExecutor::create();
    e->hip = HipExecutor::create(device_id, e->host);
    *exec = e.release();
  });
}

int spmvh_exec_destroy(spmvh_exec* exec)
{
  return guarded([&] { delete exec; });
}

int spmvh_exec_alloc(spmvh_exec* exec, size_t num_bytes, void** ptr)
{
  return guarded([&] {
    require(exec && ptr, "NULL argument");
    *ptr = exec->hip->alloc<char>(num_bytes);
  });
}

int spmvh_exec_free(spmvh_exec* exec, void* ptr)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->free(ptr);
  });
}

int spmvh_exec_memset(spmvh_exec* exec, void* ptr, int value, size_t num_bytes)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->memset<char>(static_cast<char*>(ptr), value, num_bytes);
  });
}

int spmvh_exec_copy(spmvh_exec* exec, void* dst, const void* src,
                    size_t num_bytes)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->copy<char>(static_cast<char*>(dst),
                          static_cast<const char*>(src), num_bytes);
  });
}

int spmvh_exec_copy_from_host(spmvh_exec* exec, void* dst, const void* host_src,
                              size_t num_bytes)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->copy_from<char>(static_cast<char*>(dst), exec->hip->get_host(),
                               static_cast<const char*>(host_src), num_bytes);
  });
}

int spmvh_exec_copy_to_host(spmvh_exec* exec, void* host_dst, const void* src,
                            size_t num_bytes)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->copy_to<char>(static_cast<char*>(host_dst),
                             exec->hip->get_host(),
                             static_cast<const char*>(src), num_bytes);
  });
}

int spmvh_exec_synchronize(spmvh_exec* exec)
{
  return guarded([&] {
    require(exec, "NULL argument");
    exec->hip->synchronize();
  });
}

int spmvh_exec_num_cus(spmvh_exec* exec, int* num_cus)
{
  return guarded([&] {
    require(exec && num_cus, "NULL argument");
    *num_cus = exec->hip->get_num_cus();
  });
}

int spmvh_exec_device_type(spmvh_exec* exec, int* type)
{
  return guarded([&] {
    require(exec && type, "NULL argument");
    *type = static_cast<int>(exec->hip->get_device_type());
  });
}

int spmvh_exec_context(spmvh_exec* exec, void** ctx)
{
  return guarded([&] {
    require(exec && ctx, "NULL argument");
    *ctx = exec->hip->context();
  });
}

int spmvh_host_executor_rejects_compute(void)
{
  try {
    std::shared_ptr<DeviceExecutor> host = HostExecutor::create();
    const int32_t rowptr[2] = {0, 1}, colind[1] = {0};
    const double values[1] = {1.0};
    CSRMatrix<double> m(host, 1, 1, 1, rowptr, colind, values);
    (void)m;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 0;
  }
  return 1;
}

// ---- communicators ----------------------------------------------------------------
int spmvh_comm_self(spmvh_comm** comm)
{
  return guarded([&] {
    require(comm != nullptr, "comm == NULL");
    auto c = std::make_unique<spmvh_comm>();
    c->comm = std::make_shared<SelfComm>();
    *comm = c.release();
  });
}

int spmvh_rccl_unique_id(void* id_bytes)
{
  return guarded([&] {
    require(id_bytes != nullptr, "id_bytes == NULL");
    std::vector<unsigned char> id = RcclComm::unique_id();
    std::memcpy(id_bytes, id.data(), id.size());
  });
}

int spmvh_comm_rccl(spmvh_exec* exec, int nranks, int rank, const void* id_bytes,
                    spmvh_comm** comm)
{
  return guarded([&] {
    require(exec && comm && id_bytes, "NULL argument");
    auto c = std::make_unique<spmvh_comm>();
    c->comm = std::make_shared<RcclComm>(*exec->hip, nranks, rank, id_bytes);
    *comm = c.release();
  });
}

int spmvh_comm_rccl_info(spmvh_comm* comm, int out[4], char* lib_path,
                         int lib_path_len)
{
  return guarded([&] {
    require(comm && out, "NULL argument");
    const auto* rc = dynamic_cast<const RcclComm*>(comm->comm.get());
    require(rc != nullptr, "not an RCCL communicator");
    const RcclComm::Info i = rc->info();
    out[0] = i.nranks;
    out[1] = i.rank;
    out[2] = i.rccl_version;
    out[3] = i.separate_reduction_comm;
    if (lib_path && lib_path_len > 0) {
      strncpy(lib_path, i.lib_path.c_str(), (size_t)lib_path_len - 1);
      lib_path[lib_path_len - 1] = 0;
    }
  });
}

int spmvh_comm_callback(int rank, int nranks, spmvh_allgather_fn allgather,
                        spmvh_exchange_fn exchange, spmvh_allreduce_fn allreduce,
                        void* user, spmvh_comm** comm)
{
  return guarded([&] {
    require(comm && allgather && nranks >= 1 && rank >= 0 && rank < nranks,
            "bad argument");
    CommCallbacks cb;
    cb.user = user;
    cb.allgather = allgather;
    cb.neighbor_exchange = exchange;
    cb.allreduce_sum = allreduce;
    auto c = std::make_unique<spmvh_comm>();
    c->comm = std::make_shared<CallbackComm>(rank, nranks, cb);
    *comm = c.release();
  });
}

int spmvh_comm_destroy(spmvh_comm* comm)
{
  return guarded([&] {
    if (comm && comm->comm)
      comm->comm->close_peer_reduce(); // (collective; a no-op without it)
    delete comm;
  });
}

int spmvh_comm_enable_peer_reduce(spmvh_comm* comm, spmvh_exec* exec, int* ok)
{
  return guarded([&] {
    require(comm && exec && ok && exec->hip, "NULL argument / not a HipExecutor");
    *ok = comm->comm->enable_peer_reduce(*exec->hip) ? 1 : 0;
  });
}

int spmvh_comm_ranks_share_a_process(spmvh_comm* comm, int* shared)
{
  return guarded([&] {
    require(comm && shared, "NULL argument");
    *shared = comm->comm->ranks_share_a_process() ? 1 : 0;
  });
}

int spmvh_comm_reduce_sum(spmvh_comm* comm, double* device_inout, int count,
                          void* stream)
{
  return guarded([&] {
    require(comm && device_inout && count >= 1, "NULL argument");
    comm->comm->reduce_sum(device_inout, static_cast<size_t>(count), stream);
  });
}

// ---- matrix -------------------------------------------------------------------------
int spmvh_matrix_create(spmvh_comm* comm, spmvh_exec* exec,
                        const int32_t* rowptr, const int32_t* colind,
                        const double* values, int64_t nrows_local,
                        int64_t ncols_local, const int64_t* row_ghosts,
                        int64_t num_row_ghosts, const int64_t* col_ghosts,
                        int64_t num_col_ghosts, int symmetric, int cm,
                        spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A && rowptr, "NULL argument");
    std::vector<int64_t> rg, cg;
    if (num_row_ghosts > 0)
      rg.assign(row_ghosts, row_ghosts + num_row_ghosts);
    if (num_col_ghosts > 0)
      cg.assign(col_ghosts, col_ghosts + num_col_ghosts);
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(Matrix<double>::create_matrix(comm->comm, exec->hip, rowptr,
                                             colind, values, nrows_local,
                                             ncols_local, rg, cg,
                                             symmetric != 0, to_cm(cm)));
    *A = m.release();
  });
}

int spmvh_matrix_create_poisson3d(spmvh_comm* comm, spmvh_exec* exec, int32_t n,
                                  int symmetric, int cm, spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(Matrix<double>::create_poisson3d(comm->comm, exec->hip, n,
                                                symmetric != 0, to_cm(cm)));
    *A = m.release();
  });
}

int spmvh_matrix_create_unstructured(spmvh_comm* comm, spmvh_exec* exec,
                                     int64_t nrows, int per_row, int64_t band,
                                     int far_permille, uint64_t seed,
                                     spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(Matrix<double>::create_unstructured(
        comm->comm, exec->hip, nrows, per_row, band, far_permille, seed));
    *A = m.release();
  });
}

int spmvh_matrix_create_fem_like(spmvh_comm* comm, spmvh_exec* exec,
                                 const struct spmv_hip_fem_params* params,
                                 spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A && params, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(Matrix<double>::create_fem_like(comm->comm, exec->hip, *params));
    *A = m.release();
  });
}

int spmvh_matrix_create_fem_like_sym(spmvh_comm* comm, spmvh_exec* exec,
                                     const struct spmv_hip_fem_params* params,
                                     spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A && params, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(
        Matrix<double>::create_fem_like(comm->comm, exec->hip, *params, true));
    *A = m.release();
  });
}

int spmvh_matrix_create_poisson3d_boxes(spmvh_comm* comm, spmvh_exec* exec,
                                        int32_t n, int px, int py, int pz,
                                        int symmetric, int cm, spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && A, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A.reset(Matrix<double>::create_poisson3d_boxes(
        comm->comm, exec->hip, n, px, py, pz, symmetric != 0, to_cm(cm)));
    *A = m.release();
  });
}

int spmvh_poisson3d_box_rows(int32_t n, int px, int py, int pz, int rank,
                             int64_t sizes[7], int32_t* rowptr, int32_t* colind,
                             double* values, int64_t* col_ghosts)
{
  return guarded([&] {
    require(sizes, "NULL argument");
    auto b = Matrix<double>::poisson3d_box_rows(n, px, py, pz, rank);
    sizes[0] = b.rows.rows;
    sizes[1] = b.rows.non_zeros();
    sizes[2] = static_cast<int64_t>(b.col_ghosts.size());
    sizes[3] = b.global_row_offset;
    sizes[4] = b.box[0];
    sizes[5] = b.box[1];
    sizes[6] = b.box[2];
    if (rowptr)
      std::copy(b.rows.rowptr.begin(), b.rows.rowptr.end(), rowptr);
    if (colind)
      std::copy(b.rows.colind.begin(), b.rows.colind.end(), colind);
    if (values)
      std::copy(b.rows.values.begin(), b.rows.values.end(), values);
    if (col_ghosts)
      std::copy(b.col_ghosts.begin(), b.col_ghosts.end(), col_ghosts);
  });
}

int spmvh_matrix_destroy(spmvh_matrix* A)
{
  return guarded([&] { delete A; });
}

int spmvh_matrix_rows(spmvh_matrix* A, int* rows)
{
  return guarded([&] {
    require(A && rows, "NULL argument");
    *rows = A->A->rows();
  });
}

int spmvh_matrix_cols(spmvh_matrix* A, int* cols)
{
  return guarded([&] {
    require(A && cols, "NULL argument");
    *cols = A->A->cols();
  });
}

int spmvh_matrix_non_zeros(spmvh_matrix* A, int64_t* nnz)
{
  return guarded([&] {
    require(A && nnz, "NULL argument");
    *nnz = A->A->non_zeros();
  });
}

int spmvh_matrix_format_size(spmvh_matrix* A, size_t* bytes)
{
  return guarded([&] {
    require(A && bytes, "NULL argument");
    *bytes = A->A->format_size();
  });
}

int spmvh_matrix_symmetric(spmvh_matrix* A, int* symmetric)
{
  return guarded([&] {
    require(A && symmetric, "NULL argument");
    *symmetric = A->A->symmetric() ? 1 : 0;
  });
}

int spmvh_matrix_blocks(spmvh_matrix* A, int64_t out[6])
{
  return guarded([&] {
    require(A && out, "NULL argument");
    for (int i = 0; i < 6; ++i)
      out[i] = 0;
    if (const SubMatrix<double>* l = A->A->local_block()) {
      out[0] = l->rows();
      out[1] = l->cols();
      out[2] = l->non_zeros();
    }
    if (const SubMatrix<double>* r = A->A->remote_block()) {
      out[3] = r->rows();
      out[4] = r->cols();
      out[5] = r->non_zeros();
    }
  });
}

int spmvh_matrix_plan_get(spmvh_matrix* A, int remote, const char* key,
                          int* value)
{
  return guarded([&] {
    require(A && key && value, "NULL argument");
    const SubMatrix<double>* b
        = remote ? A->A->remote_block() : A->A->local_block();
    const auto* csr = dynamic_cast<const CSRMatrix<double>*>(b);
    *value = csr ? csr->query(key) : 0;
  });
}

int spmvh_matrix_plan_set(spmvh_matrix* A, int remote, const char* key,
                          int value)
{
  return guarded([&] {
    require(A && key, "NULL argument");
    const SubMatrix<double>* b
        = remote ? A->A->remote_block() : A->A->local_block();
    const auto* csr = dynamic_cast<const CSRMatrix<double>*>(b);
    require(csr != nullptr, "no such block");
    csr->tune(key, value);
  });
}

int spmvh_matrix_release_csr(spmvh_matrix* A, int64_t* bytes_freed)
{
  return guarded([&] {
    require(A && bytes_freed, "NULL argument");
    *bytes_freed = 0;
    for (const SubMatrix<double>* b : {A->A->local_block(), A->A->remote_block()})
      if (const auto* csr = dynamic_cast<const CSRMatrix<double>*>(b))
        *bytes_freed += static_cast<int64_t>(csr->release_csr());
  });
}

int spmvh_matrix_enable_mixed(spmvh_matrix* A, int* ok)
{
  return guarded([&] {
    require(A && ok, "NULL argument");
    *ok = A->A->enable_mixed() ? 1 : 0;
  });
}

int spmvh_matrix_use_mixed(spmvh_matrix* A, int on)
{
  return guarded([&] {
    require(A, "NULL argument");
    A->A->use_mixed(on != 0);
  });
}

int spmvh_matrix_update(spmvh_matrix* A, double* x)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    A->A->col_map()->update(x);
  });
}

int spmvh_matrix_update_finalise(spmvh_matrix* A, double* x)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    A->A->col_map()->update_finalise(x);
  });
}

int spmvh_matrix_mult(spmvh_matrix* A, double* x, double* y)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    A->A->mult(x, y);
  });
}

int spmvh_split_create(const int32_t* rowptr, const int32_t* colind,
                       const double* values, int64_t nrows_local,
                       int64_t ncols_local, int64_t global_row_offset,
                       int64_t global_col_offset, const int64_t* col_ghosts,
                       int64_t num_col_ghosts, int symmetric, int cm,
                       spmvh_split** split, int64_t sizes[8])
{
  return guarded([&] {
    require(rowptr && split && sizes, "NULL argument");
    std::vector<int64_t> ghosts;
    if (num_col_ghosts > 0)
      ghosts.assign(col_ghosts, col_ghosts + num_col_ghosts);
    auto sp = std::make_unique<spmvh_split>();
    sp->s = Matrix<double>::split_rows(rowptr, colind, values, nrows_local,
                                       ncols_local, global_row_offset,
                                       global_col_offset, ghosts,
                                       symmetric != 0, to_cm(cm));
    const auto& s = sp->s;
    sizes[0] = s.local.rows;
    sizes[1] = s.local.cols;
    sizes[2] = s.local.non_zeros();
    sizes[3] = s.remote.rows;
    sizes[4] = s.remote.cols;
    sizes[5] = s.remote.non_zeros();
    sizes[6] = static_cast<int64_t>(s.col_ghosts.size());
    sizes[7] = s.nnz_full;
    *split = sp.release();
  });
}

int spmvh_matrix_f32_create(spmvh_comm* comm, spmvh_exec* exec,
                            const int32_t* rowptr, const int32_t* colind,
                            const float* values, int64_t nrows_local,
                            int64_t ncols_local, const int64_t* row_ghosts,
                            int64_t num_row_ghosts, const int64_t* col_ghosts,
                            int64_t num_col_ghosts, int symmetric, int cm,
                            spmvh_matrix_f32** A)
{
  return guarded([&] {
    require(comm && exec && A && rowptr, "NULL argument");
    std::vector<int64_t> rg, cg;
    if (num_row_ghosts > 0)
      rg.assign(row_ghosts, row_ghosts + num_row_ghosts);
    if (num_col_ghosts > 0)
      cg.assign(col_ghosts, col_ghosts + num_col_ghosts);
    auto m = std::make_unique<spmvh_matrix_f32>();
    m->A.reset(Matrix<float>::create_matrix(comm->comm, exec->hip, rowptr,
                                            colind, values, nrows_local,
                                            ncols_local, rg, cg, symmetric != 0,
                                            to_cm(cm)));
    *A = m.release();
  });
}

int spmvh_matrix_f32_destroy(spmvh_matrix_f32* A)
{
  return guarded([&] { delete A; });
}

int spmvh_matrix_f32_info(spmvh_matrix_f32* A, int* rows, int64_t* nnz,
                          int32_t* local_size, int32_t* num_ghosts)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    if (rows) *rows = A->A->rows();
    if (nnz) *nnz = A->A->non_zeros();
    if (local_size) *local_size = A->A->col_map()->local_size();
    if (num_ghosts) *num_ghosts = A->A->col_map()->num_ghosts();
  });
}

int spmvh_matrix_f32_update(spmvh_matrix_f32* A, float* x)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    A->A->col_map()->update(x);
  });
}

int spmvh_matrix_f32_mult(spmvh_matrix_f32* A, float* x, float* y)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    A->A->mult(x, y);
  });
}

namespace
{
void fill_split_sizes(const Matrix<double>::Split& s, int64_t sizes[8])
{
  sizes[0] = s.local.rows;
  sizes[1] = s.local.cols;
  sizes[2] = s.local.non_zeros();
  sizes[3] = s.remote.rows;
  sizes[4] = s.remote.cols;
  sizes[5] = s.remote.non_zeros();
  sizes[6] = static_cast<int64_t>(s.col_ghosts.size());
  sizes[7] = s.nnz_full;
}
} // namespace

int spmvh_split_create_dist(spmvh_comm* comm, const int32_t* rowptr,
                            const int32_t* colind, const double* values,
                            int64_t nrows_local, int64_t ncols_local,
                            const int64_t* row_ghosts, int64_t num_row_ghosts,
                            const int64_t* col_ghosts, int64_t num_col_ghosts,
                            int symmetric, int cm, spmvh_split** split,
                            int64_t sizes[8])
{
  return guarded([&] {
    require(comm && rowptr && split && sizes, "NULL argument");
    std::vector<int64_t> rg, cg;
    if (num_row_ghosts > 0)
      rg.assign(row_ghosts, row_ghosts + num_row_ghosts);
    if (num_col_ghosts > 0)
      cg.assign(col_ghosts, col_ghosts + num_col_ghosts);
    auto sp = std::make_unique<spmvh_split>();
    sp->s = Matrix<double>::split_rows_distributed(
        *comm->comm, rowptr, colind, values, nrows_local, ncols_local, rg, cg,
        symmetric != 0, to_cm(cm));
    fill_split_sizes(sp->s, sizes);
    *split = sp.release();
  });
}

int spmvh_split_get(spmvh_split* split, int which, int32_t* rowptr,
                    int32_t* colind, double* values)
{
  return guarded([&] {
    require(split && (which == 0 || which == 1), "bad argument");
    const CsrHost<double>& m = which == 0 ? split->s.local : split->s.remote;
    if (rowptr)
      std::copy(m.rowptr.begin(), m.rowptr.end(), rowptr);
    if (colind)
      std::copy(m.colind.begin(), m.colind.end(), colind);
    if (values)
      std::copy(m.values.begin(), m.values.end(), values);
  });
}

int spmvh_split_extra(spmvh_split* split, double* diagonal, int64_t* ghosts)
{
  return guarded([&] {
    require(split != nullptr, "NULL argument");
    if (diagonal)
      std::copy(split->s.diagonal.begin(), split->s.diagonal.end(), diagonal);
    if (ghosts)
      std::copy(split->s.col_ghosts.begin(), split->s.col_ghosts.end(), ghosts);
  });
}

int spmvh_split_destroy(spmvh_split* split)
{
  return guarded([&] { delete split; });
}

// ---- L2GMap inspection ----------------------------------------------------------------
int spmvh_l2g_sizes(spmvh_matrix* A, int32_t* local_size, int32_t* num_ghosts,
                    int64_t* global_size, int64_t* global_offset,
                    int* overlapping, int* num_neighbours, int* num_indices,
                    int* packs)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    std::shared_ptr<const L2GMap> m = A->A->col_map();
    if (local_size) *local_size = m->local_size();
    if (num_ghosts) *num_ghosts = m->num_ghosts();
    if (global_size) *global_size = m->global_size();
    if (global_offset) *global_offset = m->global_offset();
    if (overlapping) *overlapping = m->overlapping() ? 1 : 0;
    if (num_neighbours) *num_neighbours = static_cast<int>(m->neighbours().size());
    if (num_indices) *num_indices = static_cast<int>(m->indexbuf().size());
    if (packs) *packs = m->packs() ? 1 : 0;
  });
}

int spmvh_l2g_onesided(spmvh_matrix* A, int* onesided)
{
  return guarded([&] {
    require(A && onesided, "NULL argument");
    *onesided = A->A->col_map()->onesided() ? 1 : 0;
  });
}

int spmvh_l2g_ghosts(spmvh_matrix* A, int64_t* ghosts)
{
  return guarded([&] {
    require(A && ghosts, "NULL argument");
    const std::vector<int64_t>& g = A->A->col_map()->ghosts();
    std::copy(g.begin(), g.end(), ghosts);
  });
}

int spmvh_l2g_plan(spmvh_matrix* A, int32_t* neighbours, int32_t* send_count,
                   int32_t* recv_count, int32_t* send_offset,
                   int32_t* recv_offset, int32_t* indexbuf)
{
  return guarded([&] {
    require(A != nullptr, "NULL argument");
    copy_plan(*A->A->col_map(), neighbours, send_count, recv_count, send_offset,
              recv_offset, indexbuf);
  });
}

int spmvh_l2g_global_to_local(spmvh_matrix* A, int64_t global, int32_t* local)
{
  return guarded([&] {
    require(A && local, "NULL argument");
    *local = A->A->col_map()->global_to_local(global);
  });
}

int spmvh_l2g_create(spmvh_comm* comm, spmvh_exec* exec, int use_host_exec,
                     int64_t local_size, const int64_t* ghosts,
                     int64_t num_ghosts, int cm, spmvh_l2g** map)
{
  return guarded([&] {
    require(comm && map && (use_host_exec || exec), "NULL argument");
    std::vector<int64_t> g;
    if (num_ghosts > 0)
      g.assign(ghosts, ghosts + num_ghosts);
    std::shared_ptr<DeviceExecutor> e;
    if (use_host_exec)
      e = HostExecutor::create();
    else
      e = exec->hip;
    auto m = std::make_unique<spmvh_l2g>();
    m->map.reset(new L2GMap(comm->comm, local_size, g, e, to_cm(cm)));
    *map = m.release();
  });
}

int spmvh_l2g_destroy(spmvh_l2g* map)
{
  return guarded([&] { delete map; });
}

int spmvh_l2g_map_sizes(spmvh_l2g* map, int* num_neighbours, int* num_indices,
                        int* packs)
{
  return guarded([&] {
    require(map != nullptr, "NULL argument");
    if (num_neighbours)
      *num_neighbours = static_cast<int>(map->map->neighbours().size());
    if (num_indices)
      *num_indices = static_cast<int>(map->map->indexbuf().size());
    if (packs)
      *packs = map->map->packs() ? 1 : 0;
  });
}

int spmvh_l2g_map_plan(spmvh_l2g* map, int32_t* neighbours, int32_t* send_count,
                       int32_t* recv_count, int32_t* send_offset,
                       int32_t* recv_offset, int32_t* indexbuf)
{
  return guarded([&] {
    require(map != nullptr, "NULL argument");
    copy_plan(*map->map, neighbours, send_count, recv_count, send_offset,
              recv_offset, indexbuf);
  });
}

int spmvh_l2g_map_update(spmvh_l2g* map, double* x)
{
  return guarded([&] {
    require(map != nullptr, "NULL argument");
    map->map->update(x);
  });
}

int spmvh_l2g_map_reverse_update(spmvh_l2g* map, double* x)
{
  return guarded([&] {
    require(map != nullptr, "NULL argument");
    map->map->reverse_update(x);
  });
}

int spmvh_l2g_map_reverse_update_f32(spmvh_l2g* map, float* x)
{
  return guarded([&] {
    require(map != nullptr, "NULL argument");
    map->map->reverse_update(x);
  });
}

// ---- cg ---------------------------------------------------------------------------------
int spmvh_cg(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
             const double* b, double* x, int kmax, double rtol, int* num_its,
             double* rnorm_history)
{
  return guarded([&] {
    require(comm && exec && A && num_its, "NULL argument");
    std::vector<double> hist;
    *num_its = cg(*comm->comm, *exec->hip, *A->A, b, x, kmax, rtol,
                  rnorm_history ? &hist : nullptr);
    if (rnorm_history)
      std::copy(hist.begin(), hist.end(), rnorm_history);
  });
}

int spmvh_read_petsc_matrix(spmvh_comm* comm, spmvh_exec* exec,
                            const char* filename, int symmetric, int cm,
                            spmvh_matrix** A)
{
  return guarded([&] {
    require(comm && exec && filename && A, "NULL argument");
    auto m = std::make_unique<spmvh_matrix>();
    m->A = read_petsc_binary_matrix(filename, comm->comm, exec->hip,
                                    symmetric != 0, to_cm(cm));
    *A = m.release();
  });
}

int spmvh_read_petsc_vector(spmvh_comm* comm, spmvh_exec* exec,
                            const char* filename, double** device_vec,
                            int64_t* nrows_local)
{
  return guarded([&] {
    require(comm && exec && filename && device_vec, "NULL argument");
    *device_vec = read_petsc_binary_vector(*comm->comm, exec->hip.get(),
                                           filename, nrows_local);
  });
}

int spmvh_petsc_rows_read(const char* filename, int rank, int size,
                          spmvh_petsc_rows** rows, int64_t sizes[7])
{
  return guarded([&] {
    require(filename && rows && sizes && size >= 1 && rank >= 0 && rank < size,
            "bad argument");
    auto r = std::make_unique<spmvh_petsc_rows>();
    r->rows = read_petsc_binary_rows(filename, rank, size);
    sizes[0] = r->rows.nrows_global;
    sizes[1] = r->rows.ncols_global;
    sizes[2] = r->rows.nnz_global;
    sizes[3] = r->rows.row_begin;
    sizes[4] = r->rows.row_end;
    sizes[5] = static_cast<int64_t>(r->rows.values.size());
    sizes[6] = static_cast<int64_t>(r->rows.col_ghosts.size());
    *rows = r.release();
  });
}

int spmvh_petsc_rows_get(spmvh_petsc_rows* rows, int32_t* rowptr,
                         int32_t* colind, double* values, int64_t* col_ghosts)
{
  return guarded([&] {
    require(rows != nullptr, "NULL argument");
    const PetscRows& r = rows->rows;
    if (rowptr)
      std::copy(r.rowptr.begin(), r.rowptr.end(), rowptr);
    if (colind)
      std::copy(r.colind.begin(), r.colind.end(), colind);
    if (values)
      std::copy(r.values.begin(), r.values.end(), values);
    if (col_ghosts)
      std::copy(r.col_ghosts.begin(), r.col_ghosts.end(), col_ghosts);
  });
}

int spmvh_petsc_rows_destroy(spmvh_petsc_rows* rows)
{
  return guarded([&] { delete rows; });
}

int spmvh_cg_workspace_create(spmvh_exec* exec, spmvh_cg_workspace** ws)
{
  return guarded([&] {
    require(exec && ws, "NULL argument");
    auto w = std::make_unique<spmvh_cg_workspace>();
    w->exec = exec->hip;
    w->ws.reset(new CgWorkspace(*exec->hip));
    *ws = w.release();
  });
}

int spmvh_cg_workspace_destroy(spmvh_cg_workspace* ws)
{
  return guarded([&] { delete ws; });
}

int spmvh_cg_workspace_reserve_timing(spmvh_cg_workspace* ws, int iterations)
{
  return guarded([&] {
    require(ws, "NULL argument");
    ws->ws->reserve_timing(iterations);
  });
}

int spmvh_cg_mixed(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
                   const double* b, double* x, int kmax, double rtol,
                   int replace_every, int* num_its, double* rnorm_history,
                   int history_capacity, spmvh_cg_workspace* ws, int time_spmv,
                   double out_stats[6])
{
  return guarded([&] {
    require(comm && exec && A && num_its, "NULL argument");
    std::vector<double> hist;
    CgOptions opt;
    opt.time_spmv = (time_spmv & 1) != 0;
    opt.consumer_reductions = (time_spmv & 4) == 0;
    opt.mixed = true;
    opt.replace_every = replace_every;
    CgStats st;
    *num_its = cg(*comm->comm, *exec->hip, *A->A, b, x, kmax, rtol,
                  rnorm_history ? &hist : nullptr, &opt, &st,
                  ws ? ws->ws.get() : nullptr);
    if (rnorm_history)
      std::copy(hist.begin(),
                hist.begin()
                    + std::min<size_t>(hist.size(), (size_t)history_capacity),
                rnorm_history);
    if (out_stats) {
      out_stats[0] = st.spmv_ms_total;
      out_stats[1] = st.spmv_launches;
      out_stats[2] = st.replacements;
      out_stats[3] = st.true_rel_residual;
      out_stats[4] = st.continuation_iterations;
      out_stats[5] = st.final_true_rel_residual;
    }
  });
}

int spmvh_cg_ex(spmvh_comm* comm, spmvh_exec* exec, spmvh_matrix* A,
                const double* b, double* x, int kmax, double rtol, int* num_its,
                double* rnorm_history, spmvh_cg_workspace* ws, int time_spmv,
                double* spmv_ms_total, int* spmv_launches)
{
  return guarded([&] {
    require(comm && exec && A && num_its, "NULL argument");
    std::vector<double> hist;
    CgOptions opt;
    opt.time_spmv = (time_spmv & 1) != 0;
    opt.consumer_reductions = (time_spmv & 4) == 0; // bit 2 switches it off
    if ((time_spmv >> 8) & 0xff) // bits 8-15: CgOptions::poll_every (0 = default)
      opt.poll_every = (time_spmv >> 8) & 0xff;
    CgStats st;
    *num_its = cg(*comm->comm, *exec->hip, *A->A, b, x, kmax, rtol,
                  rnorm_history ? &hist : nullptr, &opt, &st,
                  ws ? ws->ws.get() : nullptr);
    if (rnorm_history)
      std::copy(hist.begin(), hist.end(), rnorm_history);
    if (spmv_ms_total)
      *spmv_ms_total = st.spmv_ms_total;
    if (spmv_launches)
      *spmv_launches = st.spmv_launches;
  });
}

} // extern "C"
