// HostExecutor / HipExecutor: see executor.h.
#include "executor.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "comm.h"
#include "csr.h"
#include "spmv_hip.h"

namespace spmv
{

void throw_on_error(int code, const char* what)
{
  if (code != SPMV_HIP_OK)
    throw std::runtime_error(std::string(what) + ": "
                             + spmv_hip_error_string(code) + " (code "
                             + std::to_string(code) + ")");
}

// a synchronisation point that found a timed-out wait (SPMV_HIP_EPEER): say
// which wait it was, on whom, and the epochs (spmv_hip_peer_error_detail)
static void throw_on_sync_error(spmv_hip_ctx* ctx, int code, const char* what)
{
  if (code != SPMV_HIP_EPEER)
    return throw_on_error(code, what);
  char detail[512] = {0};
  (void)spmv_hip_peer_error_detail(ctx, detail, sizeof(detail));
  throw std::runtime_error(std::string(what) + ": " + spmv_hip_error_string(code)
                           + " (code " + std::to_string(code) + ")"
                           + (detail[0] ? std::string(" -- ") + detail : ""));
}

// ---------------------------------------------------------------------------
// HostExecutor: memory only
// ---------------------------------------------------------------------------
namespace
{
[[noreturn]] void no_cpu_compute(const char* what)
{
  throw std::runtime_error(
      std::string("spmv::HostExecutor::") + what
      + ": this build has no CPU compute path; create the matrix with a "
        "HipExecutor");
}
} // namespace

void* HostExecutor::_alloc(size_t num_bytes) const
{
  return std::malloc(num_bytes ? num_bytes : 1);
}
void HostExecutor::_free(void* ptr) const { std::free(ptr); }
void HostExecutor::_memset(void* ptr, int value, size_t num_bytes) const
{
  if (num_bytes)
    std::memset(ptr, value, num_bytes);
}
void HostExecutor::_copy(void* dst, const void* src, size_t num_bytes) const
{
  if (num_bytes && dst != src)
    std::memcpy(dst, src, num_bytes);
}
void HostExecutor::_copy_async(void* dst, const void* src, size_t num_bytes,
                               void*) const
{
  _copy(dst, src, num_bytes);
}
void HostExecutor::_copy_from(void* dst, const DeviceExecutor& src_exec,
                              const void* src, size_t num_bytes) const
{
  if (src_exec.get_device_type() == DeviceType::cpu)
    _copy(dst, src, num_bytes);
  else
    src_exec.copy_to(static_cast<char*>(dst), *this,
                     static_cast<const char*>(src), num_bytes);
}
void HostExecutor::_copy_to(void* dst, const DeviceExecutor& dst_exec,
                            const void* src, size_t num_bytes) const
{
  dst_exec.copy_from(static_cast<char*>(dst), *this,
                     static_cast<const char*>(src), num_bytes);
}

void HostExecutor::spmv_init(CSRSpMV<float>&, const CSRMatrix<float>&) const
{
  no_cpu_compute("spmv_init");
}
void HostExecutor::spmv_init(CSRSpMV<double>&, const CSRMatrix<double>&) const
{
  no_cpu_compute("spmv_init");
}
void HostExecutor::spmv_run(const CSRSpMV<float>&, const CSRMatrix<float>&,
                            float, float*, float, float*) const
{
  no_cpu_compute("spmv_run");
}
void HostExecutor::spmv_run(const CSRSpMV<double>&, const CSRMatrix<double>&,
                            double, double*, double, double*) const
{
  no_cpu_compute("spmv_run");
}
void HostExecutor::spmv_finalize(CSRSpMV<float>&) const {}
void HostExecutor::spmv_finalize(CSRSpMV<double>&) const {}
void HostExecutor::gather_ghosts_run(int, const int32_t*, const float*,
                                     float*) const
{
  no_cpu_compute("gather_ghosts_run");
}
void HostExecutor::gather_ghosts_run(int, const int32_t*, const double*,
                                     double*) const
{
  no_cpu_compute("gather_ghosts_run");
}

// ---------------------------------------------------------------------------
// HipExecutor
// ---------------------------------------------------------------------------
HipExecutor::HipExecutor(int device_id, std::shared_ptr<DeviceExecutor> host)
    : _host(std::move(host))
{
  if (!_host || _host->get_device_type() != DeviceType::cpu)
    throw std::runtime_error("HipExecutor: host executor must be a CPU one");
  throw_on_error(spmv_hip_ctx_create(device_id, &_ctx), "spmv_hip_ctx_create");
  _dev_info.type = DeviceType::gpu; // cuda/cuda_executor.cpp:19-20
  _dev_info.id = device_id;
}

HipExecutor::~HipExecutor()
{
  // a communicator that outlives its executor: its reduction window lives in
  // this context (collective, like the communicator's own destruction)
  const std::vector<const Comm*> owners = _reduce_owners;
  for (const Comm* c : owners) {
    try {
      c->close_peer_reduce();
    } catch (...) {
    }
  }
  spmv_hip_ctx_destroy(_ctx);
}

void HipExecutor::attach_reduce_owner(const Comm* comm) const
{
  _reduce_owners.push_back(comm);
}

void HipExecutor::detach_reduce_owner(const Comm* comm) const
{
  _reduce_owners.erase(
      std::remove(_reduce_owners.begin(), _reduce_owners.end(), comm),
      _reduce_owners.end());
}

void HipExecutor::synchronize() const
{
  throw_on_sync_error(_ctx, spmv_hip_synchronize(_ctx), "spmv_hip_synchronize");
}

int HipExecutor::get_num_devices() const
{
  int n = 0;
  throw_on_error(spmv_hip_device_count(&n), "spmv_hip_device_count");
  return n;
}

int HipExecutor::get_num_cus() const
{
  int n = 0;
  throw_on_error(spmv_hip_num_cus(_ctx, &n), "spmv_hip_num_cus");
  return n;
}

void HipExecutor::set_stream(void* s)
{
  throw_on_error(spmv_hip_set_stream(_ctx, s), "spmv_hip_set_stream");
}
void HipExecutor::reset_stream() { set_stream(nullptr); }
void* HipExecutor::get_stream() const
{
  void* s = nullptr;
  throw_on_error(spmv_hip_get_stream(_ctx, &s), "spmv_hip_get_stream");
  return s;
}
void* HipExecutor::create_stream(bool high_priority) const
{
  void* s = nullptr;
  if (high_priority)
    throw_on_error(spmv_hip_stream_create_priority(_ctx, 1, &s),
                   "spmv_hip_stream_create_priority");
  else
    throw_on_error(spmv_hip_stream_create(_ctx, &s), "spmv_hip_stream_create");
  return s;
}
void HipExecutor::destroy_stream(void* s) const
{
  throw_on_error(spmv_hip_stream_destroy(_ctx, s), "spmv_hip_stream_destroy");
}
void* HipExecutor::create_event(bool timing) const
{
  void* e = nullptr;
  throw_on_error(spmv_hip_event_create(_ctx, timing ? 1 : 0, &e),
                 "spmv_hip_event_create");
  return e;
}
void HipExecutor::destroy_event(void* e) const
{
  throw_on_error(spmv_hip_event_destroy(_ctx, e), "spmv_hip_event_destroy");
}
void HipExecutor::record_event(void* e, void* s) const
{
  throw_on_error(spmv_hip_event_record(_ctx, e, s), "spmv_hip_event_record");
}
void HipExecutor::stream_wait_event(void* s, void* e) const
{
  throw_on_error(spmv_hip_stream_wait_event(_ctx, s, e),
                 "spmv_hip_stream_wait_event");
}
void HipExecutor::synchronize_stream(void* s) const
{
  throw_on_sync_error(_ctx, spmv_hip_stream_synchronize(_ctx, s),
                      "spmv_hip_stream_synchronize");
}

void HipExecutor::synchronize_event(void* e) const
{
  throw_on_sync_error(_ctx, spmv_hip_event_synchronize(_ctx, e),
                      "spmv_hip_event_synchronize");
}

void* HipExecutor::_alloc(size_t num_bytes) const
{
  void* p = nullptr;
  throw_on_error(spmv_hip_alloc(_ctx, num_bytes, &p), "spmv_hip_alloc");
  return p;
}
void HipExecutor::_free(void* ptr) const
{
  throw_on_error(spmv_hip_free(_ctx, ptr), "spmv_hip_free");
}
void HipExecutor::_memset(void* ptr, int value, size_t num_bytes) const
{
  throw_on_error(spmv_hip_memset_async(_ctx, ptr, value, num_bytes, nullptr),
                 "spmv_hip_memset_async");
}
void HipExecutor::_copy(void* dst, const void* src, size_t num_bytes) const
{
  // stream-ordered on the executor's current stream
  throw_on_error(spmv_hip_copy_d2d_async(_ctx, dst, src, num_bytes, nullptr),
                 "spmv_hip_copy_d2d_async");
}
void HipExecutor::_copy_async(void* dst, const void* src, size_t num_bytes,
                              void* stream) const
{
  throw_on_error(spmv_hip_copy_d2d_async(_ctx, dst, src, num_bytes, stream),
                 "spmv_hip_copy_d2d_async");
}
void HipExecutor::_copy_from(void* dst, const DeviceExecutor& src_exec,
                             const void* src, size_t num_bytes) const
{
  if (src_exec.get_device_type() == DeviceType::cpu) {
    // host -> device; blocking like the reference (the source may be a
    // temporary, cuda/cuda_executor.cpp:82-94)
    throw_on_error(spmv_hip_copy_h2d_async(_ctx, dst, src, num_bytes, nullptr),
                   "spmv_hip_copy_h2d_async");
    throw_on_error(spmv_hip_stream_synchronize(_ctx, nullptr),
                   "spmv_hip_stream_synchronize");
  } else if (auto* peer = dynamic_cast<const HipExecutor*>(&src_exec)) {
    throw_on_error(spmv_hip_copy_peer_async(_ctx, dst, peer->_ctx, src,
                                            num_bytes, nullptr),
                   "spmv_hip_copy_peer_async");
  } else {
    throw std::runtime_error("HipExecutor::copy_from: unknown source executor");
  }
}
void HipExecutor::_copy_to(void* dst, const DeviceExecutor& dst_exec,
                           const void* src, size_t num_bytes) const
{
  if (dst_exec.get_device_type() == DeviceType::cpu) {
    throw_on_error(spmv_hip_copy_d2h_async(_ctx, dst, src, num_bytes, nullptr),
                   "spmv_hip_copy_d2h_async");
    throw_on_error(spmv_hip_stream_synchronize(_ctx, nullptr),
                   "spmv_hip_stream_synchronize");
  } else if (auto* peer = dynamic_cast<const HipExecutor*>(&dst_exec)) {
    throw_on_error(spmv_hip_copy_peer_async(peer->_ctx, dst, _ctx, src,
                                            num_bytes, nullptr),
                   "spmv_hip_copy_peer_async");
  } else {
    throw std::runtime_error("HipExecutor::copy_to: unknown target executor");
  }
}

// double dispatch, cuda/cuda_executor.cpp:96-150
// context option "release_csr": a matrix whose plan holds it in a format of its
// own gives the device copies of colind / values back (CSRMatrix::release_csr)
template <typename T>
static void release_if_asked(spmv_hip_ctx* ctx, const CSRMatrix<T>& mat)
{
  int64_t on = 0;
  if (spmv_hip_ctx_get_option(ctx, "release_csr", &on) == SPMV_HIP_OK && on)
    (void)mat.release_csr();
}

void HipExecutor::spmv_init(CSRSpMV<float>& op, const CSRMatrix<float>& mat) const
{
  op.init(mat.rows(), mat.cols(), mat.non_zeros(), mat.rowptr(), mat.colind(),
          mat.values(), mat.symmetric(), *this);
  op.bake_values(mat.values(), mat.symmetric() ? mat.diagonal() : nullptr,
                 *this);
  release_if_asked(_ctx, mat);
}
void HipExecutor::spmv_init(CSRSpMV<double>& op,
                            const CSRMatrix<double>& mat) const
{
  op.init(mat.rows(), mat.cols(), mat.non_zeros(), mat.rowptr(), mat.colind(),
          mat.values(), mat.symmetric(), *this);
  op.bake_values(mat.values(), mat.symmetric() ? mat.diagonal() : nullptr,
                 *this);
  release_if_asked(_ctx, mat);
}
void HipExecutor::spmv_run(const CSRSpMV<float>& op, const CSRMatrix<float>& mat,
                           float alpha, float* in, float beta, float* out) const
{
  op.run(mat.rows(), mat.cols(), mat.non_zeros(), mat.rowptr(), mat.colind(),
         mat.values(), mat.diagonal(), alpha, in, beta, out, *this);
}
void HipExecutor::spmv_run(const CSRSpMV<double>& op,
                           const CSRMatrix<double>& mat, double alpha,
                           double* in, double beta, double* out) const
{
  op.run(mat.rows(), mat.cols(), mat.non_zeros(), mat.rowptr(), mat.colind(),
         mat.values(), mat.diagonal(), alpha, in, beta, out, *this);
}
void HipExecutor::spmv_finalize(CSRSpMV<float>& op) const { op.finalize(*this); }
void HipExecutor::spmv_finalize(CSRSpMV<double>& op) const { op.finalize(*this); }

void HipExecutor::gather_ghosts_run(int num_indices, const int32_t* indices,
                                    const float* in, float* out) const
{
  throw_on_error(spmv_hip_gather_f32(_ctx, num_indices, indices, in, out,
                                     nullptr),
                 "spmv_hip_gather_f32");
}
void HipExecutor::gather_ghosts_run(int num_indices, const int32_t* indices,
                                    const double* in, double* out) const
{
  throw_on_error(spmv_hip_gather_f64(_ctx, num_indices, indices, in, out,
                                     nullptr),
                 "spmv_hip_gather_f64");
}

void HipExecutor::scatter_add_run(int num_indices, const int32_t* indices,
                                  const float* in, float* out) const
{
  throw_on_error(spmv_hip_scatter_add_f32(_ctx, num_indices, indices, in, out,
                                          nullptr),
                 "spmv_hip_scatter_add_f32");
}
void HipExecutor::scatter_add_run(int num_indices, const int32_t* indices,
                                  const double* in, double* out) const
{
  throw_on_error(spmv_hip_scatter_add_f64(_ctx, num_indices, indices, in, out,
                                          nullptr),
                 "spmv_hip_scatter_add_f64");
}

} // namespace spmv
