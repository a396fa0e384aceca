// Device executors of the MI355X backend, mirroring the reference's plug-in
// seam (spmv/device_executor.h:29-151) so that CSRMatrix / L2GMap / Matrix /
// cg written against `spmv::DeviceExecutor` run unchanged on a GPU:
//
//   DeviceExecutor  abstract base, same public method set as the reference
//   HostExecutor    host memory only (malloc/memcpy): the `get_host()` side
//                   of a HipExecutor and the source of copy_from().  It has
//                   NO compute path -- its SpMV visitors throw.  (The
//                   reference's ReferenceExecutor also computes; this build
//                   keeps CPU arithmetic out of the product on purpose, it
//                   lives in oracle/ as test infrastructure.)
//   HipExecutor     one MI355X, bound to libspmv_hip.so through the C ABI in
//                   include/spmv_hip.h; mirrors CudaExecutor
//                   (spmv/cuda/cuda_executor.h:17-101).
//
// C ABI failures become std::runtime_error carrying the HIP/RCCL message
// (the reference's CUDA macros print and continue, cuda_helper.h:11-20).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct spmv_hip_ctx; // include/spmv_hip.h

namespace spmv
{

class Comm;

template <typename T>
class CSRMatrix;
template <typename T>
class CSRSpMV;

enum class DeviceType { undefined, cpu, gpu };

// Throws std::runtime_error("<what>: <decoded code>") when code != 0.
void throw_on_error(int code, const char* what);

class DeviceExecutor
{
public:
  virtual ~DeviceExecutor() = default;

  // ---- typed memory front-end (device_executor.h:38-80) -------------------
  template <typename T>
  T* alloc(size_t num_elems) const
  {
    return static_cast<T*>(_alloc(num_elems * sizeof(T)));
  }
  template <typename T>
  void memset(T* ptr, int value, size_t num_elems) const
  {
    _memset(ptr, value, num_elems * sizeof(T));
  }
  void free(void* ptr) const { _free(ptr); }
  template <typename T>
  void copy(T* dst, const T* src, size_t num_elems) const
  {
    _copy(dst, src, num_elems * sizeof(T));
  }
  template <typename T>
  void copy_async(T* dst, const T* src, size_t num_elems, void* stream) const
  {
    _copy_async(dst, src, num_elems * sizeof(T), stream);
  }
  template <typename T>
  void copy_from(T* dst, const DeviceExecutor& src_exec, const T* src,
                 size_t num_elems) const
  {
    _copy_from(dst, src_exec, src, num_elems * sizeof(T));
  }
  template <typename T>
  void copy_to(T* dst, const DeviceExecutor& dst_exec, const T* src,
               size_t num_elems) const
  {
    _copy_to(dst, dst_exec, src, num_elems * sizeof(T));
  }

  // ---- control (device_executor.h:82-85) ------------------------------------
  virtual void synchronize() const = 0;
  virtual const DeviceExecutor& get_host() const = 0;
  virtual int get_num_devices() const = 0;
  virtual int get_num_cus() const = 0;

  // ---- CSR visitors (device_executor.h:88-99) -------------------------------
  virtual void spmv_init(CSRSpMV<float>& op, const CSRMatrix<float>& mat) const = 0;
  virtual void spmv_init(CSRSpMV<double>& op, const CSRMatrix<double>& mat) const = 0;
  virtual void spmv_run(const CSRSpMV<float>& op, const CSRMatrix<float>& mat,
                        float alpha, float* in, float beta, float* out) const = 0;
  virtual void spmv_run(const CSRSpMV<double>& op, const CSRMatrix<double>& mat,
                        double alpha, double* in, double beta,
                        double* out) const = 0;
  virtual void spmv_finalize(CSRSpMV<float>& op) const = 0;
  virtual void spmv_finalize(CSRSpMV<double>& op) const = 0;

  // ---- ghost pack (device_executor.h:123-126) -------------------------------
  virtual void gather_ghosts_run(int num_indices, const int32_t* indices,
                                 const float* in, float* out) const = 0;
  virtual void gather_ghosts_run(int num_indices, const int32_t* indices,
                                 const double* in, double* out) const = 0;

  struct DeviceInfo {
    DeviceType type = DeviceType::undefined;
    int id = -1;
  };
  const DeviceInfo& get_device_info() const { return _dev_info; }
  DeviceType get_device_type() const { return _dev_info.type; }
  int get_device_id() const { return _dev_info.id; }

protected:
  // ---- byte-level back-end (device_executor.h:129-139) ----------------------
  virtual void* _alloc(size_t num_bytes) const = 0;
  virtual void _free(void* ptr) const = 0;
  virtual void _memset(void* ptr, int value, size_t num_bytes) const = 0;
  virtual void _copy(void* dst, const void* src, size_t num_bytes) const = 0;
  virtual void _copy_async(void* dst, const void* src, size_t num_bytes,
                           void* stream) const = 0;
  virtual void _copy_from(void* dst, const DeviceExecutor& src_exec,
                          const void* src, size_t num_bytes) const = 0;
  virtual void _copy_to(void* dst, const DeviceExecutor& dst_exec,
                        const void* src, size_t num_bytes) const = 0;

  DeviceInfo _dev_info;
};

// ---------------------------------------------------------------------------
class HostExecutor final : public DeviceExecutor
{
public:
  static std::unique_ptr<HostExecutor> create()
  {
    return std::unique_ptr<HostExecutor>(new HostExecutor());
  }

  void synchronize() const override {}
  const DeviceExecutor& get_host() const override { return *this; }
  int get_num_devices() const override { return 1; }
  int get_num_cus() const override { return 1; }

  void spmv_init(CSRSpMV<float>&, const CSRMatrix<float>&) const override;
  void spmv_init(CSRSpMV<double>&, const CSRMatrix<double>&) const override;
  void spmv_run(const CSRSpMV<float>&, const CSRMatrix<float>&, float, float*,
                float, float*) const override;
  void spmv_run(const CSRSpMV<double>&, const CSRMatrix<double>&, double,
                double*, double, double*) const override;
  void spmv_finalize(CSRSpMV<float>&) const override;
  void spmv_finalize(CSRSpMV<double>&) const override;
  void gather_ghosts_run(int, const int32_t*, const float*,
                         float*) const override;
  void gather_ghosts_run(int, const int32_t*, const double*,
                         double*) const override;

protected:
  void* _alloc(size_t num_bytes) const override;
  void _free(void* ptr) const override;
  void _memset(void* ptr, int value, size_t num_bytes) const override;
  void _copy(void* dst, const void* src, size_t num_bytes) const override;
  void _copy_async(void* dst, const void* src, size_t num_bytes,
                   void* stream) const override;
  void _copy_from(void* dst, const DeviceExecutor& src_exec, const void* src,
                  size_t num_bytes) const override;
  void _copy_to(void* dst, const DeviceExecutor& dst_exec, const void* src,
                size_t num_bytes) const override;

private:
  HostExecutor() { _dev_info.type = DeviceType::cpu; }
};

// ---------------------------------------------------------------------------
class HipExecutor final : public DeviceExecutor
{
public:
  // cuda/cuda_executor.h:23-30,94: create(device_id, host executor)
  static std::unique_ptr<HipExecutor>
  create(int device_id, std::shared_ptr<DeviceExecutor> host)
  {
    return std::unique_ptr<HipExecutor>(new HipExecutor(device_id, host));
  }
  ~HipExecutor() override;

  void synchronize() const override;
  const DeviceExecutor& get_host() const override { return *_host; }
  int get_num_devices() const override;
  int get_num_cus() const override;

  void spmv_init(CSRSpMV<float>& op, const CSRMatrix<float>& mat) const override;
  void spmv_init(CSRSpMV<double>& op, const CSRMatrix<double>& mat) const override;
  void spmv_run(const CSRSpMV<float>& op, const CSRMatrix<float>& mat,
                float alpha, float* in, float beta, float* out) const override;
  void spmv_run(const CSRSpMV<double>& op, const CSRMatrix<double>& mat,
                double alpha, double* in, double beta,
                double* out) const override;
  void spmv_finalize(CSRSpMV<float>& op) const override;
  void spmv_finalize(CSRSpMV<double>& op) const override;
  void gather_ghosts_run(int num_indices, const int32_t* indices,
                         const float* in, float* out) const override;
  void gather_ghosts_run(int num_indices, const int32_t* indices,
                         const double* in, double* out) const override;

  // cuda/cuda_executor.h:72-76 -- the stream every launch and async copy of
  // this executor goes to (nullptr = the device's default stream).
  void set_stream(void* hip_stream);
  void reset_stream();
  void* get_stream() const;

  // Extended API of this backend (stream/event plumbing for overlap).
  spmv_hip_ctx* context() const { return _ctx; }
  void* create_stream(bool high_priority = false) const;
  void destroy_stream(void* stream) const;
  void* create_event(bool timing = false) const;
  void destroy_event(void* event) const;
  void record_event(void* event, void* stream) const;
  void stream_wait_event(void* stream, void* event) const;
  void synchronize_stream(void* stream) const;
  void synchronize_event(void* event) const;
  // owner-side accumulate of L2GMap::reverse_update (distinct indices)
  void scatter_add_run(int num_indices, const int32_t* indices, const float* in,
                       float* out) const;
  void scatter_add_run(int num_indices, const int32_t* indices,
                       const double* in, double* out) const;
  // communicators whose peer reduction lives in this executor's context
  // (Comm::enable_peer_reduce): closed by ~HipExecutor if still open
  void attach_reduce_owner(const Comm* comm) const;
  void detach_reduce_owner(const Comm* comm) const;

protected:
  void* _alloc(size_t num_bytes) const override;
  void _free(void* ptr) const override;
  void _memset(void* ptr, int value, size_t num_bytes) const override;
  void _copy(void* dst, const void* src, size_t num_bytes) const override;
  void _copy_async(void* dst, const void* src, size_t num_bytes,
                   void* stream) const override;
  void _copy_from(void* dst, const DeviceExecutor& src_exec, const void* src,
                  size_t num_bytes) const override;
  void _copy_to(void* dst, const DeviceExecutor& dst_exec, const void* src,
                size_t num_bytes) const override;

private:
  HipExecutor(int device_id, std::shared_ptr<DeviceExecutor> host);
  std::shared_ptr<DeviceExecutor> _host;
  spmv_hip_ctx* _ctx = nullptr;
  mutable std::vector<const Comm*> _reduce_owners;
};

} // namespace spmv
