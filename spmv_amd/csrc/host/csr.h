// Local CSR block and its SpMV operator, mirroring
//   spmv/csr_kernels.h:26-82   CSRSpMV<T>::{init,run,finalize}(…, const XExecutor&)
//   spmv/sub_matrix.h:26-122   SubMatrix<T>
//   spmv/csr_matrix.{h,cpp}    CSRMatrix<T>
// Eigen is not available here; where the reference's
// constructors take an Eigen::SparseMatrix this build takes a CsrHost<T>, a
// plain host CSR triple with the same three arrays.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "executor.h"

struct spmv_hip_csr_plan;

namespace spmv
{

// Host-side CSR container standing in for Eigen::SparseMatrix<T, RowMajor>
// (outerIndexPtr / innerIndexPtr / valuePtr).
template <typename T>
struct CsrHost {
  int32_t rows = 0, cols = 0;
  std::vector<int32_t> rowptr = {0};
  std::vector<int32_t> colind;
  std::vector<T> values;
  int64_t non_zeros() const { return static_cast<int64_t>(values.size()); }
};

template <typename T>
class CSRSpMV
{
public:
  // overload set per concrete executor (csr_kernels.h:26-78)
  void init(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
            const int32_t* rowptr, const int32_t* colind, const T* values,
            bool symmetric, const HipExecutor& exec);
  void run(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
           const int32_t* rowptr, const int32_t* colind, const T* values,
           const T* diagonal, T alpha, T* in, T beta, T* out,
           const HipExecutor& exec) const;
  void finalize(const HipExecutor& exec) const;
  // Second half of init (the reference's init sees the values,
  // csr_kernels.h:28; a symmetric block's diagonal lives in SubMatrix): lets the
  // plan keep the values by offset (spmv_hip_csr_plan_bake_values_*) -- a
  // symmetric block in the symmetric lattice form, or a GENERAL block
  // (diagonal = nullptr) that turns out to be symmetric entry for entry.
  // Returns false, and changes nothing, when the block does not qualify.
  bool bake_values(const T* values, const T* diagonal,
                   const HipExecutor& exec) const;

  // Extension used by cg(): same as run() and additionally emits the
  // per-workgroup partial sums of sum_i in[i]*(alpha*(A in)_i) into
  // dot_partials (spmv_hip_dot_partials_len() doubles on the device).
  // General blocks only.
  void run_dot(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
               const int32_t* rowptr, const int32_t* colind, const T* values,
               T alpha, T* in, T beta, T* out, double* dot_partials,
               const HipExecutor& exec) const;

  // symmetric block: run() + the block's share of in . (alpha A in) as
  // per-workgroup partials (mirror identity, see include/spmv_hip.h)
  void run_dot_sym(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
                   const int32_t* rowptr, const int32_t* colind, const T* values,
                   const T* diagonal, T alpha, T* in, T beta, T* out,
                   double* dot_partials, const HipExecutor& exec) const;

  // launch-shape knob of the plan (spmv_hip_csr_plan_set); throws on an
  // unknown key
  void tune(const char* key, int value) const;
  // Mixed precision (the float visitors of
  // device_executor.h:88-99): `values32` is an fp32 copy of the block's values,
  // x / y / arithmetic stay fp64.  General blocks; dot_partials may be NULL.
  void run_mixed(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
                 const int32_t* rowptr, const int32_t* colind,
                 const float* values32, T alpha, T* in, T beta, T* out,
                 double* dot_partials, const HipExecutor& exec) const;

  // read-only counterpart (spmv_hip_csr_plan_get): which form the plan took,
  // what it cost ("lat", "lx", "slat", "sym_det", "plan_us", "plan_kib", ...)
  int query(const char* key) const;

  bool symmetric() const { return _symmetric; }
  spmv_hip_csr_plan* plan() const
  {
    return static_cast<spmv_hip_csr_plan*>(_aux_data);
  }

private:
  bool _symmetric = false;
  mutable void* _aux_data = nullptr; // spmv_hip_csr_plan* (csr_kernels.h:82)
};

template <typename T>
class SubMatrix
{
public:
  virtual ~SubMatrix() = default;
  int32_t rows() const { return _num_rows; }
  int32_t cols() const { return _num_cols; }
  int64_t non_zeros() const { return _num_non_zeros; }
  bool symmetric() const { return _symmetric; }
  const T* diagonal() const { return _diagonal; }
  T* diagonal() { return _diagonal; }
  virtual size_t format_size() const = 0;
  // out = alpha*A*in + beta*out (sub_matrix.h:112-113)
  virtual void mult(T alpha, T* in, T beta, T* out) const = 0;

protected:
  std::shared_ptr<DeviceExecutor> _exec;
  int32_t _num_rows = 0;
  int32_t _num_cols = 0;
  int64_t _num_non_zeros = 0;
  bool _symmetric = false;
  T* _diagonal = nullptr;
};

template <typename T>
class CSRMatrix final : public SubMatrix<T>
{
public:
  // csr_matrix.cpp:11-20 with CsrHost in place of the Eigen matrix
  CSRMatrix(std::shared_ptr<DeviceExecutor> exec, const CsrHost<T>* mat,
            const std::vector<T>* diagonal = nullptr, bool symmetric = false);
  // csr_matrix.cpp:22-59: HOST arrays, copied to the device via the executor
  CSRMatrix(std::shared_ptr<DeviceExecutor> exec, int32_t num_rows,
            int32_t num_cols, int64_t num_non_zeros, const int32_t* rowptr,
            const int32_t* colind, const T* values, const T* diagonal = nullptr,
            bool symmetric = false);
  // Extension: adopt arrays that already live on the device (allocated with
  // exec->alloc); the matrix takes ownership.  Used by the on-device Poisson
  // generator so a 512^3 matrix never exists on the host.
  struct AdoptDevice {};
  CSRMatrix(AdoptDevice, std::shared_ptr<DeviceExecutor> exec, int32_t num_rows,
            int32_t num_cols, int64_t num_non_zeros, int32_t* rowptr,
            int32_t* colind, T* values, T* diagonal, bool symmetric);
  ~CSRMatrix() override;
  CSRMatrix(const CSRMatrix&) = delete;
  CSRMatrix& operator=(const CSRMatrix&) = delete;

  size_t format_size() const override;
  void mult(T alpha, T* in, T beta, T* out) const override;
  // mult + fused dot partials (see CSRSpMV::run_dot); false if this block
  // cannot fuse (symmetric or empty) and nothing was launched.
  bool mult_dot(T alpha, T* in, T beta, T* out, double* dot_partials) const;

  // Mixed precision: build (once) the fp32 copy of the values; afterwards
  // use_mixed(true) makes mult / mult_dot stream that copy instead (x, y and
  // all arithmetic stay fp64).  General fp64 blocks only; a no-op elsewhere.
  void enable_mixed() const;
  void use_mixed(bool on) const { _mixed_on = on && _values32 != nullptr; }
  bool mixed_in_use() const { return _mixed_on; }

  // PLAN MEMORY.  The reference's CSRMatrix owns ONE copy of the matrix
  // (csr_matrix.cpp:34-70); a plan in a value-baking form (diagonal forms, the
  // sliced jagged form without long rows) holds a second one in its own
  // format.  release_csr() frees the device copies of colind and values when
  // the plan reports that it no longer reads them
  // (spmv_hip_csr_plan_owns_matrix / _release_matrix) and returns the bytes
  // given back (0: the plan still needs them).  Afterwards colind() / values()
  // are TOKENS the launches compare, not readable memory; enable_mixed() is a
  // no-op; mult() is unchanged, bit for bit.  The executor does it by itself
  // after spmv_init when the context option "release_csr" is set.
  size_t release_csr() const;
  bool csr_released() const { return _released; }

  const int32_t* rowptr() const { return _rowptr; }
  const int32_t* colind() const { return _colind; }
  const T* values() const { return _values; }
  const CSRSpMV<T>& op() const { return _op; }
  void tune(const char* key, int value) const { _op.tune(key, value); }
  int query(const char* key) const { return _op.query(key); }

private:
  int32_t* _rowptr = nullptr;
  int32_t* _colind = nullptr;
  T* _values = nullptr;
  mutable float* _values32 = nullptr; // enable_mixed()
  mutable bool _mixed_on = false;
  mutable bool _released = false; // _colind / _values freed (release_csr)
  CSRSpMV<T> _op;
};

} // namespace spmv
