// CSRSpMV<T> (HipExecutor overloads) and CSRMatrix<T>: see csr.h.
#include "csr.h"

#include <stdexcept>
#include <type_traits>

#include "spmv_hip.h"

namespace spmv
{

// ---------------------------------------------------------------------------
// CSRSpMV<T> -- HipExecutor overloads (pattern: csr_kernels.h:71-78)
// ---------------------------------------------------------------------------
template <typename T>
void CSRSpMV<T>::init(int32_t num_rows, int32_t num_cols,
                      int64_t num_non_zeros, const int32_t* rowptr,
                      const int32_t* colind, const T* /*values*/,
                      bool symmetric, const HipExecutor& exec)
{
  _symmetric = symmetric;
  spmv_hip_csr_plan* plan = nullptr;
  throw_on_error(spmv_hip_csr_plan_create(exec.context(), num_rows, num_cols,
                                          num_non_zeros, rowptr, colind,
                                          symmetric ? 1 : 0, SPMV_HIP_ALGO_AUTO,
                                          &plan),
                 "spmv_hip_csr_plan_create");
  _aux_data = plan;
}

template <typename T>
bool CSRSpMV<T>::bake_values(const T* values, const T* diagonal,
                             const HipExecutor& exec) const
{
  if (!plan() || !values || (_symmetric && !diagonal))
    return false;
  int rc;
  if constexpr (std::is_same<T, double>::value)
    rc = spmv_hip_csr_plan_bake_values_f64(exec.context(), plan(), values,
                                           diagonal, nullptr);
  else
    rc = spmv_hip_csr_plan_bake_values_f32(exec.context(), plan(), values,
                                           diagonal, nullptr);
  if (rc == SPMV_HIP_ENOTSUP)
    return false;
  throw_on_error(rc, "spmv_hip_csr_plan_bake_values");
  return true;
}

template <typename T>
void CSRSpMV<T>::run(int32_t num_rows, int32_t num_cols, int64_t num_non_zeros,
                     const int32_t* rowptr, const int32_t* colind,
                     const T* values, const T* diagonal, T alpha, T* in, T beta,
                     T* out, const HipExecutor& exec) const
{
  if constexpr (std::is_same<T, double>::value)
    throw_on_error(spmv_hip_csr_spmv_f64(exec.context(), plan(), num_rows,
                                         num_cols, num_non_zeros, rowptr,
                                         colind, values, diagonal, alpha, in,
                                         beta, out, nullptr, nullptr),
                   "spmv_hip_csr_spmv_f64");
  else
    throw_on_error(spmv_hip_csr_spmv_f32(exec.context(), plan(), num_rows,
                                         num_cols, num_non_zeros, rowptr,
                                         colind, values, diagonal, alpha, in,
                                         beta, out, nullptr),
                   "spmv_hip_csr_spmv_f32");
}

template <typename T>
void CSRSpMV<T>::run_dot(int32_t num_rows, int32_t num_cols,
                         int64_t num_non_zeros, const int32_t* rowptr,
                         const int32_t* colind, const T* values, T alpha, T* in,
                         T beta, T* out, double* dot_partials,
                         const HipExecutor& exec) const
{
  if constexpr (std::is_same<T, double>::value) {
    throw_on_error(spmv_hip_csr_spmv_f64(exec.context(), plan(), num_rows,
                                         num_cols, num_non_zeros, rowptr,
                                         colind, values, nullptr, alpha, in,
                                         beta, out, dot_partials, nullptr),
                   "spmv_hip_csr_spmv_f64");
  } else {
    throw std::runtime_error("CSRSpMV<float>::run_dot is not available");
  }
}

template <typename T>
void CSRSpMV<T>::run_dot_sym(int32_t num_rows, int32_t num_cols,
                             int64_t num_non_zeros, const int32_t* rowptr,
                             const int32_t* colind, const T* values,
                             const T* diagonal, T alpha, T* in, T beta, T* out,
                             double* dot_partials,
                             const HipExecutor& exec) const
{
  if constexpr (std::is_same<T, double>::value)
    throw_on_error(spmv_hip_csr_spmv_f64(exec.context(), plan(), num_rows,
                                         num_cols, num_non_zeros, rowptr,
                                         colind, values, diagonal, alpha, in,
                                         beta, out, dot_partials, nullptr),
                   "spmv_hip_csr_spmv_f64");
  else
    throw std::runtime_error("CSRSpMV<float>::run_dot_sym is not available");
}

template <typename T>
void CSRSpMV<T>::tune(const char* key, int value) const
{
  if (plan())
    throw_on_error(spmv_hip_csr_plan_set(plan(), key, value),
                   "spmv_hip_csr_plan_set");
}

template <typename T>
void CSRSpMV<T>::run_mixed(int32_t num_rows, int32_t num_cols,
                           int64_t num_non_zeros, const int32_t* rowptr,
                           const int32_t* colind, const float* values32, T alpha,
                           T* in, T beta, T* out, double* dot_partials,
                           const HipExecutor& exec) const
{
  if constexpr (std::is_same<T, double>::value)
    throw_on_error(spmv_hip_csr_spmv_f32f64(exec.context(), plan(), num_rows,
                                            num_cols, num_non_zeros, rowptr,
                                            colind, values32, alpha, in, beta,
                                            out, dot_partials, nullptr),
                   "spmv_hip_csr_spmv_f32f64");
  else
    throw std::runtime_error("CSRSpMV<float>::run_mixed is not available");
}

template <typename T>
int CSRSpMV<T>::query(const char* key) const
{
  int v = 0;
  if (plan())
    throw_on_error(spmv_hip_csr_plan_get(plan(), key, &v),
                   "spmv_hip_csr_plan_get");
  return v;
}

template <typename T>
void CSRSpMV<T>::finalize(const HipExecutor&) const
{
  spmv_hip_csr_plan_destroy(plan());
  _aux_data = nullptr;
}

// ---------------------------------------------------------------------------
// CSRMatrix<T>
// ---------------------------------------------------------------------------
template <typename T>
CSRMatrix<T>::CSRMatrix(std::shared_ptr<DeviceExecutor> exec,
                        const CsrHost<T>* mat, const std::vector<T>* diagonal,
                        bool symmetric)
    : CSRMatrix(exec, mat->rows, mat->cols, mat->non_zeros(),
                mat->rowptr.data(), mat->colind.data(), mat->values.data(),
                diagonal ? diagonal->data() : nullptr, symmetric)
{
}

template <typename T>
CSRMatrix<T>::CSRMatrix(std::shared_ptr<DeviceExecutor> exec, int32_t num_rows,
                        int32_t num_cols, int64_t num_non_zeros,
                        const int32_t* rowptr, const int32_t* colind,
                        const T* values, const T* diagonal, bool symmetric)
{
  this->_exec = exec;
  this->_num_rows = num_rows;
  this->_num_cols = num_cols;
  this->_num_non_zeros = num_non_zeros;
  this->_symmetric = symmetric;
  const DeviceExecutor& host = exec->get_host();
  if (num_non_zeros > 0) { // empty blocks own no arrays (csr_matrix.cpp:34)
    _rowptr = exec->alloc<int32_t>(num_rows + 1);
    _colind = exec->alloc<int32_t>(num_non_zeros);
    _values = exec->alloc<T>(num_non_zeros);
    exec->copy_from<int32_t>(_rowptr, host, rowptr, num_rows + 1);
    exec->copy_from<int32_t>(_colind, host, colind, num_non_zeros);
    exec->copy_from<T>(_values, host, values, num_non_zeros);
  }
  if (diagonal != nullptr) {
    this->_diagonal = exec->alloc<T>(num_rows);
    exec->copy_from<T>(this->_diagonal, host, diagonal, num_rows);
  }
  exec->spmv_init(_op, *this); // per-matrix analysis hook (csr_matrix.cpp:58)
}

template <typename T>
CSRMatrix<T>::CSRMatrix(AdoptDevice, std::shared_ptr<DeviceExecutor> exec,
                        int32_t num_rows, int32_t num_cols,
                        int64_t num_non_zeros, int32_t* rowptr, int32_t* colind,
                        T* values, T* diagonal, bool symmetric)
{
  this->_exec = exec;
  this->_num_rows = num_rows;
  this->_num_cols = num_cols;
  this->_num_non_zeros = num_non_zeros;
  this->_symmetric = symmetric;
  if (num_non_zeros > 0) {
    _rowptr = rowptr;
    _colind = colind;
    _values = values;
  } else { // keep the "empty block owns nothing" invariant
    exec->free(rowptr);
    exec->free(colind);
    exec->free(values);
  }
  this->_diagonal = diagonal;
  exec->spmv_init(_op, *this);
}

template <typename T>
CSRMatrix<T>::~CSRMatrix()
{
  // never throw from a destructor: swallow teardown errors
  try {
    this->_exec->spmv_finalize(_op);
    this->_exec->free(_rowptr);
    if (!_released) { // (else tokens: release_csr freed them)
      this->_exec->free(_colind);
      this->_exec->free(_values);
    }
    this->_exec->free(_values32);
    this->_exec->free(this->_diagonal);
  } catch (...) {
  }
}

template <typename T>
size_t CSRMatrix<T>::release_csr() const
{
  if (_released || this->_num_non_zeros == 0 || _values32 || !_op.plan())
    return 0;
  auto* hip = dynamic_cast<const HipExecutor*>(this->_exec.get());
  if (!hip)
    return 0;
  int mask = 0;
  throw_on_error(spmv_hip_csr_plan_owns_matrix(_op.plan(), &mask),
                 "spmv_hip_csr_plan_owns_matrix");
  if (mask != 3) // both or nothing: one flag, one rule
    return 0;
  throw_on_error(spmv_hip_csr_plan_release_matrix(_op.plan(), mask),
                 "spmv_hip_csr_plan_release_matrix");
  hip->synchronize(); // no launch of an earlier form still reads them
  this->_exec->free(_colind);
  this->_exec->free(_values);
  _released = true; // the pointers stay as tokens
  return (size_t)this->_num_non_zeros * (sizeof(int32_t) + sizeof(T));
}

template <typename T>
void CSRMatrix<T>::enable_mixed() const
{
  if (_released)
    return; // the fp64 values it would convert were given back
  if (_values32 || this->_symmetric || this->_num_non_zeros == 0
      || !std::is_same<T, double>::value)
    return;
  auto* hip = dynamic_cast<const HipExecutor*>(this->_exec.get());
  if (!hip)
    return;
  if constexpr (std::is_same<T, double>::value) {
    float* v32 = this->_exec->template alloc<float>(this->_num_non_zeros);
    try {
      throw_on_error(spmv_hip_convert_f64_f32(hip->context(),
                                              this->_num_non_zeros, _values, v32,
                                              nullptr),
                     "spmv_hip_convert_f64_f32");
      // a block that runs the diagonal form keeps an fp32 copy by offset too
      const int rc = spmv_hip_csr_plan_bake_values_f32f64(
          hip->context(), _op.plan(), v32, nullptr);
      if (rc != SPMV_HIP_ENOTSUP)
        throw_on_error(rc, "spmv_hip_csr_plan_bake_values_f32f64");
    } catch (...) { // nothing half-enabled, nothing leaked
      (void)spmv_hip_csr_plan_bake_values_f32f64(hip->context(), _op.plan(),
                                                 nullptr, nullptr);
      this->_exec->free(v32);
      throw;
    }
    _values32 = v32;
  }
}

template <typename T>
size_t CSRMatrix<T>::format_size() const // csr_matrix.cpp:72-78
{
  return (this->_num_rows + 1) * sizeof(int32_t)
         + this->_num_non_zeros * (sizeof(int32_t) + sizeof(T));
}

template <typename T>
void CSRMatrix<T>::mult(T alpha, T* in, T beta, T* out) const
{
  if (_mixed_on && this->_num_non_zeros > 0) {
    auto* hip = dynamic_cast<const HipExecutor*>(this->_exec.get());
    _op.run_mixed(this->_num_rows, this->_num_cols, this->_num_non_zeros,
                  _rowptr, _colind, _values32, alpha, in, beta, out, nullptr,
                  *hip);
    return;
  }
  if (this->_num_non_zeros > 0 || this->_diagonal != nullptr) // :85
    this->_exec->spmv_run(_op, *this, alpha, in, beta, out);
}

template <typename T>
bool CSRMatrix<T>::mult_dot(T alpha, T* in, T beta, T* out,
                            double* dot_partials) const
{
  if (this->_num_non_zeros == 0)
    return false;
  auto* hip = dynamic_cast<const HipExecutor*>(this->_exec.get());
  if (!hip)
    return false;
  if (this->_symmetric) {
    // the symmetric kernel produces its share from the mirror identity
    if (!std::is_same<T, double>::value)
      return false;
    _op.run_dot_sym(this->_num_rows, this->_num_cols, this->_num_non_zeros,
                    _rowptr, _colind, _values, this->_diagonal, alpha, in, beta,
                    out, dot_partials, *hip);
    return true;
  }
  if (_mixed_on)
    _op.run_mixed(this->_num_rows, this->_num_cols, this->_num_non_zeros,
                  _rowptr, _colind, _values32, alpha, in, beta, out,
                  dot_partials, *hip);
  else
    _op.run_dot(this->_num_rows, this->_num_cols, this->_num_non_zeros, _rowptr,
                _colind, _values, alpha, in, beta, out, dot_partials, *hip);
  return true;
}

template class CSRSpMV<float>;
template class CSRSpMV<double>;
template class CSRMatrix<float>;
template class CSRMatrix<double>;

} // namespace spmv
