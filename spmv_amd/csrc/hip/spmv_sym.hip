// Symmetric-storage SpMV kernels for gfx950 (strictly-lower CSR + diagonal,
// spmv/csr_kernels.cpp:26-40).  See csr_plan.h for the file map.
#include "csr_plan.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace
{

// ---------------------------------------------------------------------------
// Symmetric kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add(double* p, double v)
{
  unsafeAtomicAdd(p, v); // global_atomic_add_f64, no CAS loop
}
__device__ __forceinline__ void atomic_add(float* p, float v)
{
  unsafeAtomicAdd(p, v);
}

// diagonal-only block: out = alpha*d*x + beta*out
// (openmp/csr_kernels.openmp.cpp:222-225 covers the same nnz == 0 case)
template <typename T>
__global__ __launch_bounds__(kBlock) void diag_kernel(
    int64_t n, const T* __restrict__ diagonal, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    T y = alpha * (diagonal[i] * in[i]);
    if (beta != T(0))
      y = y + beta * out[i];
    out[i] = y;
  }
}

template <typename T, int CH, bool NT, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void csr_sym_rowblock_kernel(
    int32_t num_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const T* __restrict__ values,
    const T* __restrict__ diagonal, T alpha, const T* __restrict__ in,
    T* __restrict__ out, int num_row_blocks, DotOut dot)
{
  constexpr int V = VecOf<T>::V;
  constexpr int TILE = kBlock * CH * V;
  __shared__ double s_red[kBlock / 64];
  double dot_acc = 0.0;
  using val_t = typename VecOf<T>::val_t;
  using col_t = typename VecOf<T>::col_t;

  __shared__ T s_prod[TILE];
  __shared__ T s_val[TILE];
  __shared__ int32_t s_col[TILE];
  __shared__ int32_t s_rowptr[kRows + 1];

  const int t = threadIdx.x;
  for (int rb = blockIdx.x; rb < num_row_blocks; rb += gridDim.x) {
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);
    __syncthreads();
    if (t <= nr)
      s_rowptr[t] = rowptr[r0 + t];
    if (t == 0 && nr == kRows)
      s_rowptr[kRows] = rowptr[r0 + kRows];
    __syncthreads();

    const int32_t a = s_rowptr[0];
    const int32_t b = s_rowptr[nr];
    int32_t lo = 0, hi = 0;
    T xi = 0, sum = 0;
    if (t < nr) {
      lo = s_rowptr[t];
      hi = s_rowptr[t + 1];
      xi = in[r0 + t];
      sum = diagonal[r0 + t] * xi; // csr_kernels.cpp:28
    }

    const int64_t base0 = a & ~(V - 1);
    const int64_t jclamp = (int64_t)(b - 1) & ~(int64_t)(V - 1);
    for (int64_t base = base0; base < b; base += TILE) {
      if (base != base0)
        __syncthreads();
      if (ALIGNED && jclamp + V <= nnz) {
        // fast path: all matrix loads, then all gathers (see general kernel)
        val_t v[CH];
        col_t ci[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          const int64_t jl = j0 < jclamp ? j0 : jclamp;
          v[c] = stream_load<NT>(reinterpret_cast<const val_t*>(values + jl));
          ci[c] = stream_load<NT>(reinterpret_cast<const col_t*>(colind + jl));
        }
        T xg[CH][V];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int e = 0; e < V; ++e)
            xg[c][e] = in[ci[c][e]];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int slot = (c * kBlock + t) * V;
          const int64_t j0 = base + slot;
          val_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e)
            pv[e] = (j0 + e < b) ? v[c][e] * xg[c][e] : T(0);
          *reinterpret_cast<val_t*>(&s_prod[slot]) = pv;
          *reinterpret_cast<val_t*>(&s_val[slot]) = v[c];
          *reinterpret_cast<col_t*>(&s_col[slot]) = ci[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int slot = (c * kBlock + t) * V;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const int64_t j = base + slot + e;
            const bool live = j < b;
            const T vv = live ? values[j] : T(0);
            const int32_t cc = live ? colind[j] : 0;
            s_prod[slot + e] = live ? vv * in[cc] : T(0);
            s_val[slot + e] = vv;
            s_col[slot + e] = cc;
          }
        }
      }
      __syncthreads();
      const int32_t jlo = max((int64_t)lo, base) - base;
      const int32_t jhi = min((int64_t)hi, base + TILE) - base;
      for (int32_t k = jlo; k < jhi; ++k) {
        sum += s_prod[k];                                  // csr_kernels.cpp:34
        atomic_add(&out[s_col[k]], alpha * s_val[k] * xi); // :35
      }
    }
    if (t < nr) {
      atomic_add(&out[r0 + t], alpha * sum); // :39 (beta applied by pre-pass)
      // in . (alpha A in) with A = L + D + L^T: row i contributes
      // x_i (2 (d_i x_i + (L x)_i) - d_i x_i); the L^T terms are the mirror
      // images of the L terms, so no finished `out` is needed.
      dot_acc += (double)xi
                 * (double)(alpha * (sum + (sum - diagonal[r0 + t] * xi)));
    }
  }
  if (dot.partials) // uniform
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// Symmetric kernel with an LDS accumulation window.
//
// The plain symmetric kernel issues (entries per row + 1) global fp64 atomics
// per row and runs at the chip-wide atomic rate, not at the HBM rate.  Here a
// workgroup owns kSymRows consecutive rows and keeps a window of `out`
// covering rows [r0 - low, r0 + kSymRows) in LDS: the row owner's alpha*sum
// and every scattered term whose target falls inside the window are added
// with LDS atomics (ds_add_f64); only targets below the window go to global
// atomics.  At the end the window is added to `out` with ONE coalesced pass
// of global atomics (256 contiguous doubles per wave-instruction, the shape
// the atomic units run fastest at).  For a 7-point stencil with n <= low the
// global atomics drop from 4 to ~2.25 per row.  Works for any matrix: the
// window only decides where an add is staged.
// ---------------------------------------------------------------------------
// kSymRows = rows per workgroup (a multiple of 256, walked in sub-blocks)
template <typename T, int kSymRows, bool NT, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void csr_sym_window_kernel(
    int32_t num_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const T* __restrict__ values,
    const T* __restrict__ diagonal, T alpha, const T* __restrict__ in,
    T* __restrict__ out, int num_blocks, int low, DotOut dot)
{
  __shared__ double s_red[kBlock / 64];
  double dot_acc = 0.0;
  constexpr int V = VecOf<T>::V;
  constexpr int TILE = kBlock * V;
  using val_t = typename VecOf<T>::val_t;
  using col_t = typename VecOf<T>::col_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T* s_acc = reinterpret_cast<T*>(s_dyn); // low + kSymRows entries
  __shared__ T s_prod[TILE];
  __shared__ T s_val[TILE];
  __shared__ int32_t s_col[TILE];
  __shared__ int32_t s_rowptr[kRows + 1];

  const int t = threadIdx.x;
  const int win = low + kSymRows;
  for (int blk = blockIdx.x; blk < num_blocks; blk += gridDim.x) {
    const int64_t r_first = (int64_t)blk * kSymRows;
    const int64_t win_lo = r_first - low; // may be negative near row 0
    __syncthreads();                      // previous flush finished
    for (int j = t; j < win; j += kBlock)
      s_acc[j] = T(0);

    for (int sb = 0; sb < kSymRows / kRows; ++sb) {
      const int64_t r0 = r_first + (int64_t)sb * kRows;
      if (r0 >= num_rows)
        break; // uniform
      const int nr = (int)min((int64_t)kRows, (int64_t)num_rows - r0);
      __syncthreads(); // s_acc zeroed / previous sub-block done with LDS tiles
      if (t <= nr)
        s_rowptr[t] = rowptr[r0 + t];
      if (t == 0 && nr == kRows)
        s_rowptr[kRows] = rowptr[r0 + kRows];
      __syncthreads();

      const int32_t a = s_rowptr[0];
      const int32_t b = s_rowptr[nr];
      int32_t lo = 0, hi = 0;
      T xi = 0, sum = 0;
      if (t < nr) {
        lo = s_rowptr[t];
        hi = s_rowptr[t + 1];
        xi = in[r0 + t];
        sum = diagonal[r0 + t] * xi; // csr_kernels.cpp:28
      }
      const int64_t base0 = a & ~(V - 1);
      const int64_t jclamp = (int64_t)(b - 1) & ~(int64_t)(V - 1);
      for (int64_t base = base0; base < b; base += TILE) {
        if (base != base0)
          __syncthreads();
        const int slot = t * V;
        const int64_t j0 = base + slot;
        if (ALIGNED && jclamp + V <= nnz) {
          const int64_t jl = j0 < jclamp ? j0 : jclamp;
          val_t v = stream_load<NT>(reinterpret_cast<const val_t*>(values + jl));
          col_t ci = stream_load<NT>(reinterpret_cast<const col_t*>(colind + jl));
          T xg[V];
#pragma unroll
          for (int e = 0; e < V; ++e)
            xg[e] = in[ci[e]];
          val_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e)
            pv[e] = (j0 + e < b) ? v[e] * xg[e] : T(0);
          *reinterpret_cast<val_t*>(&s_prod[slot]) = pv;
          *reinterpret_cast<val_t*>(&s_val[slot]) = v;
          *reinterpret_cast<col_t*>(&s_col[slot]) = ci;
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const int64_t j = j0 + e;
            const bool live = j < b;
            const T vv = live ? values[j] : T(0);
            const int32_t cc = live ? colind[j] : 0;
            s_prod[slot + e] = live ? vv * in[cc] : T(0);
            s_val[slot + e] = vv;
            s_col[slot + e] = cc;
          }
        }
        __syncthreads();
        const int32_t jlo = max((int64_t)lo, base) - base;
        const int32_t jhi = min((int64_t)hi, base + TILE) - base;
        for (int32_t k = jlo; k < jhi; ++k) {
          sum += s_prod[k];                     // csr_kernels.cpp:34
          const T term = alpha * s_val[k] * xi; // :35
          const int64_t c = s_col[k];
          if (c >= win_lo && c < win_lo + win)
            atomic_add(&s_acc[c - win_lo], term); // ds_add
          else
            atomic_add(&out[c], term);
        }
      }
      if (t < nr) { // :39, beta already applied by the pre-pass
        atomic_add(&s_acc[r0 + t - win_lo], alpha * sum);
        // this row's share of in . (alpha A in), see csr_sym_rowblock_kernel
        dot_acc += (double)xi
                   * (double)(alpha * (sum + (sum - diagonal[r0 + t] * xi)));
      }
    }
    __syncthreads();
    // flush: one coalesced pass of global atomics over the window
    for (int j = t; j < win; j += kBlock) {
      const int64_t g = win_lo + j;
      const T v = s_acc[j];
      if (g >= 0 && g < num_rows && v != T(0))
        atomic_add(&out[g], v);
    }
  }
  if (dot.partials) // uniform
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

} // namespace

namespace
{

template <typename T>
int run_symmetric(const spmv_hip_csr_plan* pl, hipStream_t st,
                  const int32_t* rowptr, const int32_t* colind,
                  const T* values, const T* diagonal, T alpha, const T* in,
                  T beta, T* out, DotOut dot = DotOut())
{
  if (diagonal == nullptr)
    return SPMV_HIP_EINVAL;
  if (dot.partials && pl->nnz == 0)
    return SPMV_HIP_ENOTSUP; // diagonal-only block: caller uses a plain dot
  const int n = pl->num_rows;
  if (pl->nnz == 0) {
    const int grid = spmv_grid_for(pl->ctx, n, kBlock);
    hipLaunchKernelGGL((diag_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                       (int64_t)n, diagonal, alpha, in, beta, out);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  // pre-pass: out *= beta (zero-fill when beta == 0)
  if (beta != T(1)) {
    if (beta == T(0)) {
      SPMV_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(T) * (size_t)n, st));
    } else {
      const int grid = spmv_grid_for(pl->ctx, n, kBlock);
      hipLaunchKernelGGL((scale_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                         (int64_t)n, beta, out);
      SPMV_CHECK_LAUNCH();
    }
  }
  const bool al = aligned16(values) && aligned16(colind);
  if (pl->sym_window > 0) {
    const int srows = pl->sym_rows;
    const int nblk = (n + srows - 1) / srows;
    const size_t lds = sizeof(T) * (size_t)(pl->sym_window + srows);
    // LDS per workgroup: window + ~11.5 KB of tiles; as many workgroups per
    // CU as the 160 KB allow (<= 8)
    int per_cu = (int)((160 * 1024) / (lds + 11776));
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    if (per_cu > pl->blocks_per_cu)
      per_cu = pl->blocks_per_cu;
    int grid = pl->ctx->num_cus * per_cu;
    if (grid > nblk)
      grid = nblk;
#define SPMV_SYMW(R, NT, AL)                                                   \
  hipLaunchKernelGGL((csr_sym_window_kernel<T, R, NT, AL>), dim3(grid),        \
                     dim3(kBlock), lds, st, n, pl->nnz, rowptr, colind,        \
                     values, diagonal, alpha, in, out, nblk, pl->sym_window,   \
                     dot)
#define SPMV_SYMW_R(NT, AL)                                                    \
  do {                                                                         \
    if (srows == 512)                                                          \
      SPMV_SYMW(512, NT, AL);                                                  \
    else if (srows == 2048)                                                    \
      SPMV_SYMW(2048, NT, AL);                                                 \
    else                                                                       \
      SPMV_SYMW(1024, NT, AL);                                                 \
  } while (0)
    if (!al)
      SPMV_SYMW_R(false, false);
    else if (pl->nontemporal)
      SPMV_SYMW_R(true, true);
    else
      SPMV_SYMW_R(false, true);
#undef SPMV_SYMW_R
#undef SPMV_SYMW
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  const int nrb = (n + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * pl->blocks_per_cu;
  if (grid > nrb)
    grid = nrb;
#define SPMV_SYM(CH, NT, AL)                                                   \
  hipLaunchKernelGGL((csr_sym_rowblock_kernel<T, CH, NT, AL>), dim3(grid),     \
                     dim3(kBlock), 0, st, n, pl->nnz, rowptr, colind, values,  \
                     diagonal, alpha, in, out, nrb, dot)
  if (!al)
    SPMV_SYM(1, false, false);
  else if (pl->nontemporal)
    SPMV_SYM(1, true, true);
  else
    SPMV_SYM(1, false, true);
#undef SPMV_SYM
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

int spmv_run_symmetric_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                           const int32_t* rowptr, const int32_t* colind,
                           const double* values, const double* diagonal,
                           double alpha, const double* in, double beta,
                           double* out, DotOut dot)
{
  return run_symmetric<double>(pl, st, rowptr, colind, values, diagonal, alpha,
                               in, beta, out, dot);
}

int spmv_run_symmetric_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                           const int32_t* rowptr, const int32_t* colind,
                           const float* values, const float* diagonal,
                           float alpha, const float* in, float beta,
                           float* out)
{
  return run_symmetric<float>(pl, st, rowptr, colind, values, diagonal, alpha,
                              in, beta, out, DotOut());
}
