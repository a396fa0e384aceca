// Symmetric-storage SpMV kernels for gfx950 (strictly-lower CSR + diagonal,
// spmv/csr_kernels.cpp:26-40).  See csr_plan.h for the file map.
#include "csr_plan.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <hipcub/hipcub.hpp>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

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

// ---------------------------------------------------------------------------
// Deterministic symmetric kernel ("transposed map").
//
// The reference walks the rows in ascending order and, for a stored entry
// (r, c, v) with c < r, adds alpha*v*x[r] to out[c] AFTER out[c] was finalised
// by its own row (csr_kernels.cpp:30-39).  Seen from row i, its result is
//   y_i = fl(alpha * sum_i + beta * y0_i)            sum_i = d_i x_i, then the
//                                                    row's entries left to right
//   y_i += fl(fl(alpha * v) * x_r)                   for every stored entry
//                                                    (r, i), ascending (r, j)
// which needs no atomics if row i can find the entries of its COLUMN.  Plan
// creation builds that map once (a stable sort of the entries by column):
//   t_ptr[i] .. t_ptr[i+1]   row i's column entries, in the reference's order
//   t_pos[e]                 position of the entry in `values`
//   t_row[e]                 its row r
// The kernel is the row-block kernel run twice over the same 256 rows: the
// products of the row's own entries, then the products of its column's
// entries, both parked in LDS and added left to right by the row's lane.
// No zero-fill pass, every y written once, results bit-identical to the
// oracle.  Costs 8 B per stored entry of plan memory.
// ---------------------------------------------------------------------------
template <typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void csr_symt_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const T* __restrict__ values,
    const T* __restrict__ diagonal, const int32_t* __restrict__ t_ptr,
    const int32_t* __restrict__ t_pos, const int32_t* __restrict__ t_row,
    T alpha, const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    int num_row_blocks)
{
  constexpr int TILE = 4 * kBlock;
  __shared__ T s_prod[TILE];
  __shared__ int32_t s_ptr[kRows + 1];
  __shared__ double s_red[kBlock / 64];
  const int t = threadIdx.x;
  double dot_acc = 0.0;
  for (int rb = blockIdx.x; rb < num_row_blocks; rb += gridDim.x) {
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);
    const int32_t r = r0 + t;
    T xi = T(0), acc = T(0), cacc = T(0);
    // phase 0: the row's own entries (v * x[c]);  phase 1: its column's
    // entries ((alpha v) * x[r'])
    for (int phase = 0; phase < 2; ++phase) {
      const int32_t* ptr = phase == 0 ? rowptr : t_ptr;
      __syncthreads(); // previous pass done with s_ptr / s_prod
      if (t <= nr)
        s_ptr[t] = ptr[r0 + t];
      if (t == 0 && nr == kRows)
        s_ptr[kRows] = ptr[r0 + kRows];
      __syncthreads();
      const int32_t a = s_ptr[0], b = s_ptr[nr];
      int32_t lo = 0, hi = 0;
      if (t < nr) {
        lo = s_ptr[t];
        hi = s_ptr[t + 1];
      }
      if (phase == 0 && t < nr) {
        xi = in[r];
        acc = diagonal[r] * xi; // csr_kernels.cpp:28
      }
      for (int64_t base = a; base < b; base += TILE) {
        if (base != a)
          __syncthreads(); // row owners finished reading the previous tile
#pragma unroll
        for (int c = 0; c < TILE / kBlock; ++c) {
          const int64_t j = base + c * kBlock + t;
          T p = T(0);
          if (j < b) {
            if (phase == 0) {
              p = values[j] * in[colind[j]]; // :34
            } else {
              const T av = alpha * values[t_pos[j]]; // :35, left to right
              p = av * in[t_row[j]];
            }
          }
          s_prod[c * kBlock + t] = p;
        }
        __syncthreads();
        const int32_t jlo = (int32_t)(max((int64_t)lo, base) - base);
        const int32_t jhi = (int32_t)(min((int64_t)hi, base + TILE) - base);
        for (int32_t j = jlo; j < jhi; ++j) {
          acc += s_prod[j];
          if (phase == 1)
            cacc += s_prod[j];
        }
      }
      if (phase == 0 && t < nr) {
        const T c = alpha * acc; // :39
        cacc = c;
        acc = c;
        if (beta != T(0))
          acc = c + beta * out[r];
      }
    }
    if (t < nr) {
      out[r] = acc;
      if constexpr (DOT) // in . (alpha A in): the row's share without beta y0
        dot_acc += (double)xi * (double)cacc;
    }
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

__global__ __launch_bounds__(kBlock) void symt_rowidx_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ rowidx,
    int32_t* __restrict__ pos, int32_t* __restrict__ count,
    int32_t* __restrict__ not_lower)
{
  // one lane per row: the row index and position of every entry, the column
  // histogram, and the check that the block is strictly lower triangular
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      rowidx[j] = (int32_t)i;
      pos[j] = j;
      const int32_t c = colind[j];
      if (c >= i || c < 0)
        atomicOr(not_lower, 1);
      else
        atomicAdd(count + c, 1);
    }
  }
}

__global__ __launch_bounds__(kBlock) void symt_gather_rows_kernel(
    int64_t n, const int32_t* __restrict__ rowidx,
    const int32_t* __restrict__ t_pos, int32_t* __restrict__ t_row)
{
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    t_row[e] = rowidx[t_pos[e]];
}

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
  if (pl->sdia && pl->sdia_val && !pl->sdia_general
      && pl->sdia_elem == (int)sizeof(T) && values == pl->sdia_values0
      && diagonal == pl->sdia_diag0) {
    if constexpr (sizeof(T) == 8)
      return spmv_sdia_run_f64(pl, st, alpha, in, beta, out, dot);
    else
      return spmv_sdia_run_f32(pl, st, alpha, in, beta, out);
  }
  if (pl->slat && aligned16(values)) {
    if constexpr (sizeof(T) == 8)
      return spmv_slat_run_f64(pl, st, rowptr, values, diagonal, alpha, in, beta,
                               out, dot);
    else
      return spmv_slat_run_f32(pl, st, rowptr, values, diagonal, alpha, in, beta,
                               out);
  }
  // no lattice structure (FEM matrices): the merged matrix in the sliced jagged form
  if (pl->sym_det && pl->sym_sj && pl->sj && pl->sjt && pl->sjt->sj_val
      && pl->sjt->sj_elem == (int)sizeof(T) && values == pl->sjt->sj_values0
      && diagonal == pl->sj_diag0 && aligned16(in) && pl->num_cols >= 2) {
    if constexpr (sizeof(T) == 8)
      return spmv_sjds_run_sym_f64(pl, st, diagonal, alpha, in, beta, out, dot);
    else
      return spmv_sjds_run_sym_f32(pl, st, diagonal, alpha, in, beta, out);
  }
  if (pl->sym_det && pl->t_ptr) {
    const int nrb = (n + kRows - 1) / kRows;
    int grid = pl->ctx->num_cus * pl->blocks_per_cu;
    if (grid > pl->ctx->dot_blocks)
      grid = pl->ctx->dot_blocks;
    if (grid > nrb)
      grid = nrb;
    if (dot.partials)
      hipLaunchKernelGGL((csr_symt_kernel<T, true>), dim3(grid), dim3(kBlock), 0,
                         st, n, rowptr, colind, values, diagonal, pl->t_ptr,
                         pl->t_pos, pl->t_row, alpha, in, beta, out, dot, nrb);
    else
      hipLaunchKernelGGL((csr_symt_kernel<T, false>), dim3(grid), dim3(kBlock),
                         0, st, n, rowptr, colind, values, diagonal, pl->t_ptr,
                         pl->t_pos, pl->t_row, alpha, in, beta, out, dot, nrb);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  // atomic kernels.  pre-pass: out *= beta (zero-fill when beta == 0)
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

void spmv_symt_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->t_ptr);
  (void)hipFree(pl->t_pos);
  (void)hipFree(pl->t_row);
  pl->t_ptr = pl->t_pos = pl->t_row = nullptr;
  pl->sym_det = 0;
}

// Build the transposed map of a symmetric plan (see csr_symt_kernel).  Left
// unbuilt -- the atomic kernels then run -- when the block is not strictly
// lower triangular (the reference's finalise-then-scatter order is then not a
// per-row gather) or when memory is short.
int spmv_symt_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  spmv_symt_free(pl);
  const int32_t n = pl->num_rows;
  const int64_t nnz = pl->nnz;
  if (n == 0 || nnz == 0)
    return SPMV_HIP_OK;
  hipStream_t st = pl->ctx->stream;
  int32_t *rowidx = nullptr, *pos = nullptr, *keys = nullptr, *flag = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0, scan_bytes = 0;
  int end_bit = 1;
  while (end_bit < 31 && ((int64_t)1 << end_bit) < n)
    ++end_bit;
  hipError_t e = hipMalloc(&pl->t_ptr, sizeof(int32_t) * ((size_t)n + 1));
  if (e == hipSuccess)
    e = hipMalloc(&pl->t_pos, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess)
    e = hipMalloc(&pl->t_row, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess)
    e = hipMalloc(&rowidx, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess)
    e = hipMalloc(&pos, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess)
    e = hipMalloc(&keys, sizeof(int32_t) * (size_t)nnz);
  if (e == hipSuccess)
    e = hipMalloc(&flag, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, colind, keys, pos,
                                           pl->t_pos, nnz, 0, end_bit, st);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, pl->t_ptr,
                                         pl->t_ptr, n + 1, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, (tmp_bytes > scan_bytes ? tmp_bytes : scan_bytes) + 16);
  if (e == hipSuccess)
    e = hipMemsetAsync(pl->t_ptr, 0, sizeof(int32_t) * ((size_t)n + 1), st);
  if (e == hipSuccess)
    e = hipMemsetAsync(flag, 0, sizeof(int32_t), st);
  int32_t not_lower = 0;
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, n, kBlock);
    hipLaunchKernelGGL(symt_rowidx_kernel, dim3(grid), dim3(kBlock), 0, st, n,
                       rowptr, colind, rowidx, pos, pl->t_ptr, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&not_lower, flag, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  if (e == hipSuccess && !not_lower) {
    // entries by column, ties in their original (row, position) order: radix
    // sort is stable
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, colind, keys, pos,
                                           pl->t_pos, nnz, 0, end_bit, st);
    if (e == hipSuccess)
      e = hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, pl->t_ptr,
                                           pl->t_ptr, n + 1, st);
    if (e == hipSuccess) {
      const int grid = spmv_grid_for(pl->ctx, nnz, kBlock);
      hipLaunchKernelGGL(symt_gather_rows_kernel, dim3(grid), dim3(kBlock), 0,
                         st, nnz, rowidx, pl->t_pos, pl->t_row);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
  }
  (void)hipFree(tmp);
  (void)hipFree(rowidx);
  (void)hipFree(pos);
  (void)hipFree(keys);
  (void)hipFree(flag);
  if (e != hipSuccess || not_lower) {
    spmv_symt_free(pl);
    return (e == hipSuccess || e == hipErrorOutOfMemory) ? SPMV_HIP_OK
                                                         : static_cast<int>(e);
  }
  pl->sym_det = 1;
  return SPMV_HIP_OK;
}
