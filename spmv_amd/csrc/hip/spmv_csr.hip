// CSR and symmetric-CSR SpMV kernels for gfx950 (MI355X).
//
// Stands behind CSRSpMV<T>::init/run/finalize (spmv/csr_kernels.h:26-78);
// arithmetic follows spmv/csr_kernels.cpp:20-52.  Built with
// -ffp-contract=off so a*b+c is two roundings, as in the reference build.
//
// HBM-bound gather (0.13-0.17 flop/B): MFMA is deliberately unused.
//
// ROWBLOCK kernel (the hot one)
//   A workgroup of 256 threads owns ROWS consecutive rows.  The nnz span of
//   those rows is contiguous in CSR, so the workgroup streams it in tiles of
//   TILE entries with 16-byte-per-lane coalesced loads (values) and the
//   matching 8/16-byte colind loads, multiplies by the gathered x and parks
//   the products in LDS.  Then each thread owns one row and adds its
//   products left to right out of LDS -- the same order as the reference's
//   scalar loop, so the result is bit-identical to csr_kernels.cpp:41-51.
//   The row pointer is read once, coalesced, into LDS.  Launches are
//   grid-stride (<= 8 workgroups per CU) so the optional fused dot product
//   sum_i in[i]*out[i] leaves a fixed, small number of partials.
//
// VECTOR kernel: LPR lanes per row with a __shfl_down segmented reduction,
//   for matrices with long rows.
// SCALAR kernel: one lane per row, reference loop verbatim.
//
// This file: the kernels, their launches and the C entry points of the product.
// The plan-time builders of the LX / XW forms, the row list and the plane-walk
// tables live in spmv_csr_forms.hip, the plan API (create / bake / set / get)
// in spmv_csr_plan.hip; the symmetric-storage kernels in spmv_sym.hip, the
// lattice form in spmv_lat.hip; csr_plan.h holds what the files share.
#include "csr_plan.h"

#include <cstring>
#include <new>

namespace
{

// ---------------------------------------------------------------------------
// ROWBLOCK general kernel
//   CH      = 16-byte value loads per lane per tile (tile = 256*CH*V entries)
//   NT      = non-temporal loads for the read-once matrix stream
//   ALIGNED = values 16-B / colind 8|16-B aligned => wide loads
//   XCD     = group consecutive row blocks per XCD (see xcd_group below)
// ---------------------------------------------------------------------------
//   TV      = type of `values`; T = type of x, y and of the arithmetic
//             (TV = float, T = double: the mixed-precision SpMV)
template <typename TV, typename T, int CH, bool NT, bool ALIGNED, bool DOT,
          bool XCD>
__global__ __launch_bounds__(kBlock) void csr_rowblock_kernel(
    int32_t num_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const TV* __restrict__ values, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot, RowBlockOrder ord)
{
  constexpr int V = VecOf<TV>::V;
  constexpr int TILE = kBlock * CH * V;
  using val_t = typename VecOf<TV>::val_t;
  using col_t = typename VecOf<TV>::col_t;
  typedef T prod_t __attribute__((ext_vector_type(V)));

  __shared__ __attribute__((aligned(32))) T s_prod[TILE];
  __shared__ int32_t s_rowptr[kRows + 1];
  __shared__ double s_red[kBlock / 64];

  const int t = threadIdx.x;
  double dot_acc = 0.0;

  // XCD = false: plain order (small problems, tests)
  const int num_slots = XCD ? order_slots(ord) : ord.num_row_blocks;
  for (int it = blockIdx.x; it < num_slots; it += gridDim.x) {
    const int rb = XCD ? order_row_block(ord, it) : it;
    if (rb < 0)
      continue; // uniform per workgroup
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);

    __syncthreads(); // previous iteration done with s_rowptr / s_prod
    if (t <= nr)
      s_rowptr[t] = rowptr[r0 + t];
    if (t == 0 && nr == kRows)
      s_rowptr[kRows] = rowptr[r0 + kRows];
    __syncthreads();

    const int32_t a = s_rowptr[0];
    const int32_t b = s_rowptr[nr];
    int32_t lo = 0, hi = 0;
    if (t < nr) {
      lo = s_rowptr[t];
      hi = s_rowptr[t + 1];
    }
    T sum = 0;
    // the row's own x for the fused dot, fetched ahead of its use
    T x_own = T(0);
    if constexpr (DOT)
      if (t < nr)
        x_own = in[r0 + t];

    // tiles start V-aligned so the wide loads are naturally aligned
    const int64_t base0 = a & ~(V - 1);
    // last V-aligned slot of this row block's span: lanes past the span
    // re-read it (one cached line) instead of streaming the next block's data
    const int64_t jclamp = (int64_t)(b - 1) & ~(int64_t)(V - 1);
    for (int64_t base = base0; base < b; base += TILE) {
      if (base != base0)
        __syncthreads(); // row owners finished reading the previous tile
      // Fast path: every wide load of the tile is inside the arrays.  All
      // matrix loads are issued first, then all gathers, then the products,
      // so one lane keeps CH*(1+V) loads in flight.
      if (ALIGNED && jclamp + V <= nnz) {
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
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          prod_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e)
            pv[e] = (j0 + e < b) ? (T)v[c][e] * xg[c][e] : T(0);
          *reinterpret_cast<prod_t*>(&s_prod[(c * kBlock + t) * V]) = pv;
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          prod_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const int64_t j = j0 + e;
            pv[e] = (j < b) ? (T)values[j] * in[colind[j]] : T(0);
          }
          *reinterpret_cast<prod_t*>(&s_prod[(c * kBlock + t) * V]) = pv;
        }
      }
      // entries in [base, a) belong to earlier rows; no row of this block
      // reads them, so their (valid) products are simply ignored.
      __syncthreads();
      const int32_t jlo = max((int64_t)lo, base) - base;
      const int32_t jhi = min((int64_t)hi, base + TILE) - base;
      int32_t j = jlo;
      // four LDS reads in flight, adds strictly left to right
      for (; j + 4 <= jhi; j += 4) {
        const T p0 = s_prod[j], p1 = s_prod[j + 1], p2 = s_prod[j + 2],
                p3 = s_prod[j + 3];
        sum += p0;
        sum += p1;
        sum += p2;
        sum += p3;
      }
      for (; j < jhi; ++j)
        sum += s_prod[j];
    }

    if (t < nr) {
      const int32_t r = r0 + t;
      const T c = alpha * sum;
      T y = c;
      if (beta != T(0))
        y = c + beta * out[r];
      if (ord.nt_store) // y is not read again before it leaves the caches
        __builtin_nontemporal_store(y, &out[r]);
      else
        out[r] = y;
      if constexpr (DOT) // this block's own share: in . (alpha A in)
        dot_acc += (double)x_own * (double)c;
    }
  }

  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// ROWBLOCK with LDS-staged x windows and 16-bit local column indices ("LX").
//
// The gather `in[colind[j]]` is what separates the row-block kernel from a
// pure stream: per-lane 8-byte requests through the texture path, and the same
// x lines pulled through the L2s several times (tools/membench/spmv_probe:
// 6.3 TB/s without the gather, 5.4 with it).  For matrices whose row blocks
// reference few contiguous column ranges -- stencils, banded and well-ordered
// FEM matrices -- plan creation rewrites the column indices of every row block
// as 16-bit offsets into a small set of column WINDOWS (lx_build_kernel).
// The kernel then
//   1. copies the windows of x into LDS with coalesced 16-byte loads,
//   2. streams values (16 B) and local indices (2 B per entry instead of 4),
//   3. takes x from LDS; products, row sums and y exactly as csr_rowblock_kernel
//      (same products, same left-to-right order => bit-identical results).
// Row blocks whose columns do not fit the LDS budget keep the global gather
// (nwin < 0); the decision is per row block.
//   lx.tab[rb*kLxRec + 0]          number of windows, or -1 = direct
//   lx.tab[rb*kLxRec + 1 + k]      first column of window k (even)
//   lx.tab[rb*kLxRec + 17 + k]     offset of window k in the staged buffer
//                                  (even; entry nwin = staged length)
//   lx.lidx[j]             offset of column colind[j] in the staged buffer
// One record per row block, fetched unconditionally next to the row pointer:
// nothing in the block's prologue depends on an earlier load.
// ---------------------------------------------------------------------------

struct LxView {
  const uint16_t* lidx;
  const int32_t* tab;
};

template <typename T, bool NT, bool DOT, int CH>
__global__ __launch_bounds__(kBlock) void csr_rowblock_lx_kernel(
    int32_t num_rows, int32_t num_cols, int64_t nnz,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const T* __restrict__ values, LxView lx, T alpha, const T* __restrict__ in,
    T beta, T* __restrict__ out, DotOut dot, RowBlockOrder ord)
{
  constexpr int V = VecOf<T>::V;
  constexpr int TILE = kBlock * CH * V; // entries per barrier pair
  using val_t = typename VecOf<T>::val_t;
  using col_t = typename VecOf<T>::col_t;
  typedef T pair_t __attribute__((ext_vector_type(2)));
  typedef unsigned short lidx_t __attribute__((ext_vector_type(V)));

  __shared__ __attribute__((aligned(16))) T s_x[kLxCap];
  __shared__ __attribute__((aligned(16))) T s_prod[TILE];
  __shared__ int32_t s_rowptr[kRows + 1];
  __shared__ int32_t s_tab[kLxRec];
  __shared__ double s_red[kBlock / 64];
  const int32_t* s_wstart = s_tab + 1;
  const int32_t* s_woff = s_tab + 1 + kLxMaxWin;

  const int t = threadIdx.x;
  double dot_acc = 0.0;
  const int num_slots = order_slots(ord);
  // with an order table the entry of the NEXT slot is requested a whole row
  // block ahead: a look-up that has to be waited for stalls the workgroup at
  // the top of every block
  int rb_raw = order_slot_raw(ord, blockIdx.x, num_slots);
  for (int it = blockIdx.x; it < num_slots; it += gridDim.x) {
    const int rb = order_slot_decode(ord, rb_raw);
    rb_raw = order_slot_raw(ord, it + gridDim.x, num_slots);
    if (rb < 0)
      continue; // uniform per workgroup
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);

    __syncthreads(); // previous iteration done with every LDS array
    if (t <= nr)
      s_rowptr[t] = rowptr[r0 + t];
    if (t == 0 && nr == kRows)
      s_rowptr[kRows] = rowptr[r0 + kRows];
    if (t >= kBlock - kLxRec) // the last wave fetches the block's record
      s_tab[t - (kBlock - kLxRec)]
          = lx.tab[(int64_t)rb * kLxRec + (t - (kBlock - kLxRec))];
    __syncthreads();
    const int K = s_tab[0]; // uniform

    const int32_t a = s_rowptr[0];
    const int32_t b = s_rowptr[nr];
    int32_t lo = 0, hi = 0;
    if (t < nr) {
      lo = s_rowptr[t];
      hi = s_rowptr[t + 1];
    }
    // the row's own x for the fused dot: fetched now, used after the row sums
    // (at the end of the block its latency would be exposed)
    T x_own = T(0);
    if constexpr (DOT)
      if (t < nr)
        x_own = in[r0 + t];
    // Stage the x windows: pairs of elements, coalesced (in is 2-element
    // aligned, checked at launch; window starts and offsets are even, so a
    // pair never straddles two windows).  The staged buffer is walked as ONE
    // flat range and all of a lane's loads are issued before the first LDS
    // write: a loop over the windows made every window wait for its own load
    // (five dependent L2 round trips per row block for a 7-point stencil).
    {
      static_assert(kLxCap <= 3 * 2 * kBlock, "three pairs per lane cover it");
      const int staged = K > 0 ? s_woff[K] : 0;
      pair_t xv[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int f = 2 * t + 2 * kBlock * m;
        xv[m][0] = xv[m][1] = T(0);
        if (f < staged) {
          int k = 0;
          while (f >= s_woff[k + 1]) // K <= 16 windows
            ++k;
          const int32_t c = s_wstart[k] + (f - s_woff[k]);
          if (c + 1 < num_cols)
            xv[m] = *reinterpret_cast<const pair_t*>(in + c);
          else if (c < num_cols) // the window was rounded up past the end
            xv[m][0] = in[c];
        }
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int f = 2 * t + 2 * kBlock * m;
        if (f < staged)
          *reinterpret_cast<pair_t*>(&s_x[f]) = xv[m];
      }
    }
    T sum = 0;
    const int64_t base0 = a & ~(V - 1);
    const int64_t jclamp = (int64_t)(b - 1) & ~(int64_t)(V - 1);
    bool staged_visible = K <= 0;
    for (int64_t base = base0; base < b; base += TILE) {
      if (base != base0)
        __syncthreads(); // row owners finished reading the previous tile
      if (jclamp + V <= nnz && K >= 0) {
        // staged block: all matrix loads of the tile first, then the products
        val_t v[CH];
        lidx_t li[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          const int64_t jl = j0 < jclamp ? j0 : jclamp;
          v[c] = stream_load<NT>(reinterpret_cast<const val_t*>(values + jl));
          li[c] = stream_load<NT>(reinterpret_cast<const lidx_t*>(lx.lidx + jl));
        }
        if (!staged_visible) {
          __syncthreads(); // s_x complete (matrix loads already in flight)
          staged_visible = true;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          val_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e)
            pv[e] = (j0 + e < b) ? v[c][e] * s_x[li[c][e]] : T(0);
          *reinterpret_cast<val_t*>(&s_prod[(c * kBlock + t) * V]) = pv;
        }
      } else if (jclamp + V <= nnz) { // direct block: global gather
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          const int64_t jl = j0 < jclamp ? j0 : jclamp;
          const val_t v
              = stream_load<NT>(reinterpret_cast<const val_t*>(values + jl));
          const col_t ci
              = stream_load<NT>(reinterpret_cast<const col_t*>(colind + jl));
          T xg[V];
#pragma unroll
          for (int e = 0; e < V; ++e)
            xg[e] = in[ci[e]];
          val_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e)
            pv[e] = (j0 + e < b) ? v[e] * xg[e] : T(0);
          *reinterpret_cast<val_t*>(&s_prod[(c * kBlock + t) * V]) = pv;
        }
      } else { // the last row block of the matrix: element-wise, in bounds
        if (!staged_visible) {
          __syncthreads();
          staged_visible = true;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int64_t j0 = base + (int64_t)(c * kBlock + t) * V;
          val_t pv;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const int64_t j = j0 + e;
            T p = T(0);
            if (j < b)
              p = values[j] * (K >= 0 ? s_x[lx.lidx[j]] : in[colind[j]]);
            pv[e] = p;
          }
          *reinterpret_cast<val_t*>(&s_prod[(c * kBlock + t) * V]) = pv;
        }
      }
      __syncthreads();
      const int32_t jlo = max((int64_t)lo, base) - base;
      const int32_t jhi = min((int64_t)hi, base + TILE) - base;
      int32_t j = jlo;
      for (; j + 4 <= jhi; j += 4) {
        const T p0 = s_prod[j], p1 = s_prod[j + 1], p2 = s_prod[j + 2],
                p3 = s_prod[j + 3];
        sum += p0;
        sum += p1;
        sum += p2;
        sum += p3;
      }
      for (; j < jhi; ++j)
        sum += s_prod[j];
    }

    if (t < nr) {
      const int32_t r = r0 + t;
      const T c = alpha * sum;
      T y = c;
      if (beta != T(0))
        y = c + beta * out[r];
      out[r] = y;
      if constexpr (DOT)
        dot_acc += (double)x_own * (double)c;
    }
  }

  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// SCALAR kernel: one lane per row, the reference loop verbatim.
// ---------------------------------------------------------------------------
template <typename TV, typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void csr_scalar_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const TV* __restrict__ values, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot)
{
  __shared__ double s_red[kBlock / 64];
  double dot_acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < num_rows; i += (int64_t)gridDim.x * blockDim.x) {
    T sum = 0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
      sum += (T)values[j] * in[colind[j]];
    const T c = alpha * sum;
    T y = c;
    if (beta != T(0))
      y = c + beta * out[i];
    out[i] = y;
    if constexpr (DOT)
      dot_acc += (double)in[i] * (double)c;
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// ROWLIST kernel: for blocks whose rows are mostly empty (the "remote" block
// of a row-partitioned matrix, Matrix.cpp:354-355: only boundary rows touch
// ghost columns).  The plan holds the compacted list of non-empty rows; one
// lane per listed row, reference order.  `out` has already been scaled by
// beta for ALL rows (a no-op for the beta == 1 the reference uses here,
// Matrix.cpp:508,529,551), so out[r] = alpha*sum + out[r] rounds exactly like
// alpha*sum + beta*out[r].  With DOT the kernel emits the partials of
// sum_r in[r] * (alpha*sum_r): the block's own share of p.Ap.
// ---------------------------------------------------------------------------
template <typename TV, typename T, bool DOT>
__global__ __launch_bounds__(kBlock) void csr_rowlist_kernel(
    int32_t num_listed, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const TV* __restrict__ values, T alpha, const T* __restrict__ in,
    T* __restrict__ out, DotOut dot)
{
  __shared__ double s_red[kBlock / 64];
  double dot_acc = 0.0;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       k < num_listed; k += (int64_t)gridDim.x * blockDim.x) {
    const int32_t i = rows[k];
    T sum = 0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
      sum += (T)values[j] * in[colind[j]];
    const T c = alpha * sum;
    out[i] = c + out[i];
    if constexpr (DOT)
      dot_acc += (double)in[i] * (double)c;
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}


// ---------------------------------------------------------------------------
// VECTOR kernel: LPR lanes per row, strided walk + shuffle reduction.
// ---------------------------------------------------------------------------
template <typename TV, typename T, int LPR, bool DOT>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const TV* __restrict__ values, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot)
{
  __shared__ double s_red[kBlock / 64];
  constexpr int RPB = kBlock / LPR; // rows per workgroup
  const int sub = threadIdx.x % LPR;
  const int grp = threadIdx.x / LPR;
  double dot_acc = 0.0;
  const int64_t nblk = ((int64_t)num_rows + RPB - 1) / RPB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t i = blk * RPB + grp;
    T sum = 0;
    if (i < num_rows) {
      const int32_t lo = rowptr[i], hi = rowptr[i + 1];
      for (int32_t j = lo + sub; j < hi; j += LPR)
        sum += (T)values[j] * in[colind[j]];
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1)
      sum += __shfl_down(sum, off, LPR);
    if (i < num_rows && sub == 0) {
      const T c = alpha * sum;
      T y = c;
      if (beta != T(0))
        y = c + beta * out[i];
      out[i] = y;
      if constexpr (DOT)
        dot_acc += (double)in[i] * (double)c;
    }
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// XW or the gather kernel?  Both stream the caller's arrays untouched and give
// the same bits; XW moves 2 GB less across the fabric at 512^3 but runs two
// workgroups per CU against eight, and which of them is faster differed from
// box to box by a few per cent either way (five pairs
// 4 : 1 for XW, the round-5 driver's box the other way).  So the plan lets its
// first four launches decide -- every one of them a full, correct product:
//   launch 0 XW, 1 gather (cold: not looked at), 2 XW, 3 gather (timed by HIP
//   events on the launch's own stream); the first later launch that finds the
//   last event complete reads the two times and fixes the choice.
// Nothing is allocated, nothing runs twice, no launch waits for the host.
// (struct XwProbe: csr_plan.h; its clean-up: spmv_csr_forms.hip)
// Which kernel does this launch run (1 = XW, 0 = gather)?  *probing = the
// index of the event pair to record around it, or -1.
int xw_probe_pick(XwProbe* pb, hipStream_t st, int* probing)
{
  *probing = -1;
  if (!pb || pb->decided)
    return pb ? pb->use_xw : 1;
  if (pb->launches < 4) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // (inside a graph capture nothing is timed: XW, as without the probe)
    if (hipStreamIsCapturing(st, &cs) != hipSuccess
        || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return 1;
    }
    const int i = pb->launches;
    if (!pb->ev[i][0]
        && (hipEventCreate(&pb->ev[i][0]) != hipSuccess
            || hipEventCreate(&pb->ev[i][1]) != hipSuccess)) {
      (void)hipGetLastError();
      xw_probe_drop_events(pb);
      pb->decided = 1; // no events: the size rule stands
      return pb->use_xw;
    }
    *probing = i;
    return (i & 1) ? 0 : 1;
  }
  // the four launches are out: read them as soon as the last one has finished
  if (hipEventQuery(pb->ev[3][1]) == hipSuccess) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, pb->ev[2][0], pb->ev[2][1]) == hipSuccess
        && hipEventElapsedTime(&b, pb->ev[3][0], pb->ev[3][1]) == hipSuccess) {
      pb->us_xw = a * 1e3f;
      pb->us_gather = b * 1e3f;
      pb->use_xw = a <= b ? 1 : 0;
    }
    (void)hipGetLastError();
    xw_probe_drop_events(pb);
    pb->decided = 1;
    return pb->use_xw;
  }
  (void)hipGetLastError(); // hipErrorNotReady
  return 1;
}

template <typename TV, typename T, int CH, bool NT, bool ALIGNED, bool DOT>
int launch_rowblock_x(const spmv_hip_csr_plan* pl, hipStream_t st, int grid,
                      int nrb, const int32_t* rowptr, const int32_t* colind,
                      const TV* values, T alpha, const T* in, T beta, T* out,
                      DotOut dot)
{
  // (the plane-walk order does nothing for this kernel: one step of an XCD's
  // workgroups streams more than its L2 holds -- 2.89 against 2.91 ms at
  // 512^3, 0.217 against 0.198 at 216^3; profiles/r05_rowblock_walk.log)
  const RowBlockOrder ord = pl->row_block_order(nrb);
  if (ord.xcd_group > 0)
    hipLaunchKernelGGL((csr_rowblock_kernel<TV, T, CH, NT, ALIGNED, DOT, true>),
                       dim3(grid), dim3(kBlock), 0, st, pl->num_rows, pl->nnz,
                       rowptr, colind, values, alpha, in, beta, out, dot, ord);
  else
    hipLaunchKernelGGL((csr_rowblock_kernel<TV, T, CH, NT, ALIGNED, DOT, false>),
                       dim3(grid), dim3(kBlock), 0, st, pl->num_rows, pl->nnz,
                       rowptr, colind, values, alpha, in, beta, out, dot, ord);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

// launch grid of the row-block kernels (plain and LX)
int rowblock_grid(const spmv_hip_csr_plan* pl)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * pl->blocks_per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid < 1)
    grid = 1;
  // slots it with equal it % 8 must stay on one XCD (XCD groups)
  if (grid >= 8)
    grid -= grid % 8;
  return grid;
}

template <typename T, bool DOT>
int launch_rowblock(const spmv_hip_csr_plan* pl, hipStream_t st,
                    const int32_t* rowptr, const int32_t* colind,
                    const T* values, T alpha, const T* in, T beta, T* out,
                    DotOut dot)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const int grid = rowblock_grid(pl);
  const bool al = aligned16(values) && aligned16(colind);
  // the plan's own copy of a symmetric matrix's lower half (spmv_symdia.hip)
  if (pl->sdia && pl->sdia_val && pl->sdia_general
      && pl->sdia_elem == (int)sizeof(T) && values == pl->sdia_values0) {
    if constexpr (sizeof(T) == 8)
      return spmv_sdia_run_f64(pl, st, alpha, in, beta, out,
                               DOT ? dot : DotOut());
    else
      return spmv_sdia_run_f32(pl, st, alpha, in, beta, out);
  }
  // the plan's own copy by offset of a matrix on <= 32 diagonals
  if (pl->wdia && pl->wdia_val && pl->wdia_elem == (int)sizeof(T)
      && values == pl->wdia_values0) {
    if constexpr (sizeof(T) == 8)
      return spmv_wdia_run_f64(pl, st, alpha, in, beta, out,
                               DOT ? dot : DotOut());
    else
      return spmv_wdia_run_f32(pl, st, alpha, in, beta, out);
  }
  // the plan's own copy in sliced jagged order (ragged / long rows)
  if (pl->sj && pl->sj_val && pl->sj_elem == (int)sizeof(T)
      && values == pl->sj_values0 && aligned16(in) && pl->num_cols >= 2) {
    if constexpr (sizeof(T) == 8)
      return spmv_sjds_run_f64(pl, st, alpha, in, beta, out,
                               DOT ? dot : DotOut());
    else
      return spmv_sjds_run_f32(pl, st, alpha, in, beta, out);
  }
  if (pl->lat && aligned16(values)) {
    if constexpr (sizeof(T) == 8)
      return spmv_lat_run_f64(pl, st, rowptr, values, alpha, in, beta, out,
                              DOT ? dot : DotOut());
    else
      return spmv_lat_run_f32(pl, st, rowptr, values, alpha, in, beta, out);
  }
  if (pl->lx && pl->lxw && pl->lxw_rec && al && aligned16(in)) {
    if constexpr (sizeof(T) == 8)
      return spmv_lxw_run_f64(pl, st, rowptr, colind, values, alpha, in, beta,
                              out, DOT ? dot : DotOut());
    else
      return spmv_lxw_run_f32(pl, st, rowptr, colind, values, alpha, in, beta,
                              out);
  }
  if (pl->lx && al && aligned16(in)) {
    LxView lx{pl->lx_lidx, pl->lx_tab};
    RowBlockOrder lx_ord = pl->row_block_order(nrb);
    if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
      lx_ord.table = pl->zw_table; // plane-walk order (build_lx)
      lx_ord.num_slots = pl->zw_slots;
    }
#define SPMV_LX(NT, CH)                                                        \
  hipLaunchKernelGGL((csr_rowblock_lx_kernel<T, NT, DOT, CH>), dim3(grid),     \
                     dim3(kBlock), 0, st, pl->num_rows, pl->num_cols, pl->nnz, \
                     rowptr, colind, values, lx, alpha, in, beta, out, dot,    \
                     lx_ord)
    if (pl->nontemporal) {
      if (pl->lx_chunks == 2)
        SPMV_LX(true, 2);
      else
        SPMV_LX(true, 1);
    } else {
      if (pl->lx_chunks == 2)
        SPMV_LX(false, 2);
      else
        SPMV_LX(false, 1);
    }
#undef SPMV_LX
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  // the caller's arrays as they are, x windows staged (spmv_lxw.hip, XW) --
  // or gathered, where the plan's first launches found that faster (XwProbe)
  int probing = -1;
  if (pl->xw && pl->xw_rec && al && aligned16(in)) {
    const int use_xw = xw_probe_pick(pl->xw_probe, st, &probing);
    if (probing >= 0)
      SPMV_CHECK_HIP(hipEventRecord(pl->xw_probe->ev[probing][0], st));
    if (use_xw) {
      int rc;
      if constexpr (sizeof(T) == 8)
        rc = spmv_xw_run_f64(pl, st, rowptr, colind, values, alpha, in, beta, out,
                             DOT ? dot : DotOut());
      else
        rc = spmv_xw_run_f32(pl, st, rowptr, colind, values, alpha, in, beta, out);
      if (probing >= 0 && rc == SPMV_HIP_OK) {
        SPMV_CHECK_HIP(hipEventRecord(pl->xw_probe->ev[probing][1], st));
        pl->xw_probe->launches = probing + 1;
      }
      return rc;
    }
  }
  // (a probed gather launch: the closing event and the count)
  struct ProbeEnd {
    const spmv_hip_csr_plan* pl;
    hipStream_t st;
    int i;
    ~ProbeEnd()
    {
      if (i >= 0 && hipEventRecord(pl->xw_probe->ev[i][1], st) == hipSuccess)
        pl->xw_probe->launches = i + 1;
    }
  } probe_end{pl, st, probing};
#define SPMV_RB(CH, NT, AL)                                                    \
  return launch_rowblock_x<T, T, CH, NT, AL, DOT>(pl, st, grid, nrb, rowptr,   \
                                               colind, values, alpha, in,     \
                                               beta, out, dot)
  if (!al)
    SPMV_RB(2, false, false);
  if (pl->nontemporal) {
    if (pl->chunks == 1)
      SPMV_RB(1, true, true);
    if (pl->chunks == 4)
      SPMV_RB(4, true, true);
    SPMV_RB(2, true, true);
  }
  if (pl->chunks == 1)
    SPMV_RB(1, false, true);
  if (pl->chunks == 4)
    SPMV_RB(4, false, true);
  SPMV_RB(2, false, true);
#undef SPMV_RB
}

template <typename TV, typename T, bool DOT>
int launch_vector(const spmv_hip_csr_plan* pl, hipStream_t st,
                  const int32_t* rowptr, const int32_t* colind,
                  const TV* values, T alpha, const T* in, T beta, T* out,
                  DotOut dot)
{
  const int lpr = pl->lanes_per_row;
  const int64_t nblk = ((int64_t)pl->num_rows * lpr + kBlock - 1) / kBlock;
  int grid = pl->ctx->dot_blocks;
  if (grid > nblk)
    grid = (int)(nblk < 1 ? 1 : nblk);
#define SPMV_VEC(L)                                                            \
  hipLaunchKernelGGL((csr_vector_kernel<TV, T, L, DOT>), dim3(grid),             \
                     dim3(kBlock), 0, st, pl->num_rows, rowptr, colind, values, alpha, in, \
                     beta, out, dot)
  switch (lpr) {
  case 4: SPMV_VEC(4); break;
  case 8: SPMV_VEC(8); break;
  case 16: SPMV_VEC(16); break;
  case 32: SPMV_VEC(32); break;
  default: SPMV_VEC(64); break;
  }
#undef SPMV_VEC
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

template <typename TV, typename T, bool DOT>
int launch_scalar(const spmv_hip_csr_plan* pl, hipStream_t st,
                  const int32_t* rowptr, const int32_t* colind,
                  const TV* values, T alpha, const T* in, T beta, T* out,
                  DotOut dot)
{
  const int grid = spmv_grid_for(pl->ctx, pl->num_rows, kBlock);
  hipLaunchKernelGGL((csr_scalar_kernel<TV, T, DOT>), dim3(grid), dim3(kBlock), 0,
                     st, pl->num_rows, rowptr, colind, values, alpha, in, beta,
                     out, dot);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

template <typename TV, typename T, bool DOT>
int launch_rowlist(const spmv_hip_csr_plan* pl, hipStream_t st,
                   const int32_t* rowptr, const int32_t* colind,
                   const TV* values, T alpha, const T* in, T beta, T* out,
                   DotOut dot)
{
  const int n = pl->num_rows;
  if (beta != T(1)) { // all rows: out = beta*out (0 without reading it)
    if (beta == T(0)) {
      SPMV_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(T) * (size_t)n, st));
    } else {
      const int grid = spmv_grid_for(pl->ctx, n, kBlock);
      hipLaunchKernelGGL((scale_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                         (int64_t)n, beta, out);
      SPMV_CHECK_LAUNCH();
    }
  }
  const int grid = spmv_grid_for(pl->ctx, pl->num_listed, kBlock);
  hipLaunchKernelGGL((csr_rowlist_kernel<TV, T, DOT>), dim3(grid), dim3(kBlock), 0,
                     st, pl->num_listed, pl->row_list, rowptr, colind, values,
                     alpha, in, out, dot);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

template <typename T, bool DOT>
int run_general(const spmv_hip_csr_plan* pl, hipStream_t st,
                const int32_t* rowptr, const int32_t* colind, const T* values,
                T alpha, const T* in, T beta, T* out, DotOut dot)
{
  switch (pl->algo) {
  case SPMV_HIP_ALGO_VECTOR:
    return launch_vector<T, T, DOT>(pl, st, rowptr, colind, values, alpha, in,
                                    beta, out, dot);
  case SPMV_HIP_ALGO_SCALAR:
    return launch_scalar<T, T, DOT>(pl, st, rowptr, colind, values, alpha, in,
                                    beta, out, dot);
  case SPMV_HIP_ALGO_ROWLIST:
    return launch_rowlist<T, T, DOT>(pl, st, rowptr, colind, values, alpha, in,
                                  beta, out, dot);
  default:
    return launch_rowblock<T, DOT>(pl, st, rowptr, colind, values, alpha, in,
                                   beta, out, dot);
  }
}


// Mixed precision: fp32 `values`, fp64 vectors and arithmetic.
// General blocks only, every algorithm a general plan can have: diagonal and
// lattice forms, plain row blocks, row list, vector, scalar.
template <bool DOT>
int run_mixed(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const int32_t* colind,
                     const float* values, double alpha, const double* in,
                     double beta, double* out, DotOut dot)
{
  if (pl->algo == SPMV_HIP_ALGO_ROWLIST)
    return launch_rowlist<float, double, DOT>(pl, st, rowptr, colind, values,
                                              alpha, in, beta, out, dot);
  if (pl->algo == SPMV_HIP_ALGO_VECTOR)
    return launch_vector<float, double, DOT>(pl, st, rowptr, colind, values,
                                             alpha, in, beta, out, dot);
  if (pl->algo == SPMV_HIP_ALGO_SCALAR)
    return launch_scalar<float, double, DOT>(pl, st, rowptr, colind, values,
                                             alpha, in, beta, out, dot);
  // the plan's fp32 copy of a symmetric matrix's lower half (spmv_symdia.hip)
  if (pl->sdia && pl->sdia32_val && values == pl->sdia32_values0)
    return spmv_sdia_run_f32f64(pl, st, alpha, in, beta, out,
                                DOT ? dot : DotOut());
  // ... or of a matrix in the wide diagonal form (spmv_wdia.hip)
  if (pl->wdia && pl->wdia32_val && values == pl->wdia32_values0)
    return spmv_wdia_run_f32f64(pl, st, alpha, in, beta, out,
                                DOT ? dot : DotOut());
  // ... or in sliced jagged order (spmv_sjds.hip)
  if (pl->sj && pl->sj_val32 && values == pl->sj32_values0 && aligned16(in)
      && pl->num_cols >= 2)
    return spmv_sjds_run_f32f64(pl, st, alpha, in, beta, out, DOT ? dot : DotOut());
  if (pl->lat && aligned16(values))
    return spmv_lat_run_f32f64(pl, st, rowptr, values, alpha, in, beta, out,
                               DOT ? dot : DotOut());
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * pl->blocks_per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid < 1)
    grid = 1;
  if (grid >= 8)
    grid -= grid % 8;
  if (aligned16(values) && aligned16(colind))
    return launch_rowblock_x<float, double, 1, false, true, DOT>(
        pl, st, grid, nrb, rowptr, colind, values, alpha, in, beta, out, dot);
  return launch_rowblock_x<float, double, 1, false, false, DOT>(
      pl, st, grid, nrb, rowptr, colind, values, alpha, in, beta, out, dot);
}

// does a launch with these operands take the form that owns the matrix?
template <typename T>
bool released_launch_ok(const spmv_hip_csr_plan* pl, const T* values,
                               const T* in, const T* diagonal)
{
  if (pl->algo != SPMV_HIP_ALGO_ROWBLOCK || spmv_plan_owned_mask(pl) == 0)
    return false;
  if (pl->symmetric)
    return pl->sjt->sj_elem == (int)sizeof(T) && values == pl->sjt->sj_values0
           && diagonal == pl->sj_diag0 && aligned16(in);
  if (pl->sdia && pl->sdia_val && pl->sdia_general)
    return pl->sdia_elem == (int)sizeof(T) && values == pl->sdia_values0;
  if (pl->wdia && pl->wdia_val)
    return pl->wdia_elem == (int)sizeof(T) && values == pl->wdia_values0;
  return pl->sj_elem == (int)sizeof(T) && values == pl->sj_values0 && aligned16(in);
}

} // namespace

int spmv_rowblock_grid(const spmv_hip_csr_plan* pl) { return rowblock_grid(pl); }

extern "C" {

int spmv_hip_csr_spmv_f64(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                          int32_t num_rows, int32_t num_cols,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          const int32_t* colind, const double* values,
                          const double* diagonal, double alpha,
                          const double* in, double beta, double* out,
                          double* dot_partials, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  // the operands must be the ones the plan (and the launch shape) was built
  // for: a mismatch would index out of bounds on the device
  SPMV_REQUIRE(num_rows == plan->num_rows && num_cols == plan->num_cols
               && num_non_zeros == plan->nnz);
  // ... and the arrays it analysed, whenever their content is baked in
  SPMV_REQUIRE(!plan->structure_baked()
               || (rowptr == plan->rowptr0 && colind == plan->colind0));
  if (num_rows == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(in && out);
  SPMV_REQUIRE(num_non_zeros == 0 || (rowptr && colind && values));
  // arrays given up: only the form that owns the matrix may run
  SPMV_REQUIRE(!plan->released || released_launch_ok(plan, values, in, diagonal));
  hipStream_t st = spmv_stream(ctx, stream);
  if (plan->symmetric) {
    DotOut dot;
    dot.partials = dot_partials; // may be NULL
    dot.len = ctx->dot_blocks;
    return spmv_run_symmetric_f64(plan, st, rowptr, colind, values, diagonal,
                                 alpha, in, beta, out, dot);
  }
  if (num_non_zeros == 0) {
    // empty general block: out = beta*out (csr_kernels.cpp:44-49 with an
    // empty inner loop); CSRMatrix::mult skips the call entirely
    // (csr_matrix.cpp:85), the executor keeps the arithmetic definition.
    SPMV_REQUIRE(dot_partials == nullptr);
    const int grid = spmv_grid_for(ctx, num_rows, kBlock);
    hipLaunchKernelGGL((scale_kernel<double>), dim3(grid), dim3(kBlock), 0, st,
                       (int64_t)num_rows, beta, out);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  if (dot_partials) {
    DotOut dot;
    dot.partials = dot_partials;
    dot.len = ctx->dot_blocks;
    return run_general<double, true>(plan, st, rowptr, colind, values, alpha,
                                     in, beta, out, dot);
  }
  return run_general<double, false>(plan, st, rowptr, colind, values, alpha, in,
                                    beta, out, DotOut());
}

int spmv_hip_csr_spmv_f32f64(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                             int32_t num_rows, int32_t num_cols,
                             int64_t num_non_zeros, const int32_t* rowptr,
                             const int32_t* colind, const float* values,
                             double alpha, const double* in, double beta,
                             double* out, double* dot_partials, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx && !plan->symmetric);
  SPMV_REQUIRE(!plan->released); // (mixed launches may read the CSR arrays)
  SPMV_REQUIRE(num_rows == plan->num_rows && num_cols == plan->num_cols
               && num_non_zeros == plan->nnz);
  SPMV_REQUIRE(!plan->structure_baked()
               || (rowptr == plan->rowptr0 && colind == plan->colind0));
  if (num_rows == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(in && out);
  hipStream_t st = spmv_stream(ctx, stream);
  if (num_non_zeros == 0) {
    SPMV_REQUIRE(dot_partials == nullptr);
    const int grid = spmv_grid_for(ctx, num_rows, kBlock);
    hipLaunchKernelGGL((scale_kernel<double>), dim3(grid), dim3(kBlock), 0, st,
                       (int64_t)num_rows, beta, out);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  SPMV_REQUIRE(rowptr && colind && values);
  if (dot_partials) {
    DotOut dot;
    dot.partials = dot_partials;
    dot.len = ctx->dot_blocks;
    return run_mixed<true>(plan, st, rowptr, colind, values, alpha, in, beta,
                           out, dot);
  }
  return run_mixed<false>(plan, st, rowptr, colind, values, alpha, in, beta,
                          out, DotOut());
}

int spmv_hip_csr_spmv_f32(spmv_hip_ctx* ctx, const spmv_hip_csr_plan* plan,
                          int32_t num_rows, int32_t num_cols,
                          int64_t num_non_zeros, const int32_t* rowptr,
                          const int32_t* colind, const float* values,
                          const float* diagonal, float alpha, const float* in,
                          float beta, float* out, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  SPMV_REQUIRE(num_rows == plan->num_rows && num_cols == plan->num_cols
               && num_non_zeros == plan->nnz);
  // ... and the arrays it analysed, whenever their content is baked in
  SPMV_REQUIRE(!plan->structure_baked()
               || (rowptr == plan->rowptr0 && colind == plan->colind0));
  if (num_rows == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(in && out);
  SPMV_REQUIRE(num_non_zeros == 0 || (rowptr && colind && values));
  SPMV_REQUIRE(!plan->released || released_launch_ok(plan, values, in, diagonal));
  hipStream_t st = spmv_stream(ctx, stream);
  if (plan->symmetric)
    return spmv_run_symmetric_f32(plan, st, rowptr, colind, values, diagonal,
                                alpha, in, beta, out);
  if (num_non_zeros == 0) {
    const int grid = spmv_grid_for(ctx, num_rows, kBlock);
    hipLaunchKernelGGL((scale_kernel<float>), dim3(grid), dim3(kBlock), 0, st,
                       (int64_t)num_rows, beta, out);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
  return run_general<float, false>(plan, st, rowptr, colind, values, alpha, in,
                                   beta, out, DotOut());
}

} // extern "C"
