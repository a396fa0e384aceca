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
// The symmetric-storage kernels live in spmv_sym.hip, the lattice form in
// spmv_lat.hip; csr_plan.h holds what the three files share.
#include "csr_plan.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

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

// Plan-time analysis for the LX kernels: one workgroup per row block sorts the
// block's column indices, cuts them into windows (gap > kLxGap), and writes the
// windows and every entry's offset into the staged buffer.  Blocks with more
// than 256*ITEMS entries, more than kLxMaxWin windows or more than `cap`
// staged elements are marked direct (nwin = -1).
//   align   window starts are multiples of this many columns (2: the register
//           kernel's pair loads; 4: 16-byte LDS-DMA chunks of fp64 AND fp32 x)
//   pad     every window occupies a multiple of this many staged elements
//           (2, or kLxwPiece = whole DMA pieces)
//   wrec    != nullptr: also the record of the DMA kernel (spmv_lxw.hip):
//           span, piece list; stat[0] / stat[1] collect the largest entry count
//           and piece count of a staged block
//   xw      != 0: ONLY that record, in the XW layout (kXwRec ints: the windows
//           themselves follow the piece list; at most kXwMaxWin of them) -- no
//           16-bit indices, no register-kernel record (lidx, tab unused);
//           stat[2] counts the staged blocks
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void lx_build_kernel(
    int32_t num_rows, int32_t num_cols, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, uint16_t* __restrict__ lidx,
    int32_t* __restrict__ tab, int num_row_blocks, int end_bit, int align,
    int pad, int cap, int32_t* __restrict__ wrec, int32_t* __restrict__ stat,
    int xw)
{
  using Sort = hipcub::BlockRadixSort<int32_t, kBlock, ITEMS, int32_t>;
  using Scan = hipcub::BlockScan<int32_t, kBlock>;
  constexpr int CAP = kBlock * ITEMS;
  __shared__ union {
    typename Sort::TempStorage sort;
    typename Scan::TempStorage scan;
  } tmp;
  __shared__ int32_t s_key[CAP];
  __shared__ int32_t s_ws[kLxMaxWin], s_we[kLxMaxWin], s_wo[kLxMaxWin + 1];
  __shared__ int s_direct;
  const int t = threadIdx.x;
  // columns from here on cannot be fetched as whole aligned 16-byte chunks
  const int32_t col_limit = wrec ? (num_cols & ~(align - 1)) : INT32_MAX;
  for (int rb = blockIdx.x; rb < num_row_blocks; rb += gridDim.x) {
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);
    const int32_t a = rowptr[r0], b = rowptr[r0 + nr];
    const int cnt = b - a;
    int32_t* wr
        = wrec ? wrec + (int64_t)rb * (xw ? kXwRec : kLxwRec) : nullptr;
    const int max_win = xw ? kXwMaxWin : kLxMaxWin;
    __syncthreads(); // previous block done with the shared arrays
    if (wr && t == 0) {
      wr[0] = -1; // direct unless the analysis below succeeds
      wr[1] = a;
      wr[2] = cnt;
      wr[3] = 0;
    }
    if (cnt > CAP) {
      if (t == 0 && !xw)
        tab[(int64_t)rb * kLxRec] = -1;
      continue; // uniform
    }
    int32_t key[ITEMS], pos[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int idx = t * ITEMS + i;
      key[i] = idx < cnt ? colind[a + idx] : INT32_MAX;
      pos[i] = idx;
    }
    Sort(tmp.sort).Sort(key, pos, 0, end_bit);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      s_key[t * ITEMS + i] = key[i];
    if (t == 0)
      s_direct = 0;
    __syncthreads();
    // window starts: first valid key, or a gap larger than kLxGap
    int32_t flag[ITEMS], wid[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int idx = t * ITEMS + i;
      const bool valid = idx < cnt; // INT32_MAX padding sorts to the end
      flag[i] = valid && (idx == 0 || key[i] - s_key[idx - 1] > kLxGap) ? 1 : 0;
    }
    int32_t total_windows = 0;
    Scan(tmp.scan).InclusiveSum(flag, wid, total_windows);
    if (total_windows > max_win
        || (cnt > 0 && s_key[cnt - 1] >= col_limit)) {
      if (t == 0 && !xw)
        tab[(int64_t)rb * kLxRec] = -1;
      continue; // uniform (block-wide aggregate / shared value)
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int idx = t * ITEMS + i;
      if (idx < cnt) {
        if (flag[i])
          s_ws[wid[i] - 1] = key[i] & ~(align - 1);
        // last entry of its window: end of data or the next key starts one
        const bool last = idx == cnt - 1 || s_key[idx + 1] - key[i] > kLxGap;
        if (last)
          s_we[wid[i] - 1] = (key[i] + align) & ~(align - 1);
      }
    }
    __syncthreads();
    if (t == 0) {
      int off = 0;
      for (int k = 0; k < total_windows; ++k) {
        s_wo[k] = off;
        off += (s_we[k] - s_ws[k] + pad - 1) & ~(pad - 1);
      }
      s_wo[total_windows] = off;
      if (off > cap)
        s_direct = 1;
    }
    __syncthreads();
    if (s_direct) {
      if (t == 0 && !xw)
        tab[(int64_t)rb * kLxRec] = -1;
      continue;
    }
    if (!xw) {
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        const int idx = t * ITEMS + i;
        if (idx < cnt)
          lidx[a + pos[i]]
              = (uint16_t)(s_wo[wid[i] - 1] + (key[i] - s_ws[wid[i] - 1]));
      }
      int32_t* rec = tab + (int64_t)rb * kLxRec;
      if (t < total_windows)
        rec[1 + t] = s_ws[t];
      if (t <= total_windows)
        rec[1 + kLxMaxWin + t] = s_wo[t];
      if (t == 0)
        rec[0] = total_windows;
    } else if (t < kXwMaxWin) {
      // the windows themselves: where window t + 1 starts, and what turns a
      // column of window t into its staged position
      wr[kXwFirst0 + t] = t + 1 < total_windows ? s_ws[t + 1] : INT32_MAX;
      wr[kXwDelta0 + t] = t < total_windows ? s_wo[t] - s_ws[t] : 0;
    }
    if (wr && t == 0) {
      // the staged buffer as DMA pieces of kLxwPiece elements: source column
      // of each (pad == kLxwPiece: windows start at piece boundaries)
      int np = 0;
      for (int k = 0; k < total_windows; ++k)
        for (int c = s_ws[k]; c < s_we[k]; c += kLxwPiece)
          wr[kLxwPieces0 + np++] = c;
      wr[0] = total_windows;
      // where the block's OWN columns are staged (x_i of the fused dot): the
      // window that holds [r0, r0 + nr), if one does
      int own = -1;
      for (int k = 0; k < total_windows; ++k)
        if (s_ws[k] <= r0 && r0 + nr <= s_we[k])
          own = s_wo[k] + (r0 - s_ws[k]);
      // np = s_wo[total_windows] / kLxwPiece <= cap / kLxwPiece
      wr[3] = np | ((own + 1) << kLxwOwnShift);
      // (read first: an atomic per row block on one address is milliseconds)
      if (cnt > *(volatile int*)&stat[0])
        atomicMax(&stat[0], cnt);
      if (np > *(volatile int*)&stat[1])
        atomicMax(&stat[1], np);
      if (xw)
        atomicAdd(&stat[2], 1); // (one per staged block, spread over the grid)
    }
  }
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

struct NonEmptyRow {
  const int32_t* rowptr;
  __device__ bool operator()(int i) const { return rowptr[i + 1] > rowptr[i]; }
};

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

} // namespace

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
struct XwProbe {
  int launches = 0;
  int decided = 0;  // the choice is fixed
  int use_xw = 1;   // ... to this
  hipEvent_t ev[4][2] = {};
  float us_xw = 0.f, us_gather = 0.f;
};

namespace
{

void xw_probe_free(spmv_hip_csr_plan* pl)
{
  if (!pl->xw_probe)
    return;
  for (auto& e : pl->xw_probe->ev)
    for (hipEvent_t& h : e)
      if (h) {
        (void)hipEventDestroy(h);
        h = nullptr;
      }
  delete pl->xw_probe;
  pl->xw_probe = nullptr;
}

void xw_probe_drop_events(XwProbe* pb)
{
  for (auto& e : pb->ev)
    for (hipEvent_t& h : e)
      if (h) {
        (void)hipEventDestroy(h);
        h = nullptr;
      }
}

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


// Plane distance of a matrix on a 3-D grid that stays out of the lattice form
// (its values, boundary rows or a permutation of the entries keep it there):
// the farthest column above and below the diagonal in a row block in the middle
// of the matrix, when the two agree (and are far enough to be planes); else 0.
int64_t plane_distance(const spmv_hip_csr_plan* pl, const int32_t* rowptr,
                       const int32_t* colind)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const int mid = nrb / 2;
  const int32_t r0 = mid * kRows;
  const int nr = std::min(kRows, pl->num_rows - r0);
  if (nr <= 0)
    return 0;
  std::vector<int32_t> rp(nr + 1), ci;
  hipError_t em = hipMemcpy(rp.data(), rowptr + r0, sizeof(int32_t) * (nr + 1),
                            hipMemcpyDeviceToHost);
  const int64_t cnt = em == hipSuccess ? (int64_t)rp[nr] - rp[0] : 0;
  if (cnt <= 0 || cnt > 65536) {
    (void)hipGetLastError();
    return 0;
  }
  ci.resize((size_t)cnt);
  em = hipMemcpy(ci.data(), colind + rp[0], sizeof(int32_t) * (size_t)cnt,
                 hipMemcpyDeviceToHost);
  if (em != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int64_t up = 0, down = 0;
  for (int i = 0; i < nr; ++i)
    for (int32_t j = rp[i]; j < rp[i + 1]; ++j) {
      const int64_t d = (int64_t)ci[(size_t)(j - rp[0])] - (r0 + i);
      up = d > up ? d : up;
      down = -d > down ? -d : down;
    }
  return (up == down && up >= 2 * kRows && up <= INT32_MAX) ? up : 0;
}

// Compact the indices of the non-empty rows on the device (plan time, once).
int build_row_list(spmv_hip_csr_plan* pl, const int32_t* rowptr)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  hipStream_t st = pl->ctx->stream;
  const int n = pl->num_rows;
  const size_t cap = (size_t)(pl->nnz < n ? pl->nnz : n);
  int32_t* d_count = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  SPMV_CHECK_HIP(hipMalloc(&pl->row_list, sizeof(int32_t) * (cap ? cap : 1)));
  hipError_t e = hipMalloc(&d_count, sizeof(int32_t));
  hipcub::CountingInputIterator<int32_t> first(0);
  NonEmptyRow pred{rowptr};
  if (e == hipSuccess)
    e = hipcub::DeviceSelect::If(nullptr, tmp_bytes, first, pl->row_list,
                                 d_count, n, pred, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceSelect::If(tmp, tmp_bytes, first, pl->row_list, d_count,
                                 n, pred, st);
  int32_t count = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&count, d_count, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  (void)hipFree(d_count);
  if (e != hipSuccess) {
    (void)hipFree(pl->row_list);
    pl->row_list = nullptr;
    return static_cast<int>(e);
  }
  pl->num_listed = count;
  return SPMV_HIP_OK;
}

void free_lx(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->lx_lidx);
  (void)hipFree(pl->lx_tab);
  (void)hipFree(pl->lxw_rec);
  pl->lx_lidx = nullptr;
  pl->lx_tab = nullptr;
  pl->lxw_rec = nullptr;
  pl->lx = pl->lx_staged = pl->lx_blocks = 0;
  pl->lxw = pl->lxw_max_cnt = pl->lxw_max_pieces = 0;
}

void free_xw(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->xw_rec);
  pl->xw_rec = nullptr;
  pl->xw = pl->xw_staged = pl->xw_max_cnt = pl->xw_max_pieces = 0;
  xw_probe_free(pl);
}

// The XW records (spmv_lxw.hip): per row block its span, the DMA pieces of its
// x windows and the windows themselves.  144 B per row block; the CSR arrays
// stay the caller's.  Kept only if most blocks are staged.
int build_xw(spmv_hip_csr_plan* pl, const int32_t* rowptr, const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  free_xw(pl);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  if (nrb == 0 || pl->nnz == 0 || pl->num_cols < kLxwAlign)
    return SPMV_HIP_OK;
  hipStream_t st = pl->ctx->stream;
  int32_t* d_stat = nullptr;
  hipError_t e = hipMalloc(&pl->xw_rec, sizeof(int32_t) * (size_t)nrb * kXwRec);
  if (e == hipSuccess)
    e = hipMalloc(&d_stat, 3 * sizeof(int32_t));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_stat, 0, 3 * sizeof(int32_t), st);
  if (e == hipSuccess)
    e = hipMemsetAsync(pl->xw_rec, 0, sizeof(int32_t) * (size_t)nrb * kXwRec, st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(d_stat);
    free_xw(pl);
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  int end_bit = 1;
  while (end_bit < 31 && ((int64_t)1 << end_bit) < pl->num_cols)
    ++end_bit;
  int grid = pl->ctx->num_cus * 4;
  grid = grid > nrb ? nrb : grid;
  const double avg = (double)pl->nnz / pl->num_rows;
  if (avg <= 6.0)
    hipLaunchKernelGGL(lx_build_kernel<8>, dim3(grid), dim3(kBlock), 0, st,
                       pl->num_rows, pl->num_cols, rowptr, colind, nullptr,
                       nullptr, nrb, end_bit, kLxwAlign, kLxwPiece,
                       kLxwMaxPieces * kLxwPiece, pl->xw_rec, d_stat, 1);
  else
    hipLaunchKernelGGL(lx_build_kernel<16>, dim3(grid), dim3(kBlock), 0, st,
                       pl->num_rows, pl->num_cols, rowptr, colind, nullptr,
                       nullptr, nrb, end_bit, kLxwAlign, kLxwPiece,
                       kLxwMaxPieces * kLxwPiece, pl->xw_rec, d_stat, 1);
  e = hipGetLastError();
  int32_t h_stat[3] = {0, 0, 0};
  if (e == hipSuccess)
    e = hipMemcpyAsync(h_stat, d_stat, sizeof(h_stat), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_stat);
  if (e != hipSuccess) {
    free_xw(pl);
    return static_cast<int>(e);
  }
  pl->xw_max_cnt = h_stat[0];
  pl->xw_max_pieces = h_stat[1];
  pl->xw_staged = h_stat[2];
  if ((int64_t)pl->xw_staged * 2 < nrb) { // mostly direct blocks: the gather kernel
    free_xw(pl);
    return SPMV_HIP_OK;
  }
  pl->xw = 1;
  if (pl->ctx->xw_probe)
    pl->xw_probe = new (std::nothrow) XwProbe;
  return SPMV_HIP_OK;
}

// ... with the plane-walk order when the matrix sits on a 3-D grid
int build_xw_and_walk(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                      const int32_t* colind)
{
  int rc = build_xw(pl, rowptr, colind);
  if (rc == SPMV_HIP_OK && pl->xw) {
    const int64_t d2 = plane_distance(pl, rowptr, colind);
    if (d2 > 0) {
      pl->lattice_d2 = (int)d2;
      rc = spmv_zwalk_order_build(pl, d2, spmv_walk_grid(pl), 0, false);
    }
  }
  return rc;
}

// may this plan stage x windows over the caller's arrays?
bool xw_applies(const spmv_hip_csr_plan* pl)
{
  const spmv_hip_ctx* ctx = pl->ctx;
  return !pl->symmetric && pl->algo == SPMV_HIP_ALGO_ROWBLOCK && !pl->lat && !pl->lx
         && pl->num_rows > 0 && pl->nnz >= ctx->xw_min_nnz
         && (double)pl->nnz / pl->num_rows <= 16.0
         && (int64_t)pl->num_cols * 8 >= ctx->xw_min_x_bytes;
}

struct IsStagedRecord {
  const int32_t* tab;
  __host__ __device__ int32_t operator()(int32_t rb) const
  {
    return tab[(int64_t)rb * kLxRec] >= 0 ? 1 : 0;
  }
};

// Build the LX form (see csr_rowblock_lx_kernel).  Costs 2 B per entry plus
// 144 B per row block of device memory; kept only if most blocks are staged.
int build_lx(spmv_hip_csr_plan* pl, const int32_t* rowptr,
             const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  free_lx(pl);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  if (nrb == 0 || pl->nnz == 0)
    return SPMV_HIP_OK;
  hipStream_t st = pl->ctx->stream;
  hipError_t e = hipMalloc(&pl->lx_lidx, sizeof(uint16_t) * (pl->nnz + 8));
  if (e == hipSuccess)
    e = hipMalloc(&pl->lx_tab, sizeof(int32_t) * (size_t)nrb * kLxRec);
  if (e == hipSuccess)
    e = hipMemsetAsync(pl->lx_lidx, 0, sizeof(uint16_t) * (pl->nnz + 8), st);
  if (e == hipSuccess)
    e = hipMemsetAsync(pl->lx_tab, 0, sizeof(int32_t) * (size_t)nrb * kLxRec, st);
  if (e != hipSuccess) {
    free_lx(pl);
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  // The LDS-DMA kernel's layout (windows padded to whole DMA pieces) when the
  // context asks for it and the matrix has at least one aligned chunk of x
  const bool dma = pl->ctx->lx_dma && pl->num_cols >= kLxwAlign;
  int32_t* d_stat = nullptr;
  if (dma) {
    e = hipMalloc(&pl->lxw_rec, sizeof(int32_t) * (size_t)nrb * kLxwRec);
    if (e == hipSuccess)
      e = hipMalloc(&d_stat, 2 * sizeof(int32_t));
    if (e == hipSuccess)
      e = hipMemsetAsync(d_stat, 0, 2 * sizeof(int32_t), st);
    if (e != hipSuccess) {
      (void)hipFree(d_stat);
      free_lx(pl);
      return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
    }
  }
  const int align = dma ? kLxwAlign : 2;
  const int pad = dma ? kLxwPiece : 2;
  const int cap = dma ? kLxwMaxPieces * kLxwPiece : kLxCap;
  int end_bit = 1;
  while (end_bit < 31 && ((int64_t)1 << end_bit) < pl->num_cols)
    ++end_bit;
  int grid = pl->ctx->num_cus * 4;
  grid = grid > nrb ? nrb : grid;
  const double avg = (double)pl->nnz / pl->num_rows;
  if (avg <= 6.0)
    hipLaunchKernelGGL(lx_build_kernel<8>, dim3(grid), dim3(kBlock), 0, st,
                       pl->num_rows, pl->num_cols, rowptr, colind, pl->lx_lidx,
                       pl->lx_tab, nrb, end_bit, align, pad, cap, pl->lxw_rec,
                       d_stat, 0);
  else
    hipLaunchKernelGGL(lx_build_kernel<16>, dim3(grid), dim3(kBlock), 0, st,
                       pl->num_rows, pl->num_cols, rowptr, colind, pl->lx_lidx,
                       pl->lx_tab, nrb, end_bit, align, pad, cap, pl->lxw_rec,
                       d_stat, 0);
  e = hipGetLastError();
  int32_t h_stat[2] = {0, 0};
  if (dma && e == hipSuccess)
    e = hipMemcpyAsync(h_stat, d_stat, sizeof(h_stat), hipMemcpyDeviceToHost, st);
  // how many row blocks are staged?
  int32_t* d_count = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int32_t staged = 0;
  if (e == hipSuccess)
    e = hipMalloc(&d_count, sizeof(int32_t));
  // the window count is the first int of every record
  hipcub::CountingInputIterator<int32_t> block_ids(0);
  hipcub::TransformInputIterator<int32_t, IsStagedRecord,
                                 hipcub::CountingInputIterator<int32_t>>
      flags(block_ids, IsStagedRecord{pl->lx_tab});
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(nullptr, tmp_bytes, flags, d_count, nrb, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(tmp, tmp_bytes, flags, d_count, nrb, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&staged, d_count, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  (void)hipFree(d_count);
  (void)hipFree(d_stat);
  if (e != hipSuccess) {
    free_lx(pl);
    return static_cast<int>(e);
  }
  pl->lx_blocks = nrb;
  pl->lx_staged = staged;
  pl->lxw_max_cnt = h_stat[0];
  pl->lxw_max_pieces = h_stat[1];
  if ((int64_t)staged * 2 < nrb) { // mostly direct blocks: not worth the memory
    free_lx(pl);
    pl->lx_blocks = nrb;
    return SPMV_HIP_OK;
  }
  pl->lx = 1;
  pl->lxw = pl->lxw_rec != nullptr;
  // XCD grouping with staged x: still +3.5 % while x lives in the Infinity
  // Cache (216^3: 0.170 vs 0.176 ms), but 1.3-1.8 % slower than the plain
  // order once it does not (512^3)
  pl->xcd_group = pl->nontemporal ? 16 : 0;
  // Far column windows at a constant distance (the matrix of a 3-D grid whose
  // values or boundary rows keep it out of the lattice form): walk the row
  // blocks plane by plane, so that the far windows of a block are the ones
  // its workgroup -- or a neighbour on the same XCD -- staged one step before.
  // Plane distance = the farthest column above and below the diagonal in a
  // row block in the middle of the matrix, when the two agree.
  {
    const int64_t d2 = plane_distance(pl, rowptr, colind);
    if (d2 > 0) {
      pl->lattice_d1 = 0;
      pl->lattice_d2 = (int)d2;
      const int rc = spmv_zwalk_order_build(pl, pl->lattice_d2,
                                            spmv_walk_grid(pl), 0, false);
      if (rc != SPMV_HIP_OK)
        return rc;
    }
  }
  return SPMV_HIP_OK;
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

} // namespace

// ---------------------------------------------------------------------------
// Plane-walk order.  Planes are d2 rows apart; plane z owns the row blocks
// [B_z, B_{z+1}), B_z = ceil(z d2 / 256), and its c-th block is "column" c.  A
// walker (segment q, column c) visits column c of the planes of segment q in
// ascending z; walkers are dealt to the `grid` workgroups in rounds, 8
// consecutive columns to one XCD.  Slot layout: ((round * L + step) * grid +
// workgroup), L = planes per segment.  One segment (512^3 on 1024 workgroups:
// the identity order) keeps every far window in the workgroup's own next
// block; more segments trade a little of that for balance when the columns do
// not fill the grid evenly.  Like every order table: a permutation of the row
// blocks plus empty slots -- it changes speed, never results.
// ---------------------------------------------------------------------------
// launch grid of the lattice kernel the plan runs
int spmv_walk_grid(const spmv_hip_csr_plan* pl)
{
  if (pl->sdia && pl->sdia_val)
    return spmv_sdia_grid(pl);
  if (pl->symmetric)
    return spmv_slat_grid(pl);
  if (pl->lat_tab)
    return spmv_lat_grid(pl);
  if (pl->lxw && pl->lxw_rec)
    return spmv_lxw_grid(pl, 8);
  return (pl->xw && pl->xw_rec) ? spmv_xw_grid(pl, 8) : rowblock_grid(pl);
}

void spmv_zwalk_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->zw_table);
  pl->zw_table = nullptr;
  pl->zw_slots = pl->zw_grid = pl->zw_segments = 0;
  pl->zwalk = 0; // zw_d2 stays: a knob can rebuild
}

// The table itself: host arithmetic only (spmv_hip_zwalk_table exposes it to
// the CPU tests).  Returns false when the lattice is too small for a table to
// pay (and !force) or the slot count would not fit an int.
static bool zwalk_table(int32_t num_rows, int64_t d2, int grid, int segments,
                        bool force, std::vector<int32_t>* table, int* segs_out)
{
  const int64_t nrb = ((int64_t)num_rows + kRows - 1) / kRows;
  const int64_t nz = ((int64_t)num_rows + d2 - 1) / d2;
  const int64_t P = (d2 + kRows - 1) / kRows; // columns
  // worth it only for a real 3-D (or wide 2-D) lattice that outgrows the grid
  if (!force && (P < 8 || nz < 8 || nrb < 4 * (int64_t)grid))
    return false;
  auto first_block = [&](int64_t z) {
    const int64_t b = (z * d2 + kRows - 1) / kRows;
    return b < nrb ? b : nrb;
  };
  int64_t Q = segments;
  if (Q == 0) {
    // steps per workgroup = rounds * L, a step without the plane-ahead reuse
    // (the first of every run) counted as 1.3 steps
    double best = 0.0;
    for (int64_t q = 1; q <= nz; q *= 2) {
      const int64_t L = (nz + q - 1) / q;
      if (L < 4 && q > 1)
        break;
      const int64_t rounds = (q * P + grid - 1) / grid;
      const double cost = (double)rounds * ((double)L + 0.3);
      if (Q == 0 || cost < best) {
        best = cost;
        Q = q;
      }
    }
  }
  if (Q > nz)
    Q = nz;
  const int64_t L = (nz + Q - 1) / Q;
  Q = (nz + L - 1) / L; // no empty segments
  const int64_t W = Q * P;
  const int64_t rounds = (W + grid - 1) / grid;
  const int64_t slots = rounds * L * grid;
  if (slots > INT32_MAX)
    return false;
  const int g = 8; // consecutive columns per XCD
  const bool by_xcd = grid % (8 * g) == 0;
  table->assign((size_t)slots, -1);
  for (int64_t r = 0; r < rounds; ++r)
    for (int w = 0; w < grid; ++w) {
      int64_t idx = w;
      if (by_xcd) {
        const int x = w % 8, m = w / 8;
        idx = (int64_t)(m / g) * (8 * g) + x * g + (m % g);
      }
      const int64_t v = r * grid + idx;
      if (v >= W)
        continue;
      const int64_t q = v / P, c = v % P;
      for (int64_t s = 0; s < L; ++s) {
        const int64_t z = q * L + s;
        if (z >= nz)
          break;
        const int64_t b = first_block(z) + c;
        if (b < first_block(z + 1))
          (*table)[(size_t)((r * L + s) * grid + w)] = (int32_t)b;
      }
    }
  *segs_out = (int)Q;
  return true;
}

int spmv_zwalk_table_device(const spmv_hip_csr_plan* pl, int64_t rows,
                            int64_t d2, int grid, int segments, bool force,
                            int32_t** d_table, int* slots, int* segs)
{
  *d_table = nullptr;
  *slots = *segs = 0;
  SPMV_REQUIRE(rows > 0 && rows <= INT32_MAX && d2 > 0 && grid > 0
               && segments >= 0);
  std::vector<int32_t> table;
  if (!zwalk_table((int32_t)rows, d2, grid, segments, force, &table, segs))
    return SPMV_HIP_OK;
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  SPMV_CHECK_HIP(hipMalloc(d_table, sizeof(int32_t) * table.size()));
  hipError_t e = hipMemcpy(*d_table, table.data(), sizeof(int32_t) * table.size(),
                           hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(*d_table);
    *d_table = nullptr;
    return static_cast<int>(e);
  }
  *slots = (int)table.size();
  return SPMV_HIP_OK;
}

int spmv_zwalk_order_build(spmv_hip_csr_plan* pl, int64_t d2, int grid,
                           int segments, bool force)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  if (pl->zw_table) {
    SPMV_CHECK_HIP(hipDeviceSynchronize()); // no launch still reads the old one
    spmv_zwalk_free(pl);
  }
  SPMV_REQUIRE(d2 > 0 && grid > 0 && segments >= 0);
  pl->zw_d2 = d2;
  std::vector<int32_t> table;
  int segs = 0;
  if (!zwalk_table(pl->num_rows, d2, grid, segments, force, &table, &segs))
    return SPMV_HIP_OK;
  SPMV_CHECK_HIP(hipMalloc(&pl->zw_table, sizeof(int32_t) * table.size()));
  hipError_t e = hipMemcpy(pl->zw_table, table.data(),
                           sizeof(int32_t) * table.size(),
                           hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    spmv_zwalk_free(pl);
    return static_cast<int>(e);
  }
  pl->zw_slots = (int)table.size();
  pl->zw_grid = grid;
  pl->zw_segments = segs;
  pl->zwalk = 1;
  return SPMV_HIP_OK;
}

// Symmetric storage of a matrix without lattice structure (FEM matrices, what
// read_petsc_binary_matrix delivers with symmetric = true): the reference's
// loop (csr_kernels.cpp:26-40) seen from the row, in the sliced jagged form of
// the MERGED matrix -- per row its stored lower entries, then the entries of
// its column in the reference's order (the transposed map has them).
// ENOTSUP: the form does not apply (the transposed-map kernel stays).
template <typename T>
static int sym_sj_bake(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan, const T* values,
                       const T* diagonal, hipStream_t st)
{
  auto bake = [&](spmv_hip_csr_plan* p, const T* v, const int32_t* map) {
    if constexpr (sizeof(T) == 8)
      return spmv_sjds_bake_f64(p, v, map, st);
    else
      return spmv_sjds_bake_f32(p, v, map, st);
  };
  if (values == nullptr) { // drop the copy
    plan->sym_sj = 0;
    plan->sj = 0;
    plan->sj_diag0 = nullptr;
    return plan->sjt && plan->sjt->sj_lenperm ? bake(plan->sjt, nullptr, nullptr)
                                              : SPMV_HIP_ENOTSUP;
  }
  if (!plan->symmetric || !plan->sym_det || !plan->t_ptr || plan->slat
      || plan->nnz < ctx->sj_min_nnz || plan->num_rows < 64 || !diagonal)
    return SPMV_HIP_ENOTSUP;
  const auto t0 = std::chrono::steady_clock::now();
  if (!plan->sjt) {
    // the structure: the long rows' list (parent), the merged matrix sliced
    // jagged (child) -- spmv_sjds_plan.hip
    const int rs = spmv_sjds_sym_build(ctx, plan, st);
    if (rs != SPMV_HIP_OK)
      return rs;
  }
  plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                       std::chrono::steady_clock::now() - t0)
                       .count();
  const int rc = bake(plan->sjt, values, plan->sjv_map);
  plan->plan_us += plan->sjt->plan_us; // (the bake counted itself there)
  plan->sjt->plan_us = 0;
  if (rc != SPMV_HIP_OK) {
    plan->sym_sj = 0;
    return rc;
  }
  plan->sj_diag0 = diagonal;
  plan->sym_sj = 1;
  plan->sj = 1;
  return SPMV_HIP_OK;
}


// the arrays a launch with the baked pointers does not read (plan_owns_matrix)
static int plan_owned_mask(const spmv_hip_csr_plan* pl)
{
  if (pl->nnz == 0)
    return 0;
  // symmetric storage in the merged sliced jagged form, no long rows (those
  // are streamed from the caller's arrays): the kernel reads the merged copy,
  // the caller's row pointer and diagonal
  if (pl->symmetric)
    return pl->sym_det && pl->sym_sj && pl->sj && pl->sjt && pl->sjt->sj_val
                   && pl->sjt->sj_values0 && pl->sj_nlong == 0 && pl->num_cols >= 2
               ? 3
               : 0;
  // an fp32 twin for the mixed SpMV: its launches may fall back to CSR order
  if (pl->sdia32_val || pl->wdia32_val || pl->sj_val32)
    return 0;
  if (pl->sdia && pl->sdia_val && pl->sdia_general && pl->sdia_values0)
    return 3;
  if (pl->wdia && pl->wdia_val && pl->wdia_values0)
    return 3;
  if (pl->sj && pl->sj_val && pl->sj_values0 && pl->sj_nlong == 0 && pl->num_cols >= 2)
    return 3;
  return 0;
}

// does a launch with these operands take the form that owns the matrix?
template <typename T>
static bool released_launch_ok(const spmv_hip_csr_plan* pl, const T* values,
                               const T* in, const T* diagonal)
{
  if (pl->algo != SPMV_HIP_ALGO_ROWBLOCK || plan_owned_mask(pl) == 0)
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

extern "C" {

int spmv_hip_csr_plan_create(spmv_hip_ctx* ctx, int32_t num_rows,
                             int32_t num_cols, int64_t num_non_zeros,
                             const int32_t* rowptr, const int32_t* colind,
                             int symmetric, int algo, spmv_hip_csr_plan** plan)
{
  SPMV_REQUIRE(ctx && plan && num_rows >= 0 && num_cols >= 0
               && num_non_zeros >= 0);
  SPMV_REQUIRE(num_non_zeros == 0 || (rowptr && colind));
  // (the plan's clock counts the plan's work, not kernels of the caller still
  // running on the stream -- a device-side generator's fill, say)
  if (num_non_zeros > 0) {
    SPMV_SET_DEVICE(ctx);
    SPMV_CHECK_HIP(hipStreamSynchronize(spmv_stream(ctx, nullptr)));
  }
  const auto t_begin = std::chrono::steady_clock::now();
  // rowptr is int32 in the reference format (csr_kernels.h:28)
  if (num_non_zeros > INT32_MAX)
    return SPMV_HIP_ERANGE;
  spmv_hip_csr_plan* pl = new (std::nothrow) spmv_hip_csr_plan;
  if (!pl)
    return SPMV_HIP_ENOMEM;
  pl->ctx = ctx;
  pl->num_rows = num_rows;
  pl->num_cols = num_cols;
  pl->nnz = num_non_zeros;
  pl->symmetric = symmetric != 0;
  pl->rowptr0 = rowptr;
  pl->colind0 = colind;
  const double avg = num_rows > 0 ? (double)num_non_zeros / num_rows : 0.0;
  if (algo == SPMV_HIP_ALGO_AUTO) {
    // fewer entries than a quarter of the rows: most rows are empty, walk
    // only the non-empty ones (the remote block of a partitioned matrix)
    if (!symmetric && num_non_zeros > 0 && num_non_zeros * 4 < num_rows)
      algo = SPMV_HIP_ALGO_ROWLIST;
    else // long rows: the sliced jagged form (below) where it is built,
         // else a sub-wavefront per row
      algo = (avg <= 64.0 || num_non_zeros >= ctx->sj_min_nnz)
                 ? SPMV_HIP_ALGO_ROWBLOCK
                 : SPMV_HIP_ALGO_VECTOR;
  }
  if (algo < SPMV_HIP_ALGO_ROWBLOCK || algo > SPMV_HIP_ALGO_ROWLIST
      || (algo == SPMV_HIP_ALGO_ROWLIST && (symmetric || num_non_zeros == 0))) {
    delete pl;
    return SPMV_HIP_EINVAL;
  }
  pl->algo = algo;
  if (algo == SPMV_HIP_ALGO_ROWLIST) {
    int rc = build_row_list(pl, rowptr);
    if (rc != SPMV_HIP_OK) {
      delete pl;
      return rc;
    }
  }
  int lpr = 4;
  while (lpr < 64 && lpr < avg / 2)
    lpr *= 2;
  pl->lanes_per_row = lpr;
  // Non-temporal matrix loads keep the read-once stream out of the caches so
  // that x stays resident; measured +5 % when x fits the 256 MiB Infinity
  // Cache with room to spare (216^3) and -3 % when it does not (512^3).
  pl->nontemporal = ((int64_t)num_cols * 8 <= (int64_t)128 << 20) ? 1 : 0;
  if (!symmetric && algo == SPMV_HIP_ALGO_ROWBLOCK) {
    // Lattice form first (spmv_lat.hip): when every row block's columns are
    // row + one of <= 8 constant offsets the kernel needs no index stream at
    // all.  Otherwise the LX form: from ctx->lx_min_nnz entries on (set-up
    // time, +2 B per entry of memory) and rows short enough for the plan
    // kernel's sort; measured faster than the gather kernel at every size
    // from 128^3 to 512^3.
    int rc = SPMV_HIP_OK;
    if (num_non_zeros >= ctx->lat_min_nnz && avg <= 8.0)
      rc = spmv_lat_build(pl, rowptr, colind);
    // (ctx option "csr_in_place": the plan makes no copy of the index or value
    // stream -- neither the LX form's offsets nor the sliced jagged arrays)
    const bool copies = !ctx->csr_in_place;
    if (rc == SPMV_HIP_OK && !pl->lat && copies && num_non_zeros >= ctx->lx_min_nnz
        && avg <= 16.0 && (int64_t)num_cols * 8 <= ctx->lx_max_x_bytes)
      rc = build_lx(pl, rowptr, colind);
    // Neither: the sliced jagged form (spmv_sjds.hip) -- ragged rows, more
    // than 16 entries per row, column windows too wide for the LX form -- is
    // built by plan_bake_values, structure and values together, once the
    // diagonal forms have refused the matrix (a 27-point stencil has 27
    // entries per row too, and its analysis would be 50 ms for nothing).
    pl->sj_wanted
        = !pl->lat && !pl->lx && copies && num_non_zeros >= ctx->sj_min_nnz;
    // Neither of them and no sliced jagged form to come: the caller's CSR
    // arrays as they are.  From ctx->xw_min_nnz entries on the XW kernel
    // (spmv_lxw.hip): values and the 32-bit column indices by LDS-DMA, the x
    // windows of every row block staged -- the gather kernel fetched x across
    // the fabric 3.5 times at 512^3 -- in the plane-walk order when the
    // matrix sits on a 3-D grid.  With default options a matrix XW can stage
    // is one the LX form can stage too (the same window analysis, 16 windows
    // against 8), so XW is what "csr_in_place" plans get, what is left when
    // the LX form's 2 B per entry could not be allocated, and -- below, in
    // plan_bake_values -- what a plan whose sliced jagged form was declined
    // runs instead of the gather kernel.
    if (rc == SPMV_HIP_OK && !pl->sj_wanted && xw_applies(pl))
      rc = build_xw_and_walk(pl, rowptr, colind);
    if (rc != SPMV_HIP_OK) {
      spmv_hip_csr_plan_destroy(pl);
      return rc;
    }
  }
  if (symmetric && num_non_zeros > 0) {
    // atomic-free, bit-exact forms (the default when the block is strictly
    // lower triangular): the symmetric lattice form when the matrix has it,
    // else the transposed map
    int rc = SPMV_HIP_OK;
    const bool trace = getenv("SPMV_PLAN_TRACE") != nullptr;
    auto mark = [&](const char* what) {
      if (trace)
        fprintf(stderr, "plan_create %-10s %8.3f ms\n", what,
                std::chrono::duration<double, std::milli>(
                    std::chrono::steady_clock::now() - t_begin)
                    .count());
    };
    mark("begin");
    if (num_non_zeros >= ctx->lat_min_nnz)
      rc = spmv_slat_build(pl, rowptr, colind);
    mark("slat");
    if (rc == SPMV_HIP_OK && !pl->slat)
      rc = spmv_symt_build(pl, rowptr, colind);
    mark("symt");
    if (rc != SPMV_HIP_OK) {
      spmv_hip_csr_plan_destroy(pl);
      return rc;
    }
  }
  // what the analysis cost (every builder has synchronised its stream)
  pl->plan_us = (int)std::chrono::duration_cast<std::chrono::microseconds>(
                    std::chrono::steady_clock::now() - t_begin)
                    .count();
  *plan = pl;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_destroy(spmv_hip_csr_plan* plan)
{
  if (plan && plan->sjt) {
    (void)hipSetDevice(plan->ctx->device);
    spmv_sjds_free(plan->sjt);
    delete plan->sjt;
    plan->sjt = nullptr;
    (void)hipFree(plan->sjv_ptr);
    (void)hipFree(plan->sjv_col);
    (void)hipFree(plan->sjv_map);
    plan->sjv_ptr = plan->sjv_col = plan->sjv_map = nullptr;
  }
  if (plan
      && (plan->row_list || plan->lx_lidx || plan->lat_tab || plan->t_ptr
          || plan->slat_mask || plan->zw_table || plan->wdia_val
          || plan->sdia_val || plan->sj_lenperm || plan->xw_rec)) {
    (void)hipSetDevice(plan->ctx->device);
    (void)hipFree(plan->row_list);
    free_lx(plan);
    free_xw(plan);
    spmv_lat_free(plan);
    spmv_symt_free(plan);
    spmv_sdia_free(plan);
    spmv_wdia_free(plan);
    spmv_sjds_free(plan);
    spmv_slat_free(plan);
    spmv_zwalk_free(plan);
  }
  delete plan;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_bake_values_f64(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const double* values,
                                      const double* diagonal, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  if (values) // (as in plan_create: the caller's kernels are not plan time)
    SPMV_CHECK_HIP(hipStreamSynchronize(st));
  int rc = spmv_sdia_bake_f64(plan, values, diagonal, st);
  // a general matrix the diagonal form refuses (more than three lower
  // offsets, no lattice form): the wide diagonal form, up to 32 diagonals
  if (!plan->symmetric && (rc == SPMV_HIP_ENOTSUP || values == nullptr)) {
    const int rw = spmv_wdia_bake_f64(plan, values, st);
    rc = values == nullptr ? (rw != SPMV_HIP_OK ? rw : rc) : rw;
  } else if (!plan->symmetric && rc == SPMV_HIP_OK) {
    (void)spmv_wdia_bake_f64(plan, nullptr, st); // superseded
  }
  // a plan in the sliced jagged form keeps its own copy of the values in that
  // order (the diagonal forms never coexist with it)
  // (not from plan_values_changed: that call builds no new form and allocates
  // nothing -- a matrix the diagonal forms no longer hold goes back to the
  // CSR-order kernels, as its contract says)
  if (!plan->symmetric && values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted
      && !plan->sj_lenperm && !plan->no_new_forms) {
    // the structure of the sliced jagged form, now that it is known to be used
    const auto t0 = std::chrono::steady_clock::now();
    const int rb = spmv_sjds_build(plan, plan->rowptr0, plan->colind0,
                                   ctx->sj_wpb, ctx->sj_unit, 0);
    if (rb != SPMV_HIP_OK)
      return rb;
    plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                         std::chrono::steady_clock::now() - t0)
                         .count();
  }
  if (!plan->symmetric && plan->sj_lenperm
      && (values == nullptr || rc == SPMV_HIP_ENOTSUP)) {
    const int rj = spmv_sjds_bake_f64(plan, values, nullptr, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  // the sliced jagged form was wanted and could not be had (rows too long for
  // its length field, no memory for the copy): stage the x windows over the
  // caller's arrays rather than gather, where that applies
  if (values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted && !plan->sj_lenperm
      && !plan->no_new_forms) {
    plan->sj_wanted = false; // (declined: later bakes do not analyse it again)
    if (!plan->xw_rec && xw_applies(plan)) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rx = build_xw_and_walk(plan, plan->rowptr0, plan->colind0);
      if (rx != SPMV_HIP_OK)
        return rx;
      plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                           std::chrono::steady_clock::now() - t0)
                           .count();
    }
  }
  // symmetric storage without lattice structure: both blocks sliced jagged
  if (plan->symmetric && (values == nullptr ? plan->sjt != nullptr
                                            : rc == SPMV_HIP_ENOTSUP)) {
    const int rj = sym_sj_bake<double>(ctx, plan, values, diagonal, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  return rc;
}

int spmv_hip_csr_plan_bake_values_f32(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                      const float* values, const float* diagonal,
                                      void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  if (values)
    SPMV_CHECK_HIP(hipStreamSynchronize(st));
  int rc = spmv_sdia_bake_f32(plan, values, diagonal, st);
  if (!plan->symmetric && (rc == SPMV_HIP_ENOTSUP || values == nullptr)) {
    const int rw = spmv_wdia_bake_f32(plan, values, st);
    rc = values == nullptr ? (rw != SPMV_HIP_OK ? rw : rc) : rw;
  } else if (!plan->symmetric && rc == SPMV_HIP_OK) {
    (void)spmv_wdia_bake_f32(plan, nullptr, st);
  }
  // (not from plan_values_changed: that call builds no new form and allocates
  // nothing -- a matrix the diagonal forms no longer hold goes back to the
  // CSR-order kernels, as its contract says)
  if (!plan->symmetric && values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted
      && !plan->sj_lenperm && !plan->no_new_forms) {
    // the structure of the sliced jagged form, now that it is known to be used
    const auto t0 = std::chrono::steady_clock::now();
    const int rb = spmv_sjds_build(plan, plan->rowptr0, plan->colind0,
                                   ctx->sj_wpb, ctx->sj_unit, 0);
    if (rb != SPMV_HIP_OK)
      return rb;
    plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                         std::chrono::steady_clock::now() - t0)
                         .count();
  }
  if (!plan->symmetric && plan->sj_lenperm
      && (values == nullptr || rc == SPMV_HIP_ENOTSUP)) {
    const int rj = spmv_sjds_bake_f32(plan, values, nullptr, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  // the sliced jagged form was wanted and could not be had (rows too long for
  // its length field, no memory for the copy): stage the x windows over the
  // caller's arrays rather than gather, where that applies
  if (values && rc == SPMV_HIP_ENOTSUP && plan->sj_wanted && !plan->sj_lenperm
      && !plan->no_new_forms) {
    plan->sj_wanted = false; // (declined: later bakes do not analyse it again)
    if (!plan->xw_rec && xw_applies(plan)) {
      const auto t0 = std::chrono::steady_clock::now();
      const int rx = build_xw_and_walk(plan, plan->rowptr0, plan->colind0);
      if (rx != SPMV_HIP_OK)
        return rx;
      plan->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                           std::chrono::steady_clock::now() - t0)
                           .count();
    }
  }
  if (plan->symmetric && (values == nullptr ? plan->sjt != nullptr
                                            : rc == SPMV_HIP_ENOTSUP)) {
    const int rj = sym_sj_bake<float>(ctx, plan, values, diagonal, st);
    rc = values == nullptr ? (rj != SPMV_HIP_OK ? rj : rc) : rj;
  }
  return rc;
}

int spmv_hip_csr_plan_bake_values_f32f64(spmv_hip_ctx* ctx,
                                         spmv_hip_csr_plan* plan,
                                         const float* values32, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  SPMV_REQUIRE(!plan->released); // (its source arrays were given up)
  hipStream_t st = spmv_stream(ctx, stream);
  // whichever form holds the fp64 values (by offset, in jagged order) gets
  // its fp32 twin
  if (plan->sj_val && !plan->symmetric && !plan->sdia_val && !plan->wdia_val)
    return spmv_sjds_bake_f32f64(plan, values32, st);
  if (plan->wdia_val && !plan->sdia_val)
    return spmv_wdia_bake_f32f64(plan, values32, st);
  if (values32 == nullptr) {
    (void)spmv_wdia_bake_f32f64(plan, nullptr, st);
    (void)spmv_sjds_bake_f32f64(plan, nullptr, st);
  }
  return spmv_sdia_bake_f32f64(plan, values32, st);
}

int spmv_hip_csr_plan_owns_matrix(const spmv_hip_csr_plan* plan, int* mask)
{
  SPMV_REQUIRE(plan && mask);
  *mask = plan_owned_mask(plan);
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_release_matrix(spmv_hip_csr_plan* plan, int mask)
{
  SPMV_REQUIRE(plan && mask >= 0 && (mask & ~plan_owned_mask(plan)) == 0);
  plan->released |= mask;
  if (plan->symmetric && plan->released == 3 && plan->sjt) {
    // what only the refused paths would read goes too: the transposed map (the
    // fallback kernel) and the positions of the merged values (values_changed)
    // -- 16 B per stored entry: symmetric storage then holds the merged copy,
    // the row pointer and the diagonal, 1.75 times its own CSR bytes
    SPMV_CHECK_HIP(hipSetDevice(plan->ctx->device));
    SPMV_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(plan->t_pos);
    (void)hipFree(plan->t_row);
    plan->t_pos = plan->t_row = nullptr;
    (void)hipFree(plan->sjv_map);
    plan->sjv_map = nullptr;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_values_changed(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan,
                                     void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(plan && plan->ctx == ctx);
  // (the arrays the copies would be refreshed from are gone)
  SPMV_REQUIRE(!plan->released);
  hipStream_t st = spmv_stream(ctx, stream);
  const auto t_begin = std::chrono::steady_clock::now();
  const int plan_us0 = plan->plan_us;
  int rc = SPMV_HIP_OK;
  // the sliced jagged copy: rewritten in place
  if (plan->sj_val && plan->sj_values0) {
    rc = plan->sj_elem == 8
             ? spmv_sjds_bake_f64(plan, static_cast<const double*>(plan->sj_values0),
                                  nullptr, st)
             : spmv_sjds_bake_f32(plan, static_cast<const float*>(plan->sj_values0),
                                  nullptr, st);
    // (the fp32 twin of the mixed SpMV)
    if (rc == SPMV_HIP_OK && plan->sj_val32 && plan->sj32_values0)
      rc = spmv_sjds_bake_f32f64(plan, static_cast<const float*>(plan->sj32_values0),
                                 st);
  }
  // (symmetric storage: the merged matrix's copy)
  if (rc == SPMV_HIP_OK && plan->sjt && plan->sjt->sj_val && plan->sjt->sj_values0) {
    const void* v0 = plan->sjt->sj_values0;
    rc = plan->sjt->sj_elem == 8
             ? spmv_sjds_bake_f64(plan->sjt, static_cast<const double*>(v0),
                                  plan->sjv_map, st)
             : spmv_sjds_bake_f32(plan->sjt, static_cast<const float*>(v0),
                                  plan->sjv_map, st);
  }
  // the diagonal forms: the device checks decide the form again (a matrix
  // they no longer hold: ENOTSUP = back to the CSR-order kernels, which is a
  // correct outcome of this call)
  if (rc == SPMV_HIP_OK && (plan->sdia_val || plan->wdia_val)) {
    struct NoNewForms { // (reset on every path out of this block)
      spmv_hip_csr_plan* p;
      ~NoNewForms() { p->no_new_forms = 0; }
    } guard{plan};
    plan->no_new_forms = 1;
    const void* v32 = plan->sdia32_values0 ? plan->sdia32_values0
                                           : plan->wdia32_values0;
    if (plan->sdia_val ? plan->sdia_elem == 8 : plan->wdia_elem == 8) {
      const double* v = static_cast<const double*>(
          plan->sdia_val ? plan->sdia_values0 : plan->wdia_values0);
      const double* d = static_cast<const double*>(plan->sdia_val ? plan->sdia_diag0
                                                                  : nullptr);
      rc = spmv_hip_csr_plan_bake_values_f64(ctx, plan, v, d, stream);
      if (rc == SPMV_HIP_OK && v32) {
        rc = spmv_hip_csr_plan_bake_values_f32f64(
            ctx, plan, static_cast<const float*>(v32), stream);
        if (rc == SPMV_HIP_ENOTSUP)
          rc = SPMV_HIP_OK;
      }
    } else {
      const float* v = static_cast<const float*>(
          plan->sdia_val ? plan->sdia_values0 : plan->wdia_values0);
      const float* d = static_cast<const float*>(plan->sdia_val ? plan->sdia_diag0
                                                                 : nullptr);
      rc = spmv_hip_csr_plan_bake_values_f32(ctx, plan, v, d, stream);
    }
    if (rc == SPMV_HIP_ENOTSUP)
      rc = SPMV_HIP_OK;
  }
  SPMV_CHECK_HIP(hipStreamSynchronize(st));
  plan->plan_us = plan_us0; // (the bakes added themselves: not plan creation)
  plan->values_changed_us
      = (int)std::chrono::duration_cast<std::chrono::microseconds>(
            std::chrono::steady_clock::now() - t_begin)
            .count();
  return rc;
}

int spmv_hip_zwalk_table(int32_t num_rows, int64_t plane_rows, int grid,
                         int segments, int32_t* table, int64_t capacity,
                         int64_t* num_slots, int* segments_out)
{
  SPMV_REQUIRE(num_rows > 0 && plane_rows > 0 && grid > 0 && segments >= 0
               && num_slots && segments_out);
  std::vector<int32_t> t;
  int segs = 0;
  if (!zwalk_table(num_rows, plane_rows, grid, segments, true, &t, &segs))
    return SPMV_HIP_ERANGE;
  *num_slots = (int64_t)t.size();
  *segments_out = segs;
  if (table) {
    SPMV_REQUIRE(capacity >= (int64_t)t.size());
    std::copy(t.begin(), t.end(), table);
  }
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_algo(const spmv_hip_csr_plan* plan, int* algo)
{
  SPMV_REQUIRE(plan && algo);
  *algo = plan->algo;
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_set(spmv_hip_csr_plan* plan, const char* key, int value)
{
  SPMV_REQUIRE(plan && key);
  // arrays given up (plan_release_matrix): no key may select a kernel that
  // would read them
  if (plan->released)
    for (const char* k : {"algo", "sdia", "wdia", "sjds", "lat", "lx", "lxw", "xw",
                          "sym_det", "slat"})
      SPMV_REQUIRE(strcmp(key, k) != 0);
  if (!strcmp(key, "algo")) {
    // ROWLIST needs the list built at plan creation
    SPMV_REQUIRE(value >= SPMV_HIP_ALGO_ROWBLOCK
                 && (value <= SPMV_HIP_ALGO_SCALAR
                     || (value == SPMV_HIP_ALGO_ROWLIST && plan->row_list)));
    plan->algo = value;
  } else if (!strcmp(key, "lanes_per_row")) {
    SPMV_REQUIRE(value == 4 || value == 8 || value == 16 || value == 32
                 || value == 64);
    plan->lanes_per_row = value;
  } else if (!strcmp(key, "chunks")) {
    SPMV_REQUIRE(value == 1 || value == 2 || value == 4);
    plan->chunks = value;
  } else if (!strcmp(key, "nontemporal")) {
    plan->nontemporal = value != 0;
  } else if (!strcmp(key, "xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->xcd_group = value;
  } else if (!strcmp(key, "sym_window")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096 && value % 256 == 0);
    plan->sym_window = value;
  } else if (!strcmp(key, "sym_rows")) {
    SPMV_REQUIRE(value == 512 || value == 1024 || value == 2048);
    plan->sym_rows = value;
  } else if (!strcmp(key, "blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->blocks_per_cu = value;
    if (plan->zw_table && !plan->lat_tab && !plan->symmetric
        && !plan->sdia_val) // (LX or plain row blocks) tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "lx")) {
    // 1 needs the LX form built at plan creation (or by "lx_build")
    SPMV_REQUIRE(value == 0 || plan->lx_lidx);
    plan->lx = value != 0;
  } else if (!strcmp(key, "lxw")) {
    // the LDS-DMA kernel of the LX form (needs its records: ctx "lx_dma")
    SPMV_REQUIRE(value == 0 || plan->lxw_rec);
    plan->lxw = value != 0;
    if (plan->zw_table && plan->lx_lidx && !plan->lat_tab) // another grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "xw")) {
    // the LDS-DMA kernel on the caller's CSR arrays (needs its records)
    SPMV_REQUIRE(value == 0 || plan->xw_rec);
    plan->xw = value != 0;
    if (plan->xw_probe && value) { // asked for by name: no probe decides
      plan->xw_probe->decided = 1;
      plan->xw_probe->use_xw = 1;
    }
    if (plan->zw_table && plan->xw_rec && !plan->lat_tab && !plan->lx_lidx)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "xw_probe")) {
    // 1: (re)start the choice between XW and the gather kernel by the next
    // four launches; 0: XW from here on
    SPMV_REQUIRE((value == 0 || value == 1) && plan->xw_rec);
    if (!plan->xw_probe)
      plan->xw_probe = new (std::nothrow) XwProbe;
    SPMV_REQUIRE(plan->xw_probe);
    xw_probe_drop_events(plan->xw_probe);
    *plan->xw_probe = XwProbe();
    plan->xw_probe->decided = value ? 0 : 1;
  } else if (!strcmp(key, "lxw_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 0 && value <= kBlocksPerCU);
    plan->lxw_blocks_per_cu = value;
    if (plan->zw_table && !plan->lat_tab
        && ((plan->lxw_rec && plan->lxw) || (plan->xw_rec && plan->xw)))
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "lx_chunks")) {
    SPMV_REQUIRE(value == 1 || value == 2);
    plan->lx_chunks = value;
  } else if (!strcmp(key, "nt_store")) {
    plan->nt_store = value != 0;
  } else if (!strcmp(key, "slat")) {
    // 1 needs the symmetric lattice form built at plan creation
    SPMV_REQUIRE(value == 0 || plan->slat_mask);
    plan->slat = value != 0;
  } else if (!strcmp(key, "sdia")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    SPMV_REQUIRE(value == 0 || plan->sdia_val);
    plan->sdia = value; // the CSR-order kernel has a grid of its own
    if (plan->zw_table && plan->sdia_val)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "sjds")) {
    SPMV_REQUIRE(value == 0 || plan->sj_val || (plan->sjt && plan->sjt->sj_val));
    plan->sj = value != 0;
  } else if (!strcmp(key, "sj_phases")) { // ablation for measurements only
    SPMV_REQUIRE(value >= 1 && value <= 3);
    plan->sj_phases = value;
  } else if (!strcmp(key, "sj_long_panels")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sj_long_panels = value;
  } else if (!strcmp(key, "sj_long_table")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sj_long_table = value;
  } else if (!strcmp(key, "sj_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 0 && value <= kBlocksPerCU);
    plan->sj_blocks_per_cu = value;
  } else if (!strcmp(key, "sj_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->sj_xcd_group = value;
  } else if (!strcmp(key, "wdia")) {
    SPMV_REQUIRE(value == 0 || plan->wdia_val);
    plan->wdia = value != 0;
  } else if (!strcmp(key, "wdia_box")) {
    // lines per lane of the constant 27-point box kernel (0 = general kernel)
    SPMV_REQUIRE((value == 0 || value == 2 || value == 4) && plan->wdia_val
                 && plan->wdia_const);
    return spmv_wdia_box_build(plan, value, 0, false);
  } else if (!strcmp(key, "wdia_hbox")) {
    // the marched kernel for the half form of a 27-point box (0 = the general
    // wide diagonal kernel)
    SPMV_REQUIRE((value == 0 || value == 1) && plan->wdia_val);
    const int rh = spmv_wdia_hbox_build(plan, value);
    if (rh != SPMV_HIP_OK)
      return rh;
    if (!plan->wdia_hbox && plan->wdia_d2 > 0 && !plan->wdia_zw_table)
      return spmv_wdia_walk_build(plan, 0, false); // the general kernel's order
    return SPMV_HIP_OK;
  } else if (!strcmp(key, "wdia_hbox_segs")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->wdia_hbox_segs = value;
  } else if (!strcmp(key, "wdia_box_segments")) {
    SPMV_REQUIRE(value >= 0 && plan->wdia_box > 1);
    return spmv_wdia_box_build(plan, plan->wdia_box, value, true);
  } else if (!strcmp(key, "wdia_box_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->wdia_box_blocks_per_cu = value;
    if (plan->wdia_box > 1)
      return spmv_wdia_box_build(plan, plan->wdia_box, 0,
                                 plan->wdia_box_table != nullptr);
  } else if (!strcmp(key, "wdia_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->wdia_blocks_per_cu = value;
    if (plan->wdia_val && plan->wdia_zw_table) // the table is tied to the grid
      return spmv_wdia_walk_build(plan, 0, true);
  } else if (!strcmp(key, "wdia_zwalk")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->wdia_zwalk = value;
  } else if (!strcmp(key, "wdia_zwalk_segments")) {
    // (re)build the wide diagonal form's plane-walk table, whatever the size
    SPMV_REQUIRE(value >= 0 && plan->wdia_val && plan->wdia_d2 > 0);
    return spmv_wdia_walk_build(plan, value, true);
  } else if (!strcmp(key, "wdia_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->wdia_xcd_group = value;
  } else if (!strcmp(key, "slat_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->slat_blocks_per_cu = value;
    if (plan->zw_table && (plan->slat_mask || plan->sdia_val)) // tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "sdia_tile")) {
    // lines per lane of the constant-diagonal kernel (1, 2, 4)
    SPMV_REQUIRE((value == 1 || value == 2 || value == 4) && plan->sdia_val
                 && plan->sdia_const);
    return spmv_sdia_tile_build(plan, value, 0, false);
  } else if (!strcmp(key, "sdia_tile_segments")) {
    SPMV_REQUIRE(value >= 0 && plan->sdia_tile > 1);
    return spmv_sdia_tile_build(plan, plan->sdia_tile, value, true);
  } else if (!strcmp(key, "sdia_tile_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->sdia_tile_blocks_per_cu = value;
    if (plan->sdia_tile > 1)
      return spmv_sdia_tile_build(plan, plan->sdia_tile, 0,
                                  plan->sdia_tile_table != nullptr);
  } else if (!strcmp(key, "sdia_nt")) {
    SPMV_REQUIRE(value >= 0 && value < 32);
    plan->sdia_nt = value;
  } else if (!strcmp(key, "sdia_chain")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->sdia_chain = value; // LDS footprint, hence the grid, may change
    if (plan->zw_table && plan->sdia_val)
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else if (!strcmp(key, "zwalk")) {
    SPMV_REQUIRE(value == 0 || plan->zw_table);
    plan->zwalk = value != 0;
  } else if (!strcmp(key, "zwalk_segments")) {
    // (re)build the plane-walk table of the plan's lattice kernel with `value`
    // runs along the plane axis (0 = choose), whatever the size
    SPMV_REQUIRE(value >= 0 && plan->zw_d2 > 0);
    return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), value,
                                  true);
  } else if (!strcmp(key, "sym_det")) {
    // 1 needs the transposed map built at plan creation
    SPMV_REQUIRE(value == 0 || plan->t_ptr);
    plan->sym_det = value != 0;
  } else if (!strcmp(key, "lat")) {
    // 1 needs the lattice form built at plan creation
    SPMV_REQUIRE(value == 0 || plan->lat_tab);
    plan->lat = value != 0;
  } else if (!strcmp(key, "lat_chain")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    plan->lat_chain = value;
  } else if (!strcmp(key, "lat_xcd_group")) {
    SPMV_REQUIRE(value >= 0 && value <= 4096);
    plan->lat_xcd_group = value;
  } else if (!strcmp(key, "lat_blocks_per_cu")) {
    SPMV_REQUIRE(value >= 1 && value <= kBlocksPerCU);
    plan->lat_blocks_per_cu = value;
    if (plan->zw_table && plan->lat_tab) // the table is tied to the grid
      return spmv_zwalk_order_build(plan, plan->zw_d2, spmv_walk_grid(plan), 0,
                                    true);
  } else {
    return SPMV_HIP_EINVAL;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_csr_plan_get(const spmv_hip_csr_plan* plan, const char* key,
                          int* value)
{
  SPMV_REQUIRE(plan && key && value);
  if (!strcmp(key, "algo"))
    *value = plan->algo;
  else if (!strcmp(key, "sym_det"))
    *value = plan->sym_det;
  else if (!strcmp(key, "slat"))
    *value = plan->slat;
  else if (!strcmp(key, "sdia"))
    *value = plan->sdia && plan->sdia_val ? 1 : 0;
  else if (!strcmp(key, "sjds"))
    *value = plan->sj && (plan->sj_val || (plan->sjt && plan->sjt->sj_val)) ? 1 : 0;
  else if (!strcmp(key, "sym_sj")) // symmetric storage, both blocks sliced jagged
    *value = plan->symmetric && plan->sym_sj && plan->sj && plan->sjt
                     && plan->sjt->sj_val
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_mixed")) // the fp32 twin of the jagged copy is baked
    *value = plan->sj && plan->sj_val32 ? 1 : 0;
  else if (!strcmp(key, "sj_built"))
    *value = plan->sj_lenperm || (plan->sjt && plan->sjt->sj_lenperm) ? 1 : 0;
  else if (!strcmp(key, "sj_wpb")) // (symmetric storage: the merged matrix's)
    *value = plan->sj_lenperm ? plan->sj_wpb
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_wpb : 0;
  else if (!strcmp(key, "sj_sigma"))
    *value = plan->sj_lenperm ? plan->sj_sigma
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_sigma : 0;
  else if (!strcmp(key, "sj_unit"))
    *value = plan->sj_lenperm ? plan->sj_unit
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_unit : 0;
  else if (!strcmp(key, "sj_max_chunks"))
    *value = plan->sj_lenperm ? plan->sj_maxk
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_maxk : 0;
  else if (!strcmp(key, "sj_far_permille")) {
    // (symmetric storage: of the merged matrix)
    const spmv_hip_csr_plan* c
        = plan->sj_lenperm ? plan : (plan->sjt && plan->sjt->sj_lenperm ? plan->sjt : nullptr);
    *value = c && c->nnz > 0 ? (int)((c->sj_far * 1000 + c->nnz - 1) / c->nnz) : 0;
  }
  else if (!strcmp(key, "sj_staged_bytes_per_entry_x100")) {
    const spmv_hip_csr_plan* c
        = plan->sj_lenperm ? plan : (plan->sjt && plan->sjt->sj_lenperm ? plan->sjt : nullptr);
    *value = c && c->nnz > 0 ? (int)(c->sj_sumk * 12800 / c->nnz) : 0;
  }
  else if (!strcmp(key, "sj_pad_permille")) // entries of padding per 1000 stored
    *value = plan->sj_lenperm && plan->nnz > 0
                 ? (int)((plan->sj_units * plan->sj_unit * 1000) / plan->nnz)
                 : 0;
  // (symmetric storage: the long rows of the stored block are the PARENT's)
  else if (!strcmp(key, "sj_long_panels"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_long_sorted
                     && plan->sj_long_panels
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_sorted"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_nlong > 0
                     && plan->sj_long_sorted
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_table"))
    *value = (plan->sj_lenperm || plan->sjt) && plan->sj_long_sorted
                     && plan->sj_long_panels && plan->sj_lt_tab && plan->sj_long_table
                 ? 1
                 : 0;
  else if (!strcmp(key, "sj_long_table_kib"))
    *value = (int)((plan->sj_lt_entries * 4 + (int64_t)plan->sj_lt_nsg * 16) / 1024);
  else if (!strcmp(key, "sj_long_rows"))
    *value = (plan->sj_lenperm || plan->sjt) ? plan->sj_nlong : 0;
  else if (!strcmp(key, "sj_wide"))
    *value = plan->sj_lenperm ? plan->sj_wide_alloc
             : plan->sjt && plan->sjt->sj_lenperm ? plan->sjt->sj_wide_alloc : 0;
  else if (!strcmp(key, "sj_blocks_per_cu"))
    *value = plan->sj_blocks_per_cu;
  else if (!strcmp(key, "wdia"))
    *value = plan->wdia && plan->wdia_val ? 1 : 0;
  else if (!strcmp(key, "wdia_hbox"))
    *value = plan->wdia && plan->wdia_val ? plan->wdia_hbox : 0;
  else if (!strcmp(key, "wdia_offsets"))
    *value = plan->wdia_val ? plan->wdia_K : 0;
  else if (!strcmp(key, "xw"))
    *value = plan->xw && plan->xw_rec ? 1 : 0;
  else if (!strcmp(key, "xw_staged"))
    *value = plan->xw_staged;
  else if (!strcmp(key, "xw_pick")) // -1: the probe is still running
    *value = !(plan->xw && plan->xw_rec) ? 0
             : !plan->xw_probe           ? 1
             : plan->xw_probe->decided   ? plan->xw_probe->use_xw
                                         : -1;
  else if (!strcmp(key, "xw_probe_xw_us"))
    *value = plan->xw_probe ? (int)plan->xw_probe->us_xw : 0;
  else if (!strcmp(key, "xw_probe_gather_us"))
    *value = plan->xw_probe ? (int)plan->xw_probe->us_gather : 0;
  else if (!strcmp(key, "plan_us"))
    *value = plan->plan_us;
  else if (!strcmp(key, "values_changed_us"))
    *value = plan->values_changed_us;
  else if (!strcmp(key, "plan_kib")) {
    // device memory the plan owns beyond the caller's CSR arrays
    const int64_t n = plan->num_rows, nnz = plan->nnz;
    const int64_t nrb = (n + kRows - 1) / kRows;
    int64_t b = 0;
    if (plan->row_list)
      b += 4 * (int64_t)plan->num_listed;
    if (plan->lx_lidx)
      b += 2 * (nnz + 8) + 4 * nrb * kLxRec;
    if (plan->lxw_rec)
      b += 4 * nrb * kLxwRec;
    if (plan->xw_rec)
      b += 4 * nrb * kXwRec;
    if (plan->lat_tab)
      b += 48 * nrb + n;
    if (plan->slat_mask)
      b += n;
    const int64_t narr
        = plan->sdia_general == 2 ? 2 * plan->sdia_nd + 1 : plan->sdia_nd + 1;
    if (plan->sdia_val)
      b += narr * plan->sdia_len * plan->sdia_elem + n;
    if (plan->sdia32_val && !plan->sdia_const)
      b += narr * plan->sdia_len * 4 + n;
    if (plan->wdia_val)
      b += (int64_t)plan->wdia_narr * plan->wdia_len * plan->wdia_elem + 4 * n;
    if (plan->wdia32_val && !plan->wdia_const)
      b += (int64_t)plan->wdia_narr * plan->wdia_len * 4;
    if (plan->sj_lenperm)
      b += 4 * ((n + 63) / 64 * 64) + 8 * (int64_t)plan->sj_nblk
           + 4 * (int64_t)plan->sj_nblk * plan->sj_stride
           + (plan->sj_wide_alloc ? 4 : 2) * plan->sj_units * plan->sj_unit
           + 4 * (int64_t)plan->sj_nlong + 4 * ((n + 63) / 64 + 1)
           + 4 * plan->sj_lt_entries + 16 * (int64_t)plan->sj_lt_nsg
           + 2 * plan->sj_lt_codes_n + 8 * (int64_t)plan->sj_nlong;
    if (plan->sj_val)
      b += (int64_t)plan->sj_elem * plan->sj_units * plan->sj_unit;
    if (plan->sj_val32)
      b += 4 * plan->sj_units * plan->sj_unit;
    if (plan->sjt && plan->sjt->sj_lenperm) { // symmetric storage: the merged matrix
      const spmv_hip_csr_plan* c = plan->sjt;
      // its row pointer, the positions of its values (until released)
      b += 4 * (n + 1) + (plan->sjv_map ? 4 * c->nnz : 0);
      if (plan->sj_long_rows) // the stored block's long rows: list, table, codes
        b += 4 * (int64_t)plan->sj_nlong + 4 * plan->sj_lt_entries
             + 16 * (int64_t)plan->sj_lt_nsg + 2 * plan->sj_lt_codes_n
             + 8 * (int64_t)plan->sj_nlong;
      b += 4 * ((n + 63) / 64 * 64) + 8 * (int64_t)c->sj_nblk
           + 4 * (int64_t)c->sj_nblk * c->sj_stride
           + (c->sj_wide_alloc ? 4 : 2) * c->sj_units * c->sj_unit
           + 4 * ((n + 63) / 64 + 1);
      if (c->sj_val)
        b += (int64_t)c->sj_elem * c->sj_units * c->sj_unit;
    }
    if (plan->t_ptr)
      b += 4 * (n + 1) + (plan->t_row ? 8 * nnz : 0);
    if (plan->zw_table)
      b += 4 * (int64_t)plan->zw_slots;
    *value = (int)((b + 1023) / 1024);
  }
  else if (!strcmp(key, "sdia_offsets"))
    *value = plan->sdia_val ? plan->sdia_nd : 0;
  else if (!strcmp(key, "sdia_general"))
    *value = plan->sdia_val ? plan->sdia_general : 0;
  else if (!strcmp(key, "sdia_mixed"))
    *value = plan->sdia32_val ? 1 : 0;
  else if (!strcmp(key, "sdia_tile"))
    *value = plan->sdia_val ? plan->sdia_tile : 0;
  else if (!strcmp(key, "sdia_tile_walk"))
    *value = plan->sdia_val && plan->sdia_tile_table ? plan->sdia_tile_segments : 0;
  else if (!strcmp(key, "sdia_const"))
    *value = plan->sdia_val ? plan->sdia_const : 0;
  else if (!strcmp(key, "wdia_zwalk"))
    *value = plan->wdia_val && plan->wdia_zwalk && plan->wdia_zw_table ? 1 : 0;
  else if (!strcmp(key, "wdia_zwalk_segments"))
    *value = plan->wdia_zw_table ? plan->wdia_zw_segments : 0;
  else if (!strcmp(key, "wdia_d2"))
    *value = plan->wdia_val ? plan->wdia_d2 : 0;
  else if (!strcmp(key, "wdia_box"))
    *value = plan->wdia_val ? plan->wdia_box : 0;
  else if (!strcmp(key, "wdia_const"))
    *value = plan->wdia_val ? plan->wdia_const : 0;
  else if (!strcmp(key, "wdia_half"))
    *value = plan->wdia_val && !plan->wdia_const
                     && plan->wdia_narr < plan->wdia_K
                 ? 1
                 : 0;
  else if (!strcmp(key, "wdia_mixed"))
    *value = plan->wdia32_val ? 1 : 0;
  else if (!strcmp(key, "sdia_chain"))
    *value = plan->sdia_chain;
  else if (!strcmp(key, "sdia_nt"))
    *value = plan->sdia_nt;
  else if (!strcmp(key, "zwalk"))
    *value = plan->zwalk && plan->zw_table ? 1 : 0;
  else if (!strcmp(key, "zwalk_segments"))
    *value = plan->zw_table ? plan->zw_segments : 0;
  else if (!strcmp(key, "zwalk_grid"))
    *value = plan->zw_table ? plan->zw_grid : 0;
  else if (!strcmp(key, "lattice_d1"))
    *value = plan->lattice_d1;
  else if (!strcmp(key, "lattice_d2"))
    *value = plan->lattice_d2;
  else if (!strcmp(key, "lat"))
    *value = plan->lat;
  else if (!strcmp(key, "lat_chain"))
    *value = plan->lat_chain;
  else if (!strcmp(key, "lat_blocks"))
    *value = plan->lat_blocks;
  else if (!strcmp(key, "lx"))
    *value = plan->lx;
  else if (!strcmp(key, "lxw"))
    *value = plan->lxw && plan->lxw_rec ? 1 : 0;
  else if (!strcmp(key, "lx_staged"))
    *value = plan->lx_staged;
  else if (!strcmp(key, "lx_blocks"))
    *value = plan->lx_blocks;
  else if (!strcmp(key, "blocks_per_cu"))
    *value = plan->blocks_per_cu;
  else if (!strcmp(key, "nontemporal"))
    *value = plan->nontemporal;
  else
    return SPMV_HIP_EINVAL;
  return SPMV_HIP_OK;
}

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
