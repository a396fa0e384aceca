// Wide diagonal form of the general CSR SpMV (gfx950 / MI355X): matrices whose
// entries sit on at most 32 diagonals -- 2-D 9-point, 3-D 19- and 27-point
// stencils (HPCG's operator), any constant-offset stencil too wide for the
// lattice form (8 offsets per row block) and the diagonal form proper (3 lower
// offsets, spmv_symdia.hip).
//
// Stands behind the same CSRSpMV<T>::init/run hook as the other general
// kernels (spmv/csr_kernels.h:26-78); arithmetic and summation order are those
// of spmv/csr_kernels.cpp:41-51, so results are bit-identical to the oracle.
//
// spmv_hip_csr_plan_bake_values_* on a general plan that the diagonal form
// refused looks (on the device) for
//   * the set of distinct `col - row` over the whole matrix: at most 32,
//   * every row's entries in strictly ascending column order (then "walk the
//     set bits of the row's mask from k = 0 up" IS the row's left-to-right
//     order of csr_kernels.cpp:46-47),
//   * at least half of the rows x offsets slots filled,
// and keeps its own copy of the values BY OFFSET: K arrays v_k[i] = A(i, i +
// D_k), zero where a row has no such entry, plus one 32-bit presence mask per
// row.  The kernel then needs neither `colind` nor the row pointer: lane = row,
// every load -- v_k[i], x[i + D_k], the mask -- is coalesced and independent of
// every other, issued eight offsets at a time; 8 B per entry + 4 B per row
// instead of 12 B per entry + 4 B per row (27-point, 256^3: 3.96 GB per SpMV
// instead of 5.79 GB).  A product is added only where the mask bit is set: an
// absent entry contributes nothing, not 0 * x (x may be Inf or NaN there, and
// a sum of -0.0 must stay -0.0).
//
// No LDS, no barriers: the x loads of neighbouring offsets (dx = -1, 0, +1 of a
// stencil) hit the same lines in L1, those of neighbouring grid lines and
// planes the L2 -- the persistent grid sweeps the rows front by front.
//
// Measured (MI355X, same process, alternating): 27-point 256^3 (16.8 M rows,
// 449 M entries) 0.80-0.82 ms against 1.07 ms for the row-block gather kernel
// the matrix took before (160^3: 0.216 against 0.298); 3.96 GB per launch =
// 4.9 TB/s.  On the 7-point matrix the LDS-DMA diagonal forms stay ahead
// (512^3: 2.16-2.30 ms here against 1.74 for the full diagonal form and 1.32
// for the half form), which is why this form only takes what they refuse.
//
// Measured and dropped (round 4): x through an LDS ring of three plane windows
// (27-point 256^3 in the plane-walk order: a workgroup keeps the windows of two
// planes and loads one new window of 256 + 2 W contiguous elements per step,
// all 27 x operands from LDS) -- bit-exact, and 0.81-0.82 ms against 0.68-0.69:
// the two barriers per row block cost more than the x requests they save.
#include "csr_plan.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

namespace
{

struct WdiaOffsets {
  int32_t D[kWdiaMaxOff]; // ascending; D[k] = 0 beyond K
  // where the entry of offset k sits: array A[k], row i + S[k].  Full form:
  // A[k] = k, S[k] = 0.  HALF form (a matrix found symmetric bit for bit): only
  // the arrays of the offsets <= 0 exist, and the upper entry (i, i + d) is read
  // as the lower entry of row i + d: A[k] = the array of -d, S[k] = d.
  int32_t A[kWdiaMaxOff];
  int32_t S[kWdiaMaxOff];
};

// TV = type of the baked values (what is streamed), T = type of x, y and of
// the arithmetic.
// CONST: every diagonal is constant bit for bit -- the entry of offset k is
// cv.c[k] wherever the mask has it; no values are loaded (`sval` is unused).
struct WdiaConsts {
  double c[kWdiaMaxOff];
};

template <typename TV, typename T, bool DOT, bool CONST>
__global__ __launch_bounds__(kBlock) void csr_wdia_kernel(
    int32_t num_rows, int32_t num_cols, int64_t arr_len, int K, WdiaOffsets off,
    const TV* __restrict__ sval, const uint32_t* __restrict__ mask, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    RowBlockOrder ord, WdiaConsts cv)
{
  __shared__ double s_red[kBlock / 64];
  const int t = threadIdx.x;
  double dot_acc = 0.0;
  const int num_slots = order_slots(ord);
  for (int it = blockIdx.x; it < num_slots; it += gridDim.x) {
    const int rb = order_row_block(ord, it);
    if (rb < 0)
      continue; // uniform
    const int64_t i = (int64_t)rb * kRows + t;
    if (i >= num_rows)
      continue;
    const uint32_t m = mask[i];
    T y0 = T(0), x_own = T(0);
    if (beta != T(0))
      y0 = out[i];
    if constexpr (DOT)
      x_own = in[i];
    T sum = 0;
    // eight offsets at a time: sixteen independent loads in flight per lane
    for (int k0 = 0; k0 < K; k0 += 8) {
      T v[8], x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = T(0);
        x[j] = T(0);
        if (k < K) { // uniform
          // unconditional (no dependence on the mask load); a column the row
          // does not have is clamped into range and its product never added
          int64_t c = i + off.D[k];
          c = c < 0 ? 0 : (c >= num_cols ? (int64_t)num_cols - 1 : c);
          if constexpr (CONST) {
            v[j] = (T)cv.c[k];
          } else {
            int64_t r = i + off.S[k];
            r = r < arr_len ? r : arr_len - 1;
            v[j] = (T)sval[(int64_t)off.A[k] * arr_len + r];
          }
          x[j] = in[c];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((m >> (k0 + j)) & 1u) // csr_kernels.cpp:46-47, left to right
          sum += v[j] * x[j];
    }
    const T c = alpha * sum;
    T y = c;
    if (beta != T(0))
      y = c + beta * y0;
    out[i] = y;
    if constexpr (DOT)
      dot_acc += (double)x_own * (double)c;
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// Constant 27-point BOX stencils (offsets a P + b L + c, a, b, c in -1..1: P the
// plane and L the line distance; HPCG's operator), R lattice lines per lane.
// The kernel above asks the L2 for nine line sets of x per row block (three
// lines in each of three planes; the loads at c = -+1 hit the L1) and, like
// every kernel here, is bound by the requests a CU keeps in flight: 256^3 in
// 0.29 ms although only 0.4 GB cross the fabric.  Here a lane owns R rows one
// line apart and keeps, per plane, the R + 2 lines around them in registers
// (x at c = -1, 0, +1 each); walking from plane to plane it loads ONE plane of
// (R + 2) x 3 values per step and hands the other two on: (R + 2) / R line sets
// per row block instead of 9.  Index space and plane-walk table are those of
// csr_const_dia_tile_kernel (spmv_symdia.hip): work item j = tuple * L +
// position stands for the rows (tuple * R + r) * L + position.
// Rows are summed in ascending column order = (a, b, c) lexicographic, a term
// only where the row's mask has it: the bits of csr_kernels.cpp:41-51.
// ---------------------------------------------------------------------------
struct BoxGeom {
  int P, L;         // plane and line distance (rows)
  int64_t NJ;       // work items
  int block;        // work items per workgroup and step: 256, or the largest
                    // divisor of a plane's items in 192..256, so that planes
                    // are whole blocks and can be handed on
  int chain_blocks; // blocks of j-space per plane when whole, else 0
  double rcp_l;
};

template <typename T, bool DOT, bool TAB, int R>
__global__ __launch_bounds__(kBlock) void csr_box27_const_kernel(
    int32_t num_rows, const uint32_t* __restrict__ mask, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    RowBlockOrder ord, BoxGeom g, WdiaConsts cv)
{
  __shared__ double s_red[kBlock / 64];
  const int t = threadIdx.x;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  const int64_t last = (int64_t)num_rows - 1;
  double dot_acc = 0.0;
  // X[a][b][c]: x at plane a - 1, line b - 1 (relative to the lane's first
  // row), column offset c - 1
  T X[3][R + 2][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < R + 2; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        X[a][b][c] = T(0);
  auto at = [&](int64_t col) { // clamped: what a row does not have is not used
    return in[col < 0 ? 0 : (col > last ? last : col)];
  };
  auto load_plane = [&](int a, int64_t i0) {
#pragma unroll
    for (int b = 0; b < R + 2; ++b) {
      const int64_t base = i0 + (int64_t)(a - 1) * g.P + (int64_t)(b - 1) * g.L;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        X[a][b][c] = at(base + (c - 1));
    }
  };
  int it = blockIdx.x;
  int cur = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it, num_slots));
  int nxt_raw = order_slot_raw_t<TAB>(ord, it + stride, num_slots);
  int prev = -1; // block of the step before (its planes are in X when chained)
  while (it < num_slots) {
    const int nxt = order_slot_decode(ord, nxt_raw);
    nxt_raw = order_slot_raw_t<TAB>(ord, it + 2 * stride, num_slots);
    if (cur >= 0) { // uniform
      const bool chain = g.chain_blocks > 0 && prev >= 0
                         && cur - prev == g.chain_blocks;
      const int64_t j = (int64_t)cur * g.block + t;
      // tuple and position: j / L by reciprocal, one step of correction
      int64_t tup = (int64_t)((double)j * g.rcp_l);
      int64_t pos = j - tup * g.L;
      if (pos < 0) {
        --tup;
        pos += g.L;
      } else if (pos >= g.L) {
        ++tup;
        pos -= g.L;
      }
      const int64_t i0 = tup * R * g.L + pos;
      const bool live = t < g.block && j < g.NJ && i0 <= last;
      uint32_t m[R];
      T y0[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + (int64_t)r * g.L;
        m[r] = live && i <= last ? mask[i] : 0u;
        y0[r] = T(0);
        if (beta != T(0) && live && i <= last)
          y0[r] = out[i];
      }
      if (live) {
        if (!chain) { // uniform: a jump of the walk -- all three planes
          load_plane(0, i0);
          load_plane(1, i0);
        }
        load_plane(2, i0);
      }
      // waves whose rows all have all 27 entries (everything but the faces of
      // the grid): the same sums without the tests -- the kernel is bound by
      // its instructions now
      bool full = live;
#pragma unroll
      for (int r = 0; r < R; ++r)
        full = full && m[r] == 0x7ffffffu;
      if (__all(full)) { // uniform
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t i = i0 + (int64_t)r * g.L;
          T sum = 0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
              for (int c = 0; c < 3; ++c)
                sum += (T)cv.c[9 * a + 3 * b + c] * X[a][r + b][c];
          const T cy = alpha * sum;
          T y = cy;
          if (beta != T(0))
            y = cy + beta * y0[r];
          out[i] = y;
          if constexpr (DOT)
            dot_acc += (double)X[1][r + 1][1] * (double)cy;
        }
      } else
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t i = i0 + (int64_t)r * g.L;
        if (live && i <= last) {
          T sum = 0; // csr_kernels.cpp:45
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const int k = 9 * a + 3 * b + c;
                if ((m[r] >> k) & 1u) // :46-47, ascending column
                  sum += (T)cv.c[k] * X[a][r + b][c];
              }
          const T cy = alpha * sum; // :49
          T y = cy;
          if (beta != T(0))
            y = cy + beta * y0[r];
          out[i] = y;
          if constexpr (DOT)
            dot_acc += (double)X[1][r + 1][1] * (double)cy;
        }
      }
      // hand the planes on: what was the plane ahead is the row's own plane
      // one step later
#pragma unroll
      for (int b = 0; b < R + 2; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          X[0][b][c] = X[1][b][c];
          X[1][b][c] = X[2][b][c];
        }
    }
    prev = cur;
    cur = nxt;
    it += stride;
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// 27-point BOX stencils with VARYING coefficients in the HALF form (a matrix
// found symmetric bit for bit: 13 lower arrays + the diagonal), tiles of 1024
// rows marched DOWN the planes with the planes' values handed on through LDS.
//
// The general kernel above reads the upper entry (i, i + d) as the lower entry
// of row i + d.  For the nine offsets a plane ahead that is a second read of
// every element of nine arrays one plane-walk step after the first -- 4 MB per
// XCD and step of streamed values lie between the two, more than its L2 holds:
// 23 array reads per row cross the fabric instead of 14 (PMC: 3.09 GB at 256^3
// for 1.88 GB of values).  Here a workgroup (1024 lanes, lane = row, one per
// CU) owns the rows [s, s + 1024) of every plane of a run of planes and walks
// them from the top plane down:
//   * the lane's own 14 values of plane q are loaded ONCE, a step ahead; the
//     four in-plane arrays go to LDS at the start of step q (their upper use:
//     row i's entry (i, i + d) is the value of lane t + d), the nine
//     plane-to-plane arrays at its end -- step q - 1 reads them as ITS upper
//     entries (offset P + e: the value of lane t + e one plane up);
//   * what lies outside the tile (|e|, d <= L + 1 rows at one end) the lanes at
//     that end load for themselves: 4 + 4 registers, 9 (L + 1) / 1024 extra
//     array reads per row (2.3 at L = 256);
//   * x through a ring of three plane windows (1024 + 2 (L + 1) elements
//     each), one new window per step: all 27 operands from LDS.
// 16.3 array reads per row instead of 23, 1.5 instead of 3 of x.  Two barriers
// per step; the loads of the next plane (26 per lane) are in flight across
// them.  Rows are summed in ascending column order, a term only where the
// row's mask has it: the bits of csr_kernels.cpp:41-51.  Everything is
// addressed by INDEX (clamped where nobody needs it), so the result does not
// depend on the matrix actually being a box grid -- only the traffic does.
// ---------------------------------------------------------------------------
constexpr int kHbT = 1024; // rows per tile = lanes per workgroup

template <typename TV>
struct HalfBoxArrays {
  const TV* a[14]; // the 13 lower arrays and the diagonal, each a kernel argument
                   // of its own: scalar base + the lane's 32-bit byte offset
};

struct HalfBoxGeom {
  int P, L;    // plane and line distance (rows)
  int tiles;   // tiles per plane
  int planes;  // ceil(rows / P)
  int segs;    // runs of planes
  int seg_len; // planes per run
  int W;       // x window: kHbT + 2 (L + 1)
};

template <typename TV, typename T, bool DOT, bool BETA>
__global__ __launch_bounds__(kHbT) void csr_box27_half_kernel(
    int32_t num_rows, int32_t arr_len, HalfBoxArrays<TV> base_,
    const uint32_t* __restrict__ mask, T alpha, const T* __restrict__ in, T beta,
    T* __restrict__ out, DotOut dot, HalfBoxGeom g)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char hb_lds[];
  const TV* const* base = base_.a;
  TV* const s_far = reinterpret_cast<TV*>(hb_lds);        // [9][kHbT]: arrays 0..8
  TV* const s_near = s_far + 9 * kHbT;                    // [5][kHbT]: arrays 9..13
  T* const s_x = reinterpret_cast<T*>(s_near + 5 * kHbT); // [3][W]
  double* const s_red = reinterpret_cast<double*>(s_x + 3 * g.W); // [16]
  const int t = threadIdx.x;
  const int P = g.P, L = g.L, W = g.W;
  // (rows and array length below 2^29: every index fits an int, every byte
  // offset 32 bits -- one offset register serves the 14 loads of a row, the
  // arrays' bases are scalars)
  const int last_row = num_rows - 1;
  const int last_arr = arr_len - 1;
  double dot_acc = 0.0;
  // offsets of the nine positions of a plane, (b, c) lexicographic
  int eo[9];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      eo[3 * b + c] = (b - 1) * L + (c - 1);
  // what this lane loads for itself (outside the tile).  Upper far entries k =
  // 18 + m (m = 0..8): array 8 - m, shift eo[m]; the lanes at the low end lack m
  // = 0..3 (eo < 0), those at the high end m = 5..8.  Upper near entries k = 14
  // + j: array 12 - j, shift dn[j] = 1, L - 1, L, L + 1.
  const bool hi = __builtin_amdgcn_readfirstlane(t >> 9) != 0; // per wave
  int dn[4];
  bool he_need[4], hn_need[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = hi ? eo[5 + j] : eo[j];
    he_need[j] = hi ? (t + e >= kHbT) : (t + e < 0);
    dn[j] = eo[5 + j];
    hn_need[j] = t + dn[j] >= kHbT;
  }
  auto at = [](const auto* base, int idx) { // scalar base + 32-bit byte offset
    using E = std::remove_cv_t<std::remove_pointer_t<decltype(base)>>;
    const uint32_t off = (uint32_t)idx * (uint32_t)sizeof(E);
    return *reinterpret_cast<const E*>(reinterpret_cast<const char*>(base) + off);
  };
  auto clamp_arr = [&](int r) { return r < 0 ? 0 : (r > last_arr ? last_arr : r); };
  auto clamp_col = [&](int c) { return c < 0 ? 0 : (c > last_row ? last_row : c); };
  // the loads of step q, in two groups.  Main: own values of plane q, x window
  // of plane q - 1, mask.  Edge: outside far values of plane q + 1, outside
  // near values of plane q (a lane that needs none reads element 0).
  auto load_main = [&](int q, int s, bool last, TV (&C)[14], T (&XW)[2],
                       uint32_t& m) {
    const int row = last ? 0 : q * P + s + t;
    const int r = clamp_arr(row);
#pragma unroll
    for (int a = 0; a < 14; ++a)
      C[a] = at(base[a], r);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int w = t + j * kHbT;
      XW[j] = at(in, last ? 0 : clamp_col((q - 1) * P + s - (L + 1) + w));
    }
    m = at(mask, clamp_col(row));
  };
  auto load_edge = [&](int q, int s, bool last, TV (&HE)[4], TV (&HN)[4]) {
    const int row = q * P + s + t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = hi ? eo[5 + j] : eo[j];
      // (a select, not a branch around the load: told to the compiler by hiding
      // the index)
      int ie = he_need[j] && !last ? clamp_arr(row + P + e) : 0;
      int in_ = hn_need[j] && !last ? clamp_arr(row + dn[j]) : 0;
      asm volatile("" : "+v"(ie), "+v"(in_));
      HE[j] = at(hi ? base[3 - j] : base[8 - j], ie);
      HN[j] = at(base[12 - j], in_);
    }
  };
  // One step: the set `N` (loaded a step ago) becomes the step's own; the
  // in-plane arrays, the diagonal and the x window of plane q - 1 go to LDS;
  // the loads of step q - 1 are issued; the rows of plane q are computed
  // (COMPUTE; the first step of a run only prepares); the plane-to-plane
  // arrays go to LDS for the step below.  Straight-line code, every load
  // unconditional (`last`: nothing below to load -- element 0): the compiler's
  // counts of loads in flight stay exact and no wait covers more than it must.
  TV C[14], N[14], HE[4], HN[4];
  T XW[2];
  uint32_t mN;
  auto step = [&](int q, int s, bool last, auto compute_tag) {
    constexpr bool COMPUTE = decltype(compute_tag)::value;
#pragma unroll
    for (int a = 0; a < 14; ++a)
      C[a] = N[a];
    const uint32_t mq = mN;
#pragma unroll
    for (int a = 0; a < 5; ++a)
      s_near[a * kHbT + t] = C[9 + a];
    {
      T* const ring = s_x + ((q + 2) % 3) * W; // (q - 1) mod 3
      ring[t] = XW[0];
      if (t + kHbT < W)
        ring[t + kHbT] = XW[1];
    }
    __syncthreads();
    const int i = q * P + s + t;
    const bool live = s + t < P && i <= last_row;
    T y0 = T(0);
    if constexpr (COMPUTE && BETA) {
      // (before the loads of the next step and waited for at once: the counter
      // of loads in flight is in order)
      y0 = at(out, clamp_col(i));
      asm volatile("" : "+v"(y0));
    }
    load_main(q - 1, s, last, N, XW, mN);
    if constexpr (COMPUTE) {
      const uint32_t mm = live ? mq : 0u;
      const T* const xl = s_x + ((q + 2) % 3) * W + t + L + 1; // plane q - 1
      const T* const xc = s_x + (q % 3) * W + t + L + 1;
      const T* const xu = s_x + ((q + 1) % 3) * W + t + L + 1;
      T sum = 0; // csr_kernels.cpp:45
      // (the groups one after the other, each sum finished before the next
      // group's LDS reads: with all 41 reads of a row hoisted to the top the
      // loads in flight no longer fit the registers)
#pragma unroll
      for (int k = 0; k < 9; ++k) { // plane below: own values
        const T p = (T)C[k] * xl[eo[k]];
        sum = ((mm >> k) & 1u) ? sum + p : sum;
      }
      asm volatile("" : "+v"(sum)::"memory");
#pragma unroll
      for (int k = 9; k < 13; ++k) { // this plane, before the diagonal
        const T p = (T)s_near[(k - 9) * kHbT + t] * xc[eo[k - 9]];
        sum = ((mm >> k) & 1u) ? sum + p : sum;
      }
      {
        const T p = (T)s_near[4 * kHbT + t] * xc[0];
        sum = ((mm >> 13) & 1u) ? sum + p : sum;
      }
      asm volatile("" : "+v"(sum)::"memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) { // this plane, after it: lane t + d's value
        int src = t + dn[j];
        src = src > kHbT - 1 ? kHbT - 1 : src;
        const TV vs = s_near[(3 - j) * kHbT + src];
        const T p = (T)(hn_need[j] ? HN[j] : vs) * xc[dn[j]];
        sum = ((mm >> (14 + j)) & 1u) ? sum + p : sum;
      }
      asm volatile("" : "+v"(sum)::"memory");
#pragma unroll
      for (int mi = 0; mi < 9; ++mi) { // plane above: lane t + e's value there
        if (mi == 5)
          asm volatile("" : "+v"(sum)::"memory");
        int src = t + eo[mi];
        src = src < 0 ? 0 : (src > kHbT - 1 ? kHbT - 1 : src);
        const TV vs = s_far[(8 - mi) * kHbT + src];
        TV v = vs;
        if (mi < 4)
          v = (!hi && he_need[mi]) ? HE[mi] : vs;
        else if (mi > 4)
          v = (hi && he_need[mi - 5]) ? HE[mi - 5] : vs;
        const T p = (T)v * xu[eo[mi]];
        sum = ((mm >> (18 + mi)) & 1u) ? sum + p : sum;
      }
      asm volatile("" : "+v"(sum)::"memory");
      const T cy = alpha * sum; // :49
      T y = cy;
      if constexpr (BETA)
        y = cy + beta * y0;
      if (live) {
        out[i] = y;
        if constexpr (DOT)
          dot_acc += (double)xc[0] * (double)cy;
      }
    }
    // the edge values of the step below, now that this step's are used
    load_edge(q - 1, s, last, HE, HN);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 9; ++a) // plane q's plane-to-plane arrays for step q - 1
      s_far[a * kHbT + t] = C[a];
  };
  const int units = g.tiles * g.segs;
  int u = blockIdx.x;
  if ((gridDim.x & 7) == 0) // neighbouring tiles on one XCD (halo lines in its L2)
    u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  for (; u < units; u += gridDim.x) {
    const int tile = u % g.tiles, seg = u / g.tiles;
    const int za = seg * g.seg_len;
    const int zb = za + g.seg_len < g.planes ? za + g.seg_len : g.planes;
    if (za >= zb)
      continue; // uniform
    const int s = tile * kHbT;
    __syncthreads(); // the unit before has done with the LDS
#pragma unroll
    for (int j = 0; j < 2; ++j) { // the window of plane zb
      const int w = t + j * kHbT;
      if (w < W)
        s_x[(zb % 3) * W + w] = at(in, clamp_col(zb * P + s - (L + 1) + w));
    }
    load_main(zb, s, false, N, XW, mN);
    load_edge(zb, s, false, HE, HN);
    step(zb, s, false, std::false_type()); // plane zb's values to LDS only
    for (int q = zb - 1; q >= za; --q)
      step(q, s, q == za, std::true_type());
  }
  if constexpr (DOT) {
    __syncthreads();
    double v = dot_acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off, 64);
    if ((t & 63) == 0)
      s_red[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < kHbT / 64; ++w)
        r += s_red[w];
      dot.partials[blockIdx.x] = r;
    }
    for (int i = gridDim.x + blockIdx.x * blockDim.x + t; i < dot.len;
         i += gridDim.x * blockDim.x)
      dot.partials[i] = 0.0;
  }
}

// The row of entry j among the rows whose pointers sit in s_rp[0..nr]: the
// largest r with s_rp[r] <= j (empty rows are skipped over).
__device__ __forceinline__ int wdia_row_of(const int32_t* s_rp, int nr, int32_t j)
{
  int lo = 0, hi = nr; // s_rp[lo] <= j < s_rp[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (s_rp[mid] <= j)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// pass 1: the set of distinct col - row (capacity kWdiaMaxOff; INT32_MIN =
// free slot).  Lane = ENTRY (coalesced reads of colind; the row by bisection in
// the block's slice of the row pointer), the set kept per workgroup in LDS and
// merged into the global one at the end -- the first version (lane = row, the
// global set probed per entry) took 9.9 ms of the 27-point plan at 256^3.
// Slots are always probed from 0, so a value can only ever sit in one slot (a
// probe that started anywhere else could insert a second copy behind a free
// slot).
__device__ __forceinline__ bool wdia_set_insert(int32_t* set, int32_t d)
{
  for (int s = 0; s < kWdiaMaxOff; ++s) {
    int32_t cur = set[s];
    if (cur == INT32_MIN)
      cur = atomicCAS(set + s, INT32_MIN, d);
    if (cur == d || cur == INT32_MIN)
      return true;
  }
  return false;
}

__global__ __launch_bounds__(kBlock) void wdia_offsets_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ set,
    int32_t* __restrict__ fail)
{
  __shared__ int32_t s_rp[kRows + 1];
  __shared__ int32_t s_set[kWdiaMaxOff];
  const int t = threadIdx.x;
  if (t < kWdiaMaxOff)
    s_set[t] = INT32_MIN;
  const int nrb = (num_rows + kRows - 1) / kRows;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    if (*(volatile int32_t*)fail) // somebody found a 33rd offset
      break;
    const int r0 = rb * kRows;
    const int nr = num_rows - r0 < kRows ? num_rows - r0 : kRows;
    __syncthreads();
    for (int r = t; r <= nr; r += kBlock)
      s_rp[r] = rowptr[r0 + r];
    __syncthreads();
    const int32_t j0 = s_rp[0], j1 = s_rp[nr];
    for (int32_t j = j0 + t; j < j1; j += kBlock) {
      // (a matrix on more than 32 diagonals says so within its first row
      // block: read the flag first -- on the 10 M x 81 FEM-like matrix every
      // entry of every workgroup's first block raised it with an atomic of its
      // own, 5.8 ms of the plan; profiles/r06_plan_fem81_kernel_stats.csv)
      if (*(volatile int32_t*)fail)
        continue;
      const int r = wdia_row_of(s_rp, nr, j);
      const int64_t d64 = (int64_t)colind[j] - (r0 + r);
      if (d64 <= INT32_MIN || d64 > INT32_MAX
          || !wdia_set_insert(s_set, (int32_t)d64))
        atomicOr(fail, 1);
    }
  }
  __syncthreads();
  if (t < kWdiaMaxOff && s_set[t] != INT32_MIN && !wdia_set_insert(set, s_set[t]))
    atomicOr(fail, 1);
}

// Is the matrix symmetric, entry for entry and bit for bit?  Asked of the arrays
// by offset (all K of them baked, masks written): the entry (i, i + D_k), D_k <
// 0, and its mirror (i + D_k, i) = offset -D_k of row i + D_k must both be
// there or both be missing, with the same bits.  One coalesced pass (lane =
// row) that stops at the first mismatch -- the CSR arrays are not touched (the
// earlier check searched every entry's mirror in its row: 46 ms for the
// 27-point matrix at 256^3, more than the rest of the plan together).
// mir[k] = the array of -D_k.
struct WdiaMirror {
  int32_t q[kWdiaMaxOff];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void wdia_array_symmetry_kernel(
    int32_t num_rows, int K, WdiaOffsets off, WdiaMirror mir, int64_t arr_len,
    const T* __restrict__ sval, const uint32_t* __restrict__ mask,
    int32_t* __restrict__ fail)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (*(volatile int32_t*)fail)
      return;
    const uint32_t m = mask[i];
    bool ok = true;
    for (int k = 0; k < K && off.D[k] < 0; ++k) { // uniform
      const int64_t r = i + off.D[k]; // the mirror's row (and this entry's column)
      const int q = mir.q[k];
      const bool mine = (m >> k) & 1u;
      bool theirs = false;
      if (r >= 0)
        theirs = (mask[r] >> q) & 1u;
      if (mine != theirs) {
        ok = false;
        break;
      }
      if (mine) {
        const T a = sval[(int64_t)k * arr_len + i];
        const T b = sval[(int64_t)q * arr_len + r];
        if constexpr (sizeof(T) == 8)
          ok = __double_as_longlong(a) == __double_as_longlong(b);
        else
          ok = __float_as_int(a) == __float_as_int(b);
        if (!ok)
          break;
      }
    }
    // ... and an upper entry whose mirror row lies before row 0 cannot be
    // (column >= num_rows is excluded by the square shape): nothing to check
    if (!ok) {
      if (!*(volatile int32_t*)fail)
        atomicOr(fail, 1);
      return;
    }
  }
}

// The same question asked of the CSR arrays (a binary search per entry: slow;
// only the fp32 copy of the mixed SpMV still asks it this way).  (Rows ascend -- the
// bake pass checks that; an unsorted row merely fails the search here and the
// matrix takes the full form's checks.)  Stops at the first mismatch.
template <typename T>
__global__ __launch_bounds__(kBlock) void wdia_symmetry_csr_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const T* __restrict__ values,
    int32_t* __restrict__ fail)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (*(volatile int32_t*)fail)
      return;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int32_t c = colind[j];
      if (c == i)
        continue;
      bool ok = c >= 0 && c < num_rows;
      if (ok) {
        int32_t lo = rowptr[c], hi = rowptr[c + 1]; // binary search for i
        while (lo < hi) {
          const int32_t mid = lo + (hi - lo) / 2;
          if (colind[mid] < (int32_t)i)
            lo = mid + 1;
          else
            hi = mid;
        }
        ok = lo < rowptr[c + 1] && colind[lo] == (int32_t)i;
        if (ok) {
          const T a = values[j], b = values[lo];
          if constexpr (sizeof(T) == 8)
            ok = __double_as_longlong(a) == __double_as_longlong(b);
          else
            ok = __float_as_int(a) == __float_as_int(b);
        }
      }
      if (!ok) {
        if (!*(volatile int32_t*)fail)
          atomicOr(fail, 1);
        return;
      }
    }
  }
}

// Is every diagonal constant, bit for bit?  Pass 1 (VERIFY = false) picks,
// per offset, the bits of whichever entry gets there first; pass 2 compares
// every entry with its offset's pick and stops at the first difference.  (A
// row whose columns do not ascend fails here and again, for good, in the bake
// pass.)
struct WdiaConstProbe {
  unsigned long long bits[kWdiaMaxOff];
  int seen[kWdiaMaxOff];
  int fail;
};

template <typename T, bool VERIFY>
__global__ __launch_bounds__(kBlock) void wdia_const_kernel(
    int32_t num_rows, int K, WdiaOffsets off, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const T* __restrict__ values,
    WdiaConstProbe* __restrict__ pr, uint32_t* __restrict__ mask_out)
{
  // (pass 2 also writes the presence mask when given somewhere to put it)
  __shared__ int32_t s_D[kWdiaMaxOff];
  if (threadIdx.x < kWdiaMaxOff)
    s_D[threadIdx.x] = off.D[threadIdx.x];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (*(volatile int*)&pr->fail)
      return;
    int prev = -1;
    uint32_t m = 0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int32_t d = (int32_t)((int64_t)colind[j] - i);
      int k = prev + 1;
      while (k < K && s_D[k] != d)
        ++k;
      if (k >= K) {
        atomicOr(&pr->fail, 1);
        return;
      }
      m |= 1u << k;
      unsigned long long b;
      if constexpr (sizeof(T) == 8)
        b = (unsigned long long)__double_as_longlong(values[j]);
      else
        b = (unsigned long long)(unsigned)__float_as_int(values[j]);
      if constexpr (VERIFY) {
        if (pr->bits[k] != b) {
          atomicOr(&pr->fail, 1);
          return;
        }
      } else {
        if (*(volatile int*)&pr->seen[k] == 0
            && atomicCAS(&pr->seen[k], 0, 1) == 0)
          pr->bits[k] = b;
      }
      prev = k;
    }
    if (VERIFY && mask_out)
      mask_out[i] = m;
  }
}

template <typename T>
int wdia_const_probe(const spmv_hip_csr_plan* pl, int K, const WdiaOffsets& off,
                     const T* values, hipStream_t st, bool* yes, double* cvals,
                     uint32_t* mask_out)
{
  *yes = false;
  WdiaConstProbe* d_pr = nullptr;
  WdiaConstProbe h_pr;
  hipError_t e = hipMalloc(&d_pr, sizeof(WdiaConstProbe));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_pr, 0, sizeof(WdiaConstProbe), st);
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, pl->num_rows, kBlock);
    // pass 1 looks at the first few planes only (every diagonal of a stencil
    // shows up there); a diagonal it has not seen makes pass 2 fail unless its
    // entries are +0.0 -- then the values are streamed, as before
    int64_t far = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t a = off.D[k] < 0 ? -(int64_t)off.D[k] : (int64_t)off.D[k];
      far = a > far ? a : far;
    }
    const int64_t prefix = 4 * far + 4096;
    const int32_t pick_rows
        = prefix < pl->num_rows ? (int32_t)prefix : pl->num_rows;
    hipLaunchKernelGGL((wdia_const_kernel<T, false>), dim3(grid), dim3(kBlock), 0,
                       st, pick_rows, K, off, pl->rowptr0, pl->colind0, values,
                       d_pr, (uint32_t*)nullptr);
    hipLaunchKernelGGL((wdia_const_kernel<T, true>), dim3(grid), dim3(kBlock), 0,
                       st, pl->num_rows, K, off, pl->rowptr0, pl->colind0, values,
                       d_pr, mask_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_pr, d_pr, sizeof(WdiaConstProbe), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_pr);
  if (e != hipSuccess)
    return static_cast<int>(e);
  if (h_pr.fail)
    return SPMV_HIP_OK;
  for (int k = 0; k < kWdiaMaxOff; ++k) {
    cvals[k] = 0.0;
    if (k >= K || !h_pr.seen[k])
      continue;
    if constexpr (sizeof(T) == 8) {
      double v;
      memcpy(&v, &h_pr.bits[k], sizeof(v));
      cvals[k] = v;
    } else {
      const unsigned b = (unsigned)h_pr.bits[k];
      float v;
      memcpy(&v, &b, sizeof(v));
      cvals[k] = (double)v; // exact
    }
  }
  *yes = true;
  return SPMV_HIP_OK;
}

// pass 2: fill the arrays and the masks; fail = a row whose columns do not
// ascend strictly, or an offset outside the set.  Only the offsets k < narr
// have arrays (half form: the offsets <= 0).  Tiles of 128 rows: the tile's
// entries are read in ENTRY order (coalesced; row and offset by bisection in
// LDS) into an LDS image [offset][row], which then goes out row-contiguous,
// zeros where a row has no such entry -- the arrays need no memset, and both
// sides of the transposition are coalesced (lane = row on both sides: 14.4 ms
// of the 27-point plan at 256^3, every load instruction 64 different lines).
constexpr int kWdiaTile = 128;
constexpr int kWdiaTileStride = kWdiaTile + 1; // (offset, row) -> distinct banks

template <typename T>
__global__ __launch_bounds__(kBlock) void wdia_bake_kernel(
    int32_t num_rows, int K, int narr, WdiaOffsets off,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const T* __restrict__ values, int64_t arr_len, T* __restrict__ sval,
    uint32_t* __restrict__ mask, int32_t* __restrict__ fail)
{
  __shared__ int32_t s_D[kWdiaMaxOff];
  __shared__ int32_t s_rp[kWdiaTile + 1];
  __shared__ uint32_t s_mask[kWdiaTile];
  __shared__ T s_img[kWdiaMaxOff * kWdiaTileStride];
  const int t = threadIdx.x;
  if (t < kWdiaMaxOff)
    s_D[t] = off.D[t];
  const int ntiles = (int)(arr_len / kWdiaTile); // arr_len: whole blocks of 256
  bool bad = false;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * kWdiaTile;
    int nr = num_rows - r0;
    nr = nr < 0 ? 0 : (nr > kWdiaTile ? kWdiaTile : nr);
    __syncthreads(); // the image of the tile before is out
    for (int r = t; r <= nr; r += kBlock)
      s_rp[r] = rowptr[r0 + r];
    if (t < kWdiaTile)
      s_mask[t] = 0u;
    for (int e = t; e < narr * kWdiaTileStride; e += kBlock)
      s_img[e] = T(0);
    __syncthreads();
    const int32_t j0 = nr > 0 ? s_rp[0] : 0, j1 = nr > 0 ? s_rp[nr] : 0;
    for (int32_t j = j0 + t; j < j1; j += kBlock) {
      const int r = wdia_row_of(s_rp, nr, j);
      const int32_t c = colind[j];
      const int64_t d64 = (int64_t)c - (r0 + r);
      // the offset's place in the ascending set
      int lo = 0, hi = K;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)s_D[mid] < d64)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (lo >= K || (int64_t)s_D[lo] != d64
          || (j > s_rp[r] && colind[j - 1] >= c)) { // csr_kernels.cpp:46-47 order
        bad = true;
        continue;
      }
      atomicOr(&s_mask[r], 1u << lo);
      if (lo < narr)
        s_img[lo * kWdiaTileStride + r] = values[j];
    }
    __syncthreads();
    for (int e = t; e < narr * kWdiaTile; e += kBlock) {
      const int k = e / kWdiaTile, r = e % kWdiaTile;
      sval[(int64_t)k * arr_len + r0 + r] = s_img[k * kWdiaTileStride + r];
    }
    if (t < nr)
      mask[r0 + t] = s_mask[t];
  }
  if (bad)
    atomicOr(fail, 1);
}

void wdia_free_arrays(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->wdia32_val);
  pl->wdia32_val = nullptr;
  pl->wdia32_values0 = nullptr;
  (void)hipFree(pl->wdia_val);
  (void)hipFree(pl->wdia_mask);
  pl->wdia_val = nullptr;
  pl->wdia_mask = nullptr;
  pl->wdia_values0 = nullptr;
  pl->wdia_len = 0;
  pl->wdia_elem = 0;
  pl->wdia_K = 0;
  pl->wdia_narr = 0;
  pl->wdia_const = 0;
  pl->wdia = 0;
  (void)hipFree(pl->wdia_box_table);
  pl->wdia_box_table = nullptr;
  pl->wdia_box_slots = pl->wdia_box_grid = pl->wdia_box_segments = 0;
  pl->wdia_box = 0;
  pl->wdia_hbox = 0;
  (void)hipFree(pl->wdia_zw_table);
  pl->wdia_zw_table = nullptr;
  pl->wdia_zw_slots = pl->wdia_zw_grid = pl->wdia_zw_segments = 0;
  pl->wdia_d2 = 0;
}

int wdia_grid(const spmv_hip_csr_plan* pl)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * pl->wdia_blocks_per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid < 1)
    grid = 1;
  if (grid >= 8)
    grid -= grid % 8;
  return grid;
}

template <typename T>
int wdia_bake(spmv_hip_csr_plan* pl, const T* values, hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  wdia_free_arrays(pl);
  if (values == nullptr)
    return SPMV_HIP_OK; // dropped
  if (pl->symmetric || !pl->ctx->bake_general || pl->nnz == 0
      || pl->algo != SPMV_HIP_ALGO_ROWBLOCK || pl->nnz < pl->ctx->lat_min_nnz)
    return SPMV_HIP_ENOTSUP;
  const auto t_begin = std::chrono::steady_clock::now();
  // SPMV_WDIA_TRACE=1: the phases' wall times on stderr (each ends with a stream
  // synchronisation, so the sum exceeds the untraced plan)
  const bool trace = getenv("SPMV_WDIA_TRACE") != nullptr;
  auto t_lap = t_begin;
  auto lap = [&](const char* what) {
    if (!trace)
      return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[wdia_bake] %-28s %8.3f ms\n", what,
            std::chrono::duration<double, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  const int32_t n = pl->num_rows;
  // pass 1: the offsets
  int32_t h_set[kWdiaMaxOff + 1];
  for (int s = 0; s < kWdiaMaxOff; ++s)
    h_set[s] = INT32_MIN;
  h_set[kWdiaMaxOff] = 0; // fail flag
  int32_t* d_set = nullptr;
  hipError_t e = hipMalloc(&d_set, sizeof(h_set));
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_set, h_set, sizeof(h_set), hipMemcpyHostToDevice, st);
  const int grid = spmv_grid_for(pl->ctx, n, kBlock);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(wdia_offsets_kernel, dim3(grid), dim3(kBlock), 0, st, n,
                       pl->rowptr0, pl->colind0, d_set, d_set + kWdiaMaxOff);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(h_set, d_set, sizeof(h_set), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    (void)hipFree(d_set);
    return static_cast<int>(e);
  }
  lap("offsets");
  WdiaOffsets off;
  int K = 0;
  for (int s = 0; s < kWdiaMaxOff; ++s)
    if (h_set[s] != INT32_MIN)
      off.D[K++] = h_set[s];
  // worth the memory only while the arrays are mostly full
  if (h_set[kWdiaMaxOff] || K == 0 || (double)pl->nnz < 0.5 * (double)K * n) {
    (void)hipFree(d_set);
    return SPMV_HIP_ENOTSUP;
  }
  for (int a = 1; a < K; ++a) // insertion sort, ascending
    for (int b = a; b > 0 && off.D[b] < off.D[b - 1]; --b) {
      const int32_t tmp = off.D[b];
      off.D[b] = off.D[b - 1];
      off.D[b - 1] = tmp;
    }
  for (int k = K; k < kWdiaMaxOff; ++k)
    off.D[k] = 0;
  // HALF form?  The offsets must mirror each other and the matrix must be
  // symmetric entry for entry, bit for bit (device check, stops at the first
  // mismatch); then only the arrays of the offsets <= 0 are kept
  int narr = K;
  for (int k = 0; k < kWdiaMaxOff; ++k) {
    off.A[k] = k < K ? k : 0;
    off.S[k] = 0;
  }
  // constant diagonals: no arrays at all, whatever the symmetry
  bool is_const = false;
  uint32_t* msk = nullptr;
  e = hipMalloc(&msk, sizeof(uint32_t) * (size_t)n);
  if (e != hipSuccess) {
    (void)hipFree(d_set);
    return static_cast<int>(e);
  }
  if (pl->ctx->const_diagonals) {
    // (its second pass writes the masks on the way)
    const int rcc = wdia_const_probe<T>(pl, K, off, values, st, &is_const,
                                        pl->wdia_cval, msk);
    if (rcc != SPMV_HIP_OK) {
      (void)hipFree(d_set);
      (void)hipFree(msk);
      return rcc;
    }
    if (is_const)
      narr = 0;
  }
  lap("mask alloc + const probe");
  bool mirrored = !is_const && pl->ctx->wdia_half
                  && pl->num_rows == pl->num_cols;
  for (int k = 0; k < K && mirrored; ++k) {
    bool found = false;
    for (int q = 0; q < K; ++q)
      found = found || off.D[q] == -off.D[k];
    mirrored = found;
  }
  const bool try_half = mirrored && K > 1; // decided on the baked arrays below
  // pass 2: the copy by offset
  const int64_t len = (((int64_t)n + kRows - 1) / kRows) * kRows;
  const size_t bytes = (size_t)narr * len * sizeof(T);
  void* sval = nullptr;
  int32_t h_fail = 0;
  e = hipMalloc(&sval, bytes > 0 ? bytes : 64); // (constant: a marker only)
  lap("alloc arrays");
  if (e == hipSuccess && !is_const) {
    // (the bake kernel writes every element of the arrays, zeros included)
    e = hipMemsetAsync(d_set + kWdiaMaxOff, 0, sizeof(int32_t), st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL((wdia_bake_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                         n, K, narr, off, pl->rowptr0, pl->colind0, values, len,
                         static_cast<T*>(sval), msk, d_set + kWdiaMaxOff);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(&h_fail, d_set + kWdiaMaxOff, sizeof(int32_t),
                         hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    lap("bake");
    // HALF form?  The offsets mirror each other (try_half); is the matrix
    // symmetric entry for entry, bit for bit?  One coalesced pass over the
    // arrays just baked; if so only the arrays of the offsets <= 0 are kept
    // (they come first: D ascends) and an upper entry is read as the lower
    // entry of its column's row.
    if (e == hipSuccess && !h_fail && try_half) {
      WdiaMirror mir;
      for (int k = 0; k < kWdiaMaxOff; ++k) {
        mir.q[k] = 0;
        for (int q = 0; q < K && k < K; ++q)
          if (off.D[q] == -off.D[k])
            mir.q[k] = q;
      }
      int32_t h_asym = 1;
      e = hipMemsetAsync(d_set + kWdiaMaxOff, 0, sizeof(int32_t), st);
      if (e == hipSuccess) {
        hipLaunchKernelGGL((wdia_array_symmetry_kernel<T>), dim3(grid), dim3(kBlock),
                           0, st, n, K, off, mir, len, static_cast<const T*>(sval),
                           msk, d_set + kWdiaMaxOff);
        e = hipGetLastError();
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(&h_asym, d_set + kWdiaMaxOff, sizeof(int32_t),
                           hipMemcpyDeviceToHost, st);
      if (e == hipSuccess)
        e = hipStreamSynchronize(st);
      lap("symmetry check");
      if (e == hipSuccess && !h_asym) {
        int nh = 0;
        while (nh < K && off.D[nh] <= 0)
          ++nh;
        void* half = nullptr;
        const size_t hbytes = (size_t)nh * len * sizeof(T);
        hipError_t eh = hipMalloc(&half, hbytes);
        if (eh == hipSuccess)
          eh = hipMemcpyAsync(half, sval, hbytes, hipMemcpyDeviceToDevice, st);
        if (eh == hipSuccess)
          eh = hipStreamSynchronize(st);
        lap("alloc + copy half");
        if (eh == hipSuccess) {
          (void)hipFree(sval);
          lap("free full");
          sval = half;
          narr = nh;
          for (int k = narr; k < K; ++k) {
            off.A[k] = mir.q[k];
            off.S[k] = off.D[k];
          }
        } else { // no room for the compact copy: the full form serves
          (void)hipFree(half);
          (void)hipGetLastError();
        }
      }
    }
  }
  (void)hipFree(d_set);
  if (e != hipSuccess || h_fail) {
    (void)hipFree(sval);
    (void)hipFree(msk);
    if (e == hipErrorOutOfMemory)
      (void)hipGetLastError(); // the CSR-order kernels keep running
    return (e != hipSuccess && e != hipErrorOutOfMemory) ? static_cast<int>(e)
                                                         : SPMV_HIP_ENOTSUP;
  }
  pl->wdia_val = sval;
  pl->wdia_mask = msk;
  pl->wdia_len = len;
  pl->wdia_elem = (int)sizeof(T);
  pl->wdia_K = K;
  pl->wdia_narr = narr;
  pl->wdia_const = is_const ? 1 : 0;
  for (int k = 0; k < kWdiaMaxOff; ++k) {
    pl->wdia_D[k] = off.D[k];
    pl->wdia_A[k] = off.A[k];
    pl->wdia_S[k] = off.S[k];
  }
  pl->wdia_values0 = values;
  pl->wdia = 1;
  // plane distance of a 3-D stencil: the middle of the widest cluster of
  // |offsets| (27-point: n^2 - n - 1 ... n^2 + n + 1 -> n^2)
  {
    int64_t a[kWdiaMaxOff];
    int na = 0;
    int64_t amax = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t v = off.D[k] < 0 ? -(int64_t)off.D[k] : (int64_t)off.D[k];
      amax = v > amax ? v : amax;
    }
    for (int k = 0; k < K; ++k) {
      const int64_t v = off.D[k] < 0 ? -(int64_t)off.D[k] : (int64_t)off.D[k];
      bool seen = false;
      for (int q = 0; q < na; ++q)
        seen = seen || a[q] == v;
      if (!seen && v * 2 > amax)
        a[na++] = v;
    }
    for (int p = 1; p < na; ++p) // insertion sort, <= 32 values
      for (int q = p; q > 0 && a[q - 1] > a[q]; --q) {
        const int64_t tmp = a[q];
        a[q] = a[q - 1];
        a[q - 1] = tmp;
      }
    pl->wdia_d2 = na > 0 ? a[na / 2] : 0;
  }
  lap("bookkeeping");
  if (!is_const && narr < K) { // a box in the half form: the marched kernel
    const int rh = spmv_wdia_hbox_build(pl, 1);
    if (rh != SPMV_HIP_OK) {
      wdia_free_arrays(pl);
      return rh;
    }
  }
  // (the plane-walk table of the general kernel: 17 ms at 256^3 -- built only
  // where that kernel runs; plan_set "wdia_hbox" = 0 builds it on demand)
  if (pl->wdia_d2 > 0 && !pl->wdia_hbox) {
    const int rw = spmv_wdia_walk_build(pl, 0, false);
    if (rw != SPMV_HIP_OK) {
      wdia_free_arrays(pl);
      return rw;
    }
  }
  if (is_const) { // a constant 27-point box: several lines per lane
    const int rb = spmv_wdia_box_build(pl, pl->ctx->const_tile, 0, false);
    if (rb != SPMV_HIP_OK) {
      wdia_free_arrays(pl);
      return rb;
    }
  }
  lap("walk table / box geometry");
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count();
  return SPMV_HIP_OK;
}

// The fp32 copy of the mixed-precision SpMV (fp64 plans whose values are baked
// by offset): the same arrays filled from the caller's fp32 values; the masks
// -- the structure -- are those of the fp64 copy.
int wdia_bake_mixed(spmv_hip_csr_plan* pl, const float* values32, hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  (void)hipFree(pl->wdia32_val);
  pl->wdia32_val = nullptr;
  pl->wdia32_values0 = nullptr;
  if (values32 == nullptr)
    return SPMV_HIP_OK; // dropped
  if (!pl->wdia_val || pl->wdia_elem != 8)
    return SPMV_HIP_ENOTSUP;
  const auto t_begin = std::chrono::steady_clock::now();
  WdiaOffsets off;
  for (int k = 0; k < kWdiaMaxOff; ++k) {
    off.D[k] = pl->wdia_D[k];
    off.A[k] = pl->wdia_A[k];
    off.S[k] = pl->wdia_S[k];
  }
  const int32_t n = pl->num_rows;
  if (pl->wdia_const) {
    // constant diagonals: the fp32 array must have them too (its own constants)
    bool is_const = false;
    const int rcc = wdia_const_probe<float>(pl, pl->wdia_K, off, values32, st,
                                            &is_const, pl->wdia32_cval, nullptr);
    if (rcc != SPMV_HIP_OK)
      return rcc;
    if (!is_const)
      return SPMV_HIP_ENOTSUP;
    SPMV_CHECK_HIP(hipMalloc(&pl->wdia32_val, 64)); // the "baked" marker
    pl->wdia32_values0 = values32;
    pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                       std::chrono::steady_clock::now() - t_begin)
                       .count();
    return SPMV_HIP_OK;
  }
  if (pl->wdia_narr < pl->wdia_K) {
    // the half form needs THESE values symmetric too (nothing says the fp32
    // array is the rounded fp64 one)
    int32_t* d_asym = nullptr;
    int32_t h_asym = 1;
    hipError_t es = hipMalloc(&d_asym, sizeof(int32_t));
    if (es == hipSuccess)
      es = hipMemsetAsync(d_asym, 0, sizeof(int32_t), st);
    if (es == hipSuccess) {
      hipLaunchKernelGGL((wdia_symmetry_csr_kernel<float>),
                         dim3(spmv_grid_for(pl->ctx, n, kBlock)), dim3(kBlock), 0,
                         st, n, pl->rowptr0, pl->colind0, values32, d_asym);
      es = hipGetLastError();
    }
    if (es == hipSuccess)
      es = hipMemcpyAsync(&h_asym, d_asym, sizeof(int32_t),
                          hipMemcpyDeviceToHost, st);
    if (es == hipSuccess)
      es = hipStreamSynchronize(st);
    (void)hipFree(d_asym);
    if (es != hipSuccess)
      return static_cast<int>(es);
    if (h_asym)
      return SPMV_HIP_ENOTSUP;
  }
  const size_t bytes = (size_t)pl->wdia_narr * pl->wdia_len * sizeof(float);
  void* sval = nullptr;
  uint32_t* msk = nullptr; // rewritten with the same bits: the kernel's output
  int32_t* d_fail = nullptr;
  int32_t h_fail = 0;
  hipError_t e = hipMalloc(&sval, bytes);
  if (e == hipSuccess)
    e = hipMalloc(&msk, sizeof(uint32_t) * (size_t)n);
  if (e == hipSuccess)
    e = hipMalloc(&d_fail, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_fail, 0, sizeof(int32_t), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL((wdia_bake_kernel<float>), dim3(spmv_grid_for(pl->ctx, n, kBlock)),
                       dim3(kBlock), 0, st, n, pl->wdia_K, pl->wdia_narr, off,
                       pl->rowptr0, pl->colind0, values32, pl->wdia_len,
                       static_cast<float*>(sval), msk, d_fail);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_fail, d_fail, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_fail);
  (void)hipFree(msk);
  if (e != hipSuccess || h_fail) {
    (void)hipFree(sval);
    if (e == hipErrorOutOfMemory)
      (void)hipGetLastError();
    return (e != hipSuccess && e != hipErrorOutOfMemory) ? static_cast<int>(e)
                                                         : SPMV_HIP_ENOTSUP;
  }
  pl->wdia32_val = sval;
  pl->wdia32_values0 = values32;
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count();
  return SPMV_HIP_OK;
}

// j-space of the box kernel: work items and launch grid
int64_t box_items(const spmv_hip_csr_plan* pl, int R)
{
  const int64_t l = pl->wdia_box_L;
  const int64_t lines = ((int64_t)pl->num_rows + l - 1) / l;
  return ((lines + R - 1) / R) * l;
}

int box_block(const spmv_hip_csr_plan* pl, int R)
{
  const int64_t P = pl->wdia_box_P, L = pl->wdia_box_L;
  if (P % ((int64_t)R * L) != 0)
    return kRows;
  const int64_t plane = P / R;
  for (int b = kRows; b >= 192; --b)
    if (plane % b == 0)
      return b;
  return kRows;
}

int box_grid(const spmv_hip_csr_plan* pl)
{
  const int block = box_block(pl, pl->wdia_box);
  const int64_t nrb = (box_items(pl, pl->wdia_box) + block - 1) / block;
  int64_t grid = (int64_t)pl->ctx->num_cus * pl->wdia_box_blocks_per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid >= 8)
    grid -= grid % 8;
  return grid < 1 ? 1 : (int)grid;
}

template <typename T, bool DOT, int R>
int box_launch(const spmv_hip_csr_plan* pl, hipStream_t st, const WdiaConsts& cv,
               T alpha, const T* in, T beta, T* out, DotOut dot)
{
  BoxGeom g;
  g.P = pl->wdia_box_P;
  g.L = pl->wdia_box_L;
  g.NJ = box_items(pl, R);
  g.rcp_l = 1.0 / (double)g.L;
  g.block = box_block(pl, R);
  g.chain_blocks = 0;
  if (g.P % ((int64_t)R * g.L) == 0 && (g.P / R) % g.block == 0)
    g.chain_blocks = g.P / R / g.block;
  const int grid = box_grid(pl);
  const int nrb = (int)((g.NJ + g.block - 1) / g.block);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->wdia_xcd_group;
  if (pl->wdia_zwalk && pl->wdia_box_table && pl->wdia_box_grid == grid) {
    ord.table = pl->wdia_box_table;
    ord.num_slots = pl->wdia_box_slots;
    hipLaunchKernelGGL((csr_box27_const_kernel<T, DOT, true, R>), dim3(grid),
                       dim3(kBlock), 0, st, pl->num_rows, pl->wdia_mask, alpha, in,
                       beta, out, dot, ord, g, cv);
  } else {
    hipLaunchKernelGGL((csr_box27_const_kernel<T, DOT, false, R>), dim3(grid),
                       dim3(kBlock), 0, st, pl->num_rows, pl->wdia_mask, alpha, in,
                       beta, out, dot, ord, g, cv);
  }
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

// The marched half-box kernel: geometry, LDS footprint, launch.
HalfBoxGeom hbox_geom(const spmv_hip_csr_plan* pl)
{
  HalfBoxGeom g;
  g.P = pl->wdia_box_P;
  g.L = pl->wdia_box_L;
  g.tiles = (g.P + kHbT - 1) / kHbT;
  g.planes = (int)(((int64_t)pl->num_rows + g.P - 1) / g.P);
  // runs of planes: about one unit per CU, no run shorter than 8 planes (the
  // run's first step loads a plane it does not compute)
  int segs = pl->wdia_hbox_segs;
  if (segs <= 0) {
    segs = (pl->ctx->num_cus + g.tiles / 2) / g.tiles;
    const int most = g.planes / 8;
    segs = segs > most ? most : segs;
  }
  segs = segs < 1 ? 1 : (segs > g.planes ? g.planes : segs);
  g.seg_len = (g.planes + segs - 1) / segs;
  g.segs = (g.planes + g.seg_len - 1) / g.seg_len;
  g.W = kHbT + 2 * (g.L + 1);
  return g;
}

size_t hbox_lds(const HalfBoxGeom& g, size_t tv, size_t t)
{
  return 14 * (size_t)kHbT * tv + 3 * (size_t)g.W * t + 16 * sizeof(double);
}

template <typename TV, typename T, bool DOT>
int hbox_launch(const spmv_hip_csr_plan* pl, hipStream_t st, const TV* sval,
                T alpha, const T* in, T beta, T* out, DotOut dot)
{
  const HalfBoxGeom g = hbox_geom(pl);
  const size_t lds = hbox_lds(g, sizeof(TV), sizeof(T));
  int grid = g.tiles * g.segs;
  if (grid > pl->ctx->num_cus)
    grid = pl->ctx->num_cus;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  auto kern = beta != T(0) ? csr_box27_half_kernel<TV, T, DOT, true>
                           : csr_box27_half_kernel<TV, T, DOT, false>;
  SPMV_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  HalfBoxArrays<TV> arrays;
  for (int a = 0; a < 14; ++a)
    arrays.a[a] = sval + (int64_t)a * pl->wdia_len;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kHbT), lds, st, pl->num_rows,
                     (int32_t)pl->wdia_len, arrays, pl->wdia_mask, alpha, in, beta,
                     out, dot, g);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

template <typename TV, typename T, bool DOT>
int wdia_launch(const spmv_hip_csr_plan* pl, hipStream_t st, const TV* sval,
                T alpha, const T* in, T beta, T* out, DotOut dot)
{
  if (pl->wdia_hbox && !pl->wdia_const && pl->wdia_narr == 14
      && pl->num_rows == pl->num_cols)
    return hbox_launch<TV, T, DOT>(pl, st, sval, alpha, in, beta, out, dot);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  WdiaOffsets off;
  for (int k = 0; k < kWdiaMaxOff; ++k) {
    off.D[k] = pl->wdia_D[k];
    off.A[k] = pl->wdia_A[k];
    off.S[k] = pl->wdia_S[k];
  }
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->wdia_xcd_group;
  const int grid = wdia_grid(pl);
  if (pl->wdia_zwalk && pl->wdia_zw_table && pl->wdia_zw_grid == grid) {
    ord.table = pl->wdia_zw_table;
    ord.num_slots = pl->wdia_zw_slots;
  }
  WdiaConsts cv;
  for (int k = 0; k < kWdiaMaxOff; ++k)
    cv.c[k] = sizeof(TV) == sizeof(T) ? pl->wdia_cval[k] : pl->wdia32_cval[k];
  if (pl->wdia_const && pl->wdia_box > 1) {
    if (pl->wdia_box == 2)
      return box_launch<T, DOT, 2>(pl, st, cv, alpha, in, beta, out, dot);
    return box_launch<T, DOT, 4>(pl, st, cv, alpha, in, beta, out, dot);
  }
  if (pl->wdia_const)
    hipLaunchKernelGGL((csr_wdia_kernel<TV, T, DOT, true>), dim3(grid),
                       dim3(kBlock), 0, st, pl->num_rows, pl->num_cols,
                       pl->wdia_len, pl->wdia_K, off, sval, pl->wdia_mask, alpha,
                       in, beta, out, dot, ord, cv);
  else
    hipLaunchKernelGGL((csr_wdia_kernel<TV, T, DOT, false>), dim3(grid),
                       dim3(kBlock), 0, st, pl->num_rows, pl->num_cols,
                       pl->wdia_len, pl->wdia_K, off, sval, pl->wdia_mask, alpha,
                       in, beta, out, dot, ord, cv);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

void spmv_wdia_free(spmv_hip_csr_plan* pl) { wdia_free_arrays(pl); }

int spmv_wdia_walk_build(spmv_hip_csr_plan* pl, int segments, bool force)
{
  if (pl->wdia_zw_table) {
    SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
    SPMV_CHECK_HIP(hipDeviceSynchronize()); // no launch still reads the old one
    (void)hipFree(pl->wdia_zw_table);
    pl->wdia_zw_table = nullptr;
    pl->wdia_zw_slots = pl->wdia_zw_grid = pl->wdia_zw_segments = 0;
  }
  const int grid = wdia_grid(pl);
  const int rc = spmv_zwalk_table_device(pl, pl->num_rows, pl->wdia_d2, grid,
                                         segments, force,
                                         &pl->wdia_zw_table, &pl->wdia_zw_slots,
                                         &pl->wdia_zw_segments);
  if (rc == SPMV_HIP_OK && pl->wdia_zw_table)
    pl->wdia_zw_grid = grid;
  return rc;
}

int spmv_wdia_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       hipStream_t st)
{
  return wdia_bake<double>(pl, values, st);
}

int spmv_wdia_bake_f32(spmv_hip_csr_plan* pl, const float* values, hipStream_t st)
{
  return wdia_bake<float>(pl, values, st);
}

int spmv_wdia_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32,
                          hipStream_t st)
{
  return wdia_bake_mixed(pl, values32, st);
}

int spmv_wdia_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot)
{
  const double* sv = static_cast<const double*>(pl->wdia_val);
  if (dot.partials)
    return wdia_launch<double, double, true>(pl, st, sv, alpha, in, beta, out,
                                             dot);
  return wdia_launch<double, double, false>(pl, st, sv, alpha, in, beta, out, dot);
}

int spmv_wdia_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                         double alpha, const double* in, double beta,
                         double* out, DotOut dot)
{
  const float* sv = static_cast<const float*>(pl->wdia32_val);
  if (dot.partials)
    return wdia_launch<float, double, true>(pl, st, sv, alpha, in, beta, out,
                                            dot);
  return wdia_launch<float, double, false>(pl, st, sv, alpha, in, beta, out, dot);
}

int spmv_wdia_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out)
{
  return wdia_launch<float, float, false>(
      pl, st, static_cast<const float*>(pl->wdia_val), alpha, in, beta, out,
      DotOut());
}

// Is the offset set a 27-point box a P + b L + c (a, b, c in -1..1, sorted
// order = (a, b, c) lexicographic)?  Then (re)build the box kernel's geometry
// and plane-walk table: R lines per lane (0 or 1 = the general kernel).
int spmv_wdia_box_build(spmv_hip_csr_plan* pl, int R, int segments, bool force)
{
  SPMV_REQUIRE(R == 0 || R == 1 || R == 2 || R == 4);
  if (pl->wdia_box_table) {
    SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
    SPMV_CHECK_HIP(hipDeviceSynchronize()); // no launch still reads the old one
    (void)hipFree(pl->wdia_box_table);
    pl->wdia_box_table = nullptr;
  }
  pl->wdia_box_slots = pl->wdia_box_grid = pl->wdia_box_segments = 0;
  pl->wdia_box = 0;
  if (R <= 1 || !pl->wdia_const || pl->wdia_K != 27
      || pl->num_rows != pl->num_cols)
    return SPMV_HIP_OK;
  const int64_t P = pl->wdia_D[22], L = pl->wdia_D[16];
  if (L < 3 || P <= 2 * L + 2 || L >= (1 << 23))
    return SPMV_HIP_OK;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c)
        if (pl->wdia_D[9 * a + 3 * b + c] != (a - 1) * P + (b - 1) * L + (c - 1))
          return SPMV_HIP_OK;
  pl->wdia_box = R;
  pl->wdia_box_P = (int)P;
  pl->wdia_box_L = (int)L;
  if (P % (R * L) != 0)
    return SPMV_HIP_OK; // planes do not line up in j-space: the plain order
  const int grid = box_grid(pl);
  // the table builder counts in blocks of 256 rows: hand it the block counts
  const int block = box_block(pl, R);
  const int64_t nblocks = (box_items(pl, R) + block - 1) / block;
  const int64_t rows_eq = block == kRows ? box_items(pl, R) : nblocks * kRows;
  const int64_t plane_eq = block == kRows ? P / R : P / R / block * kRows;
  const int rc = spmv_zwalk_table_device(
      pl, rows_eq, plane_eq, grid, segments, force, &pl->wdia_box_table,
      &pl->wdia_box_slots, &pl->wdia_box_segments);
  if (rc == SPMV_HIP_OK && pl->wdia_box_table)
    pl->wdia_box_grid = grid;
  return rc;
}

// HALF form of a 27-point box a P + b L + c with varying coefficients: may the
// marched kernel (csr_box27_half_kernel) take it?  Lines of at most 511 rows
// (the two ends of a tile must not overlap, the x ring must fit the LDS), planes
// of at least a tile, at least 8 planes.
int spmv_wdia_hbox_build(spmv_hip_csr_plan* pl, int on)
{
  pl->wdia_hbox = 0;
  if (!on || !pl->wdia_val || pl->wdia_const || pl->wdia_K != 27
      || pl->wdia_narr != 14 || pl->num_rows != pl->num_cols)
    return SPMV_HIP_OK;
  const int64_t P = pl->wdia_D[22], L = pl->wdia_D[16];
  if (L < 3 || L > 511 || P <= 2 * L + 2 || P < kHbT || P > (1 << 26)
      || (int64_t)pl->num_rows < 8 * P || pl->wdia_len >= (1 << 29) - (1 << 27))
    return SPMV_HIP_OK;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int c = 0; c < 3; ++c)
        if (pl->wdia_D[9 * a + 3 * b + c] != (a - 1) * P + (b - 1) * L + (c - 1))
          return SPMV_HIP_OK;
  for (int k = 14; k < 27; ++k) // the half form's map: entry k = array 26 - k
    if (pl->wdia_A[k] != 26 - k || pl->wdia_S[k] != pl->wdia_D[k])
      return SPMV_HIP_OK;
  int lds_max = 0;
  SPMV_CHECK_HIP(hipDeviceGetAttribute(
      &lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, pl->ctx->device));
  pl->wdia_box_P = (int)P;
  pl->wdia_box_L = (int)L;
  const HalfBoxGeom g = hbox_geom(pl);
  if (hbox_lds(g, (size_t)pl->wdia_elem, (size_t)pl->wdia_elem) > (size_t)lds_max
      || hbox_lds(g, 4, 8) > (size_t)lds_max)
    return SPMV_HIP_OK;
  pl->wdia_hbox = 1;
  return SPMV_HIP_OK;
}
