// DIAGONAL form for gfx950 (MI355X): the symmetric lattice form
// (spmv_symlat.hip) with the matrix VALUES re-laid out by the plan -- one
// array per lower offset plus the diagonal -- so that the entries a row needs
// from its column are a plain shifted window of the same arrays.  SURVEY 8f n4
// ("DIA fast path for stencils"), for the symmetric storage of
// spmv/csr_kernels.cpp:26-40 and (GEN kernels, below) for a GENERAL matrix that
// a device check finds symmetric; same arithmetic, same order, same bits as
// the reference kernel of the storage the caller chose.
//
// Why: in CSR order a row of the 7-point matrix holds its three lower entries
// side by side, and row i needs, besides its own, ONE of the three of each of
// the rows i+1, i+n, i+n^2.  The symmetric lattice kernel therefore pulls
// three more full value windows per row block (72 B of values per row for 24
// used) and spends a row pointer, two masks and popcounts per entry to find
// them (2.55 ms at 512^3, 12.5 GB through the fabric for 8.6 GB algorithmic).
// By offset, v_k[i] = L(i, i + D[k]):
//   own entries      v_k[i]            window [r0, r0 + 256) of array k
//   column entries   v_k[i - D[k]]     window [r0 - D[k], ...) of the SAME array
// -- the window another row block reads as its own one line / one plane later.
// No row pointer, no positions: LDS index = window base + lane.  Per row: 24 B
// of values + 8 (diagonal) + 1 (mask) + 8 (x) + 8 (y) = 49 B from HBM.
//
// Order and reuse: the plane-walk table (spmv_zwalk_order_build) lets every
// workgroup walk one 256-row column of the lattice from plane to plane; where
// the planes are a whole number of row blocks apart the kernel then CHAINS the
// planes (RING kernels below): the far window of the farthest offset and the x
// of the plane ahead are handed to the next step in LDS / registers instead of
// being loaded again.
//
// Measured at 512^3 (MI355X, profiles/r02_pmc_summary.json): 1.35 ms = 0.79
// of the symmetric-CSR roofline (general storage: 10.3 TB/s of CSR bytes);
// 6.95 GB read + 1.07 GB written through the fabric for 5.50 + 1.07 unique,
// i.e. 5.9 TB/s of real traffic -- about what this part gives a kernel that
// mixes reads and writes (a plain copy: 4.8-5.1).
// What is left over the unique 82 lines per row block (102 are read): the x
// loads at +-n (9 lines: the neighbour columns' lines, fetched one step
// earlier) and the far window of the middle offset (9 lines: another
// workgroup's own window, sometimes on another XCD); the loads at +-1 all hit
// (profiles/r02_pmc_symdia_probes_512.json).  Removing either group entirely
// buys 1 % and 4 % of the time: the kernel runs at the rate the fabric takes
// this mix of reads and writes.  Tried and dropped: x[i +- 1] by shuffle
// (-1 %), the neighbours' x loaded one plane ahead and handed on (-4.5 %).
//
// The copy is made by spmv_hip_csr_plan_bake_values_* (the plan's only use of
// the VALUES; everything else in a plan is structure).  A launch with the
// baked pointers takes this kernel; any other `values` / `diagonal` pointer
// takes the CSR-order kernels, so a stale copy can only be used by a caller
// who rewrites the baked arrays in place without baking again (documented in
// spmv_hip.h).
//
// Kernel skeleton = csr_lattice_kernel (spmv_lat.hip): persistent grid, two
// LDS slots, all value windows of the NEXT row block by LDS-DMA, x / mask / y0
// of the next block into registers, one barrier per block.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "csr_plan.h"
#include "lat_dma.h"

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

static void sdia_free_arrays(spmv_hip_csr_plan* pl);

namespace
{

constexpr int kSdiaMaxOff = 3;

constexpr int kSdiaMaxWin = 2 * kSdiaMaxOff + 1; // own + far per offset, diagonal

struct SdiaGeom {
  int nd;                 // lower offsets
  int U[kSdiaMaxOff];     // row distance -D[k] > 0 (D ascending: U descending)
  int nwin;               // DMA windows per row block
  // Per window, packed: the kernel keeps all of this in scalar registers, and
  // one word per field overflowed them (90 spills per wave)
  int first[kSdiaMaxWin]; // its first row relative to r0
  int last[kSdiaMaxWin];  // last 16-byte chunk the window needs (its 1-KiB
                          // DMA pieces, <= 4: one per wave, = last / 64 + 1)
  int lds[kSdiaMaxWin];   // entry offset of the window inside a slot (a
                          // multiple of 128) | array of the window (bits 0-2:
                          // lower k, nd = diagonal, nd + 1 + k = upper k of
                          // the FULL form) | non-temporal (bit 3)
  int own_idx[kSdiaMaxOff]; // slot entry of v_k[r0]      (+ lane = own entry)
  int col_idx[kSdiaMaxOff]; // slot entry of v_k[r0 + U_k] (+ lane = column entry)
  int d_idx;              // slot entry of d[r0]
  int slot_entries;
  // Plane chain (RING kernels): when the farthest offset is a whole number of
  // row blocks, offset 0 has no windows in the slots; its planes live in a
  // ring of four 256-row buffers behind the slots (see the kernel)
  int chain_blocks;       // U[0] / 256 when that is whole (x is handed from
                          // plane to plane), else 0
  int ring;               // ... and the offset-0 value planes live in the ring
  int ring_off;           // entry offset of the ring
  int nt_ring, nt_store;  // non-temporal: the ring planes, the y stores
};
__host__ __device__ inline int sdia_win_arr(int lds) { return lds & 7; }
__host__ __device__ inline int sdia_win_nt(int lds) { return (lds >> 3) & 1; }
__host__ __device__ inline int sdia_win_lds(int lds) { return lds & ~127; }

template <typename T>
struct SdiaRegs {
  unsigned cm;          // bits 0..2: own entry k, bits 4..6: column entry k
  T xi, y0;
  T xl[kSdiaMaxOff];    // x[i - U_k]
  T xu[kSdiaMaxOff];    // x[i + U_k]
};

// `chain`: row block rb lies exactly one plane (U[0] rows) behind the block
// `prev` was loaded for: its x_i is prev's x[i + U_0], its x[i - U_0] prev's x_i
template <typename T>
__device__ __forceinline__ SdiaRegs<T> sdia_loads(
    int rb, const SdiaGeom& g, int t, int32_t num_rows,
    const uint8_t* __restrict__ cmask, const T* __restrict__ in, T beta,
    const T* __restrict__ out, bool chain, const SdiaRegs<T>& prev)
{
  SdiaRegs<T> q;
  q.cm = 0;
  q.xi = q.y0 = T(0);
#pragma unroll
  for (int k = 0; k < kSdiaMaxOff; ++k)
    q.xl[k] = q.xu[k] = T(0);
  if (rb < 0)
    return q;
  const int32_t i = rb * kRows + t;
  if (i < num_rows) {
    q.cm = cmask[i];
    if (chain) // uniform
      q.xi = prev.xu[0];
    else
      q.xi = in[i];
    if (beta != T(0))
      q.y0 = out[i];
#pragma unroll
    for (int k = 0; k < kSdiaMaxOff; ++k) {
      if (k < g.nd) { // uniform
        // unconditional and clamped: no dependence on the mask load; what the
        // row does not have is never used
        // (taking x[i -+ 1] from the neighbour lanes' x_i by shuffle instead
        // of loading it was measured: 1 % slower, these loads hit the caches)
        const int64_t c = (int64_t)i - g.U[k];
        if (k == 0 && chain)
          q.xl[k] = prev.xi;
        else
          q.xl[k] = in[c < 0 ? 0 : c];
        const int64_t r = (int64_t)i + g.U[k];
        q.xu[k] = in[r < num_rows ? r : (int64_t)num_rows - 1];
      }
    }
  }
  return q;
}

// RING: the plane chain.  In the plane-walk order a workgroup's next row block
// is, most of the time, exactly one plane (U[0] rows) below the current one.
// Then the far column window of offset 0 it has in LDS IS the next block's own
// window, and the x it holds for the rows one plane ahead IS the next block's
// x_i: nothing of that is loaded again (2.5-D streaming).  It takes 32 of 220
// L2 requests per row block away -- most of them were hits, the fabric reads
// barely change -- and 8 % of the time.
// The offset-0 planes sit in a ring of four buffers: two in use (own, far), up
// to two being filled for the next block (far only when chained; own and far
// after a jump).
//
// GEN: the same storage behind the GENERAL SpMV (spmv/csr_kernels.cpp:41-51)
// of a matrix the plan found to be symmetric, entry for entry and bit for bit
// (sdia_bake_general): the upper entry (i, i + u) IS the stored lower entry
// (i + u, i).  The row is summed in the general kernel's order -- ascending
// column: lower offsets, diagonal (mask bit 3: a row may lack it), upper
// offsets -- so the result has the bits of the CSR kernels while only half the
// off-diagonal values cross the fabric (49 B per row of the 7-point matrix
// instead of 73).
//
// TV = type of the baked values (what is streamed), T = type of x, y and of all
// the arithmetic.  TV = float with T = double is the mixed-precision SpMV
//  on the diagonal form: 17 instead of 33 B of matrix data per row.
// TAB: the row-block order comes from a table (see order_slot_raw_t).
template <typename TV, typename T, bool DOT, bool RING, bool GEN, bool TAB>
__global__ __launch_bounds__(kBlock) void csr_sym_dia_kernel(
    int32_t num_rows, int64_t arr_len, const TV* __restrict__ sval,
    const uint8_t* __restrict__ cmask, T alpha, const T* __restrict__ in, T beta,
    T* __restrict__ out, DotOut dot, RowBlockOrder ord, SdiaGeom g)
{
  constexpr int V = 16 / (int)sizeof(TV);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ double s_red[kBlock / 64];
  TV* const s_val = reinterpret_cast<TV*>(s_dyn);
  const unsigned lds0 = (unsigned)(uintptr_t)(
      (__attribute__((address_space(3))) void*)s_val);

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;

  // all windows of row block rb -> LDS slot: one piece per wave and window
  auto issue = [&](int rb, int slot) {
    const int64_t r0 = (int64_t)rb * kRows;
#pragma unroll
    for (int j = 0; j < kSdiaMaxWin; ++j) {
      if (j < g.nwin && wave <= (g.last[j] >> 6)) { // uniform
        const int64_t base = (r0 + g.first[j]) & ~(int64_t)(V - 1);
        // lanes past the window re-read its last chunk (one cached line)
        // instead of streaming the next row block's data
        int chunk = wave * 64 + lane;
        chunk = chunk < g.last[j] ? chunk : g.last[j];
        int64_t e = base + (int64_t)chunk * V;
        e = e < arr_len - V ? e : arr_len - V; // arrays are padded with zeros
        const TV* src = sval + (int64_t)sdia_win_arr(g.lds[j]) * arr_len + e;
        const unsigned dst
            = lds0
              + (unsigned)((slot * g.slot_entries + sdia_win_lds(g.lds[j]))
                           * (int)sizeof(TV))
              + (unsigned)wave * 1024u;
        if (sdia_win_nt(g.lds[j])) // uniform
          glds16<true>(src, dst);
        else
          glds16<false>(src, dst);
      }
    }
  };

  // 256 rows of array 0 from `first_row` -> ring buffer r
  auto ring_dma = [&](int64_t first_row, int r) {
    constexpr int pieces = kRows * (int)sizeof(TV) / 1024;
    if (wave < pieces) { // uniform
      int64_t e = first_row + (int64_t)(wave * 64 + lane) * V;
      e = e < arr_len - V ? e : arr_len - V;
      const unsigned dst = lds0
                           + (unsigned)((g.ring_off + r * kRows) * (int)sizeof(TV))
                           + (unsigned)wave * 1024u;
      if (g.nt_ring) // uniform
        glds16<true>(sval + e, dst);
      else
        glds16<false>(sval + e, dst);
    }
  };
  int R0 = 0, R1 = 1, R2 = 2, R3 = 3; // ring: own, far, free, free

  int it = blockIdx.x;
  int cur = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it, num_slots));
  int nxt_raw = order_slot_raw_t<TAB>(ord, it + stride, num_slots);
  if (cur >= 0) {
    issue(cur, 0);
    if constexpr (RING) {
      ring_dma((int64_t)cur * kRows, R0);
      ring_dma((int64_t)cur * kRows + g.U[0], R1);
    }
  }
  SdiaRegs<T> qB;
  qB.xi = T(0);
  qB.xu[0] = T(0);
  SdiaRegs<T> qA = sdia_loads<T>(cur, g, t, num_rows, cmask, in, beta, out,
                                 false, qB);
  int slot = 0;
  // y is stored ONE STEP LATE, right behind the next step's wait: that wait
  // covers everything the wave has in flight, stores included, and a store
  // issued at the end of a step would be waited for at once -- its whole round
  // trip exposed in every step (csr_lxw_kernel measured it: a fifth of the time)
  T y_late = T(0);
  int32_t i_late = -1;
  auto store_late = [&]() {
    if (i_late >= 0) {
      if (g.nt_store) // uniform
        __builtin_nontemporal_store(y_late, out + i_late);
      else
        out[i_late] = y_late;
    }
    i_late = -1;
  };
  auto step = [&](const SdiaRegs<T>& q, SdiaRegs<T>& qn) {
    // everything of this block has landed, all waves have left the previous
    // one (the builtin, not asm: see csr_lattice_kernel)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    store_late();
    const int nxt = order_slot_decode(ord, nxt_raw);
    const int nn_raw = order_slot_raw_t<TAB>(ord, it + 2 * stride, num_slots);
    const bool chain = g.chain_blocks > 0 && cur >= 0 && nxt >= 0
                       && nxt - cur == g.chain_blocks;
    if (nxt >= 0) {
      issue(nxt, slot ^ 1);
      if constexpr (RING) {
        if (chain) {
          ring_dma((int64_t)nxt * kRows + g.U[0], R2);
        } else {
          ring_dma((int64_t)nxt * kRows, R2);
          ring_dma((int64_t)nxt * kRows + g.U[0], R3);
        }
      }
    }
    qn = sdia_loads<T>(nxt, g, t, num_rows, cmask, in, beta, out, chain, q);
    const int32_t i = cur * kRows + t;
    if (cur >= 0 && i < num_rows) {
      const TV* sv = s_val + slot * g.slot_entries + t;
      T vl[kSdiaMaxOff], vu[kSdiaMaxOff];
#pragma unroll
      for (int k = 0; k < kSdiaMaxOff; ++k) {
        vl[k] = vu[k] = T(0);
        if (k < g.nd) { // uniform
          if (RING && k == 0) {
            vl[k] = (T)s_val[g.ring_off + R0 * kRows + t];
            vu[k] = (T)s_val[g.ring_off + R1 * kRows + t];
          } else {
            vl[k] = (T)sv[g.own_idx[k]];
            vu[k] = (T)sv[g.col_idx[k]];
          }
        }
      }
      const T d = (T)sv[g.d_idx];
      T y, cy;
      if constexpr (GEN) {
        T sum = 0; // csr_kernels.cpp:45
#pragma unroll
        for (int k = 0; k < kSdiaMaxOff; ++k)
          if (k < g.nd && ((q.cm >> k) & 1u)) // :46-47, ascending column
            sum += vl[k] * q.xl[k];
        if ((q.cm >> 3) & 1u)
          sum += d * q.xi;
#pragma unroll
        for (int k = kSdiaMaxOff - 1; k >= 0; --k)
          if (k < g.nd && ((q.cm >> (4 + k)) & 1u))
            sum += vu[k] * q.xu[k];
        cy = alpha * sum; // :49
        y = cy;
        if (beta != T(0))
          y = cy + beta * q.y0;
      } else {
        T sum = d * q.xi; // csr_kernels.cpp:28
#pragma unroll
        for (int k = 0; k < kSdiaMaxOff; ++k)
          if (k < g.nd && ((q.cm >> k) & 1u)) // :34, left to right
            sum += vl[k] * q.xl[k];
        const T c = alpha * sum; // :39
        y = c, cy = c;
        if (beta != T(0))
          y = c + beta * q.y0;
        // the column's entries in ascending row order: nearest row first
#pragma unroll
        for (int k = kSdiaMaxOff - 1; k >= 0; --k)
          if (k < g.nd && ((q.cm >> (4 + k)) & 1u)) { // :35
            const T term = (alpha * vu[k]) * q.xu[k];
            y += term;
            cy += term;
          }
      }
      y_late = y;
      i_late = i;
      if constexpr (DOT) // in . (alpha A in): the finished row without beta y0
        dot_acc += (double)q.xi * (double)cy;
    }
    if constexpr (RING) {
      // what was filled for the next block becomes (own, far)
      const int a = R0, b = R1;
      if (chain) {
        R0 = b, R1 = R2, R2 = R3, R3 = a;
      } else {
        R0 = R2, R1 = R3, R2 = a, R3 = b;
      }
    }
    slot ^= 1;
    cur = nxt;
    nxt_raw = nn_raw;
    it += stride;
  };
  while (it < num_slots) {
    step(qA, qB);
    if (it >= num_slots)
      break;
    step(qB, qA);
  }
  store_late();
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// CONSTANT diagonals: every entry of a diagonal has the same bits (the 7-point
// Poisson operator, any constant-coefficient stencil).  The bake keeps the
// mask byte per row and ONE number per diagonal; the kernel reads no matrix
// values at all -- 17 B per row (x, y, mask) instead of 49 -- and does the same
// multiplications and additions in the same order with the constant in a
// register.  No LDS, no barriers: x, mask and y0 of the next row block are
// loaded into registers while the current one is summed; the plane chain hands
// x from plane to plane as in the value-streaming kernel.
// ---------------------------------------------------------------------------
struct SdiaConsts {
  double c[2 * kSdiaMaxOff + 1]; // lower k | diagonal (nd) | upper (nd + 1 + k)
};

template <typename T, bool DOT, bool GEN, bool TAB>
__global__ __launch_bounds__(kBlock) void csr_const_dia_kernel(
    int32_t num_rows, const uint8_t* __restrict__ cmask, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    RowBlockOrder ord, SdiaGeom g, SdiaConsts cv)
{
  __shared__ double s_red[kBlock / 64];
  const int t = threadIdx.x;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;
  T vl[kSdiaMaxOff], vu[kSdiaMaxOff];
#pragma unroll
  for (int k = 0; k < kSdiaMaxOff; ++k) {
    vl[k] = k < g.nd ? (T)cv.c[k] : T(0);
    vu[k] = k < g.nd ? (T)cv.c[g.nd + 1 + k] : T(0);
  }
  const T d = (T)cv.c[g.nd];

  int it = blockIdx.x;
  int cur = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it, num_slots));
  int nxt_raw = order_slot_raw_t<TAB>(ord, it + stride, num_slots);
  SdiaRegs<T> qB;
  qB.xi = T(0);
  qB.xu[0] = T(0);
  SdiaRegs<T> qA = sdia_loads<T>(cur, g, t, num_rows, cmask, in, beta, out,
                                 false, qB);
  auto step = [&](const SdiaRegs<T>& q, SdiaRegs<T>& qn) {
    const int nxt = order_slot_decode(ord, nxt_raw);
    const int nn_raw = order_slot_raw_t<TAB>(ord, it + 2 * stride, num_slots);
    const bool chain = g.chain_blocks > 0 && cur >= 0 && nxt >= 0
                       && nxt - cur == g.chain_blocks;
    qn = sdia_loads<T>(nxt, g, t, num_rows, cmask, in, beta, out, chain, q);
    const int32_t i = cur * kRows + t;
    if (cur >= 0 && i < num_rows) {
      T y, cy;
      if constexpr (GEN) {
        T sum = 0; // csr_kernels.cpp:45
#pragma unroll
        for (int k = 0; k < kSdiaMaxOff; ++k)
          if (k < g.nd && ((q.cm >> k) & 1u)) // :46-47, ascending column
            sum += vl[k] * q.xl[k];
        if ((q.cm >> 3) & 1u)
          sum += d * q.xi;
#pragma unroll
        for (int k = kSdiaMaxOff - 1; k >= 0; --k)
          if (k < g.nd && ((q.cm >> (4 + k)) & 1u))
            sum += vu[k] * q.xu[k];
        cy = alpha * sum; // :49
        y = cy;
        if (beta != T(0))
          y = cy + beta * q.y0;
      } else {
        T sum = d * q.xi; // csr_kernels.cpp:28
#pragma unroll
        for (int k = 0; k < kSdiaMaxOff; ++k)
          if (k < g.nd && ((q.cm >> k) & 1u)) // :34, left to right
            sum += vl[k] * q.xl[k];
        const T c = alpha * sum; // :39
        y = c, cy = c;
        if (beta != T(0))
          y = c + beta * q.y0;
        // the column's entries in ascending row order: nearest row first
#pragma unroll
        for (int k = kSdiaMaxOff - 1; k >= 0; --k)
          if (k < g.nd && ((q.cm >> (4 + k)) & 1u)) { // :35
            const T term = (alpha * vu[k]) * q.xu[k];
            y += term;
            cy += term;
          }
      }
      if (g.nt_store) // uniform
        __builtin_nontemporal_store(y, out + i);
      else
        out[i] = y;
      if constexpr (DOT)
        dot_acc += (double)q.xi * (double)cy;
    }
    cur = nxt;
    nxt_raw = nn_raw;
    it += stride;
  };
  while (it < num_slots) {
    step(qA, qB);
    if (it >= num_slots)
      break;
    step(qB, qA);
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// Constant diagonals, R lattice lines per lane.  The kernel above asks the L2
// for three new line sets of x per row block (the plane ahead and both
// neighbour lines; own row and the plane behind are handed on by the chain) and
// it is the number of those requests, not the bytes behind them, that bounds
// it (PMC: 36 M requests per launch at 512^3, 57 in flight per CU on average
// whatever the prefetch depth,
// 1.8 GB through the fabric in 0.79 ms).  Here a lane owns R rows one LINE
// apart -- rows i0, i0 + U1, ..., i0 + (R-1) U1 with U1 the middle offset -- so
// the neighbour lines of the inner rows are the lane's own registers and only
// the two outer ones are loaded: (R + 2) / R line sets per row block of the
// plane ahead and the lines around, instead of 3.
//
// Index space: lines are runs of U1 consecutive rows; R consecutive lines form
// a tuple; work item j = tuple * U1 + position stands for the rows
// (tuple * R + r) * U1 + position, r < R.  256 consecutive j are a block; the
// order table walks the blocks of j-space plane by plane (a plane is U0 / R
// items when U0 is a multiple of R U1; then a block U0 / R items ahead holds
// the same lattice column one plane on, and the chain applies).
//
// Measured at 512^3 (MI355X): one line per lane 0.80 ms, R = 2 0.67, R = 4 0.60
// (PMC, R = 4: 25.1 M L2 read requests instead of 36.3 M, 67 in flight per CU,
// 1.66 GB + 1.07 GB through the fabric = 4.6 TB/s).  Tried and dropped: x[i-+1]
// from the neighbour lanes by shuffle with loads at the wave and line
// boundaries only -- 0.70 instead of 0.60 (those loads fetch the row's own
// lines, which the neighbour lines' workgroups need in the L2 anyway); a
// clamp-free 32-bit index path and test-free sums for the blocks and waves
// inside the lattice -- no change: with every load and store removed the kernel
// takes 0.37 ms (its instructions), under the 0.60 the memory side needs.
// (Round 5, once more with the boundary values loaded from ONE address per wave
// -- a single line each -- and every shuffle unconditional: 0.615 against 0.594
// at 512^3, 0.283 against 0.272 at 384^3, bit-equal.  The x[i -+ 1] loads hit
// the vector L1; the shuffles cost more than they save.)
// Ablations (0.65 on that box): without the x[i -+ 1] loads 0.56, without the
// outer neighbour lines 0.61, without both 0.50, without the stores 0.55.
// ---------------------------------------------------------------------------
struct SdiaTileGeom {
  int U0, U1, U2;     // row distances, descending (nd = 3)
  int64_t NJ;         // work items
  int block;          // work items per workgroup and step: 256, or the largest
                      // divisor of a plane's items in 192..256 (216^3: 243) so
                      // that planes are whole blocks and the chain applies
  int chain_blocks;   // blocks of j-space per plane when whole, else 0
  int nt_store;
  double rcp_u1;
};

#ifdef SPMV_PROBE_FOLD
// MEASUREMENT BUILD ONLY (tools/ab_build.sh B -DSPMV_PROBE_FOLD; DESIGN.md
// "folding the x / p update into the SpMV"): the memory traffic of a kernel
// that also computed p = r + beta p for its stencil neighbours and wrote p and
// x -- a second vector read at every address x is read at, one more row vector
// read, two more written -- without the arithmetic.  Results stay those of the
// plain kernel (the extra loads only feed a test that never holds).
__device__ double* g_probe_fold[3];
#endif

template <typename T, int R>
struct SdiaTileRegs {
#ifdef SPMV_PROBE_FOLD
  T junk;
#endif
  unsigned cm[R];
  T xi[R], xl0[R], xu0[R], xl2[R], xu2[R], y0[R];
  T x_below, x_above; // x[i_0 - U1], x[i_{R-1} + U1]
  int64_t i0;         // first row of the lane, -1 = none
};

template <typename T, int R>
__device__ __forceinline__ SdiaTileRegs<T, R> sdia_tile_loads(
    int jb, const SdiaTileGeom& g, int t, int32_t num_rows,
    const uint8_t* __restrict__ cmask, const T* __restrict__ in, T beta,
    const T* __restrict__ out, bool chain, const SdiaTileRegs<T, R>& prev)
{
  SdiaTileRegs<T, R> q;
  q.i0 = -1;
  q.x_below = q.x_above = T(0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    q.cm[r] = 0;
    q.xi[r] = q.xl0[r] = q.xu0[r] = q.xl2[r] = q.xu2[r] = q.y0[r] = T(0);
  }
  if (jb < 0 || t >= g.block)
    return q;
  const int64_t j = (int64_t)jb * g.block + t;
  if (j >= g.NJ)
    return q;
  // tuple and position: j / U1 by reciprocal, one step of correction
  int64_t tup = (int64_t)((double)j * g.rcp_u1);
  int64_t pos = j - tup * g.U1;
  if (pos < 0) {
    --tup;
    pos += g.U1;
  } else if (pos >= g.U1) {
    ++tup;
    pos -= g.U1;
  }
  const int64_t i0 = tup * R * g.U1 + pos;
  if (i0 >= num_rows)
    return q;
  q.i0 = i0;
  const int64_t last = (int64_t)num_rows - 1;
#ifdef SPMV_PROBE_FOLD
  const T* in2 = reinterpret_cast<const T*>(g_probe_fold[0]);
  const T* xv = reinterpret_cast<const T*>(g_probe_fold[1]);
  T junk = T(0);
  auto at = [&](int64_t c) {
    const int64_t cc = c < 0 ? 0 : (c > last ? last : c);
    junk += in2[cc];
    return in[cc];
  };
#else
  auto at = [&](int64_t c) { // clamped: what the row does not have is not used
    return in[c < 0 ? 0 : (c > last ? last : c)];
  };
#endif
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = i0 + (int64_t)r * g.U1;
    const bool have = i <= last;
    const int64_t ic = have ? i : last;
    q.cm[r] = have ? cmask[ic] : 0u;
    if (chain) { // uniform
      q.xi[r] = prev.xu0[r];
      q.xl0[r] = prev.xi[r];
    } else {
      q.xi[r] = in[ic];
#ifdef SPMV_PROBE_FOLD
      junk += in2[ic];
#endif
      q.xl0[r] = at(i - g.U0);
    }
#ifdef SPMV_PROBE_FOLD
    junk += xv[ic];
#endif
    q.xu0[r] = at(i + g.U0);
    q.xl2[r] = at(i - g.U2);
    q.xu2[r] = at(i + g.U2);
    if (beta != T(0))
      q.y0[r] = out[ic];
  }
  q.x_below = at(i0 - g.U1);
  q.x_above = at(i0 + (int64_t)R * g.U1);
#ifdef SPMV_PROBE_FOLD
  q.junk = junk;
#endif
  return q;
}

template <typename T, bool DOT, bool GEN, bool TAB, int R>
__global__ __launch_bounds__(kBlock) void csr_const_dia_tile_kernel(
    int32_t num_rows, const uint8_t* __restrict__ cmask, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    RowBlockOrder ord, SdiaTileGeom g, SdiaConsts cv)
{
  __shared__ double s_red[kBlock / 64];
  const int t = threadIdx.x;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;
  // slots of SdiaConsts with nd = 3: lower 0 1 2 | diagonal 3 | upper 4 5 6
  const T vl0 = (T)cv.c[0], vl1 = (T)cv.c[1], vl2 = (T)cv.c[2];
  const T d = (T)cv.c[3];
  const T vu0 = (T)cv.c[4], vu1 = (T)cv.c[5], vu2 = (T)cv.c[6];

  int it = blockIdx.x;
  int cur = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it, num_slots));
  int nxt_raw = order_slot_raw_t<TAB>(ord, it + stride, num_slots);
  SdiaTileRegs<T, R> qB;
#pragma unroll
  for (int r = 0; r < R; ++r)
    qB.xi[r] = qB.xu0[r] = T(0);
  SdiaTileRegs<T, R> qA = sdia_tile_loads<T, R>(cur, g, t, num_rows, cmask, in,
                                                beta, out, false, qB);
  auto step = [&](const SdiaTileRegs<T, R>& q, SdiaTileRegs<T, R>& qn) {
    const int nxt = order_slot_decode(ord, nxt_raw);
    const int nn_raw = order_slot_raw_t<TAB>(ord, it + 2 * stride, num_slots);
    const bool chain = g.chain_blocks > 0 && cur >= 0 && nxt >= 0
                       && nxt - cur == g.chain_blocks;
    qn = sdia_tile_loads<T, R>(nxt, g, t, num_rows, cmask, in, beta, out, chain,
                               q);
    if (cur >= 0 && q.i0 >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t i = q.i0 + (int64_t)r * g.U1;
        if (i < num_rows) {
          const unsigned cm = q.cm[r];
          const T xl1 = r > 0 ? q.xi[r > 0 ? r - 1 : 0] : q.x_below;
          const T xu1 = r < R - 1 ? q.xi[r < R - 1 ? r + 1 : 0] : q.x_above;
          T y, cy;
          if constexpr (GEN) {
            T sum = 0; // csr_kernels.cpp:45; :46-47 in ascending column order
            if (cm & 1u)
              sum += vl0 * q.xl0[r];
            if (cm & 2u)
              sum += vl1 * xl1;
            if (cm & 4u)
              sum += vl2 * q.xl2[r];
            if (cm & 8u)
              sum += d * q.xi[r];
            if (cm & 64u)
              sum += vu2 * q.xu2[r];
            if (cm & 32u)
              sum += vu1 * xu1;
            if (cm & 16u)
              sum += vu0 * q.xu0[r];
            cy = alpha * sum; // :49
            y = cy;
            if (beta != T(0))
              y = cy + beta * q.y0[r];
          } else {
            T sum = d * q.xi[r]; // csr_kernels.cpp:28; :34 left to right
            if (cm & 1u)
              sum += vl0 * q.xl0[r];
            if (cm & 2u)
              sum += vl1 * xl1;
            if (cm & 4u)
              sum += vl2 * q.xl2[r];
            const T c = alpha * sum; // :39
            y = c, cy = c;
            if (beta != T(0))
              y = c + beta * q.y0[r];
            // the column's entries in ascending row order: nearest row first
            if (cm & 64u) { // :35
              const T term = (alpha * vu2) * q.xu2[r];
              y += term;
              cy += term;
            }
            if (cm & 32u) {
              const T term = (alpha * vu1) * xu1;
              y += term;
              cy += term;
            }
            if (cm & 16u) {
              const T term = (alpha * vu0) * q.xu0[r];
              y += term;
              cy += term;
            }
          }
#ifdef SPMV_PROBE_FOLD
          reinterpret_cast<T*>(g_probe_fold[1])[i] = q.xi[r]; // "x"
          reinterpret_cast<T*>(g_probe_fold[2])[i] = q.xl2[r]; // "p"
          if (q.junk == T(1.2345e30)) // (never: the buffers hold zeros)
            y = q.junk;
#endif
          if (g.nt_store) // uniform
            __builtin_nontemporal_store(y, out + i);
          else
            out[i] = y;
          if constexpr (DOT)
            dot_acc += (double)q.xi[r] * (double)cy;
        }
      }
    }
    cur = nxt;
    nxt_raw = nn_raw;
    it += stride;
  };
  while (it < num_slots) {
    step(qA, qB);
    if (it >= num_slots)
      break;
    step(qB, qA);
  }
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// Is every diagonal constant?  Pass 1 (VERIFY = false) picks, per diagonal, the
// bits of whichever entry gets there first; pass 2 compares every entry with
// its diagonal's pick and stops at the first difference.  Slots as SdiaConsts.
// GENERAL: entries are classified by col - row (offsets u0 > u1 > u2); else
// symmetric storage: row i holds its lower entries in ascending column order
// (mask bit k set = offset k present) and `diagonal[i]`.
struct SdiaConstProbe {
  unsigned long long bits[2 * kSdiaMaxOff + 1];
  int seen[2 * kSdiaMaxOff + 1];
  int fail;
};

template <typename T>
__device__ __forceinline__ unsigned long long value_bits(T v)
{
  if constexpr (sizeof(T) == 8)
    return (unsigned long long)__double_as_longlong(v);
  else
    return (unsigned long long)(unsigned)__float_as_int(v);
}

template <typename T, bool GENERAL, bool VERIFY>
__global__ __launch_bounds__(kBlock) void sdia_const_kernel(
    int32_t num_rows /* to look at */, int32_t total_rows, int nd, int u0,
    int u1, int u2,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const uint8_t* __restrict__ mask, const T* __restrict__ values,
    const T* __restrict__ diagonal, SdiaConstProbe* __restrict__ pr,
    uint8_t* __restrict__ cmask_out)
{
  // (pass 2 also writes the kernel's mask byte -- own bits 0..2, diagonal 3,
  // column / upper bits 4..6 -- when given somewhere to put it)
  auto visit = [&](int slot, T v) {
    const unsigned long long b = value_bits(v);
    if constexpr (VERIFY) {
      if (pr->bits[slot] != b && !*(volatile int*)&pr->fail)
        atomicOr(&pr->fail, 1);
    } else {
      if (*(volatile int*)&pr->seen[slot] == 0
          && atomicCAS(&pr->seen[slot], 0, 1) == 0)
        pr->bits[slot] = b;
    }
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (*(volatile int*)&pr->fail)
      return;
    unsigned cm = 0;
    if constexpr (GENERAL) {
      for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
        const int64_t c = colind[j];
        if (c == i) {
          visit(nd, values[j]);
          cm |= 8u;
          continue;
        }
        const int64_t u = c < i ? i - c : c - i;
        const int k = u == u0 ? 0 : (u == u1 ? 1 : (u == u2 ? 2 : -1));
        if (k < 0 || k >= nd || c >= total_rows) {
          atomicOr(&pr->fail, 1);
          return;
        }
        visit(c < i ? k : nd + 1 + k, values[j]);
        cm |= 1u << (c < i ? k : 4 + k);
      }
    } else {
      const unsigned m = mask[i];
      int32_t j = rowptr[i];
      cm = m | 8u;
      for (int k = 0; k < nd; ++k) {
        if ((m >> k) & 1u)
          visit(k, values[j++]);
        const int64_t r = i + (k == 0 ? u0 : (k == 1 ? u1 : u2));
        if (VERIFY && r < total_rows && ((mask[r] >> k) & 1u))
          cm |= 1u << (4 + k);
      }
      visit(nd, diagonal[i]);
    }
    if (VERIFY && cmask_out)
      cmask_out[i] = (uint8_t)cm;
  }
}

// values in CSR order -> one array per offset (+ the diagonal), zero where a
// row has no entry; combined mask byte
template <typename T>
__global__ __launch_bounds__(kBlock) void sdia_bake_kernel(
    int32_t num_rows, int nd, int u0, int u1, int u2,
    const int32_t* __restrict__ rowptr, const uint8_t* __restrict__ mask,
    const T* __restrict__ values, const T* __restrict__ diagonal,
    int64_t arr_len, T* __restrict__ sval, uint8_t* __restrict__ cmask)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned m = mask[i];
    int32_t j = rowptr[i];
    unsigned cm = m;
    for (int k = 0; k < nd; ++k) {
      T v = T(0);
      if ((m >> k) & 1u)
        v = values[j++];
      if (sval) // (null: the mask only -- constant diagonals)
        sval[(int64_t)k * arr_len + i] = v;
      const int64_t r = i + (k == 0 ? u0 : (k == 1 ? u1 : u2));
      if (r < num_rows && ((mask[r] >> k) & 1u))
        cm |= 1u << (4 + k);
    }
    if (sval)
      sval[(int64_t)nd * arr_len + i] = diagonal[i];
    cmask[i] = (uint8_t)(cm | 8u);
  }
}

// ---------------------------------------------------------------------------
// General matrices: is A symmetric, entry for entry and bit for bit, with at
// most three distinct |col - row| > 0?  Then its lower half + diagonal go into
// the same arrays.
// ---------------------------------------------------------------------------
// pass 1: the set of distinct |col - row| > 0 (capacity 8, INT32_MAX = free)
__global__ __launch_bounds__(kBlock) void sdia_offsets_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ set,
    int32_t* __restrict__ fail)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (*(volatile int32_t*)fail) // (a ninth offset was found: nobody goes on)
      return;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int64_t d64 = (int64_t)colind[j] - i;
      if (d64 == 0)
        continue;
      const int32_t d = (int32_t)(d64 < 0 ? -d64 : d64);
      bool placed = false;
      for (int s = 0; s < 8 && !placed; ++s) {
        int32_t cur = set[s];
        if (cur == INT32_MAX)
          cur = atomicCAS(set + s, INT32_MAX, d);
        placed = (cur == d || cur == INT32_MAX);
      }
      if (!placed)
        atomicOr(fail, 1);
    }
  }
}

// ... the same set from the lattice form's row-block records (every entry of
// the matrix is `row + one of its block's offsets`): 48 B per 256 rows to read
// instead of every column index (512^3: 0.1 ms instead of 7)
__global__ __launch_bounds__(kBlock) void sdia_offsets_from_lattice_kernel(
    int num_row_blocks, const int32_t* __restrict__ lat_tab,
    int32_t* __restrict__ set, int32_t* __restrict__ fail)
{
  const int rb = blockIdx.x * blockDim.x + threadIdx.x;
  if (rb >= num_row_blocks)
    return;
  const int32_t* rec = lat_tab + (int64_t)rb * kLatRec;
  const int nd = rec[0];
  for (int k = 0; k < nd && k < kLatMaxOff; ++k) {
    const int64_t d64 = rec[4 + k];
    if (d64 == 0)
      continue;
    const int32_t d = (int32_t)(d64 < 0 ? -d64 : d64);
    bool placed = false;
    for (int s = 0; s < 8 && !placed; ++s) {
      int32_t cur = set[s];
      if (cur == INT32_MAX)
        cur = atomicCAS(set + s, INT32_MAX, d);
      placed = (cur == d || cur == INT32_MAX);
    }
    if (!placed)
      atomicOr(fail, 1);
  }
}

template <typename T>
__device__ __forceinline__ bool same_bits(T a, T b)
{
  if constexpr (sizeof(T) == 8)
    return __double_as_longlong(a) == __double_as_longlong(b);
  else
    return __float_as_int(a) == __float_as_int(b);
}

// pass 2: fill the arrays and the mask.  Half form: every off-diagonal entry
// (i, c) must have its mirror (c, i) with the same bits, only the lower entries
// are stored.  FULL form: no such demand, the upper entries go to arrays of
// their own (nd + 1 + k).  (Ascending columns without repeats are what the
// lattice form, a precondition, already guarantees.)
// STORE = false: the checks only (is the half form possible?), nothing written.
template <typename T, bool FULL, bool STORE>
__global__ __launch_bounds__(kBlock) void sdia_bake_general_kernel(
    int32_t num_rows, int nd, int u0, int u1, int u2,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const T* __restrict__ values, int64_t arr_len, T* __restrict__ sval,
    uint8_t* __restrict__ cmask, int32_t* __restrict__ fail, int check_mirrors)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    // a check that has failed has nothing left to learn -- and a matrix that
    // is not symmetric would otherwise send an atomic per entry to one address
    // (512^3: 150 ms of the plan)
    if (!STORE && *(volatile int32_t*)fail)
      return;
    unsigned cm = 0;
    for (int32_t j = rowptr[i]; j < rowptr[i + 1]; ++j) {
      const int64_t c = colind[j];
      const T v = values[j];
      if (c == i) {
        if (STORE && sval)
          sval[(int64_t)nd * arr_len + i] = v;
        cm |= 8u;
        continue;
      }
      const int64_t u = c < i ? i - c : c - i;
      const int k = u == u0 ? 0 : (u == u1 ? 1 : (u == u2 ? 2 : -1));
      bool ok = k >= 0 && k < nd && c < num_rows;
      if (ok && !FULL && check_mirrors) {
        ok = false;
        for (int32_t jj = rowptr[c]; jj < rowptr[c + 1]; ++jj)
          if (colind[jj] == i)
            ok = same_bits(values[jj], v);
      }
      if (!ok) {
        if (!*(volatile int32_t*)fail)
          atomicOr(fail, 1);
        continue;
      }
      if (c < i) {
        if (STORE && sval)
          sval[(int64_t)k * arr_len + i] = v;
        cm |= 1u << k;
      } else {
        if (FULL && STORE && sval)
          sval[(int64_t)(nd + 1 + k) * arr_len + i] = v;
        cm |= 1u << (4 + k);
      }
    }
    if (STORE)
      cmask[i] = (uint8_t)cm;
  }
}

// LDS geometry of one slot for element size sizeof(T)
template <typename T>
SdiaGeom sdia_geom(const spmv_hip_csr_plan* pl)
{
  constexpr int V = 16 / (int)sizeof(T);
  constexpr int per_piece = 1024 / (int)sizeof(T);
  SdiaGeom g{};
  g.nd = pl->sdia_nd;
  int w = 0, entries = 0;
  auto add = [&](int arr, int first, int rows) {
    const int lead = first & (V - 1); // alignment slack in front
    const int pieces = (lead + rows + per_piece - 1) / per_piece;
    g.first[w] = first;
    g.last[w] = (lead + rows - 1) / V;
    g.lds[w] = entries | arr; // entries: a multiple of per_piece >= 128
    entries += pieces * per_piece;
    ++w;
    return sdia_win_lds(g.lds[w - 1]) + lead; // slot entry of row r0 + first
  };
  // FULL form (a general matrix that is NOT symmetric): the upper entries have
  // arrays of their own, every array is read through its own window only
  const bool full = pl->sdia_general == 2;
  for (int k = 0; k < g.nd; ++k) {
    g.U[k] = pl->sdia_U[k];
    if (k == 0 && pl->sdia_chain && g.U[0] >= kRows && g.U[0] % kRows == 0) {
      g.chain_blocks = g.U[0] / kRows;
      if (!full) { // offset 0 lives in the ring
        g.ring = 1;
        continue;
      }
    }
    if (full) {
      g.own_idx[k] = add(k, 0, kRows);
      g.col_idx[k] = add(g.nd + 1 + k, 0, kRows);
    } else if (g.U[k] < kRows) { // the column window overlaps the own one
      g.own_idx[k] = add(k, 0, kRows + g.U[k]);
      g.col_idx[k] = g.own_idx[k] + g.U[k];
    } else {
      g.own_idx[k] = add(k, 0, kRows);
      g.col_idx[k] = add(k, g.U[k], kRows);
    }
  }
  g.d_idx = add(g.nd, 0, kRows);
  g.nwin = w;
  g.slot_entries = entries;
  g.ring_off = 2 * entries;
  // streams nobody reads a second time leave the L2 to x and the shared
  // windows (plan_set "sdia_nt": bit 0 ring planes, 1 diagonal, 2 windows of
  // near offsets, 3 windows of far offsets, 4 y stores)
  const int m = pl->sdia_nt;
  g.nt_ring = m & 1;
  g.nt_store = (m >> 4) & 1;
  for (int j = 0; j < w; ++j) {
    const int arr = sdia_win_arr(g.lds[j]);
    int nt;
    if (arr == g.nd)
      nt = (m >> 1) & 1;
    else if (g.U[arr > g.nd ? arr - g.nd - 1 : arr] < kRows)
      nt = (m >> 2) & 1;
    else
      nt = (m >> 3) & 1;
    g.lds[j] |= nt << 3;
  }
  return g;
}

template <typename T>
size_t sdia_lds_bytes(const SdiaGeom& g)
{
  return ((size_t)2 * g.slot_entries + (g.ring ? 4 * kRows : 0))
         * sizeof(T);
}

// launch grid of the baked element type
template <typename T>
int sdia_grid(const spmv_hip_csr_plan* pl)
{
  const SdiaGeom g = sdia_geom<T>(pl);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const size_t lds = sdia_lds_bytes<T>(g);
  int per_cu = (int)((160 * 1024) / (lds + 64));
  if (pl->sdia_const) // no LDS: as many workgroups as the registers allow
    per_cu = kBlocksPerCU;
  per_cu = per_cu > pl->slat_blocks_per_cu ? pl->slat_blocks_per_cu : per_cu;
  per_cu = per_cu < 1 ? 1 : per_cu;
  int grid = pl->ctx->num_cus * per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid >= 8)
    grid -= grid % 8;
  return grid < 1 ? 1 : grid;
}

// `sval` / `cmask`: the baked copy to stream (the plan's native one, or the
// fp32 copy of the mixed-precision SpMV)
// j-space of the tile kernel: work items, blocks, launch grid
int64_t sdia_tile_items(const spmv_hip_csr_plan* pl, int R)
{
  const int64_t u1 = pl->sdia_U[1];
  const int64_t lines = ((int64_t)pl->num_rows + u1 - 1) / u1;
  return ((lines + R - 1) / R) * u1;
}

// work items per workgroup and step (see SdiaTileGeom::block)
int sdia_tile_block(const spmv_hip_csr_plan* pl, int R)
{
  const int64_t u0 = pl->sdia_U[0], u1 = pl->sdia_U[1];
  if (u0 % ((int64_t)R * u1) != 0)
    return kRows;
  const int64_t plane = u0 / R;
  for (int b = kRows; b >= 192; --b)
    if (plane % b == 0)
      return b;
  return kRows;
}

int sdia_tile_grid(const spmv_hip_csr_plan* pl)
{
  const int block = sdia_tile_block(pl, pl->sdia_tile);
  const int64_t nrb = (sdia_tile_items(pl, pl->sdia_tile) + block - 1) / block;
  int64_t grid = (int64_t)pl->ctx->num_cus * pl->sdia_tile_blocks_per_cu;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > nrb)
    grid = nrb;
  if (grid >= 8)
    grid -= grid % 8;
  return grid < 1 ? 1 : (int)grid;
}

template <typename T, int R>
int sdia_tile_launch(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const SdiaConsts& cv, T alpha, const T* in, T beta, T* out,
                     DotOut dot)
{
  SdiaTileGeom g;
  g.U0 = pl->sdia_U[0];
  g.U1 = pl->sdia_U[1];
  g.U2 = pl->sdia_U[2];
  g.NJ = sdia_tile_items(pl, R);
  g.rcp_u1 = 1.0 / (double)g.U1;
  g.nt_store = (pl->sdia_nt >> 4) & 1;
  g.block = sdia_tile_block(pl, R);
  g.chain_blocks = 0;
  if (pl->sdia_chain && g.U0 % ((int64_t)R * g.U1) == 0
      && (g.U0 / R) % g.block == 0)
    g.chain_blocks = g.U0 / R / g.block;
#ifdef SPMV_PROBE_FOLD
  {
    static double* bufs[3] = {nullptr, nullptr, nullptr};
    static int64_t have_rows = 0;
    if (have_rows < pl->num_rows) {
      for (auto& b : bufs) {
        (void)hipFree(b);
        SPMV_CHECK_HIP(hipMalloc(&b, sizeof(double) * ((size_t)pl->num_rows + 64)));
        SPMV_CHECK_HIP(hipMemset(b, 0, sizeof(double) * ((size_t)pl->num_rows + 64)));
      }
      have_rows = pl->num_rows;
      SPMV_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_fold), bufs, sizeof(bufs)));
    }
  }
#endif
  const int grid = sdia_tile_grid(pl);
  const int nrb = (int)((g.NJ + g.block - 1) / g.block);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->sdia_tile_table && pl->sdia_tile_grid == grid) {
    ord.table = pl->sdia_tile_table;
    ord.num_slots = pl->sdia_tile_slots;
  }
#define SPMV_CTILE(DOTV, GENV)                                                 \
  do {                                                                         \
    if (ord.table)                                                             \
      hipLaunchKernelGGL((csr_const_dia_tile_kernel<T, DOTV, GENV, true, R>),  \
                         dim3(grid), dim3(kBlock), 0, st, pl->num_rows,        \
                         pl->sdia_cmask, alpha, in, beta, out, dot, ord, g,    \
                         cv);                                                  \
    else                                                                       \
      hipLaunchKernelGGL((csr_const_dia_tile_kernel<T, DOTV, GENV, false, R>), \
                         dim3(grid), dim3(kBlock), 0, st, pl->num_rows,        \
                         pl->sdia_cmask, alpha, in, beta, out, dot, ord, g,    \
                         cv);                                                  \
  } while (0)
  if (dot.partials) {
    if (pl->sdia_general)
      SPMV_CTILE(true, true);
    else
      SPMV_CTILE(true, false);
  } else {
    if (pl->sdia_general)
      SPMV_CTILE(false, true);
    else
      SPMV_CTILE(false, false);
  }
#undef SPMV_CTILE
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

// the constant-diagonal kernel (`cv`: the plan's constants, or those of the
// fp32 values of the mixed-precision SpMV)
template <typename T>
int sdia_const_launch(const spmv_hip_csr_plan* pl, hipStream_t st,
                      const double* cvals, T alpha, const T* in, T beta, T* out,
                      DotOut dot)
{
  if (pl->sdia_tile > 1) {
    SdiaConsts tcv;
    for (int a = 0; a < 2 * kSdiaMaxOff + 1; ++a)
      tcv.c[a] = cvals[a];
    if (pl->sdia_tile == 2)
      return sdia_tile_launch<T, 2>(pl, st, tcv, alpha, in, beta, out, dot);
    return sdia_tile_launch<T, 4>(pl, st, tcv, alpha, in, beta, out, dot);
  }
  SdiaGeom g = sdia_geom<T>(pl);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const int grid = sdia_grid<T>(pl);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
    ord.table = pl->zw_table;
    ord.num_slots = pl->zw_slots;
  }
  SdiaConsts cv;
  for (int a = 0; a < 2 * kSdiaMaxOff + 1; ++a)
    cv.c[a] = cvals[a];
#define SPMV_CDIA(DOTV, GENV)                                                  \
  do {                                                                         \
    if (ord.table)                                                             \
      hipLaunchKernelGGL((csr_const_dia_kernel<T, DOTV, GENV, true>),          \
                         dim3(grid), dim3(kBlock), 0, st, pl->num_rows,        \
                         pl->sdia_cmask, alpha, in, beta, out, dot, ord, g,    \
                         cv);                                                  \
    else                                                                       \
      hipLaunchKernelGGL((csr_const_dia_kernel<T, DOTV, GENV, false>),         \
                         dim3(grid), dim3(kBlock), 0, st, pl->num_rows,        \
                         pl->sdia_cmask, alpha, in, beta, out, dot, ord, g,    \
                         cv);                                                  \
  } while (0)
  if (dot.partials) {
    if (pl->sdia_general)
      SPMV_CDIA(true, true);
    else
      SPMV_CDIA(true, false);
  } else {
    if (pl->sdia_general)
      SPMV_CDIA(false, true);
    else
      SPMV_CDIA(false, false);
  }
#undef SPMV_CDIA
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

template <typename TV, typename T>
int sdia_launch(const spmv_hip_csr_plan* pl, hipStream_t st, const TV* sval,
                const uint8_t* cmask, T alpha, const T* in, T beta, T* out,
                DotOut dot)
{
  if (pl->sdia_const)
    return sdia_const_launch<T>(pl, st,
                                sizeof(TV) == sizeof(T) ? pl->sdia_cval
                                                        : pl->sdia32_cval,
                                alpha, in, beta, out, dot);
  const SdiaGeom g = sdia_geom<TV>(pl);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const size_t lds = sdia_lds_bytes<TV>(g);
  const int grid = sdia_grid<TV>(pl);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
    ord.table = pl->zw_table;
    ord.num_slots = pl->zw_slots;
  }
#define SPMV_SDIA_T(DOTV, RINGV, TABV)                                         \
  do {                                                                         \
    if (pl->sdia_general)                                                      \
      hipLaunchKernelGGL((csr_sym_dia_kernel<TV, T, DOTV, RINGV, true, TABV>), \
                         dim3(grid), dim3(kBlock), lds, st, pl->num_rows,      \
                         pl->sdia_len, sval, cmask, alpha, in, beta, out, dot, \
                         ord, g);                                              \
    else if constexpr (sizeof(TV) == sizeof(T))                                \
      hipLaunchKernelGGL((csr_sym_dia_kernel<TV, T, DOTV, RINGV, false, TABV>),\
                         dim3(grid), dim3(kBlock), lds, st, pl->num_rows,      \
                         pl->sdia_len, sval, cmask, alpha, in, beta, out, dot, \
                         ord, g);                                              \
    else                                                                       \
      return SPMV_HIP_ENOTSUP; /* mixed precision: general storage only */     \
  } while (0)
#define SPMV_SDIA(DOTV, RINGV)                                                 \
  do {                                                                         \
    if (ord.table)                                                             \
      SPMV_SDIA_T(DOTV, RINGV, true);                                          \
    else                                                                       \
      SPMV_SDIA_T(DOTV, RINGV, false);                                         \
  } while (0)
  if (dot.partials) {
    if (g.ring)
      SPMV_SDIA(true, true);
    else
      SPMV_SDIA(true, false);
  } else {
    if (g.ring)
      SPMV_SDIA(false, true);
    else
      SPMV_SDIA(false, false);
  }
#undef SPMV_SDIA
#undef SPMV_SDIA_T
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

// the plan's table for the kernel it runs WITHOUT a baked copy
int sdia_restore_walk(spmv_hip_csr_plan* pl)
{
  if (pl->symmetric)
    return pl->slat_mask ? spmv_zwalk_order_build(pl, -(int64_t)pl->slat_D[0],
                                                  spmv_slat_grid(pl), 0, false)
                         : SPMV_HIP_OK;
  return (pl->lat_tab && pl->lattice_d2 > 0)
             ? spmv_zwalk_order_build(pl, pl->lattice_d2, spmv_lat_grid(pl), 0,
                                      false)
             : SPMV_HIP_OK;
}

// general plan: the distinct |col - row| > 0, descending, or 0 offsets when
// there are more than three (or none)
int sdia_general_offsets(spmv_hip_csr_plan* pl, hipStream_t st, int* nd, int* U)
{
  int32_t* d_w = nullptr; // [0..7] set, [8] fail
  int32_t h_w[9];
  for (int s = 0; s < 8; ++s)
    h_w[s] = INT32_MAX;
  h_w[8] = 0;
  hipError_t e = hipMalloc(&d_w, sizeof(h_w));
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_w, h_w, sizeof(h_w), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    if (pl->lat_tab) {
      const int nrb = (pl->num_rows + kRows - 1) / kRows;
      hipLaunchKernelGGL(sdia_offsets_from_lattice_kernel,
                         dim3((nrb + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         nrb, pl->lat_tab, d_w, d_w + 8);
    } else {
      const int grid = spmv_grid_for(pl->ctx, pl->num_rows, kBlock);
      hipLaunchKernelGGL(sdia_offsets_kernel, dim3(grid), dim3(kBlock), 0, st,
                         pl->num_rows, pl->rowptr0, pl->colind0, d_w, d_w + 8);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(h_w, d_w, sizeof(h_w), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_w);
  if (e != hipSuccess)
    return static_cast<int>(e);
  *nd = 0;
  if (h_w[8])
    return SPMV_HIP_OK;
  int D[8], n = 0;
  for (int s = 0; s < 8; ++s)
    if (h_w[s] != INT32_MAX)
      D[n++] = h_w[s];
  if (n == 0 || n > kSdiaMaxOff)
    return SPMV_HIP_OK;
  for (int a = 1; a < n; ++a) // insertion sort, descending
    for (int b = a; b > 0 && D[b] > D[b - 1]; --b) {
      const int tmp = D[b];
      D[b] = D[b - 1];
      D[b - 1] = tmp;
    }
  for (int k = 0; k < n; ++k)
    U[k] = D[k];
  *nd = n;
  return SPMV_HIP_OK;
}

// Allocate and fill one baked copy for the offsets in pl->sdia_nd / sdia_U:
// SPMV_HIP_ENOTSUP when the geometry does not fit or (general) the matrix is
// not symmetric.
template <typename T>
int sdia_fill(spmv_hip_csr_plan* pl, bool general, const T* values,
              const T* diagonal, hipStream_t st, void** out_val,
              uint8_t** out_cmask, int64_t* out_len)
{
  const SdiaGeom g = sdia_geom<T>(pl);
  for (int j = 0; j < g.nwin; ++j)
    if ((g.last[j] >> 6) + 1 > kBlock / 64)
      return SPMV_HIP_ENOTSUP;
  if (sdia_lds_bytes<T>(g) > 150 * 1024)
    return SPMV_HIP_ENOTSUP;
  const int32_t n = pl->num_rows;
  // every window of every row block stays inside its array
  const int64_t len = (((int64_t)n + kRows - 1) / kRows) * kRows + 2 * kRows;
  const int narr = pl->sdia_general == 2 ? 2 * g.nd + 1 : g.nd + 1;
  const size_t bytes = (size_t)narr * len * sizeof(T);
  void* sval = nullptr;
  uint8_t* cm = nullptr;
  int32_t* d_fail = nullptr;
  int32_t h_fail = 0;
  hipError_t e = hipMalloc(&sval, bytes);
  if (e == hipSuccess)
    e = hipMalloc(&cm, (size_t)n);
  if (e == hipSuccess)
    e = hipMalloc(&d_fail, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipMemsetAsync(sval, 0, bytes, st);
  if (e == hipSuccess)
    e = hipMemsetAsync(d_fail, 0, sizeof(int32_t), st);
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, n, kBlock);
    if (general && pl->sdia_general == 2)
      hipLaunchKernelGGL((sdia_bake_general_kernel<T, true, true>), dim3(grid),
                         dim3(kBlock), 0, st, n, g.nd, g.U[0], g.U[1], g.U[2],
                         pl->rowptr0, pl->colind0, values, len,
                         static_cast<T*>(sval), cm, d_fail, 0);
    else if (general) // (the mirrors were checked by sdia_is_symmetric)
      hipLaunchKernelGGL((sdia_bake_general_kernel<T, false, true>), dim3(grid),
                         dim3(kBlock), 0, st, n, g.nd, g.U[0], g.U[1], g.U[2],
                         pl->rowptr0, pl->colind0, values, len,
                         static_cast<T*>(sval), cm, d_fail, 0);
    else
      hipLaunchKernelGGL((sdia_bake_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                         n, g.nd, g.U[0], g.U[1], g.U[2], pl->rowptr0,
                         pl->slat_mask, values, diagonal, len,
                         static_cast<T*>(sval), cm);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_fail, d_fail, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_fail);
  if (e != hipSuccess || h_fail) { // not symmetric: nothing changes
    (void)hipFree(sval);
    (void)hipFree(cm);
    return e != hipSuccess ? static_cast<int>(e) : SPMV_HIP_ENOTSUP;
  }
  *out_val = sval;
  *out_cmask = cm;
  *out_len = len;
  return SPMV_HIP_OK;
}

// general plan: can the HALF form hold this matrix (offsets in pl->sdia_U,
// every entry with its mirror, bit for bit)?  Nothing is allocated or written.
template <typename T>
int sdia_is_symmetric(spmv_hip_csr_plan* pl, const T* values, hipStream_t st,
                      bool* yes)
{
  int32_t* d_fail = nullptr;
  int32_t h_fail = 1;
  hipError_t e = hipMalloc(&d_fail, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_fail, 0, sizeof(int32_t), st);
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, pl->num_rows, kBlock);
    hipLaunchKernelGGL((sdia_bake_general_kernel<T, false, false>), dim3(grid),
                       dim3(kBlock), 0, st, pl->num_rows, pl->sdia_nd,
                       pl->sdia_U[0], pl->sdia_U[1], pl->sdia_U[2], pl->rowptr0,
                       pl->colind0, values, (int64_t)0, (T*)nullptr,
                       (uint8_t*)nullptr, d_fail, 1);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_fail, d_fail, sizeof(int32_t), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_fail);
  *yes = h_fail == 0;
  return e == hipSuccess ? SPMV_HIP_OK : static_cast<int>(e);
}

// Are the diagonals constant (offsets in pl->sdia_nd / sdia_U)?  Two passes
// over the values, nothing allocated but the 120-byte probe record.
template <typename T>
int sdia_const_probe(spmv_hip_csr_plan* pl, bool general, const T* values,
                     const T* diagonal, hipStream_t st, bool* yes, double* cvals,
                     uint8_t* cmask_out)
{
  *yes = false;
  SdiaConstProbe* d_pr = nullptr;
  SdiaConstProbe h_pr;
  hipError_t e = hipMalloc(&d_pr, sizeof(SdiaConstProbe));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_pr, 0, sizeof(SdiaConstProbe), st);
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, pl->num_rows, kBlock);
    const int nd = pl->sdia_nd, u0 = pl->sdia_U[0], u1 = pl->sdia_U[1],
              u2 = pl->sdia_U[2];
    // pass 1 looks at the first few planes only (every diagonal of a lattice
    // shows up there); a diagonal it has not seen makes pass 2 fail unless its
    // entries are +0.0 -- the values are streamed then, nothing is lost but
    // the shortcut
    const int64_t prefix = 4 * (int64_t)u0 + 4096;
    const int32_t pick_rows
        = prefix < pl->num_rows ? (int32_t)prefix : pl->num_rows;
    if (general) {
      hipLaunchKernelGGL((sdia_const_kernel<T, true, false>), dim3(grid),
                         dim3(kBlock), 0, st, pick_rows, pl->num_rows, nd, u0, u1,
                         u2,
                         pl->rowptr0, pl->colind0, (const uint8_t*)nullptr,
                         values, diagonal, d_pr, (uint8_t*)nullptr);
      hipLaunchKernelGGL((sdia_const_kernel<T, true, true>), dim3(grid),
                         dim3(kBlock), 0, st, pl->num_rows, pl->num_rows, nd, u0,
                         u1, u2,
                         pl->rowptr0, pl->colind0, (const uint8_t*)nullptr,
                         values, diagonal, d_pr, cmask_out);
    } else {
      hipLaunchKernelGGL((sdia_const_kernel<T, false, false>), dim3(grid),
                         dim3(kBlock), 0, st, pick_rows, pl->num_rows, nd, u0, u1,
                         u2,
                         pl->rowptr0, pl->colind0, pl->slat_mask, values,
                         diagonal, d_pr, (uint8_t*)nullptr);
      hipLaunchKernelGGL((sdia_const_kernel<T, false, true>), dim3(grid),
                         dim3(kBlock), 0, st, pl->num_rows, pl->num_rows, nd, u0,
                         u1, u2,
                         pl->rowptr0, pl->colind0, pl->slat_mask, values,
                         diagonal, d_pr, cmask_out);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_pr, d_pr, sizeof(SdiaConstProbe), hipMemcpyDeviceToHost,
                       st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_pr);
  if (e != hipSuccess)
    return static_cast<int>(e);
  if (h_pr.fail)
    return SPMV_HIP_OK;
  for (int a = 0; a < 2 * kSdiaMaxOff + 1; ++a) {
    cvals[a] = 0.0;
    if (!h_pr.seen[a])
      continue;
    if constexpr (sizeof(T) == 8) {
      double v;
      memcpy(&v, &h_pr.bits[a], sizeof(v));
      cvals[a] = v;
    } else {
      const unsigned b = (unsigned)h_pr.bits[a];
      float v;
      memcpy(&v, &b, sizeof(v));
      cvals[a] = (double)v; // exact
    }
  }
  if (!general) // the column entries ARE the lower entries
    for (int k = 0; k < pl->sdia_nd; ++k)
      cvals[pl->sdia_nd + 1 + k] = cvals[k];
  *yes = true;
  return SPMV_HIP_OK;
}

template <typename T>
int sdia_bake(spmv_hip_csr_plan* pl, const T* values, const T* diagonal,
              hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  const auto t_begin = std::chrono::steady_clock::now();
  // SPMV_PLAN_TRACE=1: the phases' wall times on stderr
  const bool trace = getenv("SPMV_PLAN_TRACE") != nullptr;
  auto mark = [&](const char* what) {
    if (trace)
      fprintf(stderr, "sdia_bake %-12s %8.3f ms\n", what,
              std::chrono::duration<double, std::milli>(
                  std::chrono::steady_clock::now() - t_begin)
                  .count());
  };
  const bool had = pl->sdia_val != nullptr;
  SPMV_REQUIRE((values == nullptr && diagonal == nullptr)
               || (values
                   && (pl->symmetric ? diagonal != nullptr : diagonal == nullptr)));
  // An earlier copy goes first (its pointers may be the ones passed again).
  // The plane-walk table is tied to the kernel the plan runs: with a baked
  // copy it is the diagonal-form kernel's, otherwise the one plan creation
  // built for the CSR-order lattice kernel.  Every exit that does not install
  // a new copy must therefore leave the CSR-order kernel's table: untouched
  // when there was no copy (`had` false), rebuilt when there was one.
  sdia_free_arrays(pl);
  mark("freed");
  auto unchanged = [&](int rc) {
    if (had) {
      spmv_zwalk_free(pl);
      pl->zw_d2 = 0;
      const int rw = sdia_restore_walk(pl);
      if (rw != SPMV_HIP_OK)
        return rw;
    }
    return rc;
  };
  if (values == nullptr && diagonal == nullptr) // dropped
    return unchanged(SPMV_HIP_OK);
  const bool general = !pl->symmetric;
  if (pl->nnz == 0)
    return unchanged(SPMV_HIP_ENOTSUP);
  if (general) {
    // rests on the lattice form (ascending columns without repeats, a row
    // block's worth of structure) of a square matrix
    if (!pl->ctx->bake_general || !pl->lat_tab || pl->num_rows != pl->num_cols)
      return unchanged(SPMV_HIP_ENOTSUP);
    int nd = 0, U[kSdiaMaxOff] = {0, 0, 0};
    const int rc = sdia_general_offsets(pl, st, &nd, U);
    if (rc != SPMV_HIP_OK)
      return unchanged(rc);
    if (nd == 0)
      return unchanged(SPMV_HIP_ENOTSUP);
    pl->sdia_nd = nd;
    for (int k = 0; k < kSdiaMaxOff; ++k)
      pl->sdia_U[k] = U[k];
  } else {
    // rests on the symmetric lattice analysis
    if (!pl->slat_mask)
      return unchanged(SPMV_HIP_ENOTSUP);
    pl->sdia_nd = pl->slat_nd;
    for (int k = 0; k < kSdiaMaxOff; ++k)
      pl->sdia_U[k] = -pl->slat_D[k];
  }
  void* sval = nullptr;
  uint8_t* cm = nullptr;
  int64_t len = 0;
  {
    // general storage: the half form when the matrix is symmetric bit for
    // bit (checked first, without allocating), else the full form (the
    // geometry reads the mode from the plan)
    pl->sdia_general = 0;
    pl->sdia_const = 0;
    int rc = SPMV_HIP_OK;
    // constant diagonals first: no copy of the values at all (and no need
    // for symmetry -- the upper diagonals have constants of their own)
    bool is_const = false;
    if (pl->ctx->const_diagonals) {
      // (the second pass writes the mask byte on the way: one pass fewer)
      hipError_t ea = hipMalloc(&cm, (size_t)pl->num_rows);
      if (ea == hipSuccess)
        ea = hipMalloc(&sval, 64); // the "baked" marker every check looks at
      rc = ea == hipSuccess ? SPMV_HIP_OK : static_cast<int>(ea);
      if (rc == SPMV_HIP_OK)
        rc = sdia_const_probe<T>(pl, general, values, diagonal, st, &is_const,
                                 pl->sdia_cval, cm);
      if (rc != SPMV_HIP_OK || !is_const) {
        (void)hipFree(cm);
        (void)hipFree(sval);
        cm = nullptr;
        sval = nullptr;
      }
    }
    if (rc == SPMV_HIP_OK && is_const) {
      pl->sdia_general = general ? 2 : 0;
      pl->sdia_const = 1; // (sdia_grid reads it)
    } else {
      mark("const probe");
      if (rc == SPMV_HIP_OK && general) {
        bool symmetric = false;
        rc = sdia_is_symmetric<T>(pl, values, st, &symmetric);
        pl->sdia_general = symmetric ? 1 : 2;
      }
      mark("symmetry");
      if (rc == SPMV_HIP_OK)
        rc = sdia_fill<T>(pl, general, values, diagonal, st, &sval, &cm, &len);
      mark("fill");
    }
    if (rc != SPMV_HIP_OK) {
      pl->sdia_general = 0;
      pl->sdia_nd = 0;
      return unchanged(rc);
    }
  }
  const SdiaGeom g = sdia_geom<T>(pl);
  pl->sdia_val = sval;
  pl->sdia_cmask = cm;
  pl->sdia_len = len;
  pl->sdia_elem = (int)sizeof(T);
  pl->sdia_values0 = values;
  pl->sdia_diag0 = diagonal;
  pl->sdia = 1;
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count(); // part of what the plan cost
  // measured at 512^3 (4 workgroups per CU: 1024 = the row blocks of one
  // plane, so every workgroup walks straight down z and finds its far column
  // window in the block it reads next): plain order 1.47 ms, 8 consecutive
  // row blocks per XCD 1.43
  if (!general)
    pl->lat_xcd_group = 8;
  pl->slat_blocks_per_cu = 4;
  // the plane-walk order makes that true for every size (planes = the
  // farthest offset apart)
  const int rw = spmv_zwalk_order_build(pl, g.U[0], sdia_grid<T>(pl), 0, false);
  if (rw != SPMV_HIP_OK || !pl->sdia_const)
    return rw;
  // constant diagonals of a 3-D lattice: several lines per lane
  return spmv_sdia_tile_build(pl, pl->ctx->const_tile, 0, false);
}

// The fp32 copy of the mixed-precision SpMV (general plans whose fp64 values
// are baked): the same arrays filled from the caller's fp32 values -- the same
// device check, on the fp32 bits.
int sdia_bake_mixed(spmv_hip_csr_plan* pl, const float* values32, hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  (void)hipFree(pl->sdia32_val);
  (void)hipFree(pl->sdia32_cmask);
  pl->sdia32_val = nullptr;
  pl->sdia32_cmask = nullptr;
  pl->sdia32_values0 = nullptr;
  if (values32 == nullptr)
    return SPMV_HIP_OK; // dropped
  if (!pl->sdia_val || !pl->sdia_general || pl->sdia_elem != 8)
    return SPMV_HIP_ENOTSUP;
  const auto t_begin = std::chrono::steady_clock::now();
  if (pl->sdia_const) {
    // constant diagonals: the fp32 array must have them too (its own constants)
    bool is_const = false;
    const int rcc = sdia_const_probe<float>(pl, true, values32, nullptr, st,
                                            &is_const, pl->sdia32_cval, nullptr);
    if (rcc != SPMV_HIP_OK)
      return rcc;
    if (!is_const)
      return SPMV_HIP_ENOTSUP;
    SPMV_CHECK_HIP(hipMalloc(&pl->sdia32_val, 64)); // the "baked" marker
    pl->sdia32_values0 = values32;
    pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                       std::chrono::steady_clock::now() - t_begin)
                       .count();
    return SPMV_HIP_OK;
  }
  if (pl->sdia_general == 1) {
    // the half form needs THESE values symmetric too (the caller's fp32 array
    // is normally the rounded fp64 one, but nothing says so)
    bool symmetric = false;
    const int rcs = sdia_is_symmetric<float>(pl, values32, st, &symmetric);
    if (rcs != SPMV_HIP_OK)
      return rcs;
    if (!symmetric)
      return SPMV_HIP_ENOTSUP;
  }
  void* sval = nullptr;
  uint8_t* cm = nullptr;
  int64_t len = 0;
  const int rc = sdia_fill<float>(pl, true, values32, nullptr, st, &sval, &cm,
                                  &len);
  if (rc != SPMV_HIP_OK)
    return rc;
  pl->sdia32_val = sval;
  pl->sdia32_cmask = cm;
  pl->sdia32_values0 = values32;
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count();
  return SPMV_HIP_OK;
}

} // namespace

// (Re)build the tile kernel's geometry and plane-walk table: R lines per lane
// (1 = the one-line kernel), `segments` runs along the plane axis (0 = choose).
// Needs the constant form of a lattice with three lower offsets.
int spmv_sdia_tile_build(spmv_hip_csr_plan* pl, int R, int segments, bool force)
{
  SPMV_REQUIRE(R == 1 || R == 2 || R == 4);
  if (pl->sdia_tile_table) {
    SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
    SPMV_CHECK_HIP(hipDeviceSynchronize()); // no launch still reads the old one
    (void)hipFree(pl->sdia_tile_table);
    pl->sdia_tile_table = nullptr;
  }
  pl->sdia_tile_slots = pl->sdia_tile_grid = pl->sdia_tile_segments = 0;
  pl->sdia_tile = 0;
  if (R == 1 || !pl->sdia_const || pl->sdia_nd != 3
      || pl->sdia_U[1] >= (1 << 23))
    return SPMV_HIP_OK;
  pl->sdia_tile = R;
  const int64_t u0 = pl->sdia_U[0], u1 = pl->sdia_U[1];
  if (u0 % (R * u1) != 0)
    return SPMV_HIP_OK; // planes do not line up in j-space: the plain order
  const int grid = sdia_tile_grid(pl);
  // the table builder counts in blocks of 256 rows: hand it the block counts
  // (blocks of `block` items; a plane is whole blocks or the blocks are 256)
  const int block = sdia_tile_block(pl, R);
  const int64_t nblocks = (sdia_tile_items(pl, R) + block - 1) / block;
  const int64_t plane = u0 / R;
  const int64_t rows_eq = block == kRows ? sdia_tile_items(pl, R)
                                         : nblocks * kRows;
  const int64_t plane_eq = block == kRows ? plane : plane / block * kRows;
  const int rc = spmv_zwalk_table_device(
      pl, rows_eq, plane_eq, grid, segments, force, &pl->sdia_tile_table,
      &pl->sdia_tile_slots, &pl->sdia_tile_segments);
  if (rc == SPMV_HIP_OK && pl->sdia_tile_table)
    pl->sdia_tile_grid = grid;
  return rc;
}

void spmv_sdia_free(spmv_hip_csr_plan* pl)
{
  sdia_free_arrays(pl);
  spmv_zwalk_free(pl);
  pl->zw_d2 = 0;
}

// the baked copies alone; the plane-walk table is the caller's business
static void sdia_free_arrays(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->sdia_val);
  (void)hipFree(pl->sdia_cmask);
  (void)hipFree(pl->sdia32_val);
  (void)hipFree(pl->sdia32_cmask);
  pl->sdia32_val = nullptr;
  pl->sdia32_cmask = nullptr;
  pl->sdia32_values0 = nullptr;
  pl->sdia_val = nullptr;
  pl->sdia_cmask = nullptr;
  pl->sdia_values0 = pl->sdia_diag0 = nullptr;
  pl->sdia_len = 0;
  pl->sdia_elem = 0;
  pl->sdia_const = 0;
  (void)hipFree(pl->sdia_tile_table);
  pl->sdia_tile_table = nullptr;
  pl->sdia_tile_slots = pl->sdia_tile_grid = pl->sdia_tile_segments = 0;
  pl->sdia_tile = 0;
  pl->sdia = 0;
  pl->sdia_general = 0;
  pl->sdia_nd = 0;
}

int spmv_sdia_grid(const spmv_hip_csr_plan* pl)
{
  return pl->sdia_elem == 4 ? sdia_grid<float>(pl) : sdia_grid<double>(pl);
}

int spmv_sdia_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       const double* diagonal, hipStream_t st)
{
  return sdia_bake<double>(pl, values, diagonal, st);
}

int spmv_sdia_bake_f32(spmv_hip_csr_plan* pl, const float* values,
                       const float* diagonal, hipStream_t st)
{
  return sdia_bake<float>(pl, values, diagonal, st);
}

int spmv_sdia_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32,
                          hipStream_t st)
{
  return sdia_bake_mixed(pl, values32, st);
}

int spmv_sdia_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot)
{
  return sdia_launch<double, double>(pl, st,
                                     static_cast<const double*>(pl->sdia_val),
                                     pl->sdia_cmask, alpha, in, beta, out, dot);
}

int spmv_sdia_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out)
{
  return sdia_launch<float, float>(pl, st,
                                   static_cast<const float*>(pl->sdia_val),
                                   pl->sdia_cmask, alpha, in, beta, out,
                                   DotOut());
}

int spmv_sdia_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                         double alpha, const double* in, double beta,
                         double* out, DotOut dot)
{
  return sdia_launch<float, double>(pl, st,
                                    static_cast<const float*>(pl->sdia32_val),
                                    pl->sdia32_cmask, alpha, in, beta, out, dot);
}
