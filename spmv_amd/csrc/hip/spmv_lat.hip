// Lattice form of the general CSR SpMV for gfx950 (MI355X).
//
// Stands behind the same CSRSpMV<T>::init/run hook as the other general
// kernels (spmv/csr_kernels.h:26-78); arithmetic and summation order are those
// of spmv/csr_kernels.cpp:41-51, so results are bit-identical to the oracle.
//
// Matrices assembled on a structured grid (the 7-point Poisson matrix of the
// benchmark, any stencil with <= 8 points) have a property the CSR format does
// not exploit: inside a block of consecutive rows every column index is
// `row + d` for one of a handful of constant offsets d.  Plan creation checks
// that, per block of 256 rows (lat_build_kernel):
//   * the block's distinct offsets, ascending: D[0..nd), nd <= 8
//   * per row one byte: bit k set <=> the row has an entry in column row + D[k]
//   * every row's entries must appear in ascending column order without
//     repeats, so that "walk the set bits from k = 0 up" IS the row's
//     left-to-right order of csr_kernels.cpp:46-47.
// If every row block qualifies the plan takes this form, and the kernel
//   1. reads neither colind nor (per row) the row pointer: 8 B per entry
//      (values) + 1 B per row (mask) instead of 12 B per entry + 4 B per row --
//      at 512^3 9.8 GB per SpMV instead of 13.9 GB (LX form: 12.1 GB).  A
//      row's position in `values` is the block's start + the entries of the
//      earlier waves (three 16-bit counts in the block's record) + the mask
//      popcounts of the lower lanes of its wave (four ballots);
//   2. moves `values` straight into LDS with LDS-DMA
//      (global_load_lds_dwordx4: no VGPRs, no LDS store instructions), always
//      one row block AHEAD of the one being summed (two LDS slots), so the
//      matrix stream never stops while a workgroup computes: the LX kernel
//      alternates load and compute phases and reached only ~75 % of the
//      streaming rate of the same bytes;
//   3. loads x for column row + D[k] directly, coalesced (lane = row), into
//      registers: no staging pass, no index arithmetic.
// One workgroup barrier per row block (the LX kernel needs six).
//
// Plane walk and x chain: when the plan finds a 3-D lattice the row blocks are
// walked column by column from plane to plane (spmv_zwalk_order_build) and
// the x a row holds for the plane ahead is handed to the next step in
// registers (lat_loads): 448^3 1.51 -> 1.39 ms, fabric reads at 512^3 12.2 ->
// 9.6 GB per launch.
//
// Measured on MI355X, 512^3 (profiles/r02_*): 1.98-2.07 ms against 2.41-2.62 ms
// for the LX kernel on the same boxes.  (Matrices that are also SYMMETRIC leave
// this kernel for the diagonal form, spmv_symdia.hip: 1.35 ms.)  Variants measured and dropped: waves
// with private slots and no barrier at all (equal at 16 waves per CU, 9 %
// slower at 32), two rows per lane with 16-byte x / y accesses (17 % slower),
// non-temporal y stores (-1.8 %).  What the time is made of (ablations, same
// box): without the y store 1.70 ms, without the x loads of seven of the
// eight offsets 2.0 ms (no change), without the values 1.15 ms.
//
//   lat_tab[rb*kLatRec + 0]      nd, number of offsets of row block rb (-1: not
//                                in lattice form)
//   lat_tab[rb*kLatRec + 1]      k0: D[k0] == 0 (padding included), or -1
//   lat_tab[rb*kLatRec + 2, 3]   entries in front of the waves 1, 2, 3 of the
//                                block (16 bits each): row positions come
//                                from these and the mask popcounts
//   lat_tab[rb*kLatRec + 4 + k]  D[k], ascending, padded with 0
//   lat_mask[row]                presence bits
#include "csr_plan.h"
#include "lat_dma.h"

#include <new>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

namespace
{

// LDS slot for one row block's values: 256 rows x 8 entries plus the slack of
// the 16-byte alignment of the first DMA piece, in whole 1-KiB DMA pieces
constexpr int kLatSlotBytes = (kRows * kLatMaxOff * 8 + 1024);
constexpr int kLatSlots = 2;

typedef int i32x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------
// The kernel.  lane = row.  A persistent workgroup walks its row blocks with
// everything one block ahead:
//   top     wait + ONE barrier: block k's values (DMA issued an iteration ago)
//           and its register loads have landed, for every wave; every wave has
//           left block k-1, whose LDS slot is free again
//   issue   DMA of block k+1's values into that slot; block k+1's loads into
//           registers: mask byte, x[row + D[j]] for all 8 j (D is
//           padded with 0 = the row's own x, a cache hit), y for beta != 0 --
//           all independent of each other
//   sum     block k: own row out of LDS, left to right over the set mask bits
// so a workgroup always has a whole row block (~20 KB) in flight while it
// computes, and nothing in an iteration waits for a load issued in it.
// ---------------------------------------------------------------------------
template <typename T>
struct LatRegs {
  int32_t wbase; // entries of the block in front of this lane's wave (uniform
                 // per wave)
  unsigned m;
  int k0; // uniform: xk[k0] is the row's own x (DOT), -1 = x_own was loaded
  T xk[kLatMaxOff];
  T y0, x_own;
};

struct LatBlock {
  int rb;       // row block, -1 = none
  int64_t a, b; // its span in `values`
  // the block's record (uniform): where a zero offset sits, the entries in
  // front of the waves 1..3, the offsets
  int k0;
  int32_t c12, c3;
  i32x8 D;
};

// A block's record and span travel as ONE vector load (lane l < 12: word l of
// the record; lanes 12, 13: the row pointer at the block's first row and
// behind its last) issued a whole step before they are needed, and are taken
// apart with v_readlane right behind the step's wait.  Scalar loads would
// share their counter with the LDS reads of the row sums, which then wait for
// a record coming from HBM (see csr_lxw_kernel).
__device__ __forceinline__ int32_t lat_fetch(int rb, int32_t num_rows,
                                             const int32_t* __restrict__ rowptr,
                                             const int32_t* __restrict__ tab)
{
  int32_t w = 0;
  if (rb >= 0) {
    const int l = (int)(threadIdx.x & 63);
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);
    const int32_t* p = l < kLatRec ? tab + (int64_t)rb * kLatRec + l
                                   : rowptr + (l == kLatRec ? r0 : r0 + nr);
    if (l <= kLatRec + 1)
      w = *p;
  }
  return w;
}

__device__ __forceinline__ LatBlock lat_decode(int rb, int32_t w)
{
  LatBlock blk;
  blk.rb = -1;
  blk.a = blk.b = 0;
  blk.k0 = 0;
  blk.c12 = blk.c3 = 0;
#pragma unroll
  for (int k = 0; k < kLatMaxOff; ++k)
    blk.D[k] = 0;
  if (rb >= 0) {
    blk.rb = rb;
    blk.k0 = __builtin_amdgcn_readlane(w, 1);
    blk.c12 = __builtin_amdgcn_readlane(w, 2);
    blk.c3 = __builtin_amdgcn_readlane(w, 3);
#pragma unroll
    for (int k = 0; k < kLatMaxOff; ++k)
      blk.D[k] = __builtin_amdgcn_readlane(w, 4 + k);
    blk.a = __builtin_amdgcn_readlane(w, kLatRec);
    blk.b = __builtin_amdgcn_readlane(w, kLatRec + 1);
  }
  return blk;
}

// Plane chain (`chain` > 0, uniform): the block lies exactly `chain` rows -- one
// lattice plane -- behind `prev_rb`, whose loads are `prev`.  Then its x[r]
// is prev's x[r + chain] and its x[r - chain] prev's x[r]: handed over in
// registers instead of loaded again (2 of the 7 x loads of a 7-point row).
template <typename T, bool DOT>
__device__ __forceinline__ LatRegs<T> lat_loads(
    const LatBlock& blk, int t, int32_t num_rows, int32_t num_cols,
    const uint8_t* __restrict__ mask, const T* __restrict__ in, T beta,
    const T* __restrict__ out, int chain, const LatBlock& pblk,
    const LatRegs<T>& prev)
{
  LatRegs<T> g;
  g.wbase = 0;
  g.m = 0;
  g.k0 = 0;
  g.y0 = g.x_own = T(0);
#pragma unroll
  for (int k = 0; k < kLatMaxOff; ++k)
    g.xk[k] = T(0);
  if (blk.rb < 0)
    return g;
  const int32_t r0 = blk.rb * kRows;
  const int32_t r = r0 + t;
  if (r < num_rows) {
    const i32x8 D = blk.D;
    g.k0 = blk.k0; // position of a zero offset in D, or -1
    const int32_t c12 = blk.c12, c3 = blk.c3;
    const int w = t >> 6;
    g.wbase = w == 0 ? 0 : (w == 1 ? (c12 & 0xffff) : (w == 2 ? (c12 >> 16) : c3));
    g.m = mask[r];
    T x_ahead = T(0), x_here = prev.x_own;
    if (chain > 0) { // what the previous block holds for this one
      const i32x8 Dp = pblk.D;
      bool have_ahead = false, have_here = prev.k0 < 0 && DOT;
#pragma unroll
      for (int k = 0; k < kLatMaxOff; ++k) {
        if (Dp[k] == chain) { // uniform
          x_ahead = prev.xk[k];
          have_ahead = true;
        }
        if (Dp[k] == 0) {
          x_here = prev.xk[k];
          have_here = true;
        }
      }
      if (!have_ahead || !have_here)
        chain = 0;
    }
#pragma unroll
    for (int k = 0; k < kLatMaxOff; ++k) {
      // unconditional (no dependence on the mask load); a column the row does
      // not have is clamped into range and its value ignored
      int64_t c = (int64_t)r + D[k];
      c = c < 0 ? 0 : (c >= num_cols ? num_cols - 1 : c);
      if (chain > 0 && D[k] == 0) // uniform
        g.xk[k] = x_ahead;
      else if (chain > 0 && D[k] == -chain)
        g.xk[k] = x_here;
      else
        g.xk[k] = in[c];
    }
    if (beta != T(0))
      g.y0 = out[r];
    if constexpr (DOT)
      if (g.k0 < 0) // uniform: eight offsets, none of them 0
        g.x_own = chain > 0 ? x_ahead : in[r];
  }
  return g;
}

// TV = type of `values` (what is streamed), T = type of x, y and of all the
// arithmetic.  TV = float with T = double is the mixed-precision SpMV (SURVEY
// 8f n3): half the matrix bytes, every product and sum still in fp64.
template <typename TV, typename T, bool DOT, bool NT, bool TAB>
__global__ __launch_bounds__(kBlock) void csr_lattice_kernel(
    int32_t num_rows, int32_t num_cols, int64_t nnz,
    const int32_t* __restrict__ rowptr, const TV* __restrict__ values,
    const int32_t* __restrict__ tab, const uint8_t* __restrict__ mask, T alpha,
    const T* __restrict__ in, T beta, T* __restrict__ out, DotOut dot,
    RowBlockOrder ord, int chain_rows)
{
  constexpr int V = 16 / (int)sizeof(TV);
  constexpr int SLOT = kLatSlotBytes / (int)sizeof(TV); // entries per slot
  __shared__ __attribute__((aligned(16))) TV s_val[kLatSlots * SLOT];
  __shared__ double s_red[kBlock / 64];

  const int t = threadIdx.x;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;

  // Slots it, it + stride, ...: every one is a step, an empty slot a step
  // without work.  Three blocks are in preparation: nxt (span known, DMA and
  // loads issued in the current step), the one after (its order-table entry
  // has landed; its span is fetched in the current step), and the one after
  // that (table entry requested in the current step).
  int it = blockIdx.x;
  LatBlock cur;
  int nxt_rb;
  int32_t nxt_w; // the next block's record, in flight
  {
    const int rb0 = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it, num_slots));
    nxt_rb = order_slot_decode(ord, order_slot_raw_t<TAB>(ord, it + stride, num_slots));
    const int32_t w0 = lat_fetch(rb0, num_rows, rowptr, tab);
    nxt_w = lat_fetch(nxt_rb, num_rows, rowptr, tab);
    cur = lat_decode(rb0, w0);
  }
  int nn_raw = order_slot_raw_t<TAB>(ord, it + 2 * stride, num_slots);
  if (cur.rb >= 0 && cur.b > cur.a)
    lat_issue_dma<TV, NT>(values, nnz, cur.a & ~(int64_t)(V - 1), cur.b, s_val,
                          t);
  LatRegs<T> gB;
  gB.k0 = 0;
  gB.x_own = T(0);
  LatRegs<T> gA = lat_loads<T, DOT>(cur, t, num_rows, num_cols, mask, in, beta,
                                    out, 0, cur, gB);
  int slot = 0;
  // y is stored one step late, right behind the next step's wait (which
  // covers stores too): see csr_lxw_kernel
  T y_late = T(0);
  int32_t r_late = -1;
  // one step: sums block `cur` out of registers g, loads block `nxt` into gn
  auto step = [&](const LatRegs<T>& g, LatRegs<T>& gn) {
    // Everything this wave has in flight (block k's DMA pieces and loads, the
    // previous y stores) has landed; after the barrier that holds for all
    // waves, and all of them have left block k-1.  (The builtin, not inline
    // assembly: the compiler's own wait-count bookkeeping must see it, or it
    // would wait for "its" loads again after the barrier -- and with them for
    // the DMA pieces issued there.)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    if (r_late >= 0)
      out[r_late] = y_late;
    r_late = -1;
    const LatBlock nxt = lat_decode(nxt_rb, nxt_w);
    if (nxt.rb >= 0 && nxt.b > nxt.a)
      lat_issue_dma<TV, NT>(values, nnz, nxt.a & ~(int64_t)(V - 1), nxt.b,
                            s_val + (slot ^ 1) * SLOT, t);
    // one lattice plane below the current block: x handed on in registers
    const int chain = (chain_rows > 0 && nxt.rb >= 0 && cur.rb >= 0
                       && (nxt.rb - cur.rb) * kRows == chain_rows)
                          ? chain_rows
                          : 0;
    gn = lat_loads<T, DOT>(nxt, t, num_rows, num_cols, mask, in, beta, out,
                           chain, cur, g);
    // the block after the next one: its table entry has landed with the wait
    // above; its record is needed a step from now
    const int nn_rb = order_slot_decode(ord, nn_raw);
    const int32_t nn_w = lat_fetch(nn_rb, num_rows, rowptr, tab);
    const int nnn_raw = order_slot_raw_t<TAB>(ord, it + 3 * stride, num_slots);
    const int32_t r = cur.rb * kRows + t;
    if (cur.rb >= 0 && r < num_rows) {
      // the row's first entry: block start + entries of the earlier waves +
      // entries of the lower lanes of this wave (sum of their mask
      // popcounts, <= 8 each: four ballots, one per bit of the count)
      const unsigned cnt = __popc(g.m);
      int before = 0;
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const uint64_t has = __ballot((cnt >> bit) & 1u);
        before += (int)__builtin_amdgcn_mbcnt_hi(
                      (unsigned)(has >> 32),
                      __builtin_amdgcn_mbcnt_lo((unsigned)has, 0u))
                  << bit;
      }
      const int rel = slot * SLOT + (int)(cur.a & (int64_t)(V - 1)) + g.wbase
                      + before;
      // all LDS reads first (independent), then the adds in entry order
      T v[kLatMaxOff];
#pragma unroll
      for (int k = 0; k < kLatMaxOff; ++k) {
        const int pk = __popc(g.m & ((1u << k) - 1u));
        v[k] = (T)s_val[min(rel + pk, kLatSlots * SLOT - 1)];
      }
      T sum = 0;
#pragma unroll
      for (int k = 0; k < kLatMaxOff; ++k)
        if ((g.m >> k) & 1u) // csr_kernels.cpp:46-47, left to right
          sum += v[k] * g.xk[k];
      const T c = alpha * sum;
      T y = c;
      if (beta != T(0))
        y = c + beta * g.y0;
      y_late = y;
      r_late = r;
      if constexpr (DOT) {
        T xo = g.x_own;
#pragma unroll
        for (int k = 0; k < kLatMaxOff; ++k)
          if (k == g.k0) // uniform
            xo = g.xk[k];
        dot_acc += (double)xo * (double)c;
      }
    }
    slot ^= 1;
    cur = nxt;
    nxt_rb = nn_rb;
    nxt_w = nn_w;
    nn_raw = nnn_raw;
    it += stride;
  };
  // two steps per trip: the two register sets swap roles without being copied
  // (a copy would have to wait for the loads it moves)
  while (it < num_slots) {
    step(gA, gB);
    if (it >= num_slots)
      break;
    step(gB, gA);
  }
  if (r_late >= 0)
    out[r_late] = y_late;
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// Plan-time analysis: one workgroup per row block, lane = row.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t block_min(int32_t v, int32_t* s_tmp)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v = min(v, __shfl_down(v, off, 64));
  __syncthreads(); // s_tmp free again
  if ((threadIdx.x & 63) == 0)
    s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  int32_t r = s_tmp[0];
#pragma unroll
  for (int w = 1; w < kBlock / 64; ++w)
    r = min(r, s_tmp[w]);
  return r; // every thread
}

__global__ __launch_bounds__(kBlock) void lat_build_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ tab,
    uint8_t* __restrict__ mask, int num_row_blocks, int32_t* __restrict__ ok_count)
{
  __shared__ int32_t s_tmp[kBlock / 64];
  __shared__ int32_t s_D[kLatMaxOff];
  __shared__ int s_fail;
  const int t = threadIdx.x;
  constexpr int64_t kAlign = 4; // widest 16-byte chunk (fp32): entries
  constexpr int64_t kSlotEntries = kLatSlotBytes / 8;
  int ok_mine = 0; // thread 0: row blocks this workgroup found in the form (one
                   // atomic per workgroup: half a million on one address took
                   // 4 of the kernel's 7 ms at 512^3)
  for (int rb = blockIdx.x; rb < num_row_blocks; rb += gridDim.x) {
    const int32_t r0 = rb * kRows;
    const int nr = min(kRows, num_rows - r0);
    const int32_t r = r0 + t;
    int32_t lo = 0, hi = 0;
    if (t < nr) {
      lo = rowptr[r];
      hi = rowptr[r + 1];
    }
    const int cnt = hi - lo;
    int32_t d[kLatMaxOff];
#pragma unroll
    for (int j = 0; j < kLatMaxOff; ++j)
      d[j] = (j < cnt && cnt <= kLatMaxOff) ? colind[lo + j] - r : INT32_MAX;
    __syncthreads();
    if (t == 0) {
      // the block's values plus the alignment slack must fit one LDS slot
      const int64_t a = rowptr[r0], b = rowptr[r0 + nr];
      s_fail = (b - (a & ~(kAlign - 1)) > kSlotEntries) ? 1 : 0;
    }
    __syncthreads();
    if (cnt > kLatMaxOff)
      s_fail = 1;
    // the distinct offsets, ascending: repeatedly the smallest one above the
    // last found
    int nd = 0;
    int64_t last = (int64_t)INT32_MIN - 1;
    for (int round = 0; round <= kLatMaxOff; ++round) {
      int32_t cand = INT32_MAX;
#pragma unroll
      for (int j = 0; j < kLatMaxOff; ++j)
        if ((int64_t)d[j] > last && d[j] < cand)
          cand = d[j];
      const int32_t next = block_min(cand, s_tmp);
      if (next == INT32_MAX)
        break; // uniform
      if (round == kLatMaxOff) { // a ninth offset
        if (t == 0)
          s_fail = 1;
        break;
      }
      if (t == 0)
        s_D[nd] = next;
      ++nd;
      last = next;
    }
    __syncthreads();
    // per row: entries must hit D in strictly ascending position
    unsigned m = 0;
    int prev = -1;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < kLatMaxOff; ++j) {
      if (j < cnt && cnt <= kLatMaxOff) {
        int k = 0;
        while (k < nd && s_D[k] != d[j])
          ++k;
        if (k <= prev || k >= nd)
          bad = true;
        prev = k;
        m |= 1u << k;
      }
    }
    if (bad)
      s_fail = 1;
    __syncthreads();
    const int fail = s_fail;
    if (t < nr)
      mask[r] = (uint8_t)m;
    int32_t* rec = tab + (int64_t)rb * kLatRec;
    if (t == 0) {
      rec[0] = fail ? -1 : nd;
      // where the kernel finds the row's own x among its eight x loads: a zero
      // offset, or the zero padding behind the last offset
      int k0 = nd < kLatMaxOff ? nd : -1;
      for (int k = 0; k < nd; ++k)
        if (s_D[k] == 0)
          k0 = k;
      rec[1] = k0;
      // entries in front of the block's second, third and fourth wave of 64
      // rows: with the mask bytes the kernel rebuilds every row's position
      // (row pointer = block start + this + the popcounts of the lower lanes)
      // and never loads the row pointer per row
      const int32_t a = rowptr[r0];
      const int32_t c1 = rowptr[r0 + min(64, nr)] - a;
      const int32_t c2 = rowptr[r0 + min(128, nr)] - a;
      const int32_t c3 = rowptr[r0 + min(192, nr)] - a;
      rec[2] = c1 | (c2 << 16); // <= 2176 each (slot check above)
      rec[3] = c3;
      if (!fail)
        ++ok_mine;
    }
    if (t < kLatMaxOff)
      rec[4 + t] = t < nd ? s_D[t] : 0;
  }
  if (t == 0 && ok_mine)
    atomicAdd(ok_count, ok_mine);
}

template <typename TV, typename T, bool DOT>
int lat_launch(const spmv_hip_csr_plan* pl, hipStream_t st,
               const int32_t* rowptr, const TV* values, T alpha, const T* in,
               T beta, T* out, DotOut dot)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const int grid = spmv_lat_grid(pl);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
    ord.table = pl->zw_table;
    ord.num_slots = pl->zw_slots;
  }
  // plane chain: planes a whole number of row blocks apart
  const int chain_rows = (pl->lat_chain && pl->lattice_d2 > 0
                          && pl->lattice_d2 % kRows == 0)
                             ? pl->lattice_d2
                             : 0;
  auto kern = ord.table
                  ? (pl->nontemporal ? csr_lattice_kernel<TV, T, DOT, true, true>
                                     : csr_lattice_kernel<TV, T, DOT, false, true>)
                  : (pl->nontemporal
                         ? csr_lattice_kernel<TV, T, DOT, true, false>
                         : csr_lattice_kernel<TV, T, DOT, false, false>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, st, pl->num_rows,
                     pl->num_cols, pl->nnz, rowptr, values, pl->lat_tab,
                     pl->lat_mask, alpha, in, beta, out, dot, ord, chain_rows);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

int spmv_lat_grid(const spmv_hip_csr_plan* pl)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * pl->lat_blocks_per_cu;
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

void spmv_lat_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->lat_tab);
  (void)hipFree(pl->lat_mask);
  pl->lat_tab = nullptr;
  pl->lat_mask = nullptr;
  pl->lat = 0;
}

// Build the lattice form; kept only when EVERY row block qualifies (the
// kernel has no per-block fallback).  Costs 1 B per row + 36 B per row block.
int spmv_lat_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                   const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  spmv_lat_free(pl);
  pl->lat_blocks = 0;
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  if (nrb == 0 || pl->nnz == 0)
    return SPMV_HIP_OK;
  hipStream_t st = pl->ctx->stream;
  int32_t* d_ok = nullptr;
  hipError_t e = hipMalloc(&pl->lat_tab, sizeof(int32_t) * (size_t)nrb * kLatRec);
  if (e == hipSuccess)
    e = hipMalloc(&pl->lat_mask, (size_t)pl->num_rows);
  if (e == hipSuccess)
    e = hipMalloc(&d_ok, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipMemsetAsync(d_ok, 0, sizeof(int32_t), st);
  int32_t ok = 0;
  if (e == hipSuccess) {
    int grid = pl->ctx->num_cus * 8;
    grid = grid > nrb ? nrb : grid;
    hipLaunchKernelGGL(lat_build_kernel, dim3(grid), dim3(kBlock), 0, st,
                       pl->num_rows, rowptr, colind, pl->lat_tab, pl->lat_mask,
                       nrb, d_ok);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(&ok, d_ok, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_ok);
  if (e != hipSuccess) {
    spmv_lat_free(pl);
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  pl->lat_blocks = ok;
  if (ok != nrb) {
    spmv_lat_free(pl);
    pl->lat_blocks = ok;
    return SPMV_HIP_OK;
  }
  pl->lat = 1;
  // XCD groups of 16 row blocks while x fits the Infinity Cache (216^3: 0.153
  // -> 0.138 ms); no effect beyond (512^3)
  pl->lat_xcd_group = pl->nontemporal ? 16 : 0;
  // 3-D lattice?  The offsets of a row block in the middle of the matrix: line
  // distance d1 = second smallest positive one, plane distance d2 = largest.
  int32_t rec[kLatRec];
  if (hipMemcpy(rec, pl->lat_tab + (size_t)(nrb / 2) * kLatRec, sizeof(rec),
                hipMemcpyDeviceToHost)
          == hipSuccess
      && rec[0] >= 3) {
    int pos[kLatMaxOff], np = 0;
    for (int k = 0; k < rec[0]; ++k)
      if (rec[4 + k] > 0)
        pos[np++] = rec[4 + k]; // ascending already
    if (np >= 3 && pos[1] >= 8 && pos[np - 1] % pos[1] == 0
        && pos[np - 1] / pos[1] >= 16) {
      pl->lattice_d1 = pos[1];
      pl->lattice_d2 = pos[np - 1];
      // plane-walk order (every workgroup walks a column of the lattice from
      // plane to plane: the x of the plane ahead is the next step's own, see
      // lat_loads); built for lattices that outgrow the grid
      const int rc = spmv_zwalk_order_build(pl, pl->lattice_d2,
                                            spmv_lat_grid(pl), 0, false);
      if (rc != SPMV_HIP_OK)
        return rc;
      // (An earlier order table -- every XCD sweeping a strip of grid lines
      // through all planes -- cut the fabric reads at 512^3 from 12.3 to 9.7 GB
      // and ran 6-12 % slower: higher fabric latency, and a table look-up that
      // was waited for every step; profiles/r02_pmc_lattice_512.json.)
    }
  }
  return SPMV_HIP_OK;
}

int spmv_lat_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const double* values, double alpha,
                     const double* in, double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return lat_launch<double, double, true>(pl, st, rowptr, values, alpha, in,
                                            beta, out, dot);
  return lat_launch<double, double, false>(pl, st, rowptr, values, alpha, in,
                                           beta, out, dot);
}

int spmv_lat_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                        const int32_t* rowptr, const float* values,
                        double alpha, const double* in, double beta,
                        double* out, DotOut dot)
{
  if (dot.partials)
    return lat_launch<float, double, true>(pl, st, rowptr, values, alpha, in,
                                           beta, out, dot);
  return lat_launch<float, double, false>(pl, st, rowptr, values, alpha, in,
                                          beta, out, dot);
}

int spmv_lat_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const float* values, float alpha,
                     const float* in, float beta, float* out)
{
  return lat_launch<float, float, false>(pl, st, rowptr, values, alpha, in, beta,
                                         out, DotOut());
}
