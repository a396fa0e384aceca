// LX form of the general CSR SpMV, LDS-DMA kernel (gfx950 / MI355X).
//
// Stands behind the same CSRSpMV<T>::init/run hook as the other general
// kernels (spmv/csr_kernels.h:26-78); arithmetic and summation order are those
// of spmv/csr_kernels.cpp:41-51: a row is summed left to right, mul and add
// rounded separately => bit-identical to the oracle.
//
// The LX form (spmv_csr_forms.hip) rewrites every entry's column as a 16-bit offset
// into a per-row-block set of staged x WINDOWS.  The register-staged kernel
// (csr_rowblock_lx_kernel) alternates phases -- fetch windows, wait, stream a
// tile of values + offsets, wait, multiply, park products in LDS, barrier, add
// -- and needs eight workgroups per CU to cover them; PMC passes at 512^3
// showed it at the fabric's byte ceiling (15.5 GB read + 1.07 GB written per
// launch = 6.9 TB/s for 12.1 GB requested): every x element crossed the
// fabric 3.5 times, because the three uses of a window (as a block's own
// columns, and as the far window of the blocks a plane below and above) are
// 4.9 MB of matrix stream per XCD apart -- more than its L2 holds.
//
// This kernel takes the lattice kernel's skeleton (spmv_lat.hip):
//   * lane = row.  A persistent workgroup walks its row blocks with
//     EVERYTHING one block ahead: values (8 B), offsets (2 B) and the x
//     windows of block k+1 arrive by LDS-DMA (global_load_lds_dwordx4: no
//     VGPRs, no LDS store instructions) into the second set of LDS slots while
//     block k is summed; the row pointer pair, y (beta != 0) and the row's own
//     x (fused dot) travel in a second register set.  ONE barrier per block.
//   * the row owner walks its entries in LDS: value, offset, x[offset] --
//     products are never parked.
//   * the y of a block is stored one step LATE, right behind the next
//     step's wait (which covers stores too): issued at the end of its own
//     step the store's whole round trip would be exposed in every step.
//   * a row block the plan could not stage (too many windows, too wide)
//     gathers x from global memory through `colind`, its values still by DMA;
//     one with more entries than a slot holds reads everything from global
//     memory.  The decision is per row block.
//
// LDS per workgroup (dynamic, sized by the plan): 2 x (values slot + offsets
// slot + x buffer); 60 KiB for the 7-point matrix => 2 workgroups per CU.
//
// Measured at 512^3 (7-point matrix, lattice analysis off; one process, the
// kernels alternating): 2.24-2.51 ms against 2.39-2.48 ms for the
// register-staged kernel in the plane-walk order (2.46-2.60 in the plain
// order, round 2's 0.67), i.e. the same within the spread between processes --
// 1.5 % slower at 384^3 and 512^3, equal at 216^3, 6 % faster at 256^3, 14 %
// at 128^3 -- while moving 12.0 GB instead of 13.7 GB (plain order: 15.5 GB)
// across the fabric per launch (profiles/r03_pmc_lx*).  What a step is made
// of (parts switched off one at a time, same process, 2.47 ms whole): without
// the y store 1.96 ms -- before the store moved behind the next wait; 2.24
// with it there --, without the values 1.54, without the offsets 2.15,
// without the x windows 2.40, without the row sums 2.57 (no gain: the
// arithmetic is free), nothing but the skeleton's own loads and barrier 0.85;
// one workgroup per CU instead of two 2.99.  Three things had to go before it
// reached the old kernel at all, each a wait the compiler adds for a register
// that a load may still be writing: the piece list read with v_readlane in
// the middle of the DMA issue (now extracted right after the wait), the
// order-table entry sharing a register with its computed alternative
// (template parameter TAB), and the block record fetched by SCALAR loads,
// whose counter the LDS reads share (now one vector load a step ahead).
// Dropped after measuring: copying the two window pieces a block shares with
// its predecessor in the plane walk inside LDS instead of fetching them again
// (plan-time inheritance table; 11.93 against 12.04 GB of fabric reads -- the
// L2 already served them -- and 1-7 % slower); the values two row blocks ahead
// instead of one (three value slots, 75 KiB per workgroup, still two per CU:
// 2.51 against 2.54 ms at 512^3, 3 % slower at 128^3 and 384^3 -- the bytes a
// workgroup has in flight are not what bounds the kernel either).
//
// XW variant (round 5): the same kernel on the CALLER's CSR arrays as they are.
// The 16-bit offsets are a plan-owned copy of the index stream (2 B per entry
// of plan memory, 10 B per entry streamed); the plain row-block kernel streams
// the caller's arrays untouched but gathers x entry by entry -- at 512^3 x
// crossed the fabric 3.5 times (15.6 GB read for 12.9 GB of matrix + x;
// profiles/r04_pmc_csr_rowblock_spmv_*), and walking the row blocks plane by
// plane does not help it: one step of an XCD's workgroups streams 6.8 MB
// through its 4 MB L2 (2.89 against 2.91 ms, profiles/r05_rowblock_walk.log).
// XW = true streams `colind` (32 bit) by LDS-DMA where the LX form streams its
// offsets, stages the same x windows, and turns a column into its staged
// position with the block's window list (at most eight windows: a select
// chain on uniform registers; more windows: the block gathers).  Plan memory:
// 144 B per row block.
#include "csr_plan.h"
#include "lat_dma.h"

#include <new>
#include <type_traits>

namespace
{

struct LxwBlock {
  int rb;        // row block, -1 = none
  int nwin;      // >= 0 staged, -1 direct
  int32_t a;     // span start in `values`
  int32_t cnt;   // entries
  int np;        // staged pieces
  int own;       // staged position of the block's first row's OWN column
                 // (x[rb * kRows] -- the fused dot's x), -1 = not staged
};

// A block's record travels as ONE vector load (lane l holds word l) issued a
// whole step before it is needed and is taken apart with v_readlane right
// after the step's wait.  Not scalar loads: those share their counter with
// the LDS reads, so the first LDS read of the row sums would wait for a
// record coming from HBM -- 1-2 us per step.
template <int REC>
__device__ __forceinline__ int32_t lxw_fetch(int rb,
                                             const int32_t* __restrict__ rec)
{
  int32_t w = 0;
  if (rb >= 0) {
    const int l = (int)(threadIdx.x & 63);
    w = rec[(int64_t)rb * REC + (l < REC ? l : REC - 1)];
  }
  return w;
}

__device__ __forceinline__ LxwBlock lxw_decode(int rb, int32_t w)
{
  LxwBlock b{-1, -1, 0, 0, 0, -1};
  if (rb >= 0) {
    b.rb = rb;
    b.nwin = __builtin_amdgcn_readlane(w, 0);
    b.a = __builtin_amdgcn_readlane(w, 1);
    b.cnt = __builtin_amdgcn_readlane(w, 2);
    const int32_t w3 = __builtin_amdgcn_readlane(w, 3);
    b.np = w3 & kLxwNpMask;
    b.own = (w3 >> kLxwOwnShift) - 1;
  }
  return b;
}

template <typename T>
struct LxwRegs {
  int32_t lo, hi; // the row's span
  T y0;
};

// (The fused dot's x_i is NOT among them: the block's own columns sit in one
// of its staged windows wherever the rows have a diagonal neighbourhood -- the
// plan records where, LxwBlock::own -- and are read from LDS in the block's own
// step.  As a second global load a step ahead it cost the kernel 15 % at 512^3:
// 2.53 against 2.19 ms, profiles/r05_rocprof_bench_n512_kernel_stats.csv.)
template <typename T>
__device__ __forceinline__ LxwRegs<T> lxw_loads(const LxwBlock& blk, int t,
                                                int32_t num_rows,
                                                const int32_t* __restrict__ rowptr,
                                                const T* __restrict__ in, T beta,
                                                const T* __restrict__ out)
{
  LxwRegs<T> g;
  g.lo = g.hi = 0;
  g.y0 = T(0);
  if (blk.rb >= 0) {
    const int32_t r = blk.rb * kRows + t;
    if (r < num_rows) {
      g.lo = rowptr[r];
      g.hi = rowptr[r + 1];
      if (beta != T(0))
        g.y0 = out[r];
    }
  }
  return g;
}

// TV = type of `values`, T = type of x, y and the arithmetic.
//   vcap   entries per values slot (slot bytes a multiple of 1 KiB)
//   lcap   entries per offsets slot
//   xcap   elements per x buffer (pieces * kLxwPiece)
//   TAB    the row-block order comes from a table (plane walk).  A template
//          parameter because the two sources of a slot's row block must not
//          share a register: a table entry is a LOADED value, and the compiler
//          waits for every load in flight -- the DMA pieces included -- before
//          it lets the computed alternative overwrite that register.
//   XW     the index stream is the caller's `colind` (lidx unused), the
//          record is the XW one (kXwRec ints: windows behind the pieces)
template <typename TV, typename T, bool DOT, bool NT, bool TAB, bool XW>
__global__ __launch_bounds__(kBlock) void csr_lxw_kernel(
    int32_t num_rows, int32_t num_cols, int64_t nnz,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const TV* __restrict__ values, const uint16_t* __restrict__ lidx,
    const int32_t* __restrict__ rec, T alpha, const T* __restrict__ in, T beta,
    T* __restrict__ out, DotOut dot, RowBlockOrder ord, int vcap, int lcap,
    int xcap)
{
  using IDX = typename std::conditional<XW, int32_t, uint16_t>::type;
  constexpr int REC = XW ? kXwRec : kLxwRec;
  constexpr int IA = 16 / (int)sizeof(IDX) - 1; // index chunk alignment mask
  constexpr int V = 16 / (int)sizeof(TV); // values per 16-byte chunk
  constexpr int E = 16 / (int)sizeof(T);  // x elements per 16-byte chunk
  // lanes of one DMA instruction that cover a piece of kLxwPiece elements
  constexpr int PL = kLxwPiece / E;       // 64 (fp64), 32 (fp32)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TV* const s_val = reinterpret_cast<TV*>(smem);
  IDX* const s_lidx
      = reinterpret_cast<IDX*>(smem + 2 * (size_t)vcap * sizeof(TV));
  T* const s_x = reinterpret_cast<T*>(smem + 2 * (size_t)vcap * sizeof(TV)
                                      + 2 * (size_t)lcap * sizeof(IDX));
  __shared__ double s_red[kBlock / 64];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;
  // last aligned chunk of x that lies inside the vector
  const int32_t col_last = (num_cols - E) & ~(E - 1);

  // does the block's data sit in its LDS slots?
  auto fits = [&](const LxwBlock& b) {
    return b.cnt + (b.a & (V - 1)) <= vcap && b.cnt + (b.a & IA) <= lcap;
  };
  // A wave's share of a block's piece list: pieces wave, wave + 4, ... as
  // UNIFORM values, extracted from the per-lane list right after a wait -- a
  // v_readlane of a loaded register in the middle of the DMA issue would make
  // the compiler wait for everything in flight, the DMA pieces included.
  constexpr int PW = kLxwMaxPieces / (kBlock / 64); // pieces per wave
  struct Pieces {
    int32_t col[PW]; // first column of piece wave + 4 k
  };
  auto pieces_of = [&](int32_t w) { // w: the record, one word per lane
    Pieces pc;
#pragma unroll
    for (int k = 0; k < PW; ++k)
      pc.col[k] = __builtin_amdgcn_readlane(
          w, kLxwPieces0 + wave + (kBlock / 64) * k);
    return pc;
  };
  // DMA of one block's streams and windows into slot `sl`
  auto issue = [&](const LxwBlock& b, const Pieces& pc, int sl) {
    if (b.rb < 0 || b.cnt <= 0 || !fits(b))
      return;
    const int64_t a = b.a, e = (int64_t)b.a + b.cnt;
    lat_issue_dma<TV, NT>(values, nnz, a & ~(int64_t)(V - 1), e,
                          s_val + (size_t)sl * vcap, t);
    if (b.nwin < 0)
      return;
    if constexpr (XW)
      lat_issue_dma<int32_t, NT>(colind, nnz, a & ~(int64_t)IA, e,
                                 s_lidx + (size_t)sl * lcap, t);
    else
      lat_issue_dma<uint16_t, NT>(lidx, nnz + 8, a & ~(int64_t)IA, e,
                                  s_lidx + (size_t)sl * lcap, t);
    const unsigned lds0 = (unsigned)(uintptr_t)(
        (__attribute__((address_space(3))) void*)(s_x + (size_t)sl * xcap));
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int p = wave + (kBlock / 64) * k;
      if (p >= b.np)
        continue; // uniform
      if (lane < PL) {
        int32_t c = pc.col[k] + lane * E;
        c = c < col_last ? c : col_last; // padding past the end of x
        glds16<false>(in + c,
                      lds0 + (unsigned)(p * kLxwPiece * (int)sizeof(T)));
      }
    }
  };

  // slot -> row block (raw: before the bounds check of order_slot_decode)
  auto slot_raw = [&](int i) { return order_slot_raw_t<TAB>(ord, i, num_slots); };
  int it = blockIdx.x;
  LxwBlock cur;
  int nxt_rb;
  int32_t nxt_w; // the next block's record, in flight
  int32_t cur_w0;
  {
    const int rb0 = order_slot_decode(ord, slot_raw(it));
    nxt_rb = order_slot_decode(ord, slot_raw(it + stride));
    const int32_t w0 = lxw_fetch<REC>(rb0, rec);
    nxt_w = lxw_fetch<REC>(nxt_rb, rec);
    cur = lxw_decode(rb0, w0);
    cur_w0 = w0;
    issue(cur, pieces_of(w0), 0);
  }
  int nn_raw = slot_raw(it + 2 * stride);
  LxwRegs<T> gA = lxw_loads<T>(cur, t, num_rows, rowptr, in, beta, out);
  LxwRegs<T> gB;
  int slot = 0;
  // The y of a block is STORED A STEP LATER, right after the next step's wait:
  // the wait at the top of a step covers everything the wave has in flight,
  // stores included, and a store issued at the end of a step would be waited
  // for at once -- its whole round trip exposed, every step (measured: 0.5 ms
  // of 2.47 at 512^3).
  T y_late = T(0);
  int32_t r_late = -1;

  // XW: the current block's windows as uniform values (taken from its record
  // right after a wait, like the piece list)
  struct Windows {
    int32_t f[kXwMaxWin]; // f[k]: first column of window k (f[0] unused)
    int32_t d[kXwMaxWin];
  };
  auto windows_of = [&](int32_t w) {
    Windows ws;
#pragma unroll
    for (int k = 0; k < kXwMaxWin; ++k) {
      ws.f[k] = 0;
      ws.d[k] = 0;
      if constexpr (XW) {
        if (k > 0)
          ws.f[k] = __builtin_amdgcn_readlane(w, kXwFirst0 + k - 1);
        ws.d[k] = __builtin_amdgcn_readlane(w, kXwDelta0 + k);
      }
    }
    return ws;
  };
  Windows cur_ws = windows_of(cur_w0);

  auto step = [&](const LxwRegs<T>& g, LxwRegs<T>& gn) {
    // everything this wave has in flight has landed; after the barrier that
    // holds for all waves, and all of them have left the previous block (the
    // builtin, not inline assembly: see spmv_lat.hip)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    if (r_late >= 0)
      out[r_late] = y_late;
    r_late = -1;
    const LxwBlock nxt = lxw_decode(nxt_rb, nxt_w);
    const Windows nxt_ws = windows_of(nxt_w);
    issue(nxt, pieces_of(nxt_w), slot ^ 1);
    gn = lxw_loads<T>(nxt, t, num_rows, rowptr, in, beta, out);
    // the block after the next one: its table entry has landed with the wait
    // above; its record is needed a step from now
    const int nn_rb = order_slot_decode(ord, nn_raw);
    const int32_t nn_w = lxw_fetch<REC>(nn_rb, rec);
    const int nnn_raw = slot_raw(it + 3 * stride);

    const int32_t r = cur.rb * kRows + t;
    if (cur.rb >= 0 && r < num_rows) {
      T sum = 0;
      T x_own = T(0);
      bool own_staged = false;
      int32_t j = g.lo;
      const int32_t hi = g.hi;
      if (cur.cnt > 0 && fits(cur)) {
        // index by the entry's position in `values`
        const TV* sv = s_val + (size_t)slot * vcap + (cur.a & (V - 1)) - cur.a;
        if (cur.nwin >= 0) {
          const IDX* sl = s_lidx + (size_t)slot * lcap + (cur.a & IA) - cur.a;
          const T* sx = s_x + (size_t)slot * xcap;
          if constexpr (DOT) {
            if (cur.own >= 0) { // uniform
              x_own = sx[cur.own + t];
              own_staged = true;
            }
          }
          // eight entries' LDS reads in flight (offsets, then values and x:
          // two round trips per eight entries), adds strictly left to right;
          // entries past the row's end read a valid slot and are not added
          for (; j < hi; j += 8) {
            unsigned l[8];
            T v[8], xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int32_t jj = j + k < hi ? j + k : hi - 1;
              l[k] = sl[jj];
              v[k] = (T)sv[jj];
            }
            if constexpr (XW) {
              // column -> staged position: + the delta of the last window
              // that starts at or below it (windows ascend)
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int32_t c = (int32_t)l[k];
                int32_t d = cur_ws.d[0];
#pragma unroll
                for (int q = 1; q < kXwMaxWin; ++q)
                  d = c >= cur_ws.f[q] ? cur_ws.d[q] : d;
                l[k] = (unsigned)(c + d);
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
              xv[k] = sx[l[k]];
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (j + k < hi)
                sum += v[k] * xv[k];
          }
        } else { // direct: the global gather
          for (; j < hi; ++j)
            sum += (T)sv[j] * in[colind[j]];
        }
      } else { // more entries than a slot holds: the reference loop
        for (; j < hi; ++j)
          sum += (T)values[j] * in[colind[j]];
      }
      const T c = alpha * sum;
      T y = c;
      if (beta != T(0))
        y = c + beta * g.y0;
      y_late = y;
      r_late = r;
      if constexpr (DOT) {
        if (!own_staged) // a block without a staged window over its own rows
          x_own = in[r];
        dot_acc += (double)x_own * (double)c;
      }
    }
    slot ^= 1;
    cur = nxt;
    cur_ws = nxt_ws;
    nxt_rb = nn_rb;
    nxt_w = nn_w;
    nn_raw = nnn_raw;
    it += stride;
  };
  // two steps per trip: the register sets swap roles without copies
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

struct LxwGeom {
  int vcap, lcap, xcap;
  size_t lds;
  int per_cu;
};

// idx_bytes: 2 (the LX form's offsets) or 4 (XW: the caller's column indices)
LxwGeom lxw_geom(const spmv_hip_csr_plan* pl, int elem_bytes, int idx_bytes)
{
  LxwGeom g;
  const bool xw = idx_bytes == 4;
  const int V = 16 / elem_bytes;
  const int max_cnt = xw ? pl->xw_max_cnt : pl->lxw_max_cnt;
  // slot for the largest staged block plus the slack of the 16-byte alignment
  // of its first chunk, in whole 1-KiB DMA pieces; at most 32 KiB
  int64_t vb = ((int64_t)(max_cnt + V) * elem_bytes + 1023) & ~1023ll;
  vb = vb < 1024 ? 1024 : (vb > 32768 ? 32768 : vb);
  g.vcap = (int)(vb / elem_bytes);
  int64_t lb = ((int64_t)(g.vcap + 8) * idx_bytes + 1023) & ~1023ll;
  g.lcap = (int)(lb / idx_bytes);
  int np = xw ? pl->xw_max_pieces : pl->lxw_max_pieces;
  np = np < 1 ? 1 : np;
  g.xcap = np * kLxwPiece;
  g.lds = 2 * (size_t)vb + 2 * (size_t)lb + 2 * (size_t)g.xcap * elem_bytes;
  int per = (int)((160 * 1024 - 1024) / (g.lds + 64));
  per = per < 1 ? 1 : (per > kBlocksPerCU ? kBlocksPerCU : per);
  if (pl->lxw_blocks_per_cu > 0 && pl->lxw_blocks_per_cu < per)
    per = pl->lxw_blocks_per_cu;
  g.per_cu = per;
  return g;
}

int lxw_grid(const spmv_hip_csr_plan* pl, int elem_bytes, int idx_bytes)
{
  const LxwGeom g = lxw_geom(pl, elem_bytes, idx_bytes);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  int grid = pl->ctx->num_cus * g.per_cu;
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

template <typename TV, typename T, bool DOT, bool XW>
int lxw_launch(const spmv_hip_csr_plan* pl, hipStream_t st,
               const int32_t* rowptr, const int32_t* colind, const TV* values,
               T alpha, const T* in, T beta, T* out, DotOut dot)
{
  const LxwGeom g = lxw_geom(pl, (int)sizeof(T), XW ? 4 : 2);
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const int grid = lxw_grid(pl, (int)sizeof(T), XW ? 4 : 2);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
    ord.table = pl->zw_table;
    ord.num_slots = pl->zw_slots;
  }
  auto kern
      = ord.table
            ? (pl->nontemporal ? csr_lxw_kernel<TV, T, DOT, true, true, XW>
                               : csr_lxw_kernel<TV, T, DOT, false, true, XW>)
            : (pl->nontemporal ? csr_lxw_kernel<TV, T, DOT, true, false, XW>
                               : csr_lxw_kernel<TV, T, DOT, false, false, XW>);
  if (g.lds > 48 * 1024) // beyond the default dynamic-LDS limit
    SPMV_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kern),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), g.lds, st, pl->num_rows,
                     pl->num_cols, pl->nnz, rowptr, colind, values,
                     XW ? nullptr : pl->lx_lidx, XW ? pl->xw_rec : pl->lxw_rec,
                     alpha, in, beta, out, dot, ord, g.vcap, g.lcap, g.xcap);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

int spmv_lxw_grid(const spmv_hip_csr_plan* pl, int elem_bytes)
{
  return lxw_grid(pl, elem_bytes, 2);
}

int spmv_xw_grid(const spmv_hip_csr_plan* pl, int elem_bytes)
{
  return lxw_grid(pl, elem_bytes, 4);
}

int spmv_lxw_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const int32_t* colind,
                     const double* values, double alpha, const double* in,
                     double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return lxw_launch<double, double, true, false>(pl, st, rowptr, colind, values,
                                                   alpha, in, beta, out, dot);
  return lxw_launch<double, double, false, false>(pl, st, rowptr, colind, values,
                                                  alpha, in, beta, out, dot);
}

int spmv_lxw_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                     const int32_t* rowptr, const int32_t* colind,
                     const float* values, float alpha, const float* in,
                     float beta, float* out)
{
  return lxw_launch<float, float, false, false>(pl, st, rowptr, colind, values,
                                                alpha, in, beta, out, DotOut());
}

int spmv_xw_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                    const int32_t* rowptr, const int32_t* colind,
                    const double* values, double alpha, const double* in,
                    double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return lxw_launch<double, double, true, true>(pl, st, rowptr, colind, values,
                                                  alpha, in, beta, out, dot);
  return lxw_launch<double, double, false, true>(pl, st, rowptr, colind, values,
                                                 alpha, in, beta, out, dot);
}

int spmv_xw_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                    const int32_t* rowptr, const int32_t* colind,
                    const float* values, float alpha, const float* in,
                    float beta, float* out)
{
  return lxw_launch<float, float, false, true>(pl, st, rowptr, colind, values,
                                               alpha, in, beta, out, DotOut());
}
