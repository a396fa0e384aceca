// Symmetric lattice form for gfx950 (MI355X): the symmetric-storage SpMV
// (strictly-lower CSR + diagonal, spmv/csr_kernels.cpp:26-40) for matrices
// whose lower entries sit at a handful of constant column offsets -- atomic
// free, every y written once, bit-identical to the reference's sequential
// finalise-then-scatter order, and without an index stream.
//
// The reference, seen from row i (see csr_symt_kernel in spmv_sym.hip):
//   y_i  = fl(alpha * sum_i + beta * y0_i),  sum_i = d_i x_i, then the row's
//          own entries left to right
//   y_i += fl(fl(alpha * v(r, i)) * x_r)     for the stored entries (r, i) of
//          its COLUMN, ascending r
// With lower offsets D[0] < ... < D[nd-1] < 0 (nd <= 3) the entries of column i
// can only sit in the rows r = i - D[k]; whether row r has one is bit k of
// ITS mask byte, and its position in `values` is rowptr[r] + (number of lower
// mask bits of r).  So a workgroup that owns rows [r0, r0 + 256) needs, besides
// its own values, the values of the rows [r0 - D[k], r0 - D[k] + 256): nd more
// contiguous spans of `values`.  All of them arrive by LDS-DMA one row block
// ahead (lat_dma.h), exactly like the general lattice kernel's single span;
// offsets below 256 rows share the block's own span, extended by that many
// rows.  Per lane and row block: row pointer, mask and x of the row itself and
// of its nd column rows, the diagonal -- coalesced loads into registers.
//
// HBM bytes per row of the 7-point matrix: 24 (values) + 8 (diagonal) + 4 + 1
// (row pointer, mask) + 8 (x) + 8 (y) = 53 against 77 for the general lattice
// form and 104 for plain CSR; the column spans are re-reads that the L2s and
// the Infinity Cache serve -- in CSR order each of them drags a row's three
// values along for the one that is used, which is why the DIAGONAL form
// (spmv_symdia.hip: the plan's copy of the values by offset, 1.35 ms at 512^3
// against 2.36 ms here) replaces this kernel whenever the values are baked.
// This one runs on the caller's own arrays: the C ABI without
// plan_bake_values, or a launch with other value pointers.
//
// Plan (slat_build): the global offset set (one for the whole matrix, so that
// a mask bit means the same thing in every row), one mask byte per row, and
// the largest span any window takes (its LDS sub-slot).
#include "csr_plan.h"
#include "lat_dma.h"

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

namespace
{

constexpr int kSlatMaxOff = 3; // lower offsets = mask bits
constexpr int kSlatMaxWin = 1 + kSlatMaxOff;

struct SlatGeom {
  int nd;                  // lower offsets in use
  int D[kSlatMaxOff];      // ascending, all < 0 (unused: 0)
  int ext;                 // rows the own window is extended by (offsets < 256)
  int nw;                  // windows: own + one per offset >= 256 rows
  int win_of_k[kSlatMaxOff]; // LDS sub-slot holding the column entries of
                           // offset k (0 = the own window: offset < 256 rows)
  int u_of_w[kSlatMaxWin]; // first row of window w relative to r0 (w = 0: 0)
  int sub_bytes;           // LDS bytes per window (whole DMA pieces)
};

// Window spans in `values` as raw row-pointer entries (converting them at the
// prefetch would make it wait).  Named scalars, not arrays: an array member
// indexed inside the block-search loop ends up in scratch memory.
// a0/b0: the own window; a1..a3 / b1..b3: the window of offset k = 0..2 when
// that offset has one of its own (win_of_k[k] != 0), else unused.
struct SlatBlock {
  int rb; // row block, -1 = none
  int32_t a0, a1, a2, a3;
  int32_t b0, b1, b2, b3;
};

// rows of window w (uniform; 0 = own window, extended by g.ext rows)
__device__ __forceinline__ void slat_window_rows(const SlatGeom& g, int w,
                                                 int64_t r0, int64_t nr,
                                                 int32_t num_rows,
                                                 int64_t* first, int64_t* last)
{
  // u_of_w[w] without indexing the kernel argument dynamically
  int64_t u = 0;
  u = (w == 1) ? g.u_of_w[1] : u;
  u = (w == 2) ? g.u_of_w[2] : u;
  u = (w == 3) ? g.u_of_w[3] : u;
  *first = min((int64_t)num_rows, r0 + u);
  *last = min((int64_t)num_rows, r0 + u + nr + (w == 0 ? g.ext : 0));
}

template <typename T>
struct SlatRegs {
  int32_t lo;            // rowptr[i]
  unsigned m;            // mask[i]
  T d, xi, y0;           // diagonal, x_i, y0_i
  T xl[kSlatMaxOff];     // x[i + D[k]]
  int32_t lor[kSlatMaxOff]; // rowptr[i - D[k]]
  unsigned mr[kSlatMaxOff]; // mask[i - D[k]] (0 past the last row)
  T xu[kSlatMaxOff];     // x[i - D[k]]
};

__device__ __forceinline__ SlatBlock slat_block(
    const RowBlockOrder& ord, const SlatGeom& g, int it, int num_slots,
    int32_t num_rows, const int32_t* __restrict__ rowptr, int stride,
    int* it_out)
{
  SlatBlock blk{-1, 0, 0, 0, 0, 0, 0, 0, 0};
  while (it < num_slots) {
    const int rb = order_row_block(ord, it);
    if (rb >= 0) {
      const int64_t r0 = (int64_t)rb * kRows;
      const int64_t nr = min((int64_t)kRows, (int64_t)num_rows - r0);
      int64_t f, l;
      blk.rb = rb;
      slat_window_rows(g, 0, r0, nr, num_rows, &f, &l);
      blk.a0 = rowptr[f];
      blk.b0 = rowptr[l];
      if (g.win_of_k[0]) { // uniform
        slat_window_rows(g, g.win_of_k[0], r0, nr, num_rows, &f, &l);
        blk.a1 = rowptr[f];
        blk.b1 = rowptr[l];
      }
      if (g.win_of_k[1]) {
        slat_window_rows(g, g.win_of_k[1], r0, nr, num_rows, &f, &l);
        blk.a2 = rowptr[f];
        blk.b2 = rowptr[l];
      }
      if (g.win_of_k[2]) {
        slat_window_rows(g, g.win_of_k[2], r0, nr, num_rows, &f, &l);
        blk.a3 = rowptr[f];
        blk.b3 = rowptr[l];
      }
      break;
    }
    it += stride;
  }
  *it_out = it;
  return blk;
}

template <typename T>
__device__ __forceinline__ SlatRegs<T> slat_loads(
    const SlatBlock& blk, const SlatGeom& g, int t, int32_t num_rows,
    const int32_t* __restrict__ rowptr, const uint8_t* __restrict__ mask,
    const T* __restrict__ diagonal, const T* __restrict__ in, T beta,
    const T* __restrict__ out)
{
  SlatRegs<T> q;
  q.lo = 0;
  q.m = 0;
  q.d = q.xi = q.y0 = T(0);
#pragma unroll
  for (int k = 0; k < kSlatMaxOff; ++k) {
    q.xl[k] = q.xu[k] = T(0);
    q.lor[k] = 0;
    q.mr[k] = 0;
  }
  if (blk.rb < 0)
    return q;
  const int32_t i = blk.rb * kRows + t;
  if (i < num_rows) {
    q.lo = rowptr[i];
    q.m = mask[i];
    q.d = diagonal[i];
    q.xi = in[i];
    if (beta != T(0))
      q.y0 = out[i];
#pragma unroll
    for (int k = 0; k < kSlatMaxOff; ++k) {
      if (k < g.nd) { // uniform
        // unconditional, clamped into range: no dependence on the mask loads;
        // what a row does not have is ignored later
        const int64_t c = (int64_t)i + g.D[k];
        q.xl[k] = in[c < 0 ? 0 : c];
        const int64_t r = (int64_t)i - g.D[k];
        const int64_t rr = r < num_rows ? r : (int64_t)num_rows - 1;
        q.lor[k] = rowptr[rr];
        q.mr[k] = r < num_rows ? (unsigned)mask[rr] : 0u;
        q.xu[k] = in[rr];
      }
    }
  }
  return q;
}

// ---------------------------------------------------------------------------
// The kernel: same skeleton as csr_lattice_kernel (two LDS slots, one barrier
// per row block, DMA and register loads one block ahead).
// ---------------------------------------------------------------------------
template <typename T, bool DOT, bool NT>
__global__ __launch_bounds__(kBlock) void csr_sym_lattice_kernel(
    int32_t num_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
    const T* __restrict__ values, const T* __restrict__ diagonal,
    const uint8_t* __restrict__ mask, T alpha, const T* __restrict__ in, T beta,
    T* __restrict__ out, DotOut dot, RowBlockOrder ord, SlatGeom g)
{
  constexpr int V = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ double s_red[kBlock / 64];
  T* const s_val = reinterpret_cast<T*>(s_dyn);
  const int sub = g.sub_bytes / (int)sizeof(T); // entries per window
  const int slot_entries = sub * g.nw;

  const int t = threadIdx.x;
  const int stride = gridDim.x;
  const int num_slots = order_slots(ord);
  double dot_acc = 0.0;

  auto issue1 = [&](int32_t a, int32_t b, int slot, int w) {
    if (b > a)
      lat_issue_dma<T, NT>(values, nnz, (int64_t)a & ~(int64_t)(V - 1),
                           (int64_t)b, s_val + slot * slot_entries + w * sub, t);
  };
  auto issue = [&](const SlatBlock& blk, int slot) {
    issue1(blk.a0, blk.b0, slot, 0);
    if (g.win_of_k[0])
      issue1(blk.a1, blk.b1, slot, g.win_of_k[0]);
    if (g.win_of_k[1])
      issue1(blk.a2, blk.b2, slot, g.win_of_k[1]);
    if (g.win_of_k[2])
      issue1(blk.a3, blk.b3, slot, g.win_of_k[2]);
  };

  int it = blockIdx.x, itn = 0, itnn = 0;
  SlatBlock cur = slat_block(ord, g, it, num_slots, num_rows, rowptr, stride, &it);
  SlatBlock nxt = slat_block(ord, g, it + stride, num_slots, num_rows, rowptr,
                             stride, &itn);
  if (cur.rb >= 0)
    issue(cur, 0);
  SlatRegs<T> qA = slat_loads<T>(cur, g, t, num_rows, rowptr, mask, diagonal, in,
                                 beta, out);
  SlatRegs<T> qB;
  int slot = 0;
  // y is stored one step late, right behind the next step's wait (which
  // covers stores too): see csr_lxw_kernel
  T y_late = T(0);
  int32_t i_late = -1;
  auto step = [&](const SlatRegs<T>& q, SlatRegs<T>& qn) {
    // everything of block k has landed; all waves have left block k-1 (see
    // csr_lattice_kernel for why this is the builtin)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    if (i_late >= 0)
      out[i_late] = y_late;
    i_late = -1;
    if (nxt.rb >= 0)
      issue(nxt, slot ^ 1);
    qn = slat_loads<T>(nxt, g, t, num_rows, rowptr, mask, diagonal, in, beta,
                       out);
    // the block after the next one: dependent scalar loads behind the vector
    // loads, needed an iteration from now
    const SlatBlock nn = slat_block(ord, g, itn + stride, num_slots, num_rows,
                                    rowptr, stride, &itnn);
    const int32_t i = cur.rb * kRows + t;
    if (i < num_rows) {
      const T* sv = s_val + slot * slot_entries;
      const int top = slot_entries - 1;
      // the row's own entries: all LDS reads first, then the adds in order
      const int rel = q.lo - (int32_t)((int64_t)cur.a0 & ~(int64_t)(V - 1));
      T vl[kSlatMaxOff], vu[kSlatMaxOff];
#pragma unroll
      for (int k = 0; k < kSlatMaxOff; ++k) {
        vl[k] = vu[k] = T(0);
        if (k < g.nd) { // uniform
          const unsigned below = (1u << k) - 1u;
          vl[k] = sv[min(max(rel + (int)__popc(q.m & below), 0), top)];
          // entry (r, i), r = i - D[k]: in row r it is the one of offset k
          const int w = g.win_of_k[k]; // 0: the row sits in the own window
          const int32_t ak = k == 0 ? cur.a1 : (k == 1 ? cur.a2 : cur.a3);
          const int32_t aw = w ? ak : cur.a0;
          const int relr = q.lor[k] - (int32_t)((int64_t)aw & ~(int64_t)(V - 1));
          vu[k] = sv[min(max(w * sub + relr + (int)__popc(q.mr[k] & below), 0),
                         top)];
        }
      }
      T sum = q.d * q.xi; // csr_kernels.cpp:28
#pragma unroll
      for (int k = 0; k < kSlatMaxOff; ++k)
        if (k < g.nd && ((q.m >> k) & 1u)) // :34, left to right
          sum += vl[k] * q.xl[k];
      const T c = alpha * sum; // :39
      T y = c, cy = c;
      if (beta != T(0))
        y = c + beta * q.y0;
      // the column's entries in ascending row order: largest offset first
#pragma unroll
      for (int k = kSlatMaxOff - 1; k >= 0; --k)
        if (k < g.nd && ((q.mr[k] >> k) & 1u)) { // :35
          const T term = (alpha * vu[k]) * q.xu[k];
          y += term;
          cy += term;
        }
      y_late = y;
      i_late = i;
      if constexpr (DOT) // in . (alpha A in): the finished row without beta y0
        dot_acc += (double)q.xi * (double)cy;
    }
    slot ^= 1;
    cur = nxt;
    nxt = nn;
    itn = itnn;
  };
  while (cur.rb >= 0) {
    step(qA, qB);
    if (cur.rb < 0)
      break;
    step(qB, qA);
  }
  if (i_late >= 0)
    out[i_late] = y_late;
  if constexpr (DOT)
    spmv_dot_epilogue(dot, dot_acc, s_red);
}

// ---------------------------------------------------------------------------
// Plan-time analysis
// ---------------------------------------------------------------------------
// pass 1: the set of distinct offsets col - row (capacity 8, INT32_MAX = free)
__global__ __launch_bounds__(kBlock) void slat_offsets_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int32_t* __restrict__ set,
    int32_t* __restrict__ fail)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    // a matrix that is no lattice says so in its first rows: nobody goes on
    // after that (read first -- on the 10 M-row FEM-like matrix every row of
    // this kernel raised the flag with an atomic of its own, and every entry
    // tried the eight slots: 33 of the symmetric plan's 53 ms,
    // profiles/r06_plan_fem_sym_kernel_stats.csv)
    if (*(volatile int32_t*)fail)
      return;
    const int32_t lo = rowptr[i], hi = rowptr[i + 1];
    if (hi - lo > kSlatMaxOff) {
      atomicOr(fail, 1);
      return;
    }
    for (int32_t j = lo; j < hi; ++j) {
      const int32_t d = colind[j] - (int32_t)i;
      if (d >= 0) { // not strictly lower
        atomicOr(fail, 1);
        return;
      }
      bool placed = false;
      for (int s = 0; s < 8 && !placed; ++s) {
        int32_t cur = set[s];
        if (cur == INT32_MAX)
          cur = atomicCAS(set + s, INT32_MAX, d);
        placed = (cur == d || cur == INT32_MAX);
      }
      if (!placed) {
        atomicOr(fail, 1);
        return;
      }
    }
  }
}

// pass 2: per-row mask against the global offsets (entries must hit them in
// strictly ascending position), and the largest span of any window
__global__ __launch_bounds__(kBlock) void slat_mask_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, SlatGeom g, uint8_t* __restrict__ mask,
    int32_t* __restrict__ fail, int32_t* __restrict__ max_bytes)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < num_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t lo = rowptr[i], hi = rowptr[i + 1];
    unsigned m = 0;
    int prev = -1;
    for (int32_t j = lo; j < hi; ++j) {
      const int32_t d = colind[j] - (int32_t)i;
      int k = 0;
      while (k < g.nd && g.D[k] != d)
        ++k;
      if ((k >= g.nd || k <= prev) && !*(volatile int32_t*)fail)
        atomicOr(fail, 1);
      prev = k;
      m |= 1u << k;
    }
    mask[i] = (uint8_t)m;
    if ((i % kRows) == 0) { // one lane per row block: its window spans
      const int64_t nr = min((int64_t)kRows, (int64_t)num_rows - i);
      for (int w = 0; w < g.nw; ++w) {
        const int64_t first = min((int64_t)num_rows, i + g.u_of_w[w]);
        const int64_t last = min((int64_t)num_rows,
                                 i + g.u_of_w[w] + nr + (w == 0 ? g.ext : 0));
        // widest 16-byte alignment slack (fp32: 3 entries), fp64 bytes
        const int64_t bytes
            = ((int64_t)rowptr[last] - ((int64_t)rowptr[first] & ~(int64_t)3)) * 8;
        // (read first: nearly every row block has the same spans, and two
        // million atomics on one address took 17 of the plan's 30 ms at 512^3)
        const int32_t span = (int32_t)(((bytes + 1023) >> 10) << 10);
        if (span > *(volatile int32_t*)max_bytes)
          atomicMax(max_bytes, span);
      }
    }
  }
}

int slat_grid(const spmv_hip_csr_plan* pl)
{
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const size_t lds = (size_t)2 * pl->slat_nw * pl->slat_sub_bytes;
  int per_cu = (int)((160 * 1024) / (lds + 64));
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

template <typename T>
int slat_launch(const spmv_hip_csr_plan* pl, hipStream_t st,
                const int32_t* rowptr, const T* values, const T* diagonal,
                T alpha, const T* in, T beta, T* out, DotOut dot)
{
  SlatGeom g;
  g.nd = pl->slat_nd;
  g.ext = pl->slat_ext;
  g.nw = pl->slat_nw;
  g.sub_bytes = pl->slat_sub_bytes;
  for (int k = 0; k < kSlatMaxOff; ++k) {
    g.D[k] = pl->slat_D[k];
    g.win_of_k[k] = pl->slat_win_of_k[k];
  }
  for (int w = 0; w < kSlatMaxWin; ++w)
    g.u_of_w[w] = pl->slat_u_of_w[w];
  const int nrb = (pl->num_rows + kRows - 1) / kRows;
  const size_t lds = (size_t)2 * g.nw * g.sub_bytes;
  const int grid = slat_grid(pl);
  RowBlockOrder ord = pl->row_block_order(nrb);
  ord.xcd_group = pl->lat_xcd_group;
  if (pl->zwalk && pl->zw_table && pl->zw_grid == grid) {
    ord.table = pl->zw_table;
    ord.num_slots = pl->zw_slots;
  }
#define SPMV_SLAT(DOTV, NTV)                                                   \
  hipLaunchKernelGGL((csr_sym_lattice_kernel<T, DOTV, NTV>), dim3(grid),       \
                     dim3(kBlock), lds, st, pl->num_rows, pl->nnz, rowptr,     \
                     values, diagonal, pl->slat_mask, alpha, in, beta, out,    \
                     dot, ord, g)
  if (dot.partials) {
    if (pl->nontemporal)
      SPMV_SLAT(true, true);
    else
      SPMV_SLAT(true, false);
  } else {
    if (pl->nontemporal)
      SPMV_SLAT(false, true);
    else
      SPMV_SLAT(false, false);
  }
#undef SPMV_SLAT
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

int spmv_slat_grid(const spmv_hip_csr_plan* pl)
{
  return slat_grid(pl);
}

void spmv_slat_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->slat_mask);
  pl->slat_mask = nullptr;
  pl->slat = 0;
}

// Try the symmetric lattice form: at most three distinct lower offsets in the
// whole matrix, rows in ascending column order, windows that fit the LDS.
int spmv_slat_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  spmv_slat_free(pl);
  const int32_t n = pl->num_rows;
  if (n == 0 || pl->nnz == 0)
    return SPMV_HIP_OK;
  hipStream_t st = pl->ctx->stream;
  int32_t* d_w = nullptr; // [0..7] offset set, [8] fail, [9] max bytes
  int32_t h_w[10];
  for (int s = 0; s < 8; ++s)
    h_w[s] = INT32_MAX;
  h_w[8] = h_w[9] = 0;
  hipError_t e = hipMalloc(&d_w, sizeof(h_w));
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_w, h_w, sizeof(h_w), hipMemcpyHostToDevice, st);
  const int grid = spmv_grid_for(pl->ctx, n, kBlock);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(slat_offsets_kernel, dim3(grid), dim3(kBlock), 0, st, n,
                       rowptr, colind, d_w, d_w + 8);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(h_w, d_w, sizeof(h_w), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  int nd = 0;
  int D[8];
  if (e == hipSuccess && !h_w[8]) {
    for (int s = 0; s < 8; ++s)
      if (h_w[s] != INT32_MAX)
        D[nd++] = h_w[s];
    for (int a = 1; a < nd; ++a) // insertion sort, ascending
      for (int b = a; b > 0 && D[b] < D[b - 1]; --b) {
        const int tmp = D[b];
        D[b] = D[b - 1];
        D[b - 1] = tmp;
      }
  }
  if (e != hipSuccess || h_w[8] || nd == 0 || nd > kSlatMaxOff) {
    (void)hipFree(d_w);
    return (e == hipSuccess || e == hipErrorOutOfMemory) ? SPMV_HIP_OK
                                                         : static_cast<int>(e);
  }
  // windows: offsets closer than a row block share the own window
  SlatGeom g;
  g.nd = nd;
  g.ext = 0;
  g.nw = 1;
  g.sub_bytes = 0;
  for (int w = 0; w < kSlatMaxWin; ++w)
    g.u_of_w[w] = 0;
  for (int k = 0; k < kSlatMaxOff; ++k) {
    g.D[k] = k < nd ? D[k] : 0;
    g.win_of_k[k] = 0;
    if (k < nd) {
      const int64_t u = -(int64_t)D[k];
      if (u < kRows) {
        if (u > g.ext)
          g.ext = (int)u;
      } else {
        g.win_of_k[k] = g.nw;
        g.u_of_w[g.nw] = (int)u;
        ++g.nw;
      }
    }
  }
  e = hipMalloc(&pl->slat_mask, (size_t)n);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(slat_mask_kernel, dim3(grid), dim3(kBlock), 0, st, n,
                       rowptr, colind, g, pl->slat_mask, d_w + 8, d_w + 9);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(h_w, d_w, sizeof(h_w), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_w);
  // two slots of nw windows must leave room for at least one workgroup
  const int64_t lds = (int64_t)2 * g.nw * h_w[9];
  if (e != hipSuccess || h_w[8] || h_w[9] <= 0 || lds > 150 * 1024) {
    spmv_slat_free(pl);
    return (e == hipSuccess || e == hipErrorOutOfMemory) ? SPMV_HIP_OK
                                                         : static_cast<int>(e);
  }
  pl->slat_nd = nd;
  pl->slat_ext = g.ext;
  pl->slat_nw = g.nw;
  pl->slat_sub_bytes = h_w[9];
  for (int k = 0; k < kSlatMaxOff; ++k) {
    pl->slat_D[k] = g.D[k];
    pl->slat_win_of_k[k] = g.win_of_k[k];
  }
  for (int w = 0; w < kSlatMaxWin; ++w)
    pl->slat_u_of_w[w] = g.u_of_w[w];
  pl->slat = 1;
  pl->lat_xcd_group = pl->nontemporal ? 16 : 0;
  // 3-D lattice: line distance = middle offset, plane distance = farthest
  if (nd == 3 && -(int64_t)D[1] >= 8 && D[0] % D[1] == 0 && D[0] / D[1] >= 16) {
    pl->lattice_d1 = -D[1];
    pl->lattice_d2 = -D[0];
  }
  // plane-walk order: planes = the farthest offset apart (large lattices only)
  return spmv_zwalk_order_build(pl, -(int64_t)D[0], slat_grid(pl), 0, false);
}

int spmv_slat_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                      const int32_t* rowptr, const double* values,
                      const double* diagonal, double alpha, const double* in,
                      double beta, double* out, DotOut dot)
{
  return slat_launch<double>(pl, st, rowptr, values, diagonal, alpha, in, beta,
                             out, dot);
}

int spmv_slat_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                      const int32_t* rowptr, const float* values,
                      const float* diagonal, float alpha, const float* in,
                      float beta, float* out)
{
  return slat_launch<float>(pl, st, rowptr, values, diagonal, alpha, in, beta,
                            out, DotOut());
}
