// Plan-time builders of the general CSR plans (gfx950): the LX / XW window
// analysis (lx_build_kernel), the row list of mostly-empty blocks, the
// plane-walk order tables.  Split from spmv_csr.hip in round 6; the kernels
// that read these records are in spmv_csr.hip (register-staged LX kernel) and
// spmv_lxw.hip (LDS-DMA kernel, LX and XW); the plan API that calls the
// builders is spmv_csr_plan.hip.
#include "csr_plan.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

namespace
{

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

struct NonEmptyRow {
  const int32_t* rowptr;
  __device__ bool operator()(int i) const { return rowptr[i + 1] > rowptr[i]; }
};

} // namespace

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

// Plane distance of a matrix on a 3-D grid that stays out of the lattice form
// (its values, boundary rows or a permutation of the entries keep it there):
// the farthest column above and below the diagonal in a row block in the middle
// of the matrix, when the two agree (and are far enough to be planes); else 0.
int64_t spmv_plane_distance(const spmv_hip_csr_plan* pl, const int32_t* rowptr,
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
int spmv_build_row_list(spmv_hip_csr_plan* pl, const int32_t* rowptr)
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

void spmv_free_lx(spmv_hip_csr_plan* pl)
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

void spmv_free_xw(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->xw_rec);
  pl->xw_rec = nullptr;
  pl->xw = pl->xw_staged = pl->xw_max_cnt = pl->xw_max_pieces = 0;
  xw_probe_free(pl);
}

// Most entries any 256-row block holds: the plan kernel sorts a block's
// columns with 8 keys per thread (2048) where that suffices -- half the work of
// the 16-key instantiation, which the 7-point matrix (1792 entries per block)
// was given for its AVERAGE of 6.99 entries per row (lx_build_kernel<16> 33.6 ms
// of the 512^3 plan, profiles/r06_rocprof_bench_n512_kernel_stats.csv).
struct RowBlockEntries {
  const int32_t* rowptr;
  int32_t num_rows;
  __host__ __device__ int32_t operator()(int32_t rb) const
  {
    const int64_t r0 = (int64_t)rb * kRows;
    const int64_t r1 = r0 + kRows < num_rows ? r0 + kRows : num_rows;
    return rowptr[r1] - rowptr[r0];
  }
};

static int max_block_entries(const spmv_hip_csr_plan* pl, const int32_t* rowptr, int nrb,
                             hipStream_t st, int32_t* out)
{
  *out = INT32_MAX; // (on any failure: the large instantiation)
  int32_t* d_max = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  hipcub::CountingInputIterator<int32_t> ids(0);
  hipcub::TransformInputIterator<int32_t, RowBlockEntries,
                                 hipcub::CountingInputIterator<int32_t>>
      cnt(ids, RowBlockEntries{rowptr, pl->num_rows});
  hipError_t e = hipMalloc(&d_max, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Max(nullptr, tb, cnt, d_max, nrb, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Max(tmp, tb, cnt, d_max, nrb, st);
  int32_t h = INT32_MAX;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  (void)hipFree(d_max);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return SPMV_HIP_OK; // (not fatal: *out stays at its default)
  }
  *out = h;
  return SPMV_HIP_OK;
}

// The XW records (spmv_lxw.hip): per row block its span, the DMA pieces of its
// x windows and the windows themselves.  144 B per row block; the CSR arrays
// stay the caller's.  Kept only if most blocks are staged.
int spmv_build_xw(spmv_hip_csr_plan* pl, const int32_t* rowptr, const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  spmv_free_xw(pl);
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
    spmv_free_xw(pl);
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  int end_bit = 1;
  while (end_bit < 31 && ((int64_t)1 << end_bit) < pl->num_cols)
    ++end_bit;
  int grid = pl->ctx->num_cus * 4;
  grid = grid > nrb ? nrb : grid;
  int32_t max_cnt = 0;
  (void)max_block_entries(pl, rowptr, nrb, st, &max_cnt);
  if (max_cnt <= 8 * kBlock)
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
    spmv_free_xw(pl);
    return static_cast<int>(e);
  }
  pl->xw_max_cnt = h_stat[0];
  pl->xw_max_pieces = h_stat[1];
  pl->xw_staged = h_stat[2];
  if ((int64_t)pl->xw_staged * 2 < nrb) { // mostly direct blocks: the gather kernel
    spmv_free_xw(pl);
    return SPMV_HIP_OK;
  }
  pl->xw = 1;
  if (pl->ctx->xw_probe)
    pl->xw_probe = new (std::nothrow) XwProbe;
  return SPMV_HIP_OK;
}

// ... with the plane-walk order when the matrix sits on a 3-D grid
int spmv_build_xw_and_walk(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                      const int32_t* colind)
{
  int rc = spmv_build_xw(pl, rowptr, colind);
  if (rc == SPMV_HIP_OK && pl->xw) {
    const int64_t d2 = spmv_plane_distance(pl, rowptr, colind);
    if (d2 > 0) {
      pl->lattice_d2 = (int)d2;
      rc = spmv_zwalk_order_build(pl, d2, spmv_walk_grid(pl), 0, false);
    }
  }
  return rc;
}

// may this plan stage x windows over the caller's arrays?
bool spmv_xw_applies(const spmv_hip_csr_plan* pl)
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
int spmv_build_lx(spmv_hip_csr_plan* pl, const int32_t* rowptr,
             const int32_t* colind)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  spmv_free_lx(pl);
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
    spmv_free_lx(pl);
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
      spmv_free_lx(pl);
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
  int32_t max_cnt = 0;
  (void)max_block_entries(pl, rowptr, nrb, st, &max_cnt);
  if (max_cnt <= 8 * kBlock)
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
    spmv_free_lx(pl);
    return static_cast<int>(e);
  }
  pl->lx_blocks = nrb;
  pl->lx_staged = staged;
  pl->lxw_max_cnt = h_stat[0];
  pl->lxw_max_pieces = h_stat[1];
  if ((int64_t)staged * 2 < nrb) { // mostly direct blocks: not worth the memory
    spmv_free_lx(pl);
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
    const int64_t d2 = spmv_plane_distance(pl, rowptr, colind);
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
  return (pl->xw && pl->xw_rec) ? spmv_xw_grid(pl, 8) : spmv_rowblock_grid(pl);
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

extern "C" {

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

} // extern "C"
