// Sliced jagged form, PLAN TIME (see spmv_sjds.hip for the form itself):
// which chunks of x a block stages, the sigma sort, the jagged codes and the
// plan's copy of the values, the long rows' list and the table of the
// table-driven kernel, the merged matrix of symmetric storage.
#include "sjds.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <new>

#include "plan_malloc.h" // (last: hipMalloc / hipFree below are timed)

namespace
{

// ---------------------------------------------------------------------------
// plan time
// ---------------------------------------------------------------------------
struct SjSel {
  int32_t lo; // first chunk of the bitmap's span
  int32_t wa, wb; // selected bitmap words (inclusive)
  int32_t K;  // selected chunks
};

// Which chunks of x does the block of rows [r0, r1) touch?  Bitmap over the
// 2^16 chunks around the block's diagonal position; when more than kcap are
// set, the words nearest to the diagonal are kept.  All 256 threads call it;
// s_bits[kSjSpanWords], s_pre[kSjSpanWords + 1].
__device__ SjSel sj_select(int32_t r0, int32_t r1, int32_t num_cols,
                           int32_t num_rows_all, const int32_t* __restrict__ rowptr,
                           const int32_t* __restrict__ colind, int kcap,
                           int long_thr, int64_t nnz, uint32_t* s_bits,
                           int32_t* s_pre, SjSel* s_sel)
{
  using Scan = hipcub::BlockScan<int32_t, kBlock>;
  __shared__ typename Scan::TempStorage s_scan;
  constexpr int kSpan = kSjSpanWords * 32;
  const int t = threadIdx.x;
  const int32_t nchunks = (num_cols + kSjChunk - 1) / kSjChunk;
  int32_t cc = (int32_t)(((int64_t)r0 + r1) / 2 / kSjChunk);
  int32_t lo = cc - kSpan / 2;
  if (lo > nchunks - kSpan)
    lo = nchunks - kSpan;
  if (lo < 0)
    lo = 0;
  for (int w = t; w < kSjSpanWords; w += kBlock)
    s_bits[w] = 0u;
  __syncthreads();
  // one lane per row; LONG rows are not part of the slices (phase 0 of the
  // kernel takes them), so they do not choose chunks.  Where the rows average
  // 32 entries and more a WAVE takes a row, its lanes consecutive entries:
  // one lane per row reads such rows 324 B apart (81 per row) and every line
  // comes in again and again -- 23 of the 56 ms of the 10 M x 81 plan were this
  // loop and its twin in sj_count_kernel (profiles/r06_plan_fem81_kernel_stats.csv)
  if (nnz >= (int64_t)32 * num_rows_all) { // uniform
    for (int32_t row = r0 + (t >> 6); row < r1; row += kBlock / 64) {
      const int32_t a = rowptr[row], b = rowptr[row + 1];
      if (sj_is_long(a, b, long_thr, nnz))
        continue;
      for (int32_t e = a + (t & 63); e < b; e += 64) {
        const int32_t rel = colind[e] / kSjChunk - lo;
        if (rel >= 0 && rel < kSpan) {
          const uint32_t bit = 1u << (rel & 31);
          if (!(s_bits[rel >> 5] & bit)) // (neighbours set the same bits)
            atomicOr(&s_bits[rel >> 5], bit);
        }
      }
    }
  } else {
    for (int32_t row = r0 + t; row < r1; row += kBlock) {
      const int32_t a = rowptr[row], b = rowptr[row + 1];
      if (sj_is_long(a, b, long_thr, nnz))
        continue;
      for (int32_t e = a; e < b; ++e) {
        const int32_t rel = colind[e] / kSjChunk - lo;
        if (rel >= 0 && rel < kSpan)
          atomicOr(&s_bits[rel >> 5], 1u << (rel & 31));
      }
    }
  }
  __syncthreads();
  // exclusive prefix of the words' popcounts (8 consecutive words per thread)
  constexpr int kPer = kSjSpanWords / kBlock;
  int32_t mine = 0;
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    mine += __popc(s_bits[t * kPer + q]);
  int32_t before = 0, total = 0;
  Scan(s_scan).ExclusiveSum(mine, before, total);
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    s_pre[t * kPer + q] = before;
    before += __popc(s_bits[t * kPer + q]);
  }
  if (t == 0)
    s_pre[kSjSpanWords] = total;
  __syncthreads();
  if (t == 0) {
    SjSel s;
    s.lo = lo;
    s.wa = 0;
    s.wb = kSjSpanWords - 1;
    s.K = total;
    if (total > kcap) {
      int32_t cw = (cc - lo) >> 5;
      cw = cw < 0 ? 0 : (cw > kSjSpanWords - 1 ? kSjSpanWords - 1 : cw);
      int rlo = 0, rhi = kSjSpanWords - 1; // largest radius that fits
      while (rlo < rhi) {
        const int mid = (rlo + rhi + 1) >> 1;
        const int wa = cw - mid < 0 ? 0 : cw - mid;
        const int wb = cw + mid > kSjSpanWords - 1 ? kSjSpanWords - 1 : cw + mid;
        if (s_pre[wb + 1] - s_pre[wa] <= kcap)
          rlo = mid;
        else
          rhi = mid - 1;
      }
      s.wa = cw - rlo < 0 ? 0 : cw - rlo;
      s.wb = cw + rlo > kSjSpanWords - 1 ? kSjSpanWords - 1 : cw + rlo;
      s.K = s_pre[s.wb + 1] - s_pre[s.wa];
      if (s.K > kcap) { // one word alone holds at most 32 <= kcap chunks
        s.wb = s.wa - 1;
        s.K = 0;
      }
    }
    *s_sel = s;
  }
  __syncthreads();
  return *s_sel;
}

// staged index of column `col`, or -1 = far
__device__ __forceinline__ int32_t sj_index(const SjSel& s, const uint32_t* s_bits,
                                           const int32_t* s_pre, int32_t col)
{
  const int32_t rel = col / kSjChunk - s.lo;
  const int32_t w = rel >> 5;
  if (rel < 0 || w < s.wa || w > s.wb)
    return -1;
  if (!((s_bits[w] >> (rel & 31)) & 1u))
    return -1;
  const int32_t rank
      = s_pre[w] - s_pre[s.wa] + __popc(s_bits[w] & ((1u << (rel & 31)) - 1u));
  return rank * kSjChunk + (col & (kSjChunk - 1));
}

// pass 1: per block the number of chunks kept and of far entries
template <int R>
__global__ __launch_bounds__(kBlock) void sj_count_kernel(
    int32_t num_rows, int32_t num_cols, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int kcap, int long_thr, int64_t nnz,
    int32_t* __restrict__ blk_k, int32_t* __restrict__ blk_far)
{
  __shared__ uint32_t s_bits[kSjSpanWords];
  __shared__ int32_t s_pre[kSjSpanWords + 1];
  __shared__ SjSel s_sel;
  __shared__ int32_t s_far;
  const int nblk = (num_rows + R - 1) / R;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int32_t r0 = b * R;
    const int32_t r1 = min(r0 + R, num_rows);
    if (threadIdx.x == 0)
      s_far = 0;
    const SjSel sel = sj_select(r0, r1, num_cols, num_rows, rowptr, colind, kcap,
                                long_thr, nnz, s_bits, s_pre, &s_sel);
    int32_t far = 0;
    if (nnz >= (int64_t)32 * num_rows) { // (a wave per row: see sj_select)
      for (int32_t row = r0 + (threadIdx.x >> 6); row < r1; row += kBlock / 64) {
        const int32_t a = rowptr[row], e1 = rowptr[row + 1];
        if (sj_is_long(a, e1, long_thr, nnz))
          continue;
        for (int32_t e = a + (threadIdx.x & 63); e < e1; e += 64)
          far += sj_index(sel, s_bits, s_pre, colind[e]) < 0 ? 1 : 0;
      }
    } else {
      for (int32_t row = r0 + threadIdx.x; row < r1; row += kBlock) {
        const int32_t a = rowptr[row], e1 = rowptr[row + 1];
        if (sj_is_long(a, e1, long_thr, nnz))
          continue;
        for (int32_t e = a; e < e1; ++e)
          far += sj_index(sel, s_bits, s_pre, colind[e]) < 0 ? 1 : 0;
      }
    }
    if (far)
      atomicAdd(&s_far, far);
    __syncthreads();
    if (threadIdx.x == 0) {
      blk_k[b] = sel.K;
      blk_far[b] = s_far;
    }
    __syncthreads();
  }
}

// max and sums of the two per-block arrays: out = {max K, sum K, sum far,
// blocks with far entries}
__global__ __launch_bounds__(1024) void sj_stats_kernel(
    int nblk, const int32_t* __restrict__ blk_k, const int32_t* __restrict__ blk_far,
    int64_t* __restrict__ out)
{
  using Red = hipcub::BlockReduce<int64_t, 1024>;
  __shared__ typename Red::TempStorage tmp;
  int64_t mx = 0, sk = 0, sf = 0, nf = 0;
  for (int b = threadIdx.x; b < nblk; b += 1024) {
    const int64_t k = blk_k[b], f = blk_far[b];
    mx = k > mx ? k : mx;
    sk += k;
    sf += f;
    nf += f > 0 ? 1 : 0;
  }
  mx = Red(tmp).Reduce(mx, hipcub::Max());
  __syncthreads();
  sk = Red(tmp).Sum(sk);
  __syncthreads();
  sf = Red(tmp).Sum(sf);
  __syncthreads();
  nf = Red(tmp).Sum(nf);
  if (threadIdx.x == 0) {
    out[0] = mx;
    out[1] = sk;
    out[2] = sf;
    out[3] = nf;
  }
}

// units (E entries each) every slice of 64 rows needs: its short rows, each
// padded to whole units
__global__ __launch_bounds__(kBlock) void sj_units_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr, int long_thr, int64_t nnz,
    int E, uint32_t* __restrict__ units)
{
  const int lane = threadIdx.x & 63;
  const int64_t nsl = ((int64_t)num_rows + 63) / 64;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t s = wid; s <= nsl; s += nw) { // (entry nsl: 0, the scan's total)
    const int64_t row = s * 64 + lane;
    int32_t len = 0;
    if (s < nsl && row < num_rows) {
      const int32_t a = rowptr[row], b = rowptr[row + 1];
      len = sj_is_long(a, b, long_thr, nnz) ? 0 : b - a;
    }
    uint32_t u = (uint32_t)((len + E - 1) / E);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      u += __shfl_down(u, o, 64);
    if (lane == 0)
      units[s] = u;
  }
}

// SIGMA layout (blocks of 1024 rows = 16 slices): the rows are sorted by length
// across the whole BLOCK, so that a slice holds rows of (nearly) one length --
// a slice runs as many steps as its longest row, and with rows of 5 ... 40
// entries side by side that is 2.5 times the average.  The kernel then gives a
// wave TWO slices, the k-th longest and the k-th shortest.  Per block (one
// workgroup): the (length, row in block) word of every sorted position -- the
// length from bit 10 up -- and the units of its 16 slices.
__global__ __launch_bounds__(kBlock) void sj_sigma_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr, int long_thr, int64_t nnz,
    int E, int32_t* __restrict__ lenperm, uint32_t* __restrict__ units)
{
  __shared__ int32_t s_len[kSjSigRows];
  __shared__ uint32_t s_units[kSjSigRows / 64];
  const int nblk = (num_rows + kSjSigRows - 1) / kSjSigRows;
  for (int b = blockIdx.x; b <= nblk; b += gridDim.x) {
    if (b == nblk) { // the scan's total
      if (threadIdx.x == 0)
        units[(int64_t)nblk * (kSjSigRows / 64)] = 0;
      continue;
    }
    const int32_t r0 = b * kSjSigRows;
    for (int i = threadIdx.x; i < kSjSigRows; i += kBlock) {
      int32_t v = 0; // bit 30: a LONG row (not in the slices: length 0, marked)
      if (r0 + i < num_rows) {
        const int32_t ra = rowptr[r0 + i], rb = rowptr[r0 + i + 1];
        v = sj_is_long(ra, rb, long_thr, nnz) ? (1 << 30) : rb - ra;
      }
      s_len[i] = v;
    }
    if (threadIdx.x < kSjSigRows / 64)
      s_units[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < kSjSigRows; i += kBlock) {
      const int32_t vi = s_len[i];
      const int32_t li = vi & ~(1 << 30);
      int rank = 0; // longer rows first, ties: the lower row first
      for (int j = 0; j < kSjSigRows; ++j) {
        const int32_t lj = s_len[j] & ~(1 << 30);
        rank += (lj > li || (lj == li && j < i)) ? 1 : 0;
      }
      lenperm[(int64_t)r0 + rank]
          = (int32_t)(((uint32_t)li << kSjSigBits) | (uint32_t)i
                      | ((vi >> 30) & 1 ? kSjLongFlag : 0u));
      atomicAdd(&s_units[rank / 64], (uint32_t)((li + E - 1) / E));
    }
    __syncthreads();
    if (threadIdx.x < kSjSigRows / 64)
      units[(int64_t)b * (kSjSigRows / 64) + threadIdx.x] = s_units[threadIdx.x];
    __syncthreads();
  }
}

// the slice's rows in jagged order: lane rho gets the row (0..63 within the
// slice) with the rho-th largest length (ties: the lower row first)
__device__ __forceinline__ void sj_sort_slice(int32_t len, int lane, int32_t* my_len,
                                              int* my_row)
{
  int rank = 0;
  for (int j = 0; j < 64; ++j) {
    const int32_t lj = __shfl(len, j, 64);
    rank += (lj > len || (lj == len && j < lane)) ? 1 : 0;
  }
  int32_t ml = 0;
  int mr = 0;
  for (int j = 0; j < 64; ++j) {
    const int rj = __shfl(rank, j, 64);
    const int32_t lj = __shfl(len, j, 64);
    if (rj == lane) {
      ml = lj;
      mr = j;
    }
  }
  *my_len = ml;
  *my_row = mr;
}

// pass 2: chunk lists, the (length, row) word of every jagged lane, and the
// column codes in jagged order (ubase: first unit of every slice)
template <int R>
__global__ __launch_bounds__(kBlock) void sj_fill_kernel(
    int32_t num_rows, int32_t num_cols, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, int kcap, int long_thr, int64_t nnz, int E,
    int stride, int wide_alloc, const int32_t* __restrict__ blk_far,
    const uint32_t* __restrict__ ubase, int32_t* __restrict__ blk,
    int32_t* __restrict__ chunks, int32_t* __restrict__ lenperm,
    unsigned char* __restrict__ codes, int sigma)
{
  __shared__ uint32_t s_bits[kSjSpanWords];
  __shared__ int32_t s_pre[kSjSpanWords + 1];
  __shared__ SjSel s_sel;
  const int nblk = (num_rows + R - 1) / R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int32_t r0 = b * R;
    const int32_t r1 = min(r0 + R, num_rows);
    const SjSel sel = sj_select(r0, r1, num_cols, num_rows, rowptr, colind, kcap,
                                long_thr, nnz, s_bits, s_pre, &s_sel);
    for (int w = sel.wa + threadIdx.x; w <= sel.wb; w += kBlock) {
      uint32_t bits = s_bits[w];
      int32_t rank = s_pre[w] - s_pre[sel.wa];
      while (bits) {
        const int bit = __ffs(bits) - 1;
        chunks[(int64_t)b * stride + rank] = sel.lo + w * 32 + bit;
        ++rank;
        bits &= bits - 1;
      }
    }
    // the list is padded with its last chunk up to the stride: the kernel
    // loads list entries before it knows the block's count
    __syncthreads();
    for (int c = sel.K + threadIdx.x; c < stride; c += kBlock)
      chunks[(int64_t)b * stride + c]
          = sel.K > 0 ? chunks[(int64_t)b * stride + sel.K - 1] : 0;
    const int wide = blk_far[b] > 0 ? 1 : 0;
    if (threadIdx.x == 0) {
      blk[2 * b] = sel.K;
      blk[2 * b + 1] = wide;
    }
    // the block's codes start at entry E * ubase[first slice]: 16-bit codes, or
    // 32-bit ones when the block has far entries (a plan with any wide block
    // reserves 4 bytes per entry everywhere)
    const int64_t a_b = (int64_t)ubase[r0 / 64] * E;
    uint16_t* c16 = reinterpret_cast<uint16_t*>(codes + (wide_alloc ? 4 : 2) * a_b);
    uint32_t* c32 = reinterpret_cast<uint32_t*>(codes + 4 * a_b);
    for (int sl = wave; sl < R / 64; sl += kBlock / 64) {
      const int32_t s0 = r0 + sl * 64;
      if (s0 >= num_rows && !sigma)
        break;
      int32_t mylen, src0;
      if (sigma) { // (sj_sigma_kernel sorted the block and wrote the words)
        const uint32_t w = (uint32_t)lenperm[s0 + lane] & ~kSjLongFlag;
        mylen = (int32_t)(w >> kSjSigBits);
        const int32_t grow = r0 + (int32_t)(w & (kSjSigRows - 1));
        src0 = grow < num_rows ? rowptr[grow] : 0;
      } else {
        const int32_t row = s0 + lane;
        int32_t len = 0;
        bool is_long = false; // not in the slice: length 0, marked
        if (row < num_rows) {
          const int32_t ra = rowptr[row], rb = rowptr[row + 1];
          is_long = sj_is_long(ra, rb, long_thr, nnz);
          len = is_long ? 0 : rb - ra;
        }
        int myrow;
        sj_sort_slice(len, lane, &mylen, &myrow);
        const bool my_long = __shfl((int)is_long, myrow, 64) != 0;
        lenperm[s0 + lane]
            = (int32_t)(((uint32_t)mylen << 6) | (uint32_t)myrow
                        | (my_long ? kSjLongFlag : 0u));
        src0 = s0 + myrow < num_rows ? rowptr[s0 + myrow] : 0;
      }
      const int32_t myu = (mylen + E - 1) / E;
      const int32_t maxu = __shfl(myu, 0, 64);
      int64_t off = (int64_t)ubase[s0 / 64] * E - a_b; // entries, in the block
      for (int32_t k = 0; k < maxu; ++k) {
        const bool act = k < myu;
        const int cnt = __popcll(__ballot(act));
        if (act) {
          for (int q = 0; q < E; ++q) {
            int32_t idx = 0; // a unit's padding: a valid code, never used
            int32_t col = 0;
            if (k * E + q < mylen) {
              col = colind[src0 + k * E + q];
              idx = sj_index(sel, s_bits, s_pre, col);
            }
            const int64_t at = off + (int64_t)lane * E + q;
            if (wide)
              c32[at] = idx >= 0 ? (uint32_t)idx : (0x80000000u | (uint32_t)col);
            else
              c16[at] = (uint16_t)idx;
          }
        }
        off += (int64_t)cnt * E;
      }
    }
    __syncthreads(); // the bitmap is reused by the next block
  }
}

// the plan's copy of the values in jagged order (one wave per slice)
template <typename T>
__global__ __launch_bounds__(kBlock) void sj_bake_kernel(
    int32_t num_rows, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ lenperm, const uint32_t* __restrict__ ubase, int E,
    const T* __restrict__ values, const int32_t* __restrict__ map,
    T* __restrict__ sval, int rbits)
{
  // map (symmetric storage, the transposed block): entry e of the plan's CSR
  // arrays is values[map[e]]
  const int lane = threadIdx.x & 63;
  const int64_t nsl = rbits == 6 ? ((int64_t)num_rows + 63) / 64
                                 : (((int64_t)num_rows + (1 << rbits) - 1) >> rbits)
                                       << (rbits - 6);
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t s = wid; s < nsl; s += nw) {
    const int32_t s0 = (int32_t)(s * 64);
    // (rbits = 6: the row inside its slice; 10: inside its block of 1024 rows)
    const int32_t lp = lenperm[s0 + lane];
    const int32_t mylen = (int32_t)(((uint32_t)lp & ~kSjLongFlag) >> rbits);
    const int32_t myrow = (s0 & ~((1 << rbits) - 1)) + (lp & ((1 << rbits) - 1));
    const int64_t src0 = myrow < num_rows ? rowptr[myrow] : 0;
    const int32_t myu = (mylen + E - 1) / E;
    const int32_t maxu = __shfl(myu, 0, 64);
    int64_t off = (int64_t)ubase[s] * E;
    for (int32_t k = 0; k < maxu; ++k) {
      const bool act = k < myu;
      const int cnt = __popcll(__ballot(act));
      if (act)
        for (int q = 0; q < E; ++q)
          sval[off + (int64_t)lane * E + q]
              = k * E + q < mylen
                    ? values[map ? (int64_t)map[src0 + k * E + q] : src0 + k * E + q]
                    : T(0);
      off += (int64_t)cnt * E;
    }
  }
}

__global__ __launch_bounds__(kBlock) void sj_long_key_kernel(
    int count, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rows,
    uint64_t* __restrict__ key)
{
  // runs of 2^kSjLongRunShift consecutive long rows, inside a run the longest first
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += gridDim.x * blockDim.x) {
    const uint32_t len = (uint32_t)(rowptr[rows[i] + 1] - rowptr[rows[i]]);
    key[i] = ((uint64_t)(i >> kSjLongRunShift) << 32)
             | (uint64_t)(0xFFFFFFFFu - len);
  }
}

struct SjStats {
  int64_t maxk = 0, sumk = 0, far = 0, far_blocks = 0;
};

template <int R>
int sj_count(spmv_hip_csr_plan* pl, const int32_t* rowptr, const int32_t* colind,
             int kcap, int long_thr, int32_t* d_k, int32_t* d_far,
             int64_t* d_stats, SjStats* st, hipStream_t stream)
{
  const int nblk = (pl->num_rows + R - 1) / R;
  const int grid = spmv_grid_for(pl->ctx, nblk, 1);
  hipLaunchKernelGGL((sj_count_kernel<R>), dim3(grid), dim3(kBlock), 0, stream,
                     pl->num_rows, pl->num_cols, rowptr, colind, kcap, long_thr, pl->nnz,
                     d_k, d_far);
  SPMV_CHECK_LAUNCH();
  hipLaunchKernelGGL(sj_stats_kernel, dim3(1), dim3(1024), 0, stream, nblk, d_k,
                     d_far, d_stats);
  SPMV_CHECK_LAUNCH();
  int64_t h[4] = {0, 0, 0, 0};
  SPMV_CHECK_HIP(hipMemcpyAsync(h, d_stats, sizeof(h), hipMemcpyDeviceToHost, stream));
  SPMV_CHECK_HIP(hipStreamSynchronize(stream));
  st->maxk = h[0];
  st->sumk = h[1];
  st->far = h[2];
  st->far_blocks = h[3];
  return SPMV_HIP_OK;
}

// do the columns of every listed row ascend strictly?  (*bad raised if not)
__global__ __launch_bounds__(kBlock) void sj_long_sorted_kernel(
    int count, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const int32_t* __restrict__ rows, int32_t* __restrict__ bad)
{
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t i = wid; i < count; i += nw) {
    const int32_t a = rowptr[rows[i]], b = rowptr[rows[i] + 1];
    bool ok = true;
    for (int32_t e = a + lane; e + 1 < b; e += 64)
      ok = ok && colind[e] < colind[e + 1];
    if (!ok)
      *bad = 1; // (any value: no atomic needed)
  }
}

struct SjIsLong {
  const int32_t* rowptr;
  int thr;
  int64_t nnz;
  __device__ bool operator()(int i) const
  {
    return sj_is_long(rowptr[i], rowptr[i + 1], thr, nnz);
  }
};
struct SjLongCount {
  const int32_t* rowptr;
  int thr;
  int64_t nnz;
  __device__ int operator()(int i) const
  {
    return sj_is_long(rowptr[i], rowptr[i + 1], thr, nnz) ? 1 : 0;
  }
};

} // namespace

// the rows longer than thr, ascending, sorted by length inside runs of 64
// (SPMV_HIP_ENOMEM: no memory)
int spmv_sj_build_long_list(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                       const int32_t* colind, int thr, hipStream_t st)
{
  const int n = pl->num_rows;
  hipcub::CountingInputIterator<int32_t> first(0);
  hipcub::TransformInputIterator<int, SjLongCount,
                                 hipcub::CountingInputIterator<int32_t>>
      ones(first, SjLongCount{rowptr, thr, pl->nnz});
  int32_t* d_count = nullptr;
  void* tmp = nullptr;
  size_t tb = 0, tb2 = 0;
  int32_t count = 0;
  hipError_t e = hipMalloc(&d_count, sizeof(int32_t));
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(nullptr, tb, ones, d_count, n, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(tmp, tb, ones, d_count, n, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&count, d_count, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  tmp = nullptr;
  if (e == hipSuccess && count > 0) {
    e = hipMalloc(&pl->sj_long_rows, sizeof(int32_t) * (size_t)count);
    SjIsLong pred{rowptr, thr, pl->nnz};
    if (e == hipSuccess)
      e = hipcub::DeviceSelect::If(nullptr, tb2, first, pl->sj_long_rows, d_count, n,
                                   pred, st);
    if (e == hipSuccess)
      e = hipMalloc(&tmp, tb2 ? tb2 : 16);
    if (e == hipSuccess)
      e = hipcub::DeviceSelect::If(tmp, tb2, first, pl->sj_long_rows, d_count, n,
                                   pred, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    // ... inside runs of 64 the longest first, so that the eight rows a wave
    // takes together end together and still are neighbours
    uint64_t *d_key = nullptr, *d_key2 = nullptr;
    int32_t* d_rows2 = nullptr;
    void* tmp2 = nullptr;
    size_t tb3 = 0;
    if (e == hipSuccess)
      e = hipMalloc(&d_key, sizeof(uint64_t) * (size_t)count);
    if (e == hipSuccess)
      e = hipMalloc(&d_key2, sizeof(uint64_t) * (size_t)count);
    if (e == hipSuccess)
      e = hipMalloc(&d_rows2, sizeof(int32_t) * (size_t)count);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sj_long_key_kernel,
                         dim3(spmv_grid_for(pl->ctx, count, kBlock)), dim3(kBlock), 0,
                         st, count, rowptr, pl->sj_long_rows, d_key);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb3, d_key, d_key2,
                                             pl->sj_long_rows, d_rows2, count, 0, 64,
                                             st);
    if (e == hipSuccess)
      e = hipMalloc(&tmp2, tb3 ? tb3 : 16);
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(tmp2, tb3, d_key, d_key2,
                                             pl->sj_long_rows, d_rows2, count, 0, 64,
                                             st);
    if (e == hipSuccess)
      e = hipMemcpyAsync(pl->sj_long_rows, d_rows2, sizeof(int32_t) * (size_t)count,
                         hipMemcpyDeviceToDevice, st);
    // ascending columns in every long row?  (d_count is free to be the flag)
    int32_t h_bad = 1;
    if (e == hipSuccess)
      e = hipMemsetAsync(d_count, 0, sizeof(int32_t), st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sj_long_sorted_kernel,
                         dim3(spmv_grid_for(pl->ctx, count, kBlock / 64)),
                         dim3(kBlock), 0, st, count, rowptr, colind,
                         pl->sj_long_rows, d_count);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(&h_bad, d_count, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    pl->sj_long_sorted = h_bad ? 0 : 1;
    (void)hipFree(d_key);
    (void)hipFree(d_key2);
    (void)hipFree(d_rows2);
    (void)hipFree(tmp2);
  }
  (void)hipFree(tmp);
  (void)hipFree(d_count);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_ENOMEM : static_cast<int>(e);
  }
  pl->sj_nlong = count;
  return SPMV_HIP_OK;
}

namespace
{

// --- the table of the table-driven long-row kernel --------------------------
// per supergroup (run of kSjLtRun long rows): the columns it spans -> its first
// column (a multiple of 16), its number of panels (0: more than
// kSjLtMaxPanels, the rows are not neighbours in x) and its table entries
__global__ __launch_bounds__(kBlock) void sj_lt_span_kernel(
    int nlong, int nsg, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const int32_t* __restrict__ rows,
    int32_t* __restrict__ cmin_out, int32_t* __restrict__ np_out,
    int64_t* __restrict__ cnt_out)
{
  const int lane = threadIdx.x & 63;
  const int wid = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int nw = (gridDim.x * kBlock) >> 6;
  for (int sg = wid; sg < nsg; sg += nw) {
    int32_t mn = INT32_MAX, mx = -1;
    for (int s = lane; s < kSjLtRun; s += 64) {
      const int li = sg * kSjLtRun + s;
      if (li < nlong) {
        const int32_t a = rowptr[rows[li]], b = rowptr[rows[li] + 1];
        if (b > a) {
          mn = min(mn, colind[a]);
          mx = max(mx, colind[b - 1]);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = min(mn, __shfl_xor(mn, o, 64));
      mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) {
      int32_t cmin = 0, np = 0;
      if (mx >= 0) {
        cmin = mn & ~(kSjChunk - 1);
        const int64_t n = ((int64_t)mx - cmin) / kSjLtPanel + 1;
        np = n <= kSjLtMaxPanels ? (int32_t)n : 0;
      }
      cmin_out[sg] = cmin;
      np_out[sg] = np;
      cnt_out[sg] = np ? (int64_t)(np + 1) * kSjLtRun : 0;
    }
    if (sg == 0 && lane == 0)
      cnt_out[nsg] = 0;
  }
}

// per row (slot of its supergroup) and panel boundary p = 0 ... np: the row's
// first entry whose column is >= cmin + p * panel (p = 0: the row's first
// entry; p = np: its end); slots past the list: empty ranges
__global__ __launch_bounds__(kBlock) void sj_lt_fill_kernel(
    int nlong, int nsg, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colind, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ cmin_in, const int32_t* __restrict__ np_in,
    const int64_t* __restrict__ off, int32_t* __restrict__ tab)
{
  for (int sg = blockIdx.x; sg < nsg; sg += gridDim.x) {
    const int np = np_in[sg];
    if (np == 0)
      continue;
    const int32_t cmin = cmin_in[sg];
    int32_t* out = tab + off[sg];
    for (int idx = threadIdx.x; idx < (np + 1) * kSjLtRun; idx += kBlock) {
      const int p = idx / kSjLtRun, s = idx % kSjLtRun;
      const int li = sg * kSjLtRun + s;
      int32_t res = 0;
      if (li < nlong) {
        const int32_t a = rowptr[rows[li]], b = rowptr[rows[li] + 1];
        if (p == 0) {
          res = a;
        } else if (p == np) {
          res = b;
        } else {
          const int64_t target = (int64_t)cmin + (int64_t)p * kSjLtPanel;
          int32_t x = a, y = b; // first entry in [a, b) with colind >= target
          while (x < y) {
            const int32_t mid = x + ((y - x) >> 1);
            if ((int64_t)colind[mid] < target)
              x = mid + 1;
            else
              y = mid;
          }
          res = x;
        }
      }
      out[idx] = res;
    }
  }
}

// lengths of the listed rows (+ a zero behind them, for the scan)
__global__ __launch_bounds__(kBlock) void sj_lt_len_kernel(
    int nlong, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rows,
    int64_t* __restrict__ coff)
{
  for (int i = blockIdx.x * kBlock + threadIdx.x; i <= nlong; i += gridDim.x * kBlock)
    coff[i] = i < nlong ? (int64_t)(rowptr[rows[i] + 1] - rowptr[rows[i]]) : 0;
}

// every entry's column as its position inside its panel (one wave per row;
// supergroups that are not walked by panels keep zeros)
__global__ __launch_bounds__(kBlock) void sj_lt_codes_kernel(
    int nlong, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const int32_t* __restrict__ rows, const int32_t* __restrict__ cmin_in,
    const int32_t* __restrict__ np_in, const int64_t* __restrict__ coff,
    uint16_t* __restrict__ codes)
{
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t li = wid; li < nlong; li += nw) {
    const int sg = (int)(li / kSjLtRun);
    const int32_t a = rowptr[rows[li]], b = rowptr[rows[li] + 1];
    const int32_t cmin = cmin_in[sg];
    const bool by_panels = np_in[sg] > 0;
    uint16_t* out = codes + coff[li];
    for (int32_t e = a + lane; e < b; e += 64)
      out[e - a] = by_panels ? (uint16_t)((colind[e] - cmin) % kSjLtPanel) : (uint16_t)0;
  }
}

} // namespace

void spmv_sj_lt_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->sj_lt_codes);
  (void)hipFree(pl->sj_lt_coff);
  pl->sj_lt_codes = nullptr;
  pl->sj_lt_coff = nullptr;
  (void)hipFree(pl->sj_lt_cmin);
  (void)hipFree(pl->sj_lt_np);
  (void)hipFree(pl->sj_lt_off);
  (void)hipFree(pl->sj_lt_tab);
  pl->sj_lt_cmin = pl->sj_lt_np = pl->sj_lt_tab = nullptr;
  pl->sj_lt_off = nullptr;
  pl->sj_lt_entries = 0;
  pl->sj_lt_nsg = 0;
}

// (no memory: the plan stays without the table and the older kernel runs)
int spmv_sj_build_long_table(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                        const int32_t* colind, hipStream_t st)
{
  const int nlong = pl->sj_nlong;
  if (nlong <= 0 || !pl->sj_long_sorted)
    return SPMV_HIP_OK;
  const int nsg = (nlong + kSjLtRun - 1) / kSjLtRun;
  void* tmp = nullptr;
  size_t tb = 0;
  int64_t total = 0;
  hipError_t e = hipMalloc(&pl->sj_lt_cmin, sizeof(int32_t) * (size_t)nsg);
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_lt_np, sizeof(int32_t) * (size_t)nsg);
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_lt_off, sizeof(int64_t) * (size_t)(nsg + 1));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sj_lt_span_kernel, dim3(spmv_grid_for(pl->ctx, nsg, kBlock / 64)),
                       dim3(kBlock), 0, st, nlong, nsg, rowptr, colind,
                       pl->sj_long_rows, pl->sj_lt_cmin, pl->sj_lt_np, pl->sj_lt_off);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, pl->sj_lt_off, pl->sj_lt_off,
                                         nsg + 1, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, pl->sj_lt_off, pl->sj_lt_off,
                                         nsg + 1, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total, pl->sj_lt_off + nsg, sizeof(int64_t),
                       hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_lt_tab, sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
  if (e == hipSuccess && total > 0) {
    hipLaunchKernelGGL(sj_lt_fill_kernel, dim3(spmv_grid_for(pl->ctx, nsg, 1)),
                       dim3(kBlock), 0, st, nlong, nsg, rowptr, colind,
                       pl->sj_long_rows, pl->sj_lt_cmin, pl->sj_lt_np, pl->sj_lt_off,
                       pl->sj_lt_tab);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  // the rows' columns as 16-bit panel positions (kSjLtCodes)
  int64_t ncodes = 0;
  if (e == hipSuccess && kSjLtCodes) {
    void* tmp2 = nullptr;
    size_t tb2 = 0;
    e = hipMalloc(&pl->sj_lt_coff, sizeof(int64_t) * ((size_t)nlong + 1));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sj_lt_len_kernel, dim3(spmv_grid_for(pl->ctx, nlong + 1, kBlock)),
                         dim3(kBlock), 0, st, nlong, rowptr, pl->sj_long_rows,
                         pl->sj_lt_coff);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, pl->sj_lt_coff, pl->sj_lt_coff,
                                           nlong + 1, st);
    if (e == hipSuccess)
      e = hipMalloc(&tmp2, tb2 ? tb2 : 16);
    if (e == hipSuccess)
      e = hipcub::DeviceScan::ExclusiveSum(tmp2, tb2, pl->sj_lt_coff, pl->sj_lt_coff,
                                           nlong + 1, st);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&ncodes, pl->sj_lt_coff + nlong, sizeof(int64_t),
                         hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    (void)hipFree(tmp2);
    // (loads run up to two trips past a row's end: slack behind the last row)
    const size_t nalloc = (size_t)ncodes + 4 * kSjLtTrip * (kSjLtDepth + 1);
    if (e == hipSuccess)
      e = hipMalloc(&pl->sj_lt_codes, sizeof(uint16_t) * nalloc);
    if (e == hipSuccess)
      e = hipMemsetAsync(pl->sj_lt_codes, 0, sizeof(uint16_t) * nalloc, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sj_lt_codes_kernel,
                         dim3(spmv_grid_for(pl->ctx, nlong, kBlock / 64)), dim3(kBlock), 0,
                         st, nlong, rowptr, colind, pl->sj_long_rows, pl->sj_lt_cmin,
                         pl->sj_lt_np, pl->sj_lt_coff, pl->sj_lt_codes);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
  }
  if (e != hipSuccess) {
    spmv_sj_lt_free(pl);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  pl->sj_lt_entries = total;
  pl->sj_lt_codes_n = ncodes;
  pl->sj_lt_nsg = nsg;
  return spmv_sj_lt_raise_lds();
}

namespace
{

template <typename T>
int sj_bake(spmv_hip_csr_plan* pl, const T* values, const int32_t* map, hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  if (values == nullptr) { // drop the copy
    SPMV_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(pl->sj_val);
    pl->sj_val = nullptr;
    pl->sj_values0 = nullptr;
    pl->sj_elem = 0;
    return SPMV_HIP_OK;
  }
  if (!pl->sj_lenperm || pl->nnz == 0)
    return SPMV_HIP_ENOTSUP;
  const auto t_begin = std::chrono::steady_clock::now();
  if (pl->sj_val && pl->sj_elem != (int)sizeof(T)) {
    SPMV_CHECK_HIP(hipDeviceSynchronize());
    (void)hipFree(pl->sj_val);
    pl->sj_val = nullptr;
  }
  if (!pl->sj_val) {
    const size_t entries = (size_t)(pl->sj_units + kSjSlack) * pl->sj_unit;
    hipError_t e = hipMalloc(&pl->sj_val, sizeof(T) * entries);
    if (e == hipSuccess) // the slack is read (never used): keep it finite
      e = hipMemsetAsync(static_cast<T*>(pl->sj_val)
                             + (size_t)pl->sj_units * pl->sj_unit,
                         0, sizeof(T) * (size_t)kSjSlack * pl->sj_unit, st);
    if (e != hipSuccess) {
      (void)hipFree(pl->sj_val);
      pl->sj_val = nullptr;
      (void)hipGetLastError();
      return e == hipErrorOutOfMemory ? SPMV_HIP_ENOTSUP : static_cast<int>(e);
    }
  }
  const int64_t nsl = ((int64_t)pl->num_rows + 63) / 64;
  const int grid = spmv_grid_for(pl->ctx, nsl, kBlock / 64);
  hipLaunchKernelGGL((sj_bake_kernel<T>), dim3(grid), dim3(kBlock), 0, st,
                     pl->num_rows, pl->rowptr0, pl->sj_lenperm, pl->sj_ubase,
                     pl->sj_unit, values, map, static_cast<T*>(pl->sj_val),
                     pl->sj_sigma ? kSjSigBits : 6);
  SPMV_CHECK_LAUNCH();
  SPMV_CHECK_HIP(hipStreamSynchronize(st));
  pl->sj_elem = (int)sizeof(T);
  pl->sj_values0 = values;
  pl->sj = 1;
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count();
  return SPMV_HIP_OK;
}

struct SjShortLen { // length of a row that stays in the slices (a long one: 0)
  const int32_t* rowptr;
  int thr;
  int64_t nnz;
  __device__ int32_t operator()(int i) const
  {
    const int32_t a = rowptr[i], b = rowptr[i + 1];
    return sj_is_long(a, b, thr, nnz) ? 0 : b - a;
  }
};
struct SjLongEntries {
  const int32_t* rowptr;
  int thr;
  int64_t nnz;
  __device__ int64_t operator()(int i) const
  {
    const int32_t a = rowptr[i], b = rowptr[i + 1];
    return sj_is_long(a, b, thr, nnz) ? (int64_t)(b - a) : 0;
  }
};
} // namespace

// entries in the rows the general form would take out of the slices as LONG
// (more than four times the average and more than 96 entries)
int spmv_sjds_long_entries(spmv_hip_ctx* ctx, int32_t num_rows, int64_t nnz,
                           const int32_t* rowptr, int64_t* entries, hipStream_t st)
{
  *entries = 0;
  if (num_rows < 1 || nnz < 1)
    return SPMV_HIP_OK;
  int thr = (int)(nnz * 4 / num_rows);
  thr = thr > kSjLongMin ? thr : kSjLongMin;
  hipcub::CountingInputIterator<int32_t> first(0);
  hipcub::TransformInputIterator<int64_t, SjLongEntries,
                                 hipcub::CountingInputIterator<int32_t>>
      it(first, SjLongEntries{rowptr, thr, nnz});
  int64_t* d_sum = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  hipError_t e = hipMalloc(&d_sum, sizeof(int64_t));
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(nullptr, tb, it, d_sum, num_rows, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceReduce::Sum(tmp, tb, it, d_sum, num_rows, st);
  if (e == hipSuccess)
    e = hipMemcpyAsync(entries, d_sum, sizeof(int64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(tmp);
  (void)hipFree(d_sum);
  (void)ctx;
  return e == hipSuccess ? SPMV_HIP_OK : static_cast<int>(e);
}

void spmv_sjds_free(spmv_hip_csr_plan* pl)
{
  (void)hipFree(pl->sj_lenperm);
  (void)hipFree(pl->sj_blk);
  (void)hipFree(pl->sj_chunks);
  (void)hipFree(pl->sj_codes);
  (void)hipFree(pl->sj_val);
  (void)hipFree(pl->sj_val32);
  pl->sj_val32 = nullptr;
  pl->sj32_values0 = nullptr;
  (void)hipFree(pl->sj_long_rows);
  (void)hipFree(pl->sj_ubase);
  spmv_sj_lt_free(pl);
  pl->sj_ubase = nullptr;
  pl->sj_long_rows = nullptr;
  pl->sj_nlong = 0;
  pl->sj_lenperm = pl->sj_blk = pl->sj_chunks = nullptr;
  pl->sj_codes = nullptr;
  pl->sj_val = nullptr;
  pl->sj_values0 = nullptr;
  pl->sj = pl->sj_elem = 0;
}

// Build the structure (everything but the values).  wpb_force: 4, 8, 16, or 0
// = choose; unit_force: 1, 2, 4 entries per lane and step, or 0 = choose.
// Leaves the plan without the form (SPMV_HIP_OK) when it does not pay: no
// memory, or nearly all entries far.
// no_long (the two blocks of symmetric storage, whose kernel modes exist for
// the slices only): no row leaves the slices, blocks of 8 or 16 slices, two
// entries per lane and step.
int spmv_sjds_build(spmv_hip_csr_plan* pl, const int32_t* rowptr,
                    const int32_t* colind, int wpb_force, int unit_force, int no_long)
{
  if (no_long) {
    unit_force = 2;
    if (wpb_force != 8 && wpb_force != 16)
      wpb_force = 0;
  }
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  hipStream_t st = pl->ctx->stream;
  const int n = pl->num_rows;
  if (n < 64 || pl->nnz < 1)
    return SPMV_HIP_OK;
  const PlanTrace tr(" sjds_build");
  const int kcap = pl->ctx->sj_max_chunks;
  // long rows: more than four times the average length, and more than 96
  int thr = (int)(pl->nnz * 4 / n);
  thr = thr > kSjLongMin ? thr : kSjLongMin;
  if (no_long)
    thr = INT32_MAX;
  // entries per lane and step: a unit's padding (half a unit per row) against
  // the instructions of a step.  (With the sigma layout, same box: lengths 5-40
  // 0.370 / 0.349 / 0.364 ms for 1 / 2 / 4 entries per step; 7 in every row with
  // 32-bit codes 0.327 / 0.328 / 0.317.)
  const double avg = (double)pl->nnz / n;
  // (measured, 10 M rows: lengths 5-40 0.55 / 0.41 / 0.44 ms with 1 / 2 / 4
  // entries per step; 81 per row 1.38 / 1.37 / 1.38; 7 per row 0.357 / 0.355 /
  // 0.349)
  int E = unit_force ? unit_force : (avg >= 4.0 ? 2 : 1);
  const int nblk4 = (n + 255) / 256;
  const int64_t nsl = ((int64_t)n + 63) / 64;
  int32_t* d_k = nullptr;
  int32_t* d_far3[3] = {nullptr, nullptr, nullptr}; // per candidate: the fill
                                                    // pass reads the winner's
  int64_t* d_stats = nullptr;
  hipError_t e = hipMalloc(&d_k, sizeof(int32_t) * (size_t)nblk4);
  for (int ci = 0; ci < 3 && e == hipSuccess; ++ci)
    e = hipMalloc(&d_far3[ci], sizeof(int32_t) * (size_t)nblk4);
  if (e == hipSuccess)
    e = hipMalloc(&d_stats, sizeof(int64_t) * 4);
  auto cleanup = [&]() {
    (void)hipFree(d_k);
    for (int32_t* p : d_far3)
      (void)hipFree(p);
    (void)hipFree(d_stats);
  };
  if (e != hipSuccess) {
    cleanup();
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  // Candidates: 4, 8 or 16 slices share one staged copy of x.  More rows per
  // copy = fewer staged bytes per entry, but a larger copy (fewer workgroups
  // per CU) and more rows waiting at each of the block's two barriers.  Cost =
  // bytes per entry: the matrix stream, the staged chunks (L2 traffic, priced
  // at a third), a 64-byte sector per far entry; a candidate that leaves a CU
  // fewer than 16 waves pays in proportion.
  const int cand[3] = {4, 8, 16};
  int best = 0, best_ci = 0;
  double best_cost = 0.0;
  SjStats best_st;
  auto count = [&](int wpb, int ci, SjStats* s) {
    return wpb == 4 ? sj_count<256>(pl, rowptr, colind, kcap, thr, d_k, d_far3[ci],
                                    d_stats, s, st)
           : wpb == 8
               ? sj_count<512>(pl, rowptr, colind, kcap, thr, d_k, d_far3[ci], d_stats,
                               s, st)
               : sj_count<1024>(pl, rowptr, colind, kcap, thr, d_k, d_far3[ci],
                                d_stats, s, st);
  };
  // (the largest block first: when it stages every entry -- no far ones -- and
  // leaves the CU its 16 waves, the smaller blocks can only stage more bytes
  // per entry, and their analysis passes over the matrix are saved (10 M rows x
  // 15: 12 -> 10.5 ms; at 0.8 G entries the fill pass dominates either way)
  for (int ci = 2; ci >= 0; --ci) {
    const int wpb = cand[ci];
    if (wpb_force && wpb != wpb_force)
      continue;
    if (no_long && wpb == 4)
      continue;
    if (!wpb_force && n < 64 * wpb * 8) // too few blocks for this size
      continue;
    SjStats s;
    const int rc = count(wpb, ci, &s);
    if (rc != SPMV_HIP_OK) {
      cleanup();
      return rc;
    }
    const int64_t lds = s.maxk * kSjChunk * 8 + 128;
    const int waves = sj_wgs_per_cu(wpb, lds) * wpb;
    double cost = (s.far > 0 ? 12.0 : 10.0)
                  + (double)s.sumk * 128.0 / 3.0 / (double)pl->nnz
                  + 64.0 * (double)s.far / (double)pl->nnz;
    if (waves < 16)
      cost *= 16.0 / waves;
    if (!best || cost < best_cost) {
      best = wpb;
      best_ci = ci;
      best_cost = cost;
      best_st = s;
    }
    if (s.far == 0 && waves >= 16)
      break;
  }
  if (!best) { // a matrix too small for any candidate: the smallest
    best = wpb_force ? wpb_force : (no_long ? 8 : 4);
    best_ci = best == 4 ? 0 : best == 8 ? 1 : 2;
    const int rc = count(best, best_ci, &best_st);
    if (rc != SPMV_HIP_OK) {
      cleanup();
      return rc;
    }
  }
  int32_t* d_far = d_far3[best_ci];
  // nearly all entries far: the form buys nothing
  if (best_st.far * 10 > pl->nnz * 9 && !wpb_force) {
    cleanup();
    return SPMV_HIP_OK;
  }
  tr.mark("count");
  // The SIGMA layout (blocks of 16 slices sorted by length across the block,
  // two slices per wave: sj_sigma_kernel) for blocks of 1024 rows; `nsl` then
  // counts the slices of whole blocks.
  // (measured, 10 M rows, same box: lengths 5-40 0.443 -> 0.360 ms, 7 in every
  // row 0.371 -> 0.329; 81 in every row 1.37 -> 1.40: nothing to sort there, and
  // the 16-wave workgroup streams long slices a little better)
  // many far entries (5 % or more: 32-bit codes in practically every block):
  // four entries per step -- 16-byte loads of the codes (sigma layout, 7 per
  // row with 9 % far entries: 0.328 -> 0.317 ms; without far entries two per
  // step are better, 0.349 against 0.364)
  if (!unit_force && avg >= 4.0 && best_st.far * 20 >= pl->nnz)
    E = 4;
  // ... and ragged rows of any length gain (same box, 5 M rows: lengths 20-80
  // 0.395 -> 0.366 ms, 40-120 0.593 -> 0.573): the layout is left only for rows
  // that are long AND (nearly) all alike -- the longest row that stays in the
  // slices within 10 % of the average
  // The longest row that stays in the slices.  Its length shares a 32-bit word
  // with the row's position (and bit 31, the long rows' mark): 21 bits beside
  // the 10 of the sigma layout, 25 beside a slice's 6.  A bordered matrix's
  // dense last row -- never LONG: sj_is_long keeps the rows at the arrays' end
  // in the slices -- or, in symmetric storage, a dense column can exceed that.
  int32_t h_max = 0;
  {
    hipcub::CountingInputIterator<int32_t> first(0);
    hipcub::TransformInputIterator<int32_t, SjShortLen,
                                   hipcub::CountingInputIterator<int32_t>>
        lens(first, SjShortLen{rowptr, thr, pl->nnz});
    int32_t* d_max = nullptr;
    void* tmpm = nullptr;
    size_t tbm = 0;
    hipError_t em = hipMalloc(&d_max, sizeof(int32_t));
    if (em == hipSuccess)
      em = hipcub::DeviceReduce::Max(nullptr, tbm, lens, d_max, n, st);
    if (em == hipSuccess)
      em = hipMalloc(&tmpm, tbm ? tbm : 16);
    if (em == hipSuccess)
      em = hipcub::DeviceReduce::Max(tmpm, tbm, lens, d_max, n, st);
    if (em == hipSuccess)
      em = hipMemcpyAsync(&h_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (em == hipSuccess)
      em = hipStreamSynchronize(st);
    (void)hipFree(tmpm);
    (void)hipFree(d_max);
    if (em != hipSuccess) {
      cleanup();
      (void)hipGetLastError();
      return em == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(em);
    }
  }
  if (h_max >= (1 << (31 - 6))) { // no word holds it: the form is not taken
    cleanup();
    return SPMV_HIP_OK;
  }
  tr.mark("max length");
  int sigma = 0;
  if (best == 16 && pl->ctx->sj_sigma && h_max < (1 << (31 - kSjSigBits))) {
    sigma = 1;
    if (pl->ctx->sj_sigma == 1 && avg >= 48.0 && (double)h_max <= 1.1 * avg)
      sigma = 0;
  }
  const int64_t nsl_all = sigma ? (((int64_t)n + kSjSigRows - 1) / kSjSigRows) * 16 : nsl;
  // first unit of every slice: scan of the slices' unit counts
  uint32_t total_units = 0;
  {
    void* tmp = nullptr;
    size_t tb = 0;
    e = hipMalloc(&pl->sj_ubase, sizeof(uint32_t) * (size_t)(nsl_all + 1));
    if (e == hipSuccess)
      e = hipMalloc(&pl->sj_lenperm, sizeof(int32_t) * (size_t)nsl_all * 64);
    if (e == hipSuccess) {
      if (sigma)
        hipLaunchKernelGGL(sj_sigma_kernel,
                           dim3(spmv_grid_for(pl->ctx, nsl_all / 16 + 1, 1)), dim3(kBlock),
                           0, st, n, rowptr, thr, pl->nnz, E, pl->sj_lenperm,
                           pl->sj_ubase);
      else
        hipLaunchKernelGGL(sj_units_kernel, dim3(spmv_grid_for(pl->ctx, nsl + 1, 4)),
                           dim3(kBlock), 0, st, n, rowptr, thr, pl->nnz, E,
                           pl->sj_ubase);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, pl->sj_ubase, pl->sj_ubase,
                                           (int)(nsl_all + 1), st);
    if (e == hipSuccess)
      e = hipMalloc(&tmp, tb ? tb : 16);
    if (e == hipSuccess)
      e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, pl->sj_ubase, pl->sj_ubase,
                                           (int)(nsl_all + 1), st);
    if (e == hipSuccess)
      e = hipMemcpyAsync(&total_units, pl->sj_ubase + nsl_all, sizeof(uint32_t),
                         hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
  }
  tr.mark("sigma / units");
  const int R = 64 * best;
  const int nblk = (n + R - 1) / R;
  const int stride = best_st.maxk > 0 ? (int)((best_st.maxk + 7) / 8 * 8) : 8;
  const int wide_alloc = best_st.far > 0 ? 1 : 0;
  const size_t code_bytes
      = (size_t)(wide_alloc ? 4 : 2) * ((size_t)total_units + kSjSlack) * E;
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_blk, sizeof(int32_t) * 2 * (size_t)nblk);
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_chunks, sizeof(int32_t) * (size_t)nblk * stride);
  if (e == hipSuccess)
    e = hipMalloc(&pl->sj_codes, code_bytes);
  if (e == hipSuccess) // (the slack's codes must be valid LDS indices: 0)
    e = hipMemsetAsync(pl->sj_codes, 0, code_bytes, st);
  tr.mark("alloc codes");
  if (e == hipSuccess) {
    const int grid = spmv_grid_for(pl->ctx, nblk, 1);
#define SJ_FILL(RR)                                                            \
  hipLaunchKernelGGL((sj_fill_kernel<RR>), dim3(grid), dim3(kBlock), 0, st, n,  \
                     pl->num_cols, rowptr, colind, kcap, thr, pl->nnz, E, stride, \
                     wide_alloc, d_far, pl->sj_ubase, pl->sj_blk,              \
                     pl->sj_chunks, pl->sj_lenperm, pl->sj_codes, sigma)
    if (best == 4)
      SJ_FILL(256);
    else if (best == 8)
      SJ_FILL(512);
    else
      SJ_FILL(1024);
#undef SJ_FILL
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  tr.mark("fill");
  cleanup();
  if (e != hipSuccess) {
    spmv_sjds_free(pl);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  {
    const int rc = spmv_sj_build_long_list(pl, rowptr, colind, thr, st);
    if (rc != SPMV_HIP_OK) {
      spmv_sjds_free(pl);
      return rc == SPMV_HIP_ENOMEM ? SPMV_HIP_OK : rc;
    }
    const int rc2 = spmv_sj_build_long_table(pl, rowptr, colind, st);
    if (rc2 != SPMV_HIP_OK) {
      spmv_sjds_free(pl);
      return rc2;
    }
  }
  pl->sj_long_thr = thr;
  pl->sj_unit = E;
  pl->sj_units = total_units;
  pl->sj_wpb = best;
  pl->sj_sigma = sigma;
  pl->sj_nblk = nblk;
  pl->sj_maxk = (int)best_st.maxk > 0 ? (int)best_st.maxk : 1;
  pl->sj_stride = stride;
  pl->sj_wide_alloc = wide_alloc;
  pl->sj_far = best_st.far;
  pl->sj_sumk = best_st.sumk;
  return SPMV_HIP_OK;
}

int spmv_sjds_bake_f64(spmv_hip_csr_plan* pl, const double* values,
                       const int32_t* map, hipStream_t st)
{
  return sj_bake<double>(pl, values, map, st);
}
int spmv_sjds_bake_f32(spmv_hip_csr_plan* pl, const float* values, const int32_t* map,
                       hipStream_t st)
{
  return sj_bake<float>(pl, values, map, st);
}

// Mixed precision: the fp32 twin of the jagged copy (fp64 vectors and
// arithmetic; the long rows read the caller's fp32 CSR values).  values32 ==
// nullptr drops it.
int spmv_sjds_bake_f32f64(spmv_hip_csr_plan* pl, const float* values32, hipStream_t st)
{
  SPMV_CHECK_HIP(hipSetDevice(pl->ctx->device));
  if (values32 == nullptr) {
    if (pl->sj_val32) {
      SPMV_CHECK_HIP(hipDeviceSynchronize());
      (void)hipFree(pl->sj_val32);
    }
    pl->sj_val32 = nullptr;
    pl->sj32_values0 = nullptr;
    return SPMV_HIP_OK;
  }
  if (!pl->sj_lenperm || !pl->sj_val || pl->sj_elem != 8 || pl->symmetric
      || pl->nnz == 0)
    return SPMV_HIP_ENOTSUP;
  const auto t_begin = std::chrono::steady_clock::now();
  if (!pl->sj_val32) {
    const size_t entries = (size_t)(pl->sj_units + kSjSlack) * pl->sj_unit;
    hipError_t e = hipMalloc(&pl->sj_val32, sizeof(float) * entries);
    if (e == hipSuccess)
      e = hipMemsetAsync(static_cast<float*>(pl->sj_val32)
                             + (size_t)pl->sj_units * pl->sj_unit,
                         0, sizeof(float) * (size_t)kSjSlack * pl->sj_unit, st);
    if (e != hipSuccess) {
      (void)hipFree(pl->sj_val32);
      pl->sj_val32 = nullptr;
      (void)hipGetLastError();
      return e == hipErrorOutOfMemory ? SPMV_HIP_ENOTSUP : static_cast<int>(e);
    }
  }
  const int64_t nsl = ((int64_t)pl->num_rows + 63) / 64;
  const int grid = spmv_grid_for(pl->ctx, nsl, kBlock / 64);
  hipLaunchKernelGGL((sj_bake_kernel<float>), dim3(grid), dim3(kBlock), 0, st,
                     pl->num_rows, pl->rowptr0, pl->sj_lenperm, pl->sj_ubase,
                     pl->sj_unit, values32, (const int32_t*)nullptr,
                     static_cast<float*>(pl->sj_val32), pl->sj_sigma ? kSjSigBits : 6);
  SPMV_CHECK_LAUNCH();
  SPMV_CHECK_HIP(hipStreamSynchronize(st));
  pl->sj32_values0 = values32;
  pl->plan_us += (int)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now() - t_begin)
                     .count();
  return SPMV_HIP_OK;
}

namespace
{

// lengths of the merged rows (+ a zero behind them, for the scan)
// (a row whose stored lower part is LONG keeps only its column's entries: the
// long-row kernels take the lower part from the caller's arrays)
__global__ __launch_bounds__(kBlock) void sj_sym_len_kernel(
    int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ t_ptr,
    int long_thr, int64_t nnz, int32_t* __restrict__ vptr)
{
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n;
       i += (int64_t)gridDim.x * kBlock) {
    int32_t len = 0;
    if (i < n) {
      const int32_t a = rowptr[i], b = rowptr[i + 1];
      len = (sj_is_long(a, b, long_thr, nnz) ? 0 : b - a) + (t_ptr[i + 1] - t_ptr[i]);
    }
    vptr[i] = len;
  }
}

// columns of the merged rows and where their values are in the caller's array
// (one wave per row)
__global__ __launch_bounds__(kBlock) void sj_sym_merge_kernel(
    int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const int32_t* __restrict__ t_ptr, const int32_t* __restrict__ t_row,
    const int32_t* __restrict__ t_pos, const int32_t* __restrict__ vptr,
    int long_thr, int64_t nnz, int32_t* __restrict__ vcol, int32_t* __restrict__ vmap)
{
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t i = wid; i < n; i += nw) {
    const int32_t a = rowptr[i];
    const int32_t nl = sj_is_long(a, rowptr[i + 1], long_thr, nnz) ? 0 : rowptr[i + 1] - a;
    const int32_t ta = t_ptr[i], nu = t_ptr[i + 1] - ta;
    const int32_t d = vptr[i];
    for (int32_t k = lane; k < nl; k += 64) {
      vcol[d + k] = colind[a + k];
      vmap[d + k] = a + k;
    }
    for (int32_t k = lane; k < nu; k += 64) {
      vcol[d + nl + k] = t_row[ta + k];
      vmap[d + nl + k] = t_pos[ta + k];
    }
  }
}
} // namespace

// The merged matrix of a symmetric plan with its transposed map: row pointer,
// columns and value positions, owned by the caller (hipFree).  ENOMEM: no memory.
// long_thr: rows of the stored lower block with more entries keep only their
// column's entries (INT32_MAX: every row whole); *total: the merged entries.
int spmv_sjds_sym_merge(const spmv_hip_csr_plan* pl, int long_thr, int32_t** vptr,
                        int32_t** vcol, int32_t** vmap, int64_t* total, hipStream_t st)
{
  const int32_t n = pl->num_rows;
  const int64_t nnz2 = 2 * pl->nnz;
  if (nnz2 > INT32_MAX)
    return SPMV_HIP_ENOTSUP;
  *vptr = *vcol = *vmap = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  hipError_t e = hipMalloc(vptr, sizeof(int32_t) * ((size_t)n + 1));
  if (e == hipSuccess)
    e = hipMalloc(vcol, sizeof(int32_t) * (size_t)nnz2);
  if (e == hipSuccess)
    e = hipMalloc(vmap, sizeof(int32_t) * (size_t)nnz2);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sj_sym_len_kernel, dim3(spmv_grid_for(pl->ctx, n + 1, kBlock)),
                       dim3(kBlock), 0, st, n, pl->rowptr0, pl->t_ptr, long_thr, pl->nnz,
                       *vptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, *vptr, *vptr, n + 1, st);
  if (e == hipSuccess)
    e = hipMalloc(&tmp, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, *vptr, *vptr, n + 1, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sj_sym_merge_kernel,
                       dim3(spmv_grid_for(pl->ctx, n, kBlock / 64)), dim3(kBlock), 0, st,
                       n, pl->rowptr0, pl->colind0, pl->t_ptr, pl->t_row, pl->t_pos,
                       *vptr, long_thr, pl->nnz, *vcol, *vmap);
    e = hipGetLastError();
  }
  int32_t h_total = 0;
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_total, *vptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  *total = h_total;
  (void)hipFree(tmp);
  if (e != hipSuccess) {
    (void)hipFree(*vptr);
    (void)hipFree(*vcol);
    (void)hipFree(*vmap);
    *vptr = *vcol = *vmap = nullptr;
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_ENOMEM : static_cast<int>(e);
  }
  return SPMV_HIP_OK;
}

// Symmetric storage of a matrix without lattice structure: everything of the
// merged form but the values (sym_sj_bake in spmv_csr_plan.hip bakes those).  The
// rows of the stored lower block that are LONG (more than four times the merged
// matrix's average length, and more than 96 entries) are listed in the PARENT
// plan, with the table of the table-driven kernel, exactly as a general
// matrix's long rows are: the long-row kernels read them from the caller's
// arrays.  Long COLUMNS stay inside the slices (as rows of the merged matrix):
// a matrix with more than sym_sj_long_permille of its entries in them keeps the
// transposed-map kernel.  ENOTSUP: the form does not apply.
int spmv_sjds_sym_build(spmv_hip_ctx* ctx, spmv_hip_csr_plan* plan, hipStream_t st)
{
  const PlanTrace tr("sym_build");
  int64_t la = 0, lb = 0;
  int rl = spmv_sjds_long_entries(ctx, plan->num_rows, plan->nnz, plan->t_ptr, &lb, st);
  if (rl == SPMV_HIP_OK && !ctx->sym_sj_long_rows)
    rl = spmv_sjds_long_entries(ctx, plan->num_rows, plan->nnz, plan->rowptr0, &la, st);
  if (rl != SPMV_HIP_OK)
    return rl;
  if ((la + lb) * 1000 > (int64_t)ctx->sym_sj_long_permille * 2 * plan->nnz)
    return SPMV_HIP_ENOTSUP;
  // (the merged matrix has 2 nnz entries, long rows included)
  int thr = (int)(2 * plan->nnz * 4 / plan->num_rows);
  thr = thr > kSjLongMin ? thr : kSjLongMin;
  if (!ctx->sym_sj_long_rows)
    thr = INT32_MAX;
  // (one clean-up for every way out without the form: the parent's long-row
  // state must not outlive a form that was not built -- ADVICE r05)
  auto drop_long = [&]() {
    spmv_sj_lt_free(plan);
    (void)hipFree(plan->sj_long_rows);
    plan->sj_long_rows = nullptr;
    plan->sj_nlong = 0;
    plan->sj_long_thr = INT32_MAX;
  };
  drop_long();
  plan->sj_long_thr = thr;
  if (thr != INT32_MAX) {
    int rc = spmv_sj_build_long_list(plan, plan->rowptr0, plan->colind0, thr, st);
    if (rc == SPMV_HIP_OK && plan->sj_nlong > 0)
      rc = spmv_sj_build_long_table(plan, plan->rowptr0, plan->colind0, st);
    if (rc != SPMV_HIP_OK) {
      drop_long();
      return rc == SPMV_HIP_ENOMEM ? SPMV_HIP_ENOTSUP : rc;
    }
  }
  if (plan->sj_nlong == 0)
    plan->sj_long_thr = thr = INT32_MAX;
  tr.mark("long rows");
  // the merged matrix: per row its (short) lower part, then its column's entries
  int64_t total = 0;
  const int rm = spmv_sjds_sym_merge(plan, thr, &plan->sjv_ptr, &plan->sjv_col,
                                     &plan->sjv_map, &total, st);
  if (rm != SPMV_HIP_OK) {
    drop_long();
    return rm == SPMV_HIP_ENOMEM ? SPMV_HIP_ENOTSUP : rm;
  }
  tr.mark("merge");
  spmv_hip_csr_plan* ch = new (std::nothrow) spmv_hip_csr_plan;
  int rb = ch ? SPMV_HIP_OK : SPMV_HIP_ENOMEM;
  if (ch) {
    ch->ctx = ctx;
    ch->num_rows = plan->num_rows;
    ch->num_cols = plan->num_cols;
    ch->nnz = total;
    ch->symmetric = false;
    ch->rowptr0 = plan->sjv_ptr;
    ch->colind0 = plan->sjv_col;
    rb = total > 0 ? spmv_sjds_build(ch, plan->sjv_ptr, plan->sjv_col, ctx->sj_wpb, 2, 1)
                   : SPMV_HIP_OK;
  }
  tr.mark("sjds_build");
  // (the columns were for the analysis only: the kernel reads its codes)
  (void)hipFree(plan->sjv_col);
  plan->sjv_col = nullptr;
  if (ch)
    ch->colind0 = nullptr;
  if (rb != SPMV_HIP_OK || !ch->sj_lenperm) {
    if (ch) {
      spmv_sjds_free(ch);
      delete ch;
    }
    (void)hipFree(plan->sjv_ptr);
    (void)hipFree(plan->sjv_map);
    plan->sjv_ptr = plan->sjv_map = nullptr;
    drop_long();
    return rb != SPMV_HIP_OK ? rb : SPMV_HIP_ENOTSUP;
  }
  plan->sjt = ch;
  return SPMV_HIP_OK;
}
