// Sliced jagged form ("SJDS") of a general CSR matrix for gfx950 (MI355X):
// the kernel for matrices WITHOUT lattice structure -- ragged rows, FEM
// matrices with 15-80 entries per row, a tail of rows with thousands of
// entries -- behind CSRSpMV<T>::run (spmv/csr_kernels.cpp:41-51).
//
// What the row-block gather kernel (spmv_csr.hip) pays for on such matrices is
// the gather itself: one lane per ENTRY means 64 lanes of a wave ask for 64
// different 128-byte lines of x, and every one of them is a request to the L2
// (10 M rows x 15 entries: 150 M requests, the L2's whole request rate for
// half a millisecond).  Here
//
//   * lane = ROW.  The plan re-lays the matrix out (plan_bake_values: the
//     plan's own copy of the values; spmv_hip_csr_plan_values_changed after an
//     update in place) in slices of 64 consecutive rows, each slice stored as
//     jagged diagonals: the slice's rows sorted by length (descending, stable),
//     then UNIT k (E = 1, 2 or 4 consecutive entries) of every row that has
//     one, side by side.  Step k of a wave is one coalesced load of the values
//     (8 E bytes per lane) and one of the column codes by the first cnt_k
//     lanes; rows are padded to whole units only.  A lane adds its own row's
//     products in the row's order: the bits of the reference loop, no LDS
//     parking of products, no barrier inside a slice.  The step's address is
//     scalar arithmetic (slice base + units so far) plus a constant per-lane
//     offset: what limited the first version of this kernel was the number of
//     vector instructions per entry, not bytes.
//     SIGMA layout (blocks of 1024 rows, unless the rows are long and all
//     alike): a slice runs as many steps as its LONGEST row -- with lengths
//     5 ... 40 side by side 2.5 times the average, and the slices' phase is three
//     quarters of the kernel (a build with clocks in it, -DSJ_PROBE; rows of
//     one length: 0.35 instead of 0.43 ms).  There the rows are sorted by
//     length across the whole BLOCK (the staged copy of x is the block's
//     anyway), a slice holds rows of nearly one length, and a wave of the
//     8-wave workgroup takes TWO slices, the k-th longest and the k-th
//     shortest: every wave the same work, two workgroups per CU.  Lengths 5-40:
//     0.443 -> 0.360 ms; 7 in every row: 0.371 -> 0.329.
//   * x comes from LDS.  Per block of WPB slices (256, 512 or 1024 rows) the
//     plan lists the 16-column chunks (128 B) of x the block's entries touch
//     -- up to 432 of them, nearest to the diagonal first -- and rewrites every
//     column as a 16-bit index into the staged copy.  Entries outside the list
//     ("far") keep their column and are gathered from memory; a block that has
//     any uses 32-bit codes.  Staging is coalesced 16-byte loads, shared by all
//     rows of the block; the gather becomes an LDS read.
//   * long rows.  A row with more than four times the average length (and
//     more than 96 entries) stays out of the slices: the kernel's first phase
//     gives eight such rows to one wave, eight lanes each, streamed from the
//     caller's CSR arrays (values and columns two load groups ahead, x one);
//     a group's eight products are added to the row's sum one by one in the
//     row's order -- the reference's bits.  The rows are taken in ROW order
//     (neighbours share their x lines in the L2), sorted by length only inside
//     runs of 64.  Inside a slice, past the second-longest row, the wave takes
//     over lane 0's row 64 entries at a time (v_readlane + add).
//
// Bytes per launch: nnz * (8 + 2) + rows * (4 + 8) + x (+ the chunk lists)
// against CSR's nnz * 12 + rows * 12 + x.
#include "csr_plan.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <type_traits>

namespace
{

constexpr int kSjChunk = 16;       // columns per staged chunk
constexpr int kSjSpanWords = 2048; // bitmap words of the plan analysis: 65,536
                                   // chunks = 2^20 columns around the block
constexpr int kSjGroup = 8;        // entries per lane and load group (two groups
                                   // in flight): 8 / E steps
constexpr int kSjTailMin = 48;     // entries past the second-longest row from
                                   // which the wave takes lane 0's row over
constexpr int kSjLongMin = 96;     // LONG rows: more than 4 x the average and
                                   // more than this many entries
constexpr uint32_t kSjLongFlag = 0x80000000u; // ... marked in their lenperm word
constexpr int kSjSlack = 128;      // units of slack behind the jagged arrays: a
                                   // step past a slice's end reads, never uses

// LONG rows (the long-row kernel takes them, the slices leave them out): more
// than `thr` entries -- and not within kSjLongPad entries of the arrays' end,
// so that the kernel's loads may run past a row's end without a clamp
constexpr int kSjLongPad = 160;
__host__ __device__ __forceinline__ bool sj_is_long(int32_t a, int32_t b, int thr,
                                                    int64_t nnz)
{
  return b - a > thr && (int64_t)b + kSjLongPad <= nnz;
}

// E consecutive entries of a row: one aligned load
template <typename X, int E>
struct __attribute__((aligned(sizeof(X) * E))) SjUnit {
  X e[E];
};

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
                           const int32_t* __restrict__ rowptr,
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
  // kernel takes them), so they do not choose chunks
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
    const SjSel sel = sj_select(r0, r1, num_cols, rowptr, colind, kcap, long_thr,
                                nnz, s_bits, s_pre, &s_sel);
    int32_t far = 0;
    for (int32_t row = r0 + threadIdx.x; row < r1; row += kBlock) {
      const int32_t a = rowptr[row], e1 = rowptr[row + 1];
      if (sj_is_long(a, e1, long_thr, nnz))
        continue;
      for (int32_t e = a; e < e1; ++e)
        far += sj_index(sel, s_bits, s_pre, colind[e]) < 0 ? 1 : 0;
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
constexpr int kSjSigRows = 1024;
constexpr int kSjSigBits = 10;
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
    const SjSel sel = sj_select(r0, r1, num_cols, rowptr, colind, kcap, long_thr,
                                nnz, s_bits, s_pre, &s_sel);
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

// The long rows keep their row order (neighbours in x share the staged panels)
// and are sorted by length inside runs of 2^6 = one workgroup's 64 rows.
// Measured on the 1 % tail (same box): runs of 16 / 32 / 64 / 128 / 256 / 1024
// rows 0.45 / 0.44 / 0.44 / 0.48 / 0.58 / 1.24 ms -- what the longer runs gain in
// waves that end together they lose several times over in panels (the rows of
// a workgroup are no longer neighbours); 4-wave workgroups 0.50-0.65.
// The table-driven kernel (csr_sjds_longt_kernel) takes supergroups of 64 RS
// rows, an 8-lane group RS of them (RS = 1: see the measurements there): the
// runs are its supergroups.
#ifndef SJ_LT_RS
#define SJ_LT_RS 1
#endif
#ifndef SJ_LT_G
#define SJ_LT_G 8
#endif
#define SJ_LT_G_ SJ_LT_G
constexpr int kSjLtRS = SJ_LT_RS;        // rows per group and supergroup
constexpr int kSjLtRun = 512 / SJ_LT_G_ * kSjLtRS; // rows per supergroup
static_assert(kSjLtRun == 64 || kSjLtRun == 128 || kSjLtRun == 256 || kSjLtRun == 512,
              "runs of 64 ... 512 rows");
#ifndef SJ_LONG_RUN_SHIFT
#define SJ_LONG_RUN_SHIFT (kSjLtRun == 64 ? 6 : kSjLtRun == 128 ? 7 : kSjLtRun == 256 ? 8 : 9)
#endif
constexpr int kSjLongRunShift = SJ_LONG_RUN_SHIFT;

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

// ---------------------------------------------------------------------------
// the SpMV kernel
// ---------------------------------------------------------------------------
// TV: the type of the stored values (fp32 under fp64 vectors and arithmetic:
// the mixed-precision SpMV)
template <typename T, typename TV = T>
struct SjArgs {
  int32_t num_rows, num_cols;
  int32_t nblk;
  int32_t maxk;      // staged chunks the LDS buffer holds
  int32_t stride;    // chunk-list entries per block
  int32_t wide_alloc;
  const int32_t* rowptr;  // the caller's (long rows)
  const uint32_t* ubase;  // first unit of every slice
  const int32_t* lenperm;
  const int32_t* blk;     // per block: chunks, wide
  const int32_t* chunks;
  const unsigned char* codes;
  const TV* val;          // jagged order
  // long rows (phase 0): straight from the caller's CSR arrays
  int32_t phases; // measurement only (plan_set "sj_phases"): 1 = long rows, 2 = slices
  int32_t nlong;
  int32_t long_sorted; // every long row's columns ascend: x by panels
  int32_t long_panel;  // ... of this many columns
  const int32_t* long_rows;
  const int32_t* colind;
  const TV* values;
  // the table-driven long-row kernel (csr_sjds_longt_kernel)
  const int32_t* lt_cmin;
  const int32_t* lt_np;
  const int64_t* lt_off;
  const int32_t* lt_tab;
  const uint16_t* lt_codes; // per entry of the listed rows: column - its panel's first
  const int64_t* lt_coff;   // per listed row: its first code
};

template <typename T>
__device__ __forceinline__ T sj_readlane(T v, int j);
template <>
__device__ __forceinline__ double sj_readlane<double>(double v, int j)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float sj_readlane<float>(float v, int j)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// One slice.  vs / cs: the slice's first unit of values / codes (uniform);
// mylen = this lane's row length in entries (lanes sorted by it, descending).
// Returns the lane's row sum.
//
// Every load is UNCONDITIONAL (an inactive lane re-reads the step's first
// unit; a step past the slice's end reads into the arrays' slack) and every
// use a select: with branches around the loads the compiler waits for ALL
// loads in flight at each join, and the two groups no longer overlap.  The
// step's address is the uniform `vs + off` plus the lane's own constant: no
// vector arithmetic per step but one select.
//
// MODE 0 = a general matrix.  MODE 3 = symmetric storage (csr_kernels.cpp:26-40
// seen from the row): the lane's "row" is the row's nlow stored (lower) entries
// followed by the entries of its COLUMN in the reference's order (ascending
// (r, j)); the sum starts at d_i x_i and takes the lower products as they are;
// where the column's entries begin it TURNS into y_i = fl(alpha sum + beta y0_i)
// and every further product is fl(fl(alpha v) x) -- the reference's
// out[col] += alpha * val * in[i].  Returns y_i.  (No wave take-over of a long
// row there: the turn would fall inside it.)
template <typename T, typename TV, typename CODE, bool WIDE, int E, int MODE>
__device__ __forceinline__ T sj_slice(const SjUnit<TV, E>* __restrict__ vs,
                                      const SjUnit<CODE, E>* __restrict__ cs,
                                      int32_t mylen, int lane,
                                      const T* __restrict__ s_x,
                                      const T* __restrict__ in, T init, T alpha,
                                      int32_t nlow = 0, T by0 = T(0),
                                      bool has_beta = false)
{
  constexpr int U = kSjGroup / E; // steps per group
  const int32_t myu = (mylen + E - 1) / E;
  const int32_t maxu = __builtin_amdgcn_readlane(myu, 0);
  const int32_t u1 = __builtin_amdgcn_readlane(myu, 1);
  const int32_t maxlen = __builtin_amdgcn_readlane(mylen, 0);
  // the jagged part ends at the second-longest row when lane 0's row goes on
  // for long enough to be worth taking over by the whole wave
  const int32_t kmain
      = (MODE == 0 && maxlen - u1 * E >= kSjTailMin) ? u1 : maxu; // units
  const int32_t mymain = myu < kmain ? myu : kmain;
  T sum = init;
  SjUnit<TV, E> va[U], vc[U];
  SjUnit<CODE, E> ca[U], cc[U];
  uint32_t off = 0; // units behind the slice's first (uniform)

#define SJ_ISSUE(V, C, K0)                                                     \
  _Pragma("unroll") for (int u = 0; u < U; ++u)                                \
  {                                                                            \
    const bool act = (K0) + u < mymain;                                        \
    const int cnt = __popcll(__ballot(act));                                   \
    const int li = act ? lane : 0;                                             \
    V[u] = (vs + off)[li];                                                     \
    C[u] = (cs + off)[li];                                                     \
    off += cnt;                                                                \
  }
#define SJ_CONSUME(V, C, K0)                                                   \
  {                                                                            \
    T xs[U][E];                                                                \
    if constexpr (WIDE) {                                                      \
      T xg[U][E], xl[U][E];                                                    \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                            \
          _Pragma("unroll") for (int q = 0; q < E; ++q)                        \
      {                                                                        \
        const uint32_t c = (uint32_t)C[u].e[q];                                \
        const bool far = (c >> 31) != 0 && ((K0) + u) * E + q < mylen;         \
        xg[u][q] = in[far ? (c & 0x7fffffffu) : 0u];                           \
      }                                                                        \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                            \
          _Pragma("unroll") for (int q = 0; q < E; ++q)                        \
      {                                                                        \
        const uint32_t c = (uint32_t)C[u].e[q];                                \
        xl[u][q] = s_x[(c >> 31) ? 0u : c];                                    \
      }                                                                        \
      /* all LDS reads are wanted whatever the codes say: without this the  \
         compiler moves each read under its own test, and every join waits  \
         for everything in flight */                                        \
      static_assert(U * E == 8, "eight operands below");                       \
      asm volatile("" ::"v"(xl[0][0]), "v"(xl[(1 / E) % U][1 % E]),            \
                   "v"(xl[(2 / E) % U][2 % E]), "v"(xl[(3 / E) % U][3 % E]),   \
                   "v"(xl[(4 / E) % U][4 % E]), "v"(xl[(5 / E) % U][5 % E]),   \
                   "v"(xl[(6 / E) % U][6 % E]), "v"(xl[(7 / E) % U][7 % E]));  \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                            \
          _Pragma("unroll") for (int q = 0; q < E; ++q)                        \
      {                                                                        \
        const uint32_t c = (uint32_t)C[u].e[q];                                \
        xs[u][q] = (c >> 31) ? xg[u][q] : xl[u][q];                            \
      }                                                                        \
    } else {                                                                   \
      _Pragma("unroll") for (int u = 0; u < U; ++u)                            \
          _Pragma("unroll") for (int q = 0; q < E; ++q) xs[u][q]               \
          = s_x[C[u].e[q]];                                                    \
    }                                                                          \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                              \
        _Pragma("unroll") for (int q = 0; q < E; ++q)                          \
    {                                                                          \
      const int32_t kk = ((K0) + u) * E + q;                                   \
      T vq = (T)V[u].e[q];                                                     \
      if constexpr (MODE == 3) {                                               \
        const T as = alpha * sum;                                              \
        const T turned = has_beta ? as + by0 : as;                             \
        sum = (kk == nlow && kk < mylen) ? turned : sum;                       \
        vq = kk >= nlow ? alpha * vq : vq;                                     \
      }                                                                        \
      const T nxt = sum + vq * xs[u][q];                                       \
      const bool use = (K0) + u < mymain && kk < mylen;                        \
      sum = use ? nxt : sum;                                                   \
    }                                                                          \
  }

  if (maxu > 0) {
    SJ_ISSUE(va, ca, 0)
    int32_t k = 0;
    for (; k + 2 * U < kmain; k += 2 * U) {
      SJ_ISSUE(vc, cc, k + U)
      SJ_CONSUME(va, ca, k)
      SJ_ISSUE(va, ca, k + 2 * U)
      SJ_CONSUME(vc, cc, k + U)
    }
    SJ_ISSUE(vc, cc, k + U)
    SJ_CONSUME(va, ca, k)
    SJ_CONSUME(vc, cc, k + U)
  }
#undef SJ_ISSUE
#undef SJ_CONSUME

  if (kmain < maxu) {
    // lane 0's row goes on alone: its remaining entries are contiguous.  The
    // wave loads and multiplies 64 at a time; the products are added to the
    // row's sum one by one, in order (every lane computes the same sum).
    const int32_t rem = maxlen - kmain * E;
    const TV* vt = reinterpret_cast<const TV*>(vs + off);
    const CODE* ct = reinterpret_cast<const CODE*>(cs + off);
    T t = sj_readlane<T>(sum, 0);
    for (int32_t j0 = 0; j0 < rem; j0 += 64) {
      const int32_t j = j0 + lane < rem ? j0 + lane : 0;
      const T v = (T)vt[j];
      T x;
      if constexpr (WIDE) {
        const uint32_t c = (uint32_t)ct[j];
        const T xg = in[(c >> 31) ? (c & 0x7fffffffu) : 0u];
        const T xl = s_x[(c >> 31) ? 0u : c];
        x = (c >> 31) ? xg : xl;
      } else {
        x = s_x[ct[j]];
      }
      const T p = v * x; // lanes past the row's end: never added
      const int n = rem - j0 < 64 ? rem - j0 : 64;
      if (n == 64) {
#pragma unroll
        for (int q = 0; q < 64; ++q)
          t += sj_readlane<T>(p, q);
      } else {
        for (int q = 0; q < n; ++q)
          t += sj_readlane<T>(p, q);
      }
    }
    if (lane == 0)
      sum = t;
  }
  if constexpr (MODE == 3) { // a row without column entries turns at its end
    const T as = alpha * sum;
    const T turned = has_beta ? as + by0 : as;
    sum = nlow == mylen ? turned : sum;
  }
  return sum;
}

// EIGHT long rows by one wave, eight lanes each: lane l of a group reads entries
// 8 s + l of its row (64 + 32 bytes per row and step, straight from the
// caller's CSR arrays), the group's eight products are added to the row's sum
// one by one in the row's order (every lane of the group keeps the sum).
// Values and columns travel two load groups ahead, x one.  [a, b) = the lane's
// row (b == a: no row); returns the row's sum.
constexpr int kSjLpr = 8;
#ifndef SJ_PANEL_U
#define SJ_PANEL_U 4
#endif
constexpr int kSjPanelU = SJ_PANEL_U; // steps per trip of the panel walk
#ifndef SJ_LONG_SETS
#define SJ_LONG_SETS 1
#endif
// sets of eight rows per wave that share the staged panels.  Measured on the
// 1 % tail (same box, alternating builds): 1 / 2 / 4 sets 0.44 / 0.48 / 0.57 ms --
// half the x staged per entry does not pay for the longer walk per panel (121
// / 146 registers): the kernel is bound by the latency of its trips, not by
// the panels' bytes
constexpr int kSjLongSets = SJ_LONG_SETS;
constexpr int kSjLU = 4; // steps per load group of the long-row phase
template <typename T, typename TV>
__device__ __forceinline__ T sj_long_rows8(const TV* __restrict__ val,
                                          const int32_t* __restrict__ col,
                                          int64_t a, int64_t b, int lane,
                                          const T* __restrict__ in)
{
  const int l = lane & (kSjLpr - 1);
  const int32_t len = (int32_t)(b - a);
  int32_t maxlen = len; // over the wave's eight rows
#pragma unroll
  for (int o = 32; o >= kSjLpr; o >>= 1) {
    const int32_t other = __shfl_xor(maxlen, o, 64);
    maxlen = other > maxlen ? other : maxlen;
  }
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  const int64_t last = b > a ? b - 1 : a; // (no row: a valid address all the same)
  auto at = [&](int32_t s) {
    const int64_t e = a + (int64_t)s * kSjLpr + l;
    return e < last ? e : last;
  };
  constexpr int U = kSjLU;
  T vA[U], vB[U], vC[U], xA[U], xB[U];
  int32_t cB[U], cC[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    vA[u] = val[at(u)];
    cB[u] = col[at(u)]; // (group 0's columns, used at once)
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    xA[u] = in[cB[u]];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    vB[u] = val[at(U + u)];
    cB[u] = col[at(U + u)];
  }
  T t = T(0);
  for (int32_t s = 0; s * kSjLpr < maxlen; s += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vC[u] = val[at(s + 2 * U + u)];
      cC[u] = col[at(s + 2 * U + u)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      xB[u] = in[cB[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T p = vA[u] * xA[u];
      T pj[kSjLpr];
#pragma unroll
      for (int j = 0; j < kSjLpr; ++j)
        pj[j] = __shfl(p, j, kSjLpr);
#pragma unroll
      for (int j = 0; j < kSjLpr; ++j) {
        const T nxt = t + pj[j];
        t = (s + u) * kSjLpr + j < len ? nxt : t;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vA[u] = vB[u];
      xA[u] = xB[u];
      vB[u] = vC[u];
      cB[u] = cC[u];
    }
  }
  return t;
}

// The LONG rows (a launch of its own behind the slices' kernel: its registers
// are its own).  WPB waves per workgroup, eight rows per wave.
template <typename T, typename TV, int WPB, bool DOT, bool PANELS>
__global__ __launch_bounds__(64 * WPB) void csr_sjds_long_kernel(
    SjArgs<T, TV> A, T alpha, const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot, int dot_slot0)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  T* s_x = reinterpret_cast<T*>(s_raw);
  __shared__ double s_red[WPB];
  constexpr int NT = 64 * WPB;
  typedef T pair_t __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double dot_acc = 0.0;
  // phase 0: the long rows, eight per wave (eight lanes each).  The list is in
  // row order (sorted by length inside runs of 64): a workgroup takes a
  // contiguous run of SUPERGROUPS of 8 WPB rows, workgroups of one XCD
  // neighbouring runs.
  //
  // PANELS (the plan found every long row's columns ascending): a long row's
  // entries sit one per cache line over a window far wider than a slice's --
  // gathered from memory each entry drags a line of x through the L2 (110 M
  // lines for the 1 % tail of the benchmark's matrix: 1.0 ms, twice the rest
  // of the product).  The rows of a supergroup are neighbours, their windows
  // overlap: the workgroup walks the columns they span in panels of x that
  // fit the LDS buffer, stages each panel once with coalesced loads, and every
  // row adds the products of ITS entries inside the panel -- ascending
  // columns, so the row's own order, the reference's bits.
  {
    __shared__ int32_t s_cmin, s_cmax;
    __shared__ __attribute__((aligned(16))) T s_scr[WPB * 64]; // per wave: products
    // RS sets of eight rows per wave: a supergroup is 8 WPB RS rows that share
    // the staged panels (kSjLongSets)
    constexpr int RS = kSjLongSets;
    const int nitems = (A.nlong + 7) / 8;
    const int nsg = (nitems + WPB * RS - 1) / (WPB * RS);
    const int g8 = gridDim.x >= 8 && (gridDim.x & 7) == 0;
    const int chunk = g8 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)
                         : blockIdx.x;
    // (balanced contiguous runs: the first nsg mod grid workgroups take one more)
    const int per = nsg / gridDim.x, rem = nsg % gridDim.x;
    const int sg0 = chunk * per + min(chunk, rem);
    const int sg1 = sg0 + per + (chunk < rem ? 1 : 0);
    const int panel = A.long_panel; // columns of x the LDS buffer holds
    for (int sg = sg0; sg < sg1; ++sg) { // uniform per workgroup
      bool have_row[RS];
      int32_t row[RS];
      int64_t ra[RS], rb[RS];
      T sum[RS];
#pragma unroll
      for (int h = 0; h < RS; ++h) {
        const int item = (sg * RS + h) * WPB + wave;
        const int g = item * 8 + (lane >> 3);
        have_row[h] = g < A.nlong;
        row[h] = A.long_rows[have_row[h] ? g : A.nlong - 1];
        ra[h] = A.rowptr[row[h]];
        rb[h] = have_row[h] ? (int64_t)A.rowptr[row[h] + 1] : ra[h];
        sum[h] = T(0);
      }
      bool by_panels = PANELS; // (an instantiation per path: the two together
                               //  need 180 registers)
      int32_t cmin = 0, cmax = -1;
      if constexpr (PANELS) { // (uniform) the columns the supergroup spans
        if (t == 0) {
          s_cmin = INT32_MAX;
          s_cmax = -1;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < RS; ++h)
          if (have_row[h] && rb[h] > ra[h] && (lane & 7) == 0) {
            atomicMin(&s_cmin, A.colind[ra[h]]);
            atomicMax(&s_cmax, A.colind[rb[h] - 1]);
          }
        __syncthreads();
        cmin = s_cmin & ~(kSjChunk - 1);
        cmax = s_cmax;
        __syncthreads();
        // a span of more than 64 panels: the rows are not neighbours in x
        by_panels = cmax >= cmin && (int64_t)cmax - cmin < (int64_t)64 * panel;
      }
      if constexpr (!PANELS) {
#pragma unroll
        for (int h = 0; h < RS; ++h)
          sum[h] = sj_long_rows8<T, TV>(A.values, A.colind, ra[h], rb[h], lane, in);
      } else if (!by_panels) { // rows that are not neighbours in x: rare, slow
#pragma unroll
        for (int h = 0; h < RS; ++h)
          if ((lane & 7) == 0)
            for (int64_t i = ra[h]; i < rb[h]; ++i)
              sum[h] += A.values[i] * in[A.colind[i]];
      } else {
        const int l = lane & 7;
        int64_t e[RS]; // the row's first entry not yet added (same in its 8 lanes)
#pragma unroll
        for (int h = 0; h < RS; ++h)
          e[h] = ra[h];
        for (int64_t p0 = cmin; p0 <= cmax; p0 += panel) {
          // stage x[p0, p0 + panel): 2 elements per lane and round
          const int64_t cend = (int64_t)A.num_cols;
          const int64_t clast = (cend - 2) & ~(int64_t)1;
          for (int64_t q = 2 * t; q < panel; q += 2 * NT) {
            const int64_t col = p0 + q;
            pair_t xv = *reinterpret_cast<const pair_t*>(
                in + (col < clast ? col : clast));
            if (col + 1 == cend)
              xv[0] = in[col];
            *reinterpret_cast<pair_t*>(&s_x[q]) = xv;
          }
          __syncthreads();
          const int64_t pend = p0 + panel;
          // the row's entries below pend, four steps of eight per trip: the
          // loads assume whole steps (a step the panel's end cuts short ends
          // the trip early; what was loaded past it is loaded again with the
          // next panel)
          // (the NEXT trip's loads are issued before this trip's sums, assuming
          // it ends whole; a trip the panel's end cuts short drops them)
          constexpr int U = kSjPanelU;
          // (no clamps: a long row ends at least kSjLongPad entries before the
          // arrays do -- sj_is_long -- and what lies past its end is never used;
          // one address per stream and trip, the steps at immediate offsets)
          static_assert(2 * U * 8 + 8 <= kSjLongPad, "loads past a row's end");
#pragma unroll
          for (int h = 0; h < RS; ++h) {
            T v[U], vn[U];
            int32_t c[U], cn[U];
            int64_t eh = e[h];
            const int64_t rbh = rb[h];
            T acc = sum[h];
            {
              const TV* vp = A.values + eh + l;
              const int32_t* cp = A.colind + eh + l;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                v[u] = vp[u * 8];
                c[u] = cp[u * 8];
              }
            }
            bool more = true;
            while (__any(more)) {
              {
                const TV* vp = A.values + eh + l + U * 8;
                const int32_t* cp = A.colind + eh + l + U * 8;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  vn[u] = vp[u * 8];
                  cn[u] = cp[u * 8];
                }
              }
              bool open = more; // this group's steps so far were whole
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int64_t i = eh + l; // (eh advances with the steps)
                const bool ok = open && i < rbh && c[u] < pend;
                const T x = s_x[ok ? (int32_t)(c[u] - p0) : 0];
                // a lane without an entry contributes +0.0: the sum starts at
                // +0.0 and can never become -0.0, so adding it changes no bit
                const T pr = ok ? v[u] * x : T(0);
                // valid lanes are a prefix of the group: ascending columns
                const uint64_t bal = __ballot(ok);
                const int nv = __popcll((bal >> (lane & ~7)) & 0xFFull);
                // the group's eight products through the wave's LDS scratch (one
                // store, four broadcast loads) and onto the sum one by one
                T* scr = s_scr + wave * 64;
                scr[lane] = pr;
                typedef T vec2 __attribute__((ext_vector_type(2)));
                const vec2* gp = reinterpret_cast<const vec2*>(scr + (lane & ~7));
                const vec2 q0 = gp[0], q1 = gp[1], q2 = gp[2], q3 = gp[3];
                acc += q0[0];
                acc += q0[1];
                acc += q1[0];
                acc += q1[1];
                acc += q2[0];
                acc += q2[1];
                acc += q3[0];
                acc += q3[1];
                eh += nv;
                open = open && nv == 8;
              }
              more = open;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                v[u] = vn[u];
                c[u] = cn[u];
              }
            }
            e[h] = eh;
            sum[h] = acc;
          }
          __syncthreads(); // everybody is done with this panel
        }
      }
#pragma unroll
      for (int h = 0; h < RS; ++h)
        if (have_row[h] && (lane & 7) == 0) {
          const T c = alpha * sum[h];
          T y = c;
          if (beta != T(0))
            y = c + beta * out[row[h]];
          out[row[h]] = y;
          if constexpr (DOT)
            dot_acc += (double)in[row[h]] * (double)c;
        }
    }
  }
  if constexpr (DOT) {
    double v = dot_acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      v += __shfl_down(v, o, 64);
    if (lane == 0)
      s_red[wave] = v;
    __syncthreads();
    if (t == 0) {
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < WPB; ++w)
        r += s_red[w];
      dot.partials[dot_slot0 + blockIdx.x] = r; // behind the slices' partials
    }
  }
}


// ---------------------------------------------------------------------------
// The LONG rows whose columns ascend, table-driven ("marched" through panels).
//
// What the kernel above spends its time on (ISA + timings): the panel's staging
// loop ran one load at a time (load, wait, LDS write: the panel size was a
// run-time number and the loop not unrolled); every step of eight entries took
// a trip through LDS for the products (one store, four 16-byte broadcast reads
// per lane) on top of the read of x; and a step the panel's end cut short
// dropped what had been loaded past it.  Here
//
//   * the plan knows where every row crosses every panel boundary
//     (sj_lt_fill_kernel: one bisection per row and boundary), so a row's
//     range inside a panel is known before anything is loaded: loads run a
//     trip ahead whatever the columns are, nothing is loaded twice, and `ok`
//     is an index comparison; with the panels fixed at plan time the plan also
//     keeps every entry's column as a 16-bit position inside its panel (2
//     instead of 4 bytes per entry streamed, and no subtraction);
//   * a lane loads FOUR consecutive entries of the trip's 32 (16-byte loads:
//     one 256-byte piece of the values per row instead of four 64-byte ones);
//     the row's sum travels down the group's eight lanes by DPP (row_shr:1):
//     in round r lane r adds its four products to the sum it was handed, in
//     entry order -- the reference's bits (csr_kernels.cpp:41-51), no LDS, no
//     broadcasts (every lane executes all 32 additions; only the one holding
//     the true sum matters);
//   * the panel (a compile-time size) is requested in one go -- eight 16-byte
//     loads per lane in flight -- before the barrier that frees the buffer.
//
// Measured on the 1 % tail of the benchmark's matrix (110 M entries in 100 k
// rows; same box, alternating builds; the older kernel 0.447 ms):
//   rows per 8-lane group, walked one after the other inside a panel and dealt
//   in serpentine order (SJ_LT_RS)             1 / 2 / 4: 0.352 / 0.375 / 0.534 ms
//   lanes per row x entries per lane (SJ_LT_G x SJ_LT_EPL)
//                      8 x 4 / 8 x 8 / 4 x 8 / 4 x 4 / 2 x 8: 0.343 / 0.356 /
//                                                        0.415 / 0.414 / 0.69
//   trips of loads in flight (SJ_LT_DEPTH)                   1 / 2: 0.375 / 0.376
//   supergroups from an atomic queue instead of static runs: 0.367 against 0.355
//   the matrix loaded non-temporally (nt):                   0.458 against 0.352
//   an XCD's workgroups on consecutive supergroups (interleaved) instead of
//   contiguous runs per workgroup:                    0.331-0.339 against 0.343-0.351
//   the columns as 16-bit panel positions (the plan's own array, SJ_LT_CODES)
//   instead of the caller's colind:                          0.294 against 0.328
// -- whatever makes a supergroup wider (more rows: more panels, more staged x)
// loses.  A build with clocks in it (SJ_LT_PROBE) shows where the time goes:
// 1.03 us per trip of a wave whether its neighbours are busy or idle, 0.34 us
// with the loads of the matrix taken out (SJ_LT_PROBE_NOLOAD; without the
// additions or without the LDS reads: within 10 %); workgroups with three and
// with four supergroups end together.  The kernel is bound by the stream of
// the matrix: 1.32 GB of values and columns + 0.5 GB of panels in 0.33-0.35 ms.
// ---------------------------------------------------------------------------
constexpr int kSjLtMaxPanels = 64;       // wider supergroups: rows one by one
#ifndef SJ_LT_PANEL_COLS
#define SJ_LT_PANEL_COLS 8192
#endif
// columns of x per panel: 64 KiB of fp64, two workgroups per CU; a multiple of
// the 1024 columns one round of the workgroup's 16-byte loads stages
constexpr int kSjLtPanel = SJ_LT_PANEL_COLS;
#ifndef SJ_LT_G
#define SJ_LT_G 8
#endif
#ifndef SJ_LT_EPL
#define SJ_LT_EPL 4
#endif
constexpr int kSjLtG = SJ_LT_G;           // lanes per row (a power of two <= 16)
constexpr int kSjLtEpl = SJ_LT_EPL;       // entries per lane and trip (4 or 8)
#ifndef SJ_LT_DEPTH
#define SJ_LT_DEPTH 1
#endif
constexpr int kSjLtDepth = SJ_LT_DEPTH;   // trips of loads in flight ahead
constexpr int kSjLtTrip = kSjLtG * kSjLtEpl; // ... per group and trip
#ifndef SJ_LT_CODES
#define SJ_LT_CODES 1
#endif
// the long rows' columns as 16-bit positions inside their panel (the plan's own
// array, 2 B per entry) instead of the caller's 4-byte colind: 10 instead of 12
// bytes per entry streamed
constexpr bool kSjLtCodes = SJ_LT_CODES != 0;
static_assert(kSjLtPanel <= 65536, "16-bit panel positions");
static_assert(kSjLtTrip * (kSjLtDepth + 1) + 8 <= kSjLongPad, "loads past a row's end");
static_assert(kSjLtPanel % 1024 == 0, "whole staging rounds");

template <typename X, int N>
struct __attribute__((packed, aligned(sizeof(X)))) SjPack {
  X e[N];
};

// DPP move within rows of 16 lanes: lanes without a source keep `old`
template <int CTRL>
__device__ __forceinline__ double sj_dpp(double old, double src)
{
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src),
                                             CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src),
                                             CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float sj_dpp(float old, float src)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xF, false));
}
constexpr int kDppRowShr1 = 0x111; // lane l <- lane l - 1
constexpr int kDppRowShlBack = 0x100 + kSjLtG - 1; // lane l <- lane l + G - 1
static_assert(kSjLtG == 2 || kSjLtG == 4 || kSjLtG == 8 || kSjLtG == 16, "DPP rows");

template <typename T, typename TV, bool DOT>
__global__ __launch_bounds__(512, 4) void csr_sjds_longt_kernel(
    SjArgs<T, TV> A, T alpha, const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot, int dot_slot0)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  T* s_x = reinterpret_cast<T*>(s_raw);
  __shared__ double s_red[8];
  constexpr int NT = 512, RS = kSjLtRS, RUN = kSjLtRun, PANEL = kSjLtPanel;
  constexpr int EPL = kSjLtEpl, TRIP = kSjLtTrip;
  constexpr int NST = PANEL / 2 / NT; // staging loads per lane
  typedef T pair_t __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int G = kSjLtG, NG = 512 / G; // lanes per row, groups per workgroup
  const int l = lane & (G - 1), g = wave * (64 / G) + lane / G;
  double dot_acc = 0.0;
  const int nsg = (A.nlong + RUN - 1) / RUN;
#ifndef SJ_LT_INTERLEAVE
#define SJ_LT_INTERLEAVE 1
#endif
  // An XCD (blockIdx mod 8: its own L2) takes a contiguous eighth of the
  // supergroups, and its G workgroups take them INTERLEAVED (j, j + G, ...): at
  // any time they walk G consecutive supergroups, whose panels of x overlap
  // (neighbours shift by 64 long rows' worth of columns) and meet in that L2.
  // (Contiguous runs per workgroup, SJ_LT_INTERLEAVE = 0: a workgroup's next
  // supergroup finds its predecessor's panels evicted by the matrix stream.)
  const int g8 = gridDim.x >= 8 && (gridDim.x & 7) == 0;
  int sg0, sg1, sgstep;
  if (g8 && SJ_LT_INTERLEAVE) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int per8 = nsg / 8, rem8 = nsg % 8;
    const int lo = x * per8 + min(x, rem8);
    sg0 = lo + j;
    sg1 = lo + per8 + (x < rem8 ? 1 : 0);
    sgstep = G;
  } else {
    const int chunk = g8 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)
                         : blockIdx.x;
    const int per = nsg / gridDim.x, rem = nsg % gridDim.x;
    sg0 = chunk * per + min(chunk, rem);
    sg1 = sg0 + per + (chunk < rem ? 1 : 0);
    sgstep = 1;
  }
  const int64_t cend = (int64_t)A.num_cols;
#ifdef SJ_LT_PROBE
  long long pr_stage = 0, pr_loop = 0, pr_trips = 0, pr_panels = 0;
  const long long pr_begin = wall_clock64();
#endif
  for (int sg = sg0; sg < sg1; sg += sgstep) { // uniform per workgroup
    // the group's rows: rank g of the supergroup (RS > 1: g, 2 NG - 1 - g, ...)
    int32_t slot[RS], row[RS];
    bool have[RS];
    T acc[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) {
      slot[j] = NG * j + ((j & 1) ? NG - 1 - g : g);
      const int li = sg * RUN + slot[j];
      have[j] = li < A.nlong;
      row[j] = A.long_rows[have[j] ? li : A.nlong - 1];
      acc[j] = T(0);
    }
    const int np = A.lt_np[sg];
    if (np == 0) { // a supergroup whose rows are not neighbours in x: rare, slow
      if (l == 0) {
#pragma unroll
        for (int j = 0; j < RS; ++j)
          if (have[j]) {
            const int64_t a = A.rowptr[row[j]], b = A.rowptr[row[j] + 1];
            T sum = T(0);
            for (int64_t i = a; i < b; ++i)
              sum += A.values[i] * in[A.colind[i]];
            acc[j] = sum;
          }
      }
    } else {
      const int32_t cmin = A.lt_cmin[sg];
      const int32_t* tab = A.lt_tab + A.lt_off[sg];
      int32_t lo[RS], hi[RS];
      int64_t cb[RS]; // the row's codes: entry e's is at lt_codes[cb + e]
#pragma unroll
      for (int j = 0; j < RS; ++j) {
        lo[j] = tab[slot[j]];
        hi[j] = tab[RUN + slot[j]];
        cb[j] = 0;
        if constexpr (kSjLtCodes) {
          const int li = sg * RUN + slot[j];
          cb[j] = A.lt_coff[li < A.nlong ? li : A.nlong - 1] - lo[j];
        }
      }
      for (int p = 0; p < np; ++p) {
        const int32_t p0 = cmin + p * PANEL; // (<= the supergroup's last column)
        // the boundary behind the next panel: back by the time it is needed
        int32_t hin[RS];
        {
          const int pn = p + 2 <= np ? p + 2 : np;
#pragma unroll
          for (int j = 0; j < RS; ++j)
            hin[j] = tab[(int64_t)pn * RUN + slot[j]];
        }
        // the group's trips in this panel: row 0's range, then row 1's, ...
        // state = (row j, first entry pos, the range's end); j == RS: done
        auto settle = [&](int& j, int32_t& pos, int32_t& end) {
#pragma unroll
          for (int q = 0; q < RS; ++q) {
            const bool ex = pos >= end && j < RS;
            j += ex ? 1 : 0;
            int32_t nl = 0, nh = 0;
#pragma unroll
            for (int r = 1; r < RS; ++r) {
              nl = j == r ? lo[r] : nl;
              nh = j == r ? hi[r] : nh;
            }
            pos = ex ? nl : pos;
            end = ex ? nh : end;
          }
        };
        // D trips of loads in flight ahead of the one being consumed: a ring of
        // D + 1 register sets, the loop unrolled over it
        constexpr int D = kSjLtDepth;
        typedef typename std::conditional<kSjLtCodes, uint16_t, int32_t>::type code_t;
        SjPack<TV, EPL> vv[D + 1];
        SjPack<code_t, EPL> cc[D + 1];
        auto issue = [&](SjPack<TV, EPL>& v, SjPack<code_t, EPL>& c, int32_t pos,
                         int j) {
          // (no clamp: a long row ends kSjLongPad entries before the arrays do;
          // a finished group reads entries 0 ...)
#ifdef SJ_LT_PROBE_NOLOAD
          const int64_t e = EPL * l + (pos & 1);
#else
          const int64_t e = (int64_t)pos + EPL * l;
#endif
          v = *reinterpret_cast<const SjPack<TV, EPL>*>(A.values + e);
          if constexpr (kSjLtCodes) {
            int64_t base = cb[0];
#pragma unroll
            for (int r = 1; r < RS; ++r)
              base = j == r ? cb[r] : base;
            // (a finished group reads the array's first codes)
            const int64_t ce = j < RS ? base + e : (int64_t)EPL * l;
            c = *reinterpret_cast<const SjPack<code_t, EPL>*>(
                reinterpret_cast<const code_t*>(A.lt_codes) + ce);
          } else {
            c = *reinterpret_cast<const SjPack<code_t, EPL>*>(
                reinterpret_cast<const code_t*>(A.colind) + e);
          }
        };
        auto consume = [&](const SjPack<TV, EPL>& v, const SjPack<code_t, EPL>& c,
                           int j, int32_t pos, int32_t end) {
          T pr[EPL], xs[EPL];
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            const bool ok = pos + EPL * l + k < end;
            const int32_t xi = kSjLtCodes ? (int32_t)c.e[k] : (int32_t)c.e[k] - p0;
#ifdef SJ_LT_PROBE_NOLDS
            xs[k] = (T)(ok ? xi : 0);
#else
            xs[k] = s_x[ok ? xi : 0];
#endif
          }
          // (every LDS read is wanted whatever `ok` says: left to itself the
          // compiler moves each read under its own test, and every join waits
          // for everything in flight)
          static_assert(EPL == 4 || EPL == 8, "the operands below");
          asm volatile("" ::"v"(xs[0]), "v"(xs[1]), "v"(xs[2]), "v"(xs[3]));
          if constexpr (EPL == 8)
            asm volatile("" ::"v"(xs[4 % EPL]), "v"(xs[5 % EPL]), "v"(xs[6 % EPL]),
                         "v"(xs[7 % EPL]));
#pragma unroll
          for (int k = 0; k < EPL; ++k) {
            const bool ok = pos + EPL * l + k < end;
            // a lane without an entry contributes +0.0: the sum starts at +0.0
            // and can never become -0.0, so adding it changes no bit
            const T prod = (T)v.e[k] * xs[k];
            pr[k] = ok ? prod : T(0);
          }
          T tsum = acc[0];
#pragma unroll
          for (int r = 1; r < RS; ++r)
            tsum = j == r ? acc[r] : tsum;
          // the sum walks down the group's lanes: in round r lane r holds it
#ifdef SJ_LT_PROBE_NOCHAIN
          tsum += (pr[0] + pr[1]) + (pr[2] + pr[3]);
#else
#pragma unroll
          for (int r = 0; r < G; ++r) {
            T s = tsum;
#pragma unroll
            for (int k = 0; k < EPL; ++k)
              s += pr[k];
            tsum = r < G - 1 ? sj_dpp<kDppRowShr1>(s, s) : sj_dpp<kDppRowShlBack>(s, s);
          }
#endif
          // (lane 0 of the group has it; the others' copies are never used)
#pragma unroll
          for (int r = 0; r < RS; ++r)
            acc[r] = j == r ? tsum : acc[r];
        };
        int jq[D + 1];
        int32_t posq[D + 1], endq[D + 1];
        jq[0] = 0, posq[0] = lo[0], endq[0] = hi[0];
        settle(jq[0], posq[0], endq[0]);
#pragma unroll
        for (int d = 1; d <= D; ++d) {
          jq[d] = jq[d - 1], posq[d] = posq[d - 1] + TRIP, endq[d] = endq[d - 1];
          settle(jq[d], posq[d], endq[d]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
          issue(vv[d], cc[d], posq[d], jq[d]);
#ifdef SJ_LT_PROBE
        const long long pc0 = wall_clock64();
#endif
        // the panel: every load in flight before the barrier that frees the buffer
        pair_t xv[NST];
#pragma unroll
        for (int m = 0; m < NST; ++m) {
          const int q = 2 * t + 2 * NT * m;
          const int64_t col = (int64_t)p0 + q;
          // (an odd number of columns: the last one comes as the second element
          // of the pair in front of it -- no branch around a load)
          const SjPack<T, 2> ld = *reinterpret_cast<const SjPack<T, 2>*>(
              in + (col < cend - 2 ? col : cend - 2));
          xv[m][0] = col == cend - 1 ? ld.e[1] : ld.e[0];
          xv[m][1] = ld.e[1];
        }
        __syncthreads(); // everybody is done with the previous panel
#pragma unroll
        for (int m = 0; m < NST; ++m)
          *reinterpret_cast<pair_t*>(&s_x[2 * t + 2 * NT * m]) = xv[m];
        __syncthreads();
#ifdef SJ_LT_PROBE
        const long long pc1 = wall_clock64();
        int ptrips = 0;
#endif
        bool go = __any(jq[0] < RS);
        while (go) {
#pragma unroll
          for (int u = 0; u <= D; ++u) {
            if (go) { // (uniform) slot u is consumed, slot u + D (mod D + 1) is free
              issue(vv[(u + D) % (D + 1)], cc[(u + D) % (D + 1)], posq[D], jq[D]);
              consume(vv[u], cc[u], jq[0], posq[0], endq[0]);
#pragma unroll
              for (int d = 0; d < D; ++d)
                jq[d] = jq[d + 1], posq[d] = posq[d + 1], endq[d] = endq[d + 1];
              posq[D] += TRIP;
              settle(jq[D], posq[D], endq[D]);
              go = __any(jq[0] < RS);
#ifdef SJ_LT_PROBE
              ++ptrips;
#endif
            }
          }
        }
#ifdef SJ_LT_PROBE
        {
          const long long pc2 = wall_clock64();
          pr_stage += pc1 - pc0, pr_loop += pc2 - pc1, pr_trips += ptrips, ++pr_panels;
        }
#endif
#pragma unroll
        for (int j = 0; j < RS; ++j) {
          lo[j] = hi[j];
          hi[j] = hin[j];
        }
      }
    }
    if (l == 0) {
#pragma unroll
      for (int j = 0; j < RS; ++j)
        if (have[j]) {
          const T c = alpha * acc[j];
          T y = c;
          if (beta != T(0))
            y = c + beta * out[row[j]];
          out[row[j]] = y;
          if constexpr (DOT)
            dot_acc += (double)in[row[j]] * (double)c;
        }
    }
  }
#ifdef SJ_LT_PROBE
  // (100 MHz ticks) per wave 0 and 7 of a few workgroups
  if (lane == 0 && (wave == 0 || wave == 7)
      && (blockIdx.x == 0 || blockIdx.x == 3 || blockIdx.x == 300 || blockIdx.x == 509))
    printf("LTPROBE wg %d wave %d sgs %d panels %lld trips %lld stage %lld loop %lld total %lld\n",
           (int)blockIdx.x, wave, (sg1 - sg0 + sgstep - 1) / sgstep, pr_panels, pr_trips,
           pr_stage, pr_loop,
           wall_clock64() - pr_begin);
#endif
  if constexpr (DOT) {
    double v = dot_acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      v += __shfl_down(v, o, 64);
    if (lane == 0)
      s_red[wave] = v;
    __syncthreads();
    if (t == 0) {
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w)
        r += s_red[w];
      dot.partials[dot_slot0 + blockIdx.x] = r; // behind the slices' partials
    }
  }
}

// SIG (sigma layout, WPB = 8): blocks of 1024 rows sorted by length across the
// block, a wave takes the slices `wave` and `15 - wave` one after the other.
template <typename T, typename TV, int WPB, int E, bool DOT, int MODE, bool SIG = false>
__global__ __launch_bounds__(64 * WPB, 4) void csr_sjds_kernel(
    SjArgs<T, TV> A, T alpha, const T* __restrict__ in, T beta, T* __restrict__ out,
    DotOut dot, RowBlockOrder ord, const T* __restrict__ diagonal,
    const int32_t* __restrict__ low_rowptr)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  T* s_x = reinterpret_cast<T*>(s_raw);
  __shared__ double s_red[WPB];
  constexpr int NT = 64 * WPB;
  constexpr int SPW = SIG ? 2 : 1;      // slices per wave and block
  constexpr int R = 64 * WPB * SPW;     // rows per block
  constexpr int RB = SIG ? kSjSigBits : 6; // the row's bits in its (length, row) word
  static_assert(!SIG || R == kSjSigRows, "sigma blocks are 1024 rows");
  typedef T pair_t __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int part = t & 7, cg = t >> 3;
  constexpr int CG = NT / 8;
  double dot_acc = 0.0;
  const int num_slots = order_slots(ord);
  // the chunk numbers of a block are requested a whole block ahead (the lists
  // are padded to the stride, so the request does not need the block's count)
  int32_t ch[4];
  auto request_chunks = [&](int bb) {
    const int32_t* cl = A.chunks + (int64_t)bb * A.stride;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int c = cg + m * CG;
      ch[m] = cl[c < A.stride ? c : A.stride - 1];
    }
  };
  {
    const int b0 = blockIdx.x < num_slots ? order_row_block(ord, blockIdx.x) : -1;
    if (b0 >= 0)
      request_chunks(b0);
  }
#ifdef SJ_PROBE
  long long pr_wait = 0, pr_stage = 0, pr_bar = 0, pr_slice = 0, pr_blocks = 0;
  const long long pr_begin = wall_clock64();
#endif
  for (int it = blockIdx.x; it < num_slots && (A.phases & 2); it += gridDim.x) {
    const int b = order_row_block(ord, it);
    const int itn = it + gridDim.x;
    const int bn = itn < num_slots ? order_row_block(ord, itn) : -1;
    if (b < 0) { // uniform per workgroup
      if (bn >= 0)
        request_chunks(bn);
      continue;
    }
    const int K = A.blk[2 * b];
    const int wide = A.blk[2 * b + 1];
    const int32_t r0 = b * R;
    // the slices' own loads first: they do not depend on the staged x
    // (a slice past the end of the matrix reads the last one's, unused).  The
    // sigma layout gives the wave the slices `wave` and `2 WPB - 1 - wave` of
    // the block (sorted by length across the block: the longest rows with the
    // shortest), every slice of the block being there.
    // (symmetric storage with alpha = 1, beta = 0 -- Matrix::mult, cg(): 1 * s is
    // s and fl(1 * v) is v, so nothing turns: the merged row's products are
    // added to d_i x_i as they are, the plain slice with a starting value)
    const bool sym_plain = MODE == 3 && alpha == T(1) && beta == T(0);
    bool have[SPW], in_slice[SPW];
    int32_t lp[SPW], myrow[SPW], nlow[SPW];
    uint32_t ub_slice[SPW];
    T x_own[SPW], y0[SPW], init[SPW];
    // (uniform per wave, and told so: the slice's base pointers then live in
    // scalar registers and a step's address costs no vector arithmetic)
    const uint32_t ub_block
        = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.ubase[r0 / 64]);
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
      // (odd waves taking their short slice first: measured, no difference)
      const int sl = h == 0 ? wave : 2 * WPB - 1 - wave;
      const int32_t s0 = r0 + sl * 64;
      have[h] = SIG || s0 < A.num_rows;
      const int32_t s0c = have[h] ? s0 : ((A.num_rows - 1) / 64) * 64;
      const uint32_t lpw = (uint32_t)A.lenperm[s0c + lane];
      lp[h] = (int32_t)(lpw & ~kSjLongFlag);
      in_slice[h] = (lpw & kSjLongFlag) == 0; // else phase 0 wrote its y
      ub_slice[h] = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.ubase[s0c / 64]);
      myrow[h] = (SIG ? r0 : s0c) + (lp[h] & ((1 << RB) - 1));
      x_own[h] = y0[h] = init[h] = T(0);
      nlow[h] = 0;
      // (symmetric storage: the second slice's row data are loaded when its
      // turn comes -- both sets live across the first slice cost spills)
      if (MODE == 3 && h > 0)
        continue;
      const int32_t myrow_c = myrow[h] < A.num_rows ? myrow[h] : A.num_rows - 1;
      if constexpr (DOT || MODE == 3)
        x_own[h] = in[myrow_c];
      if (beta != T(0))
        y0[h] = out[myrow_c];
      if constexpr (MODE == 3) {
        init[h] = diagonal[myrow_c] * x_own[h];
        if (!sym_plain)
          nlow[h] = low_rowptr[myrow_c + 1] - low_rowptr[myrow_c];
      }
    }
    const int32_t* cl = A.chunks + (int64_t)b * A.stride;
    // the block's chunks of x: 8 lanes per chunk, 2 elements per lane, four
    // chunks per lane in flight; lanes past the list repeat its last chunk
#ifdef SJ_PROBE
    const long long pq0 = wall_clock64();
#endif
    __syncthreads(); // the previous block's slices are done with s_x
#ifdef SJ_PROBE
    const long long pq1 = wall_clock64();
#endif
    for (int c0 = cg; c0 < K; c0 += 4 * CG) {
      pair_t xv[4];
      const int64_t cmax = ((int64_t)A.num_cols - 2) & ~(int64_t)1;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t col = (int64_t)ch[m] * kSjChunk + part * 2;
        xv[m] = *reinterpret_cast<const pair_t*>(in + (col < cmax ? col : cmax));
        if (col + 1 == A.num_cols) // an odd number of columns: the last one
          xv[m][0] = in[col];
      }
      int32_t chn[4];
      if (c0 + 4 * CG < K) { // (uniform) the next round's chunk numbers
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int c = c0 + 4 * CG + m * CG;
          chn[m] = cl[c < K ? c : K - 1];
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int c = c0 + m * CG;
        *reinterpret_cast<pair_t*>(&s_x[(c < K ? c : K - 1) * kSjChunk + part * 2])
            = xv[m];
      }
      if (c0 + 4 * CG < K) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          ch[m] = chn[m];
      }
    }
    if (bn >= 0) // the next block's chunk numbers: back by the time they are used
      request_chunks(bn);
#ifdef SJ_PROBE
    const long long pq2 = wall_clock64();
#endif
    __syncthreads();
#ifdef SJ_PROBE
    const long long pq3 = wall_clock64();
#endif
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
      if (!have[h])
        continue;
      if (MODE == 3 && h > 0) {
        const int32_t myrow_c = myrow[h] < A.num_rows ? myrow[h] : A.num_rows - 1;
        x_own[h] = in[myrow_c];
        if (beta != T(0))
          y0[h] = out[myrow_c];
        init[h] = diagonal[myrow_c] * x_own[h];
        if (!sym_plain)
          nlow[h] = low_rowptr[myrow_c + 1] - low_rowptr[myrow_c];
      }
      const int32_t mylen = lp[h] >> RB;
      const SjUnit<TV, E>* vs
          = reinterpret_cast<const SjUnit<TV, E>*>(A.val) + ub_slice[h];
      const int64_t a_b = (int64_t)ub_block * E;
      T sum;
      if (wide) {
        const SjUnit<uint32_t, E>* cs
            = reinterpret_cast<const SjUnit<uint32_t, E>*>(A.codes + 4 * a_b)
              + (ub_slice[h] - ub_block);
        if (sym_plain)
          sum = sj_slice<T, TV, uint32_t, true, E, 0>(vs, cs, mylen, lane, s_x, in,
                                                      init[h], alpha);
        else
          sum = sj_slice<T, TV, uint32_t, true, E, MODE>(vs, cs, mylen, lane, s_x, in,
                                                         init[h], alpha, nlow[h],
                                                         beta * y0[h], beta != T(0));
      } else {
        const SjUnit<uint16_t, E>* cs
            = reinterpret_cast<const SjUnit<uint16_t, E>*>(
                  A.codes + (A.wide_alloc ? 4 : 2) * a_b)
              + (ub_slice[h] - ub_block);
        if (sym_plain)
          sum = sj_slice<T, TV, uint16_t, false, E, 0>(vs, cs, mylen, lane, s_x, in,
                                                       init[h], alpha);
        else
          sum = sj_slice<T, TV, uint16_t, false, E, MODE>(vs, cs, mylen, lane, s_x, in,
                                                          init[h], alpha, nlow[h],
                                                          beta * y0[h], beta != T(0));
      }
      if (myrow[h] < A.num_rows && in_slice[h]) {
        const T c = MODE == 3 ? sum : alpha * sum;
        T y = c;
        if (beta != T(0) && MODE != 3)
          y = c + beta * y0[h];
        out[myrow[h]] = y;
        if constexpr (DOT)
          dot_acc += (double)x_own[h] * (double)c;
      }
    }
#ifdef SJ_PROBE
    {
      const long long pq4 = wall_clock64();
      pr_wait += pq1 - pq0, pr_stage += pq2 - pq1, pr_bar += pq3 - pq2,
          pr_slice += pq4 - pq3, ++pr_blocks;
    }
#endif
  }
#ifdef SJ_PROBE
  // (100 MHz ticks) waves 0 and WPB - 1 of a few workgroups
  if (lane == 0 && (wave == 0 || wave == WPB - 1)
      && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 200))
    printf("SJPROBE wg %d wave %d blocks %lld wait %lld stage %lld barrier %lld slice %lld total %lld\n",
           (int)blockIdx.x, wave, pr_blocks, pr_wait, pr_stage, pr_bar, pr_slice,
           wall_clock64() - pr_begin);
#endif
  if constexpr (DOT) {
    // the workgroup's partial (fixed tree: deterministic), the array's unused
    // tail cleared -- spmv_dot_epilogue for WPB waves
    double v = dot_acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      v += __shfl_down(v, o, 64);
    if (lane == 0)
      s_red[wave] = v;
    __syncthreads();
    if (t == 0) {
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < WPB; ++w)
        r += s_red[w];
      dot.partials[blockIdx.x] = r;
    }
    for (int i = gridDim.x + blockIdx.x * NT + t; i < dot.len; i += gridDim.x * NT)
      dot.partials[i] = 0.0;
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

// workgroups of a CU that fit `lds` bytes each, and the waves they bring (the
// kernel's registers allow 16 waves per CU)
int sj_wgs_per_cu(int wpb, int64_t lds)
{
  int wgs = (int)((160 * 1024 - 2048) / (lds > 1 ? lds : 1));
  const int by_waves = 16 / wpb;
  wgs = wgs < by_waves ? wgs : by_waves;
  return wgs < 1 ? 1 : wgs;
}

template <typename T, int WPB, int E, bool DOT, int MODE = 0, typename TV = T,
          bool SIG = false>
int sj_launch(const spmv_hip_csr_plan* pl, hipStream_t st, T alpha, const T* in,
              T beta, T* out, DotOut dot, const T* diagonal = nullptr,
              const int32_t* low_rowptr = nullptr)
{
  constexpr bool kMixed = !std::is_same<T, TV>::value; // the fp32 copy of the values
  SjArgs<T, TV> A;
  A.num_rows = pl->num_rows;
  A.num_cols = pl->num_cols;
  A.nblk = pl->sj_nblk;
  A.maxk = pl->sj_maxk;
  A.stride = pl->sj_stride;
  A.wide_alloc = pl->sj_wide_alloc;
  A.rowptr = pl->rowptr0;
  A.ubase = pl->sj_ubase;
  A.lenperm = pl->sj_lenperm;
  A.blk = pl->sj_blk;
  A.chunks = pl->sj_chunks;
  A.codes = pl->sj_codes;
  A.val = static_cast<const TV*>(kMixed ? pl->sj_val32 : pl->sj_val);
  A.phases = pl->sj_phases;
  A.nlong = pl->sj_nlong;
  A.long_sorted = pl->sj_long_sorted && pl->sj_long_panels;
  A.long_rows = pl->sj_long_rows;
  A.colind = pl->colind0;
  A.values = static_cast<const TV*>(kMixed ? pl->sj32_values0 : pl->sj_values0);
  A.lt_cmin = pl->sj_lt_cmin;
  A.lt_np = pl->sj_lt_np;
  A.lt_off = pl->sj_lt_off;
  A.lt_tab = pl->sj_lt_tab;
  A.lt_codes = pl->sj_lt_codes;
  A.lt_coff = pl->sj_lt_coff;
  const size_t lds = (size_t)pl->sj_maxk * kSjChunk * sizeof(T) + 16;
  int wgs = pl->sj_blocks_per_cu > 0 ? pl->sj_blocks_per_cu
                                     : sj_wgs_per_cu(WPB, (int64_t)lds + 64);
  int grid = pl->ctx->num_cus * wgs;
  if (grid > pl->ctx->dot_blocks)
    grid = pl->ctx->dot_blocks;
  if (grid > pl->sj_nblk)
    grid = pl->sj_nblk;
  if (grid >= 8)
    grid -= grid % 8; // slots it with equal it % 8 stay on one XCD
  if (grid < 1)
    grid = 1;
  RowBlockOrder ord;
  ord.table = nullptr;
  ord.num_slots = 0;
  ord.xcd_group = grid >= 8 ? pl->sj_xcd_group : 0;
  ord.num_row_blocks = pl->sj_nblk;
  ord.nt_store = 0;
  if (A.phases & 2) {
    hipLaunchKernelGGL((csr_sjds_kernel<T, TV, WPB, E, DOT, MODE, SIG>), dim3(grid),
                       dim3(64 * WPB), lds, st, A, alpha, in, beta, out, dot, ord,
                       diagonal, low_rowptr);
    SPMV_CHECK_LAUNCH();
  }
  if constexpr (MODE != 0) // (symmetric storage: built without long rows)
    return pl->sj_nlong > 0 ? SPMV_HIP_EINVAL : SPMV_HIP_OK;
  if (pl->sj_nlong > 0 && (A.phases & 1) && A.long_sorted && pl->sj_lt_tab
      && pl->sj_long_table) {
    // the long rows by the table-driven kernel: 8-wave workgroups, two per CU,
    // contiguous runs of supergroups; dot partials behind the slices'
    const int nsg = pl->sj_lt_nsg;
    const size_t llds = (size_t)kSjLtPanel * sizeof(T);
    int lgrid = pl->ctx->num_cus * 2;
    if (lgrid > nsg)
      lgrid = nsg;
    if (DOT && lgrid > pl->ctx->dot_blocks - grid)
      lgrid = pl->ctx->dot_blocks - grid;
    if (lgrid >= 8)
      lgrid -= lgrid % 8;
    if (lgrid < 1)
      lgrid = 1;
    // (more dynamic LDS than a launch gets by default: the attribute was
    // raised when the table was built, on the plan's device)
    hipLaunchKernelGGL((csr_sjds_longt_kernel<T, TV, DOT>), dim3(lgrid), dim3(512), llds,
                       st, A, alpha, in, beta, out, dot, (A.phases & 2) ? grid : 0);
    SPMV_CHECK_LAUNCH();
  } else if (pl->sj_nlong > 0 && (A.phases & 1)) {
    // the long rows: 8-wave workgroups, 64 rows each; their dot
    // partials go behind the slices' (whose kernel cleared the array's tail)
#ifndef SJ_LONG_WAVES
#define SJ_LONG_WAVES 8
#endif
    constexpr int LW = SJ_LONG_WAVES;
    const int nsg = ((pl->sj_nlong + 7) / 8 + LW * kSjLongSets - 1) / (LW * kSjLongSets);
#ifndef SJ_PANEL_COLS
#define SJ_PANEL_COLS 7680
#endif
    // panels of 7680 columns (60 KiB of fp64), trips of 4 steps (103
    // registers: 16 waves per CU, two workgroups).  Measured on the 1 % tail of
    // the benchmark's matrix (110 M entries), same box: panels of 2048 / 4096 /
    // 6144 / 7680 / 9216 columns 0.60 / 0.52 / 0.48 / 0.44 / 0.44 ms; trips of
    // 8 steps (180 registers, one workgroup per CU) 0.58; 16-wave workgroups
    // with panels of 12288 / 16384 columns 0.50 / 0.49; loads three trips ahead
    // in a ring of four register sets 0.45-0.57.  The launch FORKED onto a
    // helper stream beside the slices' (independent rows of y): 1.03 ms for
    // the pair against 0.85 one after the other -- the two persistent grids
    // take each other's CUs
    A.long_panel = SJ_PANEL_COLS;
    const size_t llds = (size_t)A.long_panel * sizeof(T) + 16;
    int lwgs = (int)((160 * 1024 - 2048) / ((int64_t)llds + LW * 512 + 256));
    lwgs = lwgs < 1 ? 1 : (lwgs > 4 ? 4 : lwgs);
    int lgrid = pl->ctx->num_cus * lwgs;
    if (lgrid > nsg)
      lgrid = nsg;
    if (DOT && lgrid > pl->ctx->dot_blocks - grid)
      lgrid = pl->ctx->dot_blocks - grid;
    if (lgrid >= 8)
      lgrid -= lgrid % 8;
    if (lgrid < 1)
      lgrid = 1;
    if (llds > 64 * 1024) { // more dynamic LDS than a launch gets by default
      static bool raised = false;
      if (!raised) {
        SPMV_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&csr_sjds_long_kernel<T, TV, LW, DOT, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
        SPMV_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&csr_sjds_long_kernel<T, TV, LW, DOT, false>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
        raised = true;
      }
    }
    if (A.long_sorted)
      hipLaunchKernelGGL((csr_sjds_long_kernel<T, TV, LW, DOT, true>), dim3(lgrid),
                         dim3(64 * LW), llds, st, A, alpha, in, beta, out, dot,
                         (A.phases & 2) ? grid : 0);
    else
      hipLaunchKernelGGL((csr_sjds_long_kernel<T, TV, LW, DOT, false>), dim3(lgrid),
                         dim3(64 * LW), llds, st, A, alpha, in, beta, out, dot,
                         (A.phases & 2) ? grid : 0);
    SPMV_CHECK_LAUNCH();
  }
  return SPMV_HIP_OK;
}

template <typename T, int E, bool DOT, typename TV = T>
int sj_run_e(const spmv_hip_csr_plan* pl, hipStream_t st, T alpha, const T* in,
             T beta, T* out, DotOut dot)
{
  switch (pl->sj_wpb) {
  case 4: return sj_launch<T, 4, E, DOT, 0, TV>(pl, st, alpha, in, beta, out, dot);
  case 8: return sj_launch<T, 8, E, DOT, 0, TV>(pl, st, alpha, in, beta, out, dot);
  default:
    if (pl->sj_sigma) // blocks of 1024 rows sorted across the block: 8 waves, 2 slices each
      return sj_launch<T, 8, E, DOT, 0, TV, true>(pl, st, alpha, in, beta, out, dot);
    return sj_launch<T, 16, E, DOT, 0, TV>(pl, st, alpha, in, beta, out, dot);
  }
}

template <typename T, bool DOT, typename TV = T>
int sj_run(const spmv_hip_csr_plan* pl, hipStream_t st, T alpha, const T* in,
           T beta, T* out, DotOut dot)
{
  switch (pl->sj_unit) {
  case 1: return sj_run_e<T, 1, DOT, TV>(pl, st, alpha, in, beta, out, dot);
  case 2: return sj_run_e<T, 2, DOT, TV>(pl, st, alpha, in, beta, out, dot);
  default: return sj_run_e<T, 4, DOT, TV>(pl, st, alpha, in, beta, out, dot);
  }
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

// the rows longer than thr, ascending, sorted by length inside runs of 64
// (SPMV_HIP_ENOMEM: no memory)
int sj_build_long_list(spmv_hip_csr_plan* pl, const int32_t* rowptr,
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

void sj_lt_free(spmv_hip_csr_plan* pl)
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

// the table-driven kernel's panel exceeds the dynamic LDS a launch gets by
// default: raise the limit for every instantiation, on the current device
int sj_lt_raise_lds()
{
#define SJ_LT_RAISE(...)                                                       \
  SPMV_CHECK_HIP(hipFuncSetAttribute(                                          \
      reinterpret_cast<const void*>(&csr_sjds_longt_kernel<__VA_ARGS__>),      \
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kSjLtPanel * sizeof(double))))
  SJ_LT_RAISE(double, double, false);
  SJ_LT_RAISE(double, double, true);
  SJ_LT_RAISE(double, float, false);
  SJ_LT_RAISE(double, float, true);
  SJ_LT_RAISE(float, float, false);
#undef SJ_LT_RAISE
  return SPMV_HIP_OK;
}

// (no memory: the plan stays without the table and the older kernel runs)
int sj_build_long_table(spmv_hip_csr_plan* pl, const int32_t* rowptr,
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
    sj_lt_free(pl);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  pl->sj_lt_entries = total;
  pl->sj_lt_codes_n = ncodes;
  pl->sj_lt_nsg = nsg;
  return sj_lt_raise_lds();
}

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

} // namespace

namespace
{
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
  sj_lt_free(pl);
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
  int sigma = 0;
  if (best == 16 && pl->ctx->sj_sigma) {
    sigma = 1;
    if (pl->ctx->sj_sigma == 1 && avg >= 48.0) {
      hipcub::CountingInputIterator<int32_t> first(0);
      hipcub::TransformInputIterator<int32_t, SjShortLen,
                                     hipcub::CountingInputIterator<int32_t>>
          lens(first, SjShortLen{rowptr, thr, pl->nnz});
      int32_t* d_max = nullptr;
      void* tmpm = nullptr;
      size_t tbm = 0;
      int32_t h_max = 0;
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
        (void)hipGetLastError();
        h_max = 0;
      }
      if ((double)h_max <= 1.1 * avg)
        sigma = 0;
    }
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
  cleanup();
  if (e != hipSuccess) {
    spmv_sjds_free(pl);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? SPMV_HIP_OK : static_cast<int>(e);
  }
  {
    const int rc = sj_build_long_list(pl, rowptr, colind, thr, st);
    if (rc != SPMV_HIP_OK) {
      spmv_sjds_free(pl);
      return rc == SPMV_HIP_ENOMEM ? SPMV_HIP_OK : rc;
    }
    const int rc2 = sj_build_long_table(pl, rowptr, colind, st);
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

int spmv_sjds_run_f32f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                         const double* in, double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return sj_run<double, true, float>(pl, st, alpha, in, beta, out, dot);
  return sj_run<double, false, float>(pl, st, alpha, in, beta, out, dot);
}

// Symmetric storage (csr_kernels.cpp:26-40) in ONE pass over ONE sliced jagged
// structure (plan->sjt): the "rows" of the merged matrix -- a row's stored lower
// entries followed by the entries of its column in the reference's order
// (spmv_sjds_sym_merge) -- by the slices' kernel in MODE 3.
namespace
{
template <typename T, bool DOT>
int sj_run_sym(const spmv_hip_csr_plan* pl, hipStream_t st, const T* diagonal, T alpha,
               const T* in, T beta, T* out, DotOut dot)
{
  const spmv_hip_csr_plan* m = pl->sjt;
  if (m->sj_unit != 2)
    return SPMV_HIP_EINVAL;
  if (m->sj_wpb == 8)
    return sj_launch<T, 8, 2, DOT, 3>(m, st, alpha, in, beta, out, dot, diagonal,
                                      pl->rowptr0);
  if (m->sj_wpb == 16 && m->sj_sigma)
    return sj_launch<T, 8, 2, DOT, 3, T, true>(m, st, alpha, in, beta, out, dot, diagonal,
                                               pl->rowptr0);
  if (m->sj_wpb == 16)
    return sj_launch<T, 16, 2, DOT, 3>(m, st, alpha, in, beta, out, dot, diagonal,
                                       pl->rowptr0);
  return SPMV_HIP_EINVAL;
}

// lengths of the merged rows (+ a zero behind them, for the scan)
__global__ __launch_bounds__(kBlock) void sj_sym_len_kernel(
    int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ t_ptr,
    int32_t* __restrict__ vptr)
{
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n;
       i += (int64_t)gridDim.x * kBlock)
    vptr[i] = i < n ? (rowptr[i + 1] - rowptr[i]) + (t_ptr[i + 1] - t_ptr[i]) : 0;
}

// columns of the merged rows and where their values are in the caller's array
// (one wave per row)
__global__ __launch_bounds__(kBlock) void sj_sym_merge_kernel(
    int32_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const int32_t* __restrict__ t_ptr, const int32_t* __restrict__ t_row,
    const int32_t* __restrict__ t_pos, const int32_t* __restrict__ vptr,
    int32_t* __restrict__ vcol, int32_t* __restrict__ vmap)
{
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t i = wid; i < n; i += nw) {
    const int32_t a = rowptr[i], nl = rowptr[i + 1] - a;
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
int spmv_sjds_sym_merge(const spmv_hip_csr_plan* pl, int32_t** vptr, int32_t** vcol,
                        int32_t** vmap, hipStream_t st)
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
                       dim3(kBlock), 0, st, n, pl->rowptr0, pl->t_ptr, *vptr);
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
                       *vptr, *vcol, *vmap);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
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

int spmv_sjds_run_sym_f64(const spmv_hip_csr_plan* pl, hipStream_t st,
                          const double* diagonal, double alpha, const double* in,
                          double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return sj_run_sym<double, true>(pl, st, diagonal, alpha, in, beta, out, dot);
  return sj_run_sym<double, false>(pl, st, diagonal, alpha, in, beta, out, dot);
}
int spmv_sjds_run_sym_f32(const spmv_hip_csr_plan* pl, hipStream_t st,
                          const float* diagonal, float alpha, const float* in,
                          float beta, float* out)
{
  return sj_run_sym<float, false>(pl, st, diagonal, alpha, in, beta, out, DotOut());
}

int spmv_sjds_run_f64(const spmv_hip_csr_plan* pl, hipStream_t st, double alpha,
                      const double* in, double beta, double* out, DotOut dot)
{
  if (dot.partials)
    return sj_run<double, true>(pl, st, alpha, in, beta, out, dot);
  return sj_run<double, false>(pl, st, alpha, in, beta, out, dot);
}

int spmv_sjds_run_f32(const spmv_hip_csr_plan* pl, hipStream_t st, float alpha,
                      const float* in, float beta, float* out)
{
  return sj_run<float, false>(pl, st, alpha, in, beta, out, DotOut());
}
