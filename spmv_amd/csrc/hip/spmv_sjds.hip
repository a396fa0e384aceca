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
#include "sjds.h"

namespace
{

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
  // ROWLDS (symmetric storage, alpha = 1, beta = 0): a row's own data travel
  // through LDS.  In the sorted blocks a wave's 64 rows lie anywhere in the
  // block's 1024: its loads of the diagonal, of x_i and of the two row-pointer
  // words and its store of y_i are 64 different cache lines each -- five such
  // instructions per slice are four times the L2 requests of the slice's whole
  // entry stream.  Here the workgroup stages d_i x_i (or, for a row whose lower
  // part the long-row kernel took, the y it left) for the block's rows with
  // coalesced loads next to the chunks of x, the lanes pick theirs up in LDS,
  // leave y_i there, and the block's y goes out coalesced while the next
  // block's x is staged (the fused dot is taken there too).
  // YLDS: the general kernel in the sigma layout hands its y over the same way
  // (alpha sum in LDS; beta y0 and the fused dot are taken at the flush).
  constexpr bool ROWLDS = MODE == 4;
  constexpr bool YLDS = ROWLDS || (MODE == 0 && SIG);
  __shared__ T s_init[ROWLDS ? R : 1];
  __shared__ T s_y[YLDS ? R : 1];
  // (general storage: the rows the long-row kernel writes are not the slices'
  // to store; one bit per row of the block, two sets in turn -- the flush reads
  // one while the next block's slices fill the other)
  constexpr bool SKIPS = YLDS && MODE == 0;
  __shared__ uint32_t s_skip[SKIPS ? 2 : 1][SKIPS ? R / 32 : 1];
  int skip_set = 0;
  int32_t flush_r0 = -1; // first row of the block whose y sits in s_y
  auto flush_y = [&]() {
    if constexpr (YLDS) {
      if (flush_r0 >= 0) // uniform
        for (int i = t; i < R; i += NT) {
          const int32_t row = flush_r0 + i;
          if (row < A.num_rows) {
            const T c = s_y[i];
            if constexpr (MODE == 0) {
              if ((s_skip[skip_set ^ 1][i >> 5] >> (i & 31)) & 1u)
                continue; // a LONG row: the long-row kernel writes its y
              T y = c;
              if (beta != T(0))
                y = c + beta * out[row];
              out[row] = y;
            } else {
              out[row] = c;
            }
            if constexpr (DOT)
              dot_acc += (double)in[row] * (double)c;
          }
        }
    }
  };
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
    // MODE 4 = MODE 3 with alpha = 1, beta = 0 as its own instantiation: only the
    // plain slice is compiled in (the two paths in one kernel spilled: 127
    // registers + 28 B of scratch per lane)
    constexpr bool SYM = MODE == 3 || MODE == 4;
    constexpr bool sym_plain = MODE == 4;
    // symmetric storage, one row: where its sum starts and how many stored
    // lower entries precede its column's.  A row whose lower part is LONG gave
    // it to the long-row kernel (launched before this one): it starts from the
    // y that kernel left -- fl(alpha (d x + lower)) + fl(beta y0), the value the
    // reference holds where the column's entries begin -- and never turns
    // (nlow = -1: every entry of the merged row is one of the column's).
    auto sym_row = [&](int32_t row, T x_row, T& init_out, int32_t& nlow_out) {
      const int32_t la = low_rowptr[row], lb = low_rowptr[row + 1];
      const bool lng = sj_is_long(la, lb, A.sym_long_thr, A.sym_nnz);
      init_out = lng ? out[row] : diagonal[row] * x_row;
      nlow_out = lng ? -1 : (sym_plain ? 0 : lb - la);
    };
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
      if (SYM && h > 0)
        continue;
      if constexpr (YLDS)
        continue; // (the row's data come / go through LDS)
      const int32_t myrow_c = myrow[h] < A.num_rows ? myrow[h] : A.num_rows - 1;
      if constexpr (DOT || SYM)
        x_own[h] = in[myrow_c];
      if (beta != T(0))
        y0[h] = out[myrow_c];
      if constexpr (SYM)
        sym_row(myrow_c, x_own[h], init[h], nlow[h]);
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
    if constexpr (YLDS && !ROWLDS) {
      flush_y(); // (the previous block's y: its slices are done)
      if (t < R / 32)
        s_skip[skip_set][t] = 0u;
      flush_r0 = r0;
    }
    if constexpr (ROWLDS) {
      flush_y(); // (the previous block's y: its slices are done)
      for (int i = t; i < R; i += NT) {
        const int32_t row = r0 + i;
        T v = T(0);
        if (row < A.num_rows) {
          // (no long rows in the stored block -- the threshold is at its
          // default: nothing to look up in the row pointer, 4 B per row saved)
          bool lng = false;
          if (A.sym_long_thr != INT32_MAX) { // uniform
            const int32_t la = low_rowptr[row], lb = low_rowptr[row + 1];
            lng = sj_is_long(la, lb, A.sym_long_thr, A.sym_nnz);
          }
          v = lng ? out[row] : diagonal[row] * in[row];
        }
        s_init[i] = v;
      }
      flush_r0 = r0;
    }
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
      if constexpr (ROWLDS)
        init[h] = s_init[myrow[h] - r0];
      if (SYM && !ROWLDS && h > 0) {
        const int32_t myrow_c = myrow[h] < A.num_rows ? myrow[h] : A.num_rows - 1;
        x_own[h] = in[myrow_c];
        if (beta != T(0))
          y0[h] = out[myrow_c];
        sym_row(myrow_c, x_own[h], init[h], nlow[h]);
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
        if constexpr (sym_plain)
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
        if constexpr (sym_plain)
          sum = sj_slice<T, TV, uint16_t, false, E, 0>(vs, cs, mylen, lane, s_x, in,
                                                       init[h], alpha);
        else
          sum = sj_slice<T, TV, uint16_t, false, E, MODE>(vs, cs, mylen, lane, s_x, in,
                                                          init[h], alpha, nlow[h],
                                                          beta * y0[h], beta != T(0));
      }
      if constexpr (ROWLDS) {
        if (in_slice[h]) // (every row of a symmetric child plan is)
          s_y[myrow[h] - r0] = sum;
      } else if constexpr (YLDS) {
        if (in_slice[h])
          s_y[myrow[h] - r0] = alpha * sum;
        else if (myrow[h] < A.num_rows)
          atomicOr(&s_skip[skip_set][(myrow[h] - r0) >> 5], 1u << ((myrow[h] - r0) & 31));
      } else if (myrow[h] < A.num_rows && in_slice[h]) {
        const T c = SYM ? sum : alpha * sum;
        T y = c;
        if (beta != T(0) && !SYM)
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
    if constexpr (SKIPS)
      skip_set ^= 1; // (the next flush reads the set this block filled)
  }
#ifdef SJ_PROBE
  // (100 MHz ticks) waves 0 and WPB - 1 of a few workgroups
  if (lane == 0 && (wave == 0 || wave == WPB - 1)
      && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 200))
    printf("SJPROBE wg %d wave %d blocks %lld wait %lld stage %lld barrier %lld slice %lld total %lld\n",
           (int)blockIdx.x, wave, pr_blocks, pr_wait, pr_stage, pr_bar, pr_slice,
           wall_clock64() - pr_begin);
#endif
  if constexpr (YLDS) {
    __syncthreads(); // the last block's slices are done
    flush_y();
  }
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

template <typename T, int WPB, int E, bool DOT, int MODE = 0, typename TV = T,
          bool SIG = false>
int sj_launch(const spmv_hip_csr_plan* pl, hipStream_t st, T alpha, const T* in,
              T beta, T* out, DotOut dot, const T* diagonal = nullptr,
              const int32_t* low_rowptr = nullptr,
              const spmv_hip_csr_plan* parent = nullptr)
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
  // (MODE 4 keeps the block's d x and y in static LDS beside the staged x)
  const int64_t lds_static
      = (MODE == 4 ? 2 : (MODE == 0 && SIG ? 1 : 0))
        * (int64_t)(64 * WPB * (SIG ? 2 : 1)) * (int64_t)sizeof(T);
  int wgs = pl->sj_blocks_per_cu > 0
                ? pl->sj_blocks_per_cu
                : sj_wgs_per_cu(WPB, (int64_t)lds + lds_static + 64);
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
  if constexpr (MODE == 3 || MODE == 4) {
    // symmetric storage: the long rows of the stored lower block FIRST, by the
    // long-row kernels on the caller's arrays (the parent plan lists them);
    // the slices' kernel below starts those rows from the y they leave
    if (parent && parent->sj_nlong > 0) {
      A.sym_long_thr = parent->sj_long_thr;
      A.sym_nnz = parent->nnz;
      if (A.phases & 1) {
        SjArgs<T, TV> L = A;
        L.rowptr = parent->rowptr0;
        L.colind = parent->colind0;
        L.values = static_cast<const TV*>(pl->sj_values0); // the caller's (baked from)
        L.nlong = parent->sj_nlong;
        L.long_sorted = parent->sj_long_sorted && parent->sj_long_panels;
        L.long_rows = parent->sj_long_rows;
        L.lt_cmin = parent->sj_lt_cmin;
        L.lt_np = parent->sj_lt_np;
        L.lt_off = parent->sj_lt_off;
        L.lt_tab = parent->sj_lt_tab;
        L.lt_codes = parent->sj_lt_codes;
        L.lt_coff = parent->sj_lt_coff;
        L.sym_diag = diagonal;
        int rl;
        if constexpr (sizeof(T) == 8)
          rl = spmv_sj_long_launch_f64(parent, L, st, alpha, in, beta, out, DotOut(), 0,
                                       0);
        else
          rl = spmv_sj_long_launch_f32(parent, L, st, alpha, in, beta, out, DotOut(), 0,
                                       0);
        if (rl != SPMV_HIP_OK)
          return rl;
      }
    }
  }
  if (A.phases & 2) {
    hipLaunchKernelGGL((csr_sjds_kernel<T, TV, WPB, E, DOT, MODE, SIG>), dim3(grid),
                       dim3(64 * WPB), lds, st, A, alpha, in, beta, out, dot, ord,
                       diagonal, low_rowptr);
    SPMV_CHECK_LAUNCH();
  }
  if constexpr (MODE != 0) // (symmetric storage: built without long rows)
    return pl->sj_nlong > 0 ? SPMV_HIP_EINVAL : SPMV_HIP_OK;
  if (pl->sj_nlong > 0 && (A.phases & 1)) {
    // the long rows: a launch of its own (spmv_sjds_long.hip), its dot partials
    // behind the slices' (whose kernel cleared the array's tail)
    const int slot0 = (A.phases & 2) ? grid : 0;
    const DotOut ld = DOT ? dot : DotOut();
    if constexpr (kMixed)
      return spmv_sj_long_launch_f32f64(pl, A, st, alpha, in, beta, out, ld, slot0,
                                        pl->ctx->dot_blocks - grid);
    else if constexpr (sizeof(T) == 8)
      return spmv_sj_long_launch_f64(pl, A, st, alpha, in, beta, out, ld, slot0,
                                     pl->ctx->dot_blocks - grid);
    else
      return spmv_sj_long_launch_f32(pl, A, st, alpha, in, beta, out, ld, slot0,
                                     pl->ctx->dot_blocks - grid);
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
} // namespace

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
  // (alpha = 1, beta = 0 -- Matrix::mult, cg(): 1 * s is s and fl(1 * v) is v,
  // nothing turns: an instantiation of its own, MODE 4)
  const bool plain = alpha == T(1) && beta == T(0);
#define SJ_SYM(WPB, SIGV)                                                      \
  (plain ? sj_launch<T, WPB, 2, DOT, 4, T, SIGV>(m, st, alpha, in, beta, out, dot, \
                                                 diagonal, pl->rowptr0, pl)   \
         : sj_launch<T, WPB, 2, DOT, 3, T, SIGV>(m, st, alpha, in, beta, out, dot, \
                                                 diagonal, pl->rowptr0, pl))
  if (m->sj_wpb == 8)
    return SJ_SYM(8, false);
  if (m->sj_wpb == 16 && m->sj_sigma)
    return SJ_SYM(8, true);
  if (m->sj_wpb == 16)
    return SJ_SYM(16, false);
#undef SJ_SYM
  return SPMV_HIP_EINVAL;
}
} // namespace

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
