// Sliced jagged form, the LONG ROWS' kernels (see spmv_sjds.hip for the form):
// rows the slices leave out -- more than four times the average length and more
// than 96 entries -- streamed from the caller's CSR arrays, eight lanes per row.
#include "sjds.h"

namespace
{

// EIGHT long rows by one wave, eight lanes each: lane l of a group reads entries
// 8 s + l of its row (64 + 32 bytes per row and step, straight from the
// caller's CSR arrays), the group's eight products are added to the row's sum
// one by one in the row's order (every lane of the group keeps the sum).
// Values and columns travel two load groups ahead, x one.  [a, b) = the lane's
// row (b == a: no row); returns the row's sum.
template <typename T, typename TV>
__device__ __forceinline__ T sj_long_rows8(const TV* __restrict__ val,
                                          const int32_t* __restrict__ col,
                                          int64_t a, int64_t b, int lane,
                                          const T* __restrict__ in, T init)
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
  T t = init; // (symmetric storage: d_i x_i)
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
        sum[h] = A.sym_diag ? A.sym_diag[row[h]] * in[row[h]] : T(0);
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
          sum[h] = sj_long_rows8<T, TV>(A.values, A.colind, ra[h], rb[h], lane, in,
                                        sum[h]);
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
                // a lane without an entry contributes -0.0, the identity of the
                // addition for EVERY sum (+0.0 would turn a sum of -0.0 -- a
                // symmetric row starts at d_i x_i, which can be that -- into
                // +0.0)
                const T pr = ok ? v[u] * x : -T(0);
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
      // (the sum starts in lane 0 of the group; symmetric storage: at d_i x_i)
      acc[j] = A.sym_diag ? A.sym_diag[row[j]] * in[row[j]] : T(0);
    }
    const int np = A.lt_np[sg];
    if (np == 0) { // a supergroup whose rows are not neighbours in x: rare, slow
      if (l == 0) {
#pragma unroll
        for (int j = 0; j < RS; ++j)
          if (have[j]) {
            const int64_t a = A.rowptr[row[j]], b = A.rowptr[row[j] + 1];
            T sum = acc[j];
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
            // a lane without an entry contributes -0.0, the identity of the
            // addition for every sum, a -0.0 included (symmetric storage starts
            // a row at d_i x_i)
            const T prod = (T)v.e[k] * xs[k];
            pr[k] = ok ? prod : -T(0);
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

template <typename T, typename TV, bool DOT>
int sj_launch_long(const spmv_hip_csr_plan* pl, SjArgs<T, TV> A, hipStream_t st, T alpha,
                   const T* in, T beta, T* out, DotOut dot, int dot_slot0, int dot_room)
{
  if (pl->sj_nlong <= 0)
    return SPMV_HIP_OK;
  if (A.long_sorted && pl->sj_lt_tab && pl->sj_long_table) {
    // the long rows by the table-driven kernel: 8-wave workgroups, two per CU,
    // contiguous runs of supergroups; dot partials behind the slices'
    const int nsg = pl->sj_lt_nsg;
    const size_t llds = (size_t)kSjLtPanel * sizeof(T);
    int lgrid = pl->ctx->num_cus * 2;
    if (lgrid > nsg)
      lgrid = nsg;
    if (DOT && lgrid > dot_room)
      lgrid = dot_room;
    if (lgrid >= 8)
      lgrid -= lgrid % 8;
    if (lgrid < 1)
      lgrid = 1;
    // (more dynamic LDS than a launch gets by default: the attribute was
    // raised when the table was built, on the plan's device)
    hipLaunchKernelGGL((csr_sjds_longt_kernel<T, TV, DOT>), dim3(lgrid), dim3(512), llds,
                       st, A, alpha, in, beta, out, dot, dot_slot0);
    SPMV_CHECK_LAUNCH();
    return SPMV_HIP_OK;
  }
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
  if (DOT && lgrid > dot_room)
    lgrid = dot_room;
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
                       dim3(64 * LW), llds, st, A, alpha, in, beta, out, dot, dot_slot0);
  else
    hipLaunchKernelGGL((csr_sjds_long_kernel<T, TV, LW, DOT, false>), dim3(lgrid),
                       dim3(64 * LW), llds, st, A, alpha, in, beta, out, dot, dot_slot0);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // namespace

// the table-driven kernel's panel exceeds the dynamic LDS a launch gets by
// default: raise the limit for every instantiation, on the current device
int spmv_sj_lt_raise_lds()
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

int spmv_sj_long_launch_f64(const spmv_hip_csr_plan* pl, SjArgs<double, double> A,
                            hipStream_t st, double alpha, const double* in, double beta,
                            double* out, DotOut dot, int dot_slot0, int dot_room)
{
  if (dot.partials)
    return sj_launch_long<double, double, true>(pl, A, st, alpha, in, beta, out, dot,
                                                dot_slot0, dot_room);
  return sj_launch_long<double, double, false>(pl, A, st, alpha, in, beta, out, dot,
                                               dot_slot0, dot_room);
}

int spmv_sj_long_launch_f32(const spmv_hip_csr_plan* pl, SjArgs<float, float> A,
                            hipStream_t st, float alpha, const float* in, float beta,
                            float* out, DotOut dot, int dot_slot0, int dot_room)
{
  return sj_launch_long<float, float, false>(pl, A, st, alpha, in, beta, out, dot,
                                             dot_slot0, dot_room);
}

int spmv_sj_long_launch_f32f64(const spmv_hip_csr_plan* pl, SjArgs<double, float> A,
                               hipStream_t st, double alpha, const double* in,
                               double beta, double* out, DotOut dot, int dot_slot0,
                               int dot_room)
{
  if (dot.partials)
    return sj_launch_long<double, float, true>(pl, A, st, alpha, in, beta, out, dot,
                                               dot_slot0, dot_room);
  return sj_launch_long<double, float, false>(pl, A, st, alpha, in, beta, out, dot,
                                              dot_slot0, dot_room);
}
