// PROBE, second version (round 6): march_probe.hip with its loads PIPELINED.
// The first version ran at 2.5 TB/s -- 15.6 us per 1024-row step in ~15
// DEPENDENT global round trips (descriptor -> lengths -> stream bases -> entries,
// five streams one after the other, three barriers).  Here everything a step
// needs is in registers when the step begins (descriptors two steps ahead, the
// lanes' lengths, stream bases and chunk numbers one step ahead) and ALL of a
// step's global loads -- the chunks of x, the first groups of all five streams
// -- are issued at its top, before the first barrier.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace
{
#ifndef MARCH_NT
#define MARCH_NT 1024
#define MARCH_STASH 9216
#define MARCH_XCAP 6144
#endif
constexpr int NT = MARCH_NT;
constexpr int NSL = NT / 64; // slices (waves) of a tile
constexpr int kStash = MARCH_STASH; // products
constexpr int kXcap = MARCH_XCAP;  // staged x elements
constexpr int GA = 8;        // entries of a lane in flight per group, stream A
constexpr int GS = 2;        // ... the streamed parts (Bs, Cs)
constexpr int GC = 4;        // ... the captured parts (Bc, Cc): codes only
constexpr int NCH = (kXcap / 16 * 8 + NT - 1) / NT; // staging loads per thread

struct StepDesc { // 32 bytes
  int32_t tile, chunk0, nchunks, own0, row0, pad0, pad1, pad2;
};

struct MarchArgs2 {
  int nunits;
  const int32_t* unit_step0;
  const StepDesc* steps;    // + two idle descriptors behind the last one
  const int32_t* chunks;    // (padded: a step's list may be read NT / 8 past its end)
  const uint64_t* meta;     // per tile and lane (+ one idle tile at the end)
  const double* dval;       // ... the row's diagonal entry
  const uint32_t* sbp;      // per tile and slice: first entry of the 5 streams (8 words)
  const double* a_val;
  const uint32_t* a_code;   // staged x position | position in the tile's stream << 16
  const uint16_t* bc_code;
  const double* bs_val;
  const uint16_t* bs_code;
  const uint16_t* cc_code;
  const double* cs_val;
  const uint16_t* cs_code;
  int64_t num_cols;
};

template <int G, typename C, bool VAL>
struct Grp {
  double v[VAL ? G : 1];
  C c[G];
  uint32_t act; // bit u: entry u exists
};

// issue the loads of entries k0 .. k0 + G - 1 of the lane's row
template <int G, typename C, bool VAL>
__device__ __forceinline__ void issue(Grp<G, C, VAL>& g, int len, int k0,
                                      uint32_t& base, const double* __restrict__ val,
                                      const C* __restrict__ code)
{
  g.act = 0u;
#pragma unroll
  for (int u = 0; u < G; ++u) {
    const bool act = k0 + u < len;
    const uint64_t m = __ballot(act);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(
        (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t pos = base + (act ? below : 0u);
    base += (uint32_t)__popcll(m);
    g.c[u] = code[pos];
    if constexpr (VAL)
      g.v[u] = val[pos];
    g.act |= act ? (1u << u) : 0u;
  }
}

__global__ __launch_bounds__(NT) void march2_kernel(MarchArgs2 a,
                                                    const double* __restrict__ x,
                                                    double* __restrict__ y,
                                                    double* __restrict__ dot_partials)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* stash = smem;
  double* sx = smem + kStash;
  __shared__ double s_red[NT / 64];
  const int v = threadIdx.x, lane = v & 63, slice = v >> 6;
  double dot_acc = 0.0;
  for (int u = blockIdx.x; u < a.nunits; u += gridDim.x) {
    const int s0 = a.unit_step0[u], ns = a.unit_step0[u + 1] - s0;
    // what a step needs, one step ahead
    StepDesc d0 = a.steps[s0], d1 = a.steps[s0 + 1];
    auto lane_data = [&](const StepDesc& d, uint64_t& meta, double& dv, uint32_t* sb,
                         int32_t* ch) {
      const int t = d.tile >= 0 ? d.tile : 0;
      meta = d.tile >= 0 ? a.meta[(int64_t)t * NT + v] : 0ull;
      dv = a.dval[(int64_t)t * NT + v];
      const uint4* q = reinterpret_cast<const uint4*>(a.sbp + ((int64_t)t * NSL + slice) * 8);
      const uint4 q0 = q[0], q1 = q[1];
      sb[0] = q0.x, sb[1] = q0.y, sb[2] = q0.z, sb[3] = q0.w, sb[4] = q1.x;
#pragma unroll
      for (int m = 0; m < NCH; ++m)
        ch[m] = a.chunks[d.chunk0 + ((v + m * NT) >> 3)];
    };
    uint64_t meta, pmeta = 0;
    double dv;
    uint32_t sb[5], psb[5] = {0, 0, 0, 0, 0};
    int32_t ch[NCH];
    lane_data(d0, meta, dv, sb, ch);
    double pend = 0.0, pend_x = 0.0;
    int prow = -1;
    bool have_p = false;
    for (int si = 0; si < ns; ++si) {
      const StepDesc d2 = a.steps[s0 + si + 2]; // (two idle ones behind the last)
      // ---- this step's loads, all of them --------------------------------
      double xs[NCH][2];
#pragma unroll
      for (int m = 0; m < NCH; ++m) {
        const int i = v + m * NT;
        const int64_t col = (int64_t)ch[m] * 16 + (i & 7) * 2;
        const int64_t c0 = col < a.num_cols ? col : a.num_cols - 1;
        const int64_t c1 = col + 1 < a.num_cols ? col + 1 : a.num_cols - 1;
        xs[m][0] = x[c0];
        xs[m][1] = x[c1];
      }
      const bool hasT = d0.tile >= 0; // uniform
      const int lenA = (int)(meta & 0xffu), lenBc = (int)((meta >> 8) & 0xffu);
      const int lenBs = (int)((meta >> 16) & 0xffu);
      const int lenCc = (int)((pmeta >> 24) & 0xffu), lenCs = (int)((pmeta >> 32) & 0xffu);
      const uint32_t tbase = __builtin_amdgcn_readfirstlane(
          a.sbp[(int64_t)(hasT ? d0.tile : 0) * NSL * 8]);
      uint32_t bA = __builtin_amdgcn_readfirstlane(sb[0]);
      uint32_t bBc = __builtin_amdgcn_readfirstlane(sb[1]);
      uint32_t bBs = __builtin_amdgcn_readfirstlane(sb[2]);
      uint32_t bCc = __builtin_amdgcn_readfirstlane(psb[3]);
      uint32_t bCs = __builtin_amdgcn_readfirstlane(psb[4]);
      Grp<GA, uint32_t, true> gA0, gA1;
      Grp<GC, uint16_t, false> gBc, gCc;
      Grp<GS, uint16_t, true> gBs, gCs;
      issue<GA, uint32_t, true>(gA0, lenA, 0, bA, a.a_val, a.a_code);
      issue<GC, uint16_t, false>(gBc, lenBc, 0, bBc, nullptr, a.bc_code);
      issue<GS, uint16_t, true>(gBs, lenBs, 0, bBs, a.bs_val, a.bs_code);
      issue<GC, uint16_t, false>(gCc, lenCc, 0, bCc, nullptr, a.cc_code);
      issue<GS, uint16_t, true>(gCs, lenCs, 0, bCs, a.cs_val, a.cs_code);
      // ---- the next step's lane data -----------------------------------------
      uint64_t nmeta;
      double ndv;
      uint32_t nsb[5];
      int32_t nch[NCH];
      lane_data(d1, nmeta, ndv, nsb, nch);
      // ---- stage x ------------------------------------------------------------
      __syncthreads(); // the previous step's consumers are done with stash, sx
#pragma unroll
      for (int m = 0; m < NCH; ++m) {
        const int i = v + m * NT;
        if (i < d0.nchunks * 8) {
          sx[(i >> 3) * 16 + (i & 7) * 2] = xs[m][0];
          sx[(i >> 3) * 16 + (i & 7) * 2 + 1] = xs[m][1];
        }
      }
      // (the second group of A: behind the staging stores, whose registers it
      // takes over; its round trip runs under the barrier and the first group)
      issue<GA, uint32_t, true>(gA1, lenA, GA, bA, a.a_val, a.a_code);
      __syncthreads();
      // ---- A: the lower part, products to the stash --------------------------
      const bool valid = (meta >> 63) != 0;
      const int loc = (int)((meta >> 40) & 0xffffu);
      double x_own = 0.0, s = 0.0;
      if (hasT) {
        if (valid) {
          x_own = sx[d0.own0 + loc];
          s = dv * x_own;
        }
        auto consumeA = [&](const Grp<GA, uint32_t, true>& g) {
#pragma unroll
          for (int q = 0; q < GA; ++q) {
            const bool act = (g.act >> q) & 1u;
            const double xv = sx[g.c[q] & 0xffffu];
            s = act ? s + g.v[q] * xv : s;
            if (act)
              stash[g.c[q] >> 16] = g.v[q] * x_own;
          }
        };
        consumeA(gA0);
        consumeA(gA1);
        for (int k0 = 2 * GA; __ballot(k0 < lenA) != 0ull; k0 += GA) { // (rare)
          issue<GA, uint32_t, true>(gA0, lenA, k0, bA, a.a_val, a.a_code);
          consumeA(gA0);
        }
      }
      __syncthreads(); // the tile's products are in the stash
      // ---- B: T's rows take their column's early entries ---------------------
      if (hasT) {
#pragma unroll
        for (int q = 0; q < GC; ++q)
          s = ((gBc.act >> q) & 1u) ? s + stash[gBc.c[q]] : s;
        for (int k0 = GC; __ballot(k0 < lenBc) != 0ull; k0 += GC) {
          issue<GC, uint16_t, false>(gBc, lenBc, k0, bBc, nullptr, a.bc_code);
#pragma unroll
          for (int q = 0; q < GC; ++q)
            s = ((gBc.act >> q) & 1u) ? s + stash[gBc.c[q]] : s;
        }
#pragma unroll
        for (int q = 0; q < GS; ++q)
          s = ((gBs.act >> q) & 1u) ? s + gBs.v[q] * sx[gBs.c[q]] : s;
        for (int k0 = GS; __ballot(k0 < lenBs) != 0ull; k0 += GS) {
          issue<GS, uint16_t, true>(gBs, lenBs, k0, bBs, a.bs_val, a.bs_code);
#pragma unroll
          for (int q = 0; q < GS; ++q)
            s = ((gBs.act >> q) & 1u) ? s + gBs.v[q] * sx[gBs.c[q]] : s;
        }
      }
      // ---- C: the previous tile's rows are finished --------------------------
      if (have_p) {
#pragma unroll
        for (int q = 0; q < GC; ++q)
          pend = ((gCc.act >> q) & 1u) ? pend + stash[gCc.c[q]] : pend;
        for (int k0 = GC; __ballot(k0 < lenCc) != 0ull; k0 += GC) {
          issue<GC, uint16_t, false>(gCc, lenCc, k0, bCc, nullptr, a.cc_code);
#pragma unroll
          for (int q = 0; q < GC; ++q)
            pend = ((gCc.act >> q) & 1u) ? pend + stash[gCc.c[q]] : pend;
        }
#pragma unroll
        for (int q = 0; q < GS; ++q)
          pend = ((gCs.act >> q) & 1u) ? pend + gCs.v[q] * sx[gCs.c[q]] : pend;
        for (int k0 = GS; __ballot(k0 < lenCs) != 0ull; k0 += GS) {
          issue<GS, uint16_t, true>(gCs, lenCs, k0, bCs, a.cs_val, a.cs_code);
#pragma unroll
          for (int q = 0; q < GS; ++q)
            pend = ((gCs.act >> q) & 1u) ? pend + gCs.v[q] * sx[gCs.c[q]] : pend;
        }
        if (prow >= 0) {
          y[prow] = pend;
          dot_acc += pend_x * pend;
        }
      }
      // ---- rotate ----------------------------------------------------------------
      pend = s;
      pend_x = x_own;
      prow = (hasT && valid) ? d0.row0 + loc : -1;
      have_p = hasT;
      pmeta = meta;
#pragma unroll
      for (int q = 0; q < 5; ++q)
        psb[q] = sb[q], sb[q] = nsb[q];
      meta = nmeta;
      dv = ndv;
#pragma unroll
      for (int m = 0; m < NCH; ++m)
        ch[m] = nch[m];
      d0 = d1;
      d1 = d2;
    }
  }
  if (dot_partials) {
    double r = dot_acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      r += __shfl_down(r, o, 64);
    if (lane == 0)
      s_red[slice] = r;
    __syncthreads();
    if (v == 0) {
      double t = 0.0;
      for (int w = 0; w < NT / 64; ++w)
        t += s_red[w];
      dot_partials[blockIdx.x] = t;
    }
  }
}
} // namespace

extern "C" int march2_spmv(int grid, int nunits, const int32_t* unit_step0,
                           const void* steps, const int32_t* chunks,
                           const uint64_t* meta, const double* dval,
                           const uint32_t* sbp, const double* a_val,
                           const uint32_t* a_code, const uint16_t* bc_code,
                           const double* bs_val, const uint16_t* bs_code,
                           const uint16_t* cc_code, const double* cs_val,
                           const uint16_t* cs_code, int64_t num_cols, const double* x,
                           double* y, double* dot_partials, void* stream)
{
  MarchArgs2 a;
  a.nunits = nunits;
  a.unit_step0 = unit_step0;
  a.steps = static_cast<const StepDesc*>(steps);
  a.chunks = chunks;
  a.meta = meta;
  a.dval = dval;
  a.sbp = sbp;
  a.a_val = a_val, a.a_code = a_code, a.bc_code = bc_code;
  a.bs_val = bs_val, a.bs_code = bs_code, a.cc_code = cc_code;
  a.cs_val = cs_val, a.cs_code = cs_code;
  a.num_cols = num_cols;
  const size_t lds = sizeof(double) * (kStash + kXcap);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(march2_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess)
    return (int)e;
  hipLaunchKernelGGL(march2_kernel, dim3(grid), dim3(NT), lds,
                     static_cast<hipStream_t>(stream), a, x, y, dot_partials);
  return (int)hipGetLastError();
}
