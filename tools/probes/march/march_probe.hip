// PROBE (round 6, not part of the product): the MARCHED exchange for symmetric
// storage of a matrix without lattice structure as a
// stand-alone kernel on a plan built on the HOST (march_build.py), to measure
// -- before a device-side builder is written -- whether streaming a stored value
// once where both its rows sit in one workgroup beats the merged sliced jagged
// form (spmv_sjds.hip, 20 B per stored entry) on the same matrix.
//
//   y = (L + D + L^T) x for alpha = 1, beta = 0 (Matrix::mult, cg()), summed in
//   the reference's order (csr_kernels.cpp:26-40 seen from the row): d_i x_i,
//   the row's stored lower entries in order, then the entries of its column in
//   ascending (r, position).
//
// A workgroup of 1024 lanes (lane = row of a TILE of <= 1024 rows) walks a UNIT:
// the tiles [a + l S, a + l S + 1024), l = l0 .. l1, S = the matrix's far
// offset.  Step s (tile T, previous tile P):
//   stage   the 16-column chunks of x the step needs                     barrier
//   A       T's lower entries (stream JA: value 8 B + code 2 B): the row's sum;
//           every product v * x_r (r = the lane's own row) goes to the STASH in
//           LDS at the entry's position in the tile's stream               barrier
//   B       T's rows go on with their column's entries: those whose source row
//           is in T read the stash (stream JBc: 2 B), then the streamed ones
//           before the next tile (JBs: 10 B); the sum stays in a register
//   C       P's rows (the same lanes, one step later) take the entries whose
//           source row is in T from the stash (JCc: 2 B), then the streamed rest
//           (JCs: 10 B), and store y                                         barrier
// A captured entry costs 10 + 2 B instead of 20.
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
constexpr int kXcap = MARCH_XCAP;   // staged x elements
constexpr int G = 8;          // entries of a lane in flight

struct MarchArgs {
  int nunits;
  const int32_t* unit_step0; // [nunits + 1]
  const int32_t* step_tile;  // per step: tile, or -1 (the unit's drain step)
  const int32_t* step_chunk0; // [nsteps + 1] into chunks
  const int32_t* step_own0;  // staged position of the tile's first row
  const int32_t* chunks;
  const int32_t* tile_row0;  // first row of a tile
  const uint64_t* meta;      // per tile and lane: lenA | lenBc<<8 | lenBs<<16 |
                             // lenCc<<24 | lenCs<<32 | row in tile<<40 | valid<<63
  const uint32_t* sb[5];     // per stream, tile and slice: first entry
  const double* a_val;
  const uint16_t* a_code;
  const uint16_t* bc_code;
  const double* bs_val;
  const uint16_t* bs_code;
  const uint16_t* cc_code;
  const double* cs_val;
  const uint16_t* cs_code;
  const double* diag;
  int64_t num_cols;
};

// one jagged stream of the lane's row, G entries in flight.
//   VAL: the stream has values; FROM_STASH: the code indexes the stash, else
//   the staged x; TO_STASH: v * x_own goes to stash[position in the tile]
template <bool VAL, bool FROM_STASH, bool TO_STASH>
__device__ __forceinline__ double march_pass(double s, int len, uint32_t base,
                                             const double* __restrict__ val,
                                             const uint16_t* __restrict__ code,
                                             const double* sx, double* stash,
                                             uint32_t tbase, double x_own)
{
  for (int k0 = 0; __ballot(k0 < len) != 0ull; k0 += G) {
    uint32_t pos[G];
    bool act[G];
    double vv[G];
    uint16_t cc[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      act[u] = k0 + u < len;
      const uint64_t m = __ballot(act[u]);
      const uint32_t below = __builtin_amdgcn_mbcnt_hi(
          (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      pos[u] = base + (act[u] ? below : 0u);
      base += (uint32_t)__popcll(m);
      cc[u] = code[pos[u]];
      if constexpr (VAL)
        vv[u] = val[pos[u]];
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const double xv = FROM_STASH ? stash[cc[u]] : sx[cc[u]];
      const double p = VAL ? vv[u] * xv : xv;
      s = act[u] ? s + p : s;
      if constexpr (TO_STASH) {
        if (act[u])
          stash[pos[u] - tbase] = vv[u] * x_own;
      }
    }
  }
  return s;
}

__global__ __launch_bounds__(NT) void march_kernel(MarchArgs a,
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
    double pend = 0.0, pend_x = 0.0;
    int prow = -1, ptile = -1;
    uint64_t pmeta = 0;
    for (int si = 0; si < ns; ++si) {
      const int st = s0 + si;
      const int T = a.step_tile[st];
      const int c0 = a.step_chunk0[st], nc = a.step_chunk0[st + 1] - c0;
      __syncthreads(); // the previous step's consumers are done with stash, sx
      for (int i = v; i < nc * 8; i += NT) {
        const int64_t col = (int64_t)a.chunks[c0 + (i >> 3)] * 16 + (i & 7) * 2;
        double x0 = 0.0, x1 = 0.0;
        if (col < a.num_cols)
          x0 = x[col];
        if (col + 1 < a.num_cols)
          x1 = x[col + 1];
        sx[(i >> 3) * 16 + (i & 7) * 2] = x0;
        sx[(i >> 3) * 16 + (i & 7) * 2 + 1] = x1;
      }
      uint64_t meta = 0;
      int row = -1;
      double dd = 0.0;
      if (T >= 0) {
        meta = a.meta[(int64_t)T * NT + v];
        if (meta >> 63) {
          row = a.tile_row0[T] + (int)((meta >> 40) & 0xffffu);
          dd = a.diag[row];
        }
      }
      __syncthreads();
      double s = 0.0, x_own = 0.0;
      if (T >= 0) { // (uniform)
        if (row >= 0) {
          x_own = sx[a.step_own0[st] + (int)((meta >> 40) & 0xffffu)];
          s = dd * x_own;
        }
        const uint32_t tbase = a.sb[0][(int64_t)T * NSL];
        s = march_pass<true, false, true>(s, (int)(meta & 0xffu),
                                          a.sb[0][(int64_t)T * NSL + slice], a.a_val,
                                          a.a_code, sx, stash, tbase, x_own);
      }
      __syncthreads(); // the tile's products are in the stash
      if (T >= 0) {
        s = march_pass<false, true, false>(s, (int)((meta >> 8) & 0xffu),
                                           a.sb[1][(int64_t)T * NSL + slice], nullptr,
                                           a.bc_code, sx, stash, 0u, 0.0);
        s = march_pass<true, false, false>(s, (int)((meta >> 16) & 0xffu),
                                           a.sb[2][(int64_t)T * NSL + slice], a.bs_val,
                                           a.bs_code, sx, stash, 0u, 0.0);
      }
      if (ptile >= 0) {
        pend = march_pass<false, true, false>(pend, (int)((pmeta >> 24) & 0xffu),
                                              a.sb[3][(int64_t)ptile * NSL + slice],
                                              nullptr, a.cc_code, sx, stash, 0u, 0.0);
        pend = march_pass<true, false, false>(pend, (int)((pmeta >> 32) & 0xffu),
                                              a.sb[4][(int64_t)ptile * NSL + slice],
                                              a.cs_val, a.cs_code, sx, stash, 0u, 0.0);
        if (prow >= 0) {
          y[prow] = pend;
          dot_acc += pend_x * pend;
        }
      }
      pend = s;
      pend_x = x_own;
      prow = row;
      pmeta = meta;
      ptile = T;
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

extern "C" int march_spmv(int grid, int nunits, const int32_t* unit_step0,
                          const int32_t* step_tile, const int32_t* step_chunk0,
                          const int32_t* step_own0, const int32_t* chunks,
                          const int32_t* tile_row0, const uint64_t* meta,
                          const uint32_t* sb0, const uint32_t* sb1, const uint32_t* sb2,
                          const uint32_t* sb3, const uint32_t* sb4, const double* a_val,
                          const uint16_t* a_code, const uint16_t* bc_code,
                          const double* bs_val, const uint16_t* bs_code,
                          const uint16_t* cc_code, const double* cs_val,
                          const uint16_t* cs_code, const double* diag, int64_t num_cols,
                          const double* x, double* y, double* dot_partials, void* stream)
{
  MarchArgs a;
  a.nunits = nunits;
  a.unit_step0 = unit_step0;
  a.step_tile = step_tile;
  a.step_chunk0 = step_chunk0;
  a.step_own0 = step_own0;
  a.chunks = chunks;
  a.tile_row0 = tile_row0;
  a.meta = meta;
  a.sb[0] = sb0, a.sb[1] = sb1, a.sb[2] = sb2, a.sb[3] = sb3, a.sb[4] = sb4;
  a.a_val = a_val, a.a_code = a_code, a.bc_code = bc_code;
  a.bs_val = bs_val, a.bs_code = bs_code, a.cc_code = cc_code;
  a.cs_val = cs_val, a.cs_code = cs_code;
  a.diag = diag;
  a.num_cols = num_cols;
  const size_t lds = sizeof(double) * (kStash + kXcap);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(march_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess)
    return (int)e;
  hipLaunchKernelGGL(march_kernel, dim3(grid), dim3(NT), lds,
                     static_cast<hipStream_t>(stream), a, x, y, dot_partials);
  return (int)hipGetLastError();
}

extern "C" int march_limits(int* stash, int* xcap)
{
  *stash = kStash;
  *xcap = kXcap;
  return 0;
}
