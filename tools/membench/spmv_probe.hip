// Ablation probe for the row-block CSR SpMV: the same launch shape as
// csr_rowblock_kernel (256 rows per workgroup, 512-entry tiles, 16-B value +
// 8-B column loads) with the stages switched on one at a time, on a 7-entries-
// per-row periodic stencil (same memory behaviour as the 7-point Poisson
// matrix).  Shows where the rate drops between "pure stream" and "full SpMV".
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/membench/spmv_probe tools/membench/spmv_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr int kRows = 256, kTile = 512;

// staged-x layout of MODE 8: five windows per 256-row block
//   [r0-n^2,+256) [r0-n,+256) [r0-2,+260) [r0+n,+256) [r0+n^2,+256)
__device__ __host__ inline int win_off(int w) { return w < 3 ? 256 * w : 256 * w + 4; }
constexpr int kStaged = 4 * 256 + 260;

// MODE 12 layout: five windows per 64-row wave block
//   [r0-n^2,+64) [r0-n,+64) [r0-2,+68) [r0+n,+64) [r0+n^2,+64)
__device__ __host__ inline int wwin_off(int w) { return w < 3 ? 64 * w : 64 * w + 4; }
constexpr int kWStaged = 4 * 64 + 68;

__global__ void gen(int n, long N, int* rowptr, int* colind, double* values,
                    double* x, unsigned short* lidx, unsigned short* lidx64)
{
  const long stride = (long)gridDim.x * blockDim.x;
  const long n2 = (long)n * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const long off[7] = {-n2, -(long)n, -1, 0, 1, n, n2};
    rowptr[i] = (int)(7 * i);
    for (int k = 0; k < 7; ++k) {
      long c = i + off[k];
      c = c < 0 ? c + N : (c >= N ? c - N : c);
      colind[7 * i + k] = (int)c;
      values[7 * i + k] = k == 3 ? 6.0 : -1.0;
    }
    {
      const int t = (int)(i % kRows);
      const int l[7] = {win_off(0) + t, win_off(1) + t, win_off(2) + t + 1,
                        win_off(2) + t + 2, win_off(2) + t + 3, win_off(3) + t,
                        win_off(4) + t};
      for (int k = 0; k < 7; ++k)
        lidx[7 * i + k] = (unsigned short)l[k];
      const int tw = (int)(i % 64);
      const int lw[7] = {wwin_off(0) + tw, wwin_off(1) + tw, wwin_off(2) + tw + 1,
                         wwin_off(2) + tw + 2, wwin_off(2) + tw + 3,
                         wwin_off(3) + tw, wwin_off(4) + tw};
      for (int k = 0; k < 7; ++k)
        lidx64[7 * i + k] = (unsigned short)lw[k];
    }
    x[i] = 1.0 + 1e-3 * (double)(i % 1000);
    if (i == N - 1)
      rowptr[N] = (int)(7 * N);
  }
}

template <bool NT, typename P>
__device__ __forceinline__ P sld(const P* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// SCHED: 0 grid-stride, 1 XCD groups of 16 row blocks, 2 contiguous range per WG
__device__ const long* g_wtab = nullptr; // MODE 11: 8 longs per row block
__device__ const int* g_order = nullptr; // SCHED 3: table, -1 = skip
__device__ int g_order_len = 0;

template <int SCHED>
__device__ __forceinline__ bool next_block(int it, int nrb, int& rb)
{
  if (SCHED == 3) {
    const int q = it * gridDim.x + blockIdx.x;
    if (q >= g_order_len)
      return false;
    rb = g_order[q];
    return true;
  }
  if (SCHED == 0) {
    rb = it * gridDim.x + blockIdx.x;
    return rb < nrb;
  } else if (SCHED == 1) {
    constexpr int G = 16;
    const int q = it * gridDim.x + blockIdx.x; // virtual index
    const int super = q / (8 * G), r = q % (8 * G);
    rb = super * 8 * G + (r & 7) * G + (r >> 3);
    return q < nrb && rb < nrb; // nrb is a multiple of 128 in this probe
  } else {
    const int per = (nrb + gridDim.x - 1) / gridDim.x;
    rb = blockIdx.x * per + it;
    return it < per && rb < nrb;
  }
}

// MODE 0 stream, 1 +rowptr, 2 +gather, 3 +LDS row sums (full), 4 = 3 without
// the trailing barrier per tile (double-buffered LDS)
template <int MODE, bool NT, int SCHED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void probe(int nrows, const int* __restrict__ rowptr,
                                             const int* __restrict__ colind,
                                             const double* __restrict__ values,
                                             const double* __restrict__ x,
                                             double* __restrict__ y,
                                             const unsigned short* __restrict__ lidx,
                                             int n,
                                             const unsigned short* __restrict__ lidx64)
{
  if (MODE == 12) {
    // wave-granular staged-x SpMV: every wave owns 64-row blocks, its own
    // windows and product tile in LDS; no workgroup barrier anywhere
    __shared__ double w_x[4][kWStaged];
    __shared__ double w_prod[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* sx = w_x[wave];
    double* sp = w_prod[wave];
    const long N = nrows, n2l = (long)n * n;
    const long nwb = N / 64;                                  // 64-row blocks
    const long wstride = (long)gridDim.x * 4;
    for (long wb = (long)blockIdx.x * 4 + wave; wb < nwb; wb += wstride) {
      const long r0 = wb * 64;
      const long a = 7 * r0, b = a + 448;
      const long starts[5] = {r0 - n2l, r0 - n, r0 - 2, r0 + n, r0 + n2l};
      // stage: 162 pairs, all loads first
      f64x2 xv[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int f = lane + 64 * m; // pair index
        xv[m] = f64x2{0.0, 0.0};
        if (f < kWStaged / 2) {
          const int w = f < 32 ? 0 : f < 64 ? 1 : f < 98 ? 2 : f < 130 ? 3 : 4;
          const int e = f - wwin_off(w) / 2;
          const int len = w == 2 ? 68 : 64;
          long st = starts[w];
          st = st < 0 ? 0 : (st + len > N ? N - len : st);
          xv[m] = *reinterpret_cast<const f64x2*>(x + st + 2 * e);
        }
      }
      // both tiles' matrix loads (2 x (16 B + 4 B) per lane)
      f64x2 v[2][2];
      unsigned int li[2][2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          long j = a + tl * 256 + c * 128 + 2 * lane;
          j = j < b - 2 ? j : b - 2;
          v[tl][c] = sld<NT>(reinterpret_cast<const f64x2*>(values + j));
          li[tl][c] = sld<NT>(reinterpret_cast<const unsigned int*>(lidx64 + j));
        }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int f = lane + 64 * m;
        if (f < kWStaged / 2)
          *reinterpret_cast<f64x2*>(&sx[2 * f]) = xv[m];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double acc = 0.0;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const long j = a + tl * 256 + c * 128 + 2 * lane;
          const double p0 = j < b ? v[tl][c].x * sx[li[tl][c] & 0xffffu] : 0.0;
          const double p1 = j + 1 < b ? v[tl][c].y * sx[li[tl][c] >> 16] : 0.0;
          *reinterpret_cast<f64x2*>(&sp[c * 128 + 2 * lane]) = f64x2{p0, p1};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rlo = 7 * lane - tl * 256, rhi = rlo + 7;
        const int klo = rlo > 0 ? rlo : 0, khi = rhi < 256 ? rhi : 256;
        for (int k = klo; k < khi; ++k)
          acc += sp[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      y[r0 + lane] = acc;
    }
    return;
  }
  __shared__ double s_x[MODE >= 8 && MODE != 9 && MODE != 12 ? kStaged : 2];
  __shared__ double s_prod[2][kTile];
  __shared__ int s_rp[kRows + 1];
  __shared__ long s_win[8];
  const int t = threadIdx.x;
  const int nrb = nrows / kRows;
  int rb;
  int rb_next = 0; // SCHED 4 = SCHED 3 with the table entry fetched one
                   // iteration ahead (off the dependent chain)
  if (SCHED == 4)
    rb_next = (int)blockIdx.x < g_order_len ? g_order[blockIdx.x] : -2;
  for (int it = 0;; ++it) {
    if (SCHED == 4) {
      rb = rb_next;
      if (rb == -2)
        break;
      const long qn = (long)(it + 1) * gridDim.x + blockIdx.x;
      rb_next = qn < g_order_len ? g_order[qn] : -2;
    } else if (!next_block<SCHED>(it, nrb, rb)) {
      break;
    }
    if ((SCHED == 3 || SCHED == 4) && rb < 0)
      continue; // padding slot
    const long r0 = (long)rb * kRows;
    long a = 7 * r0, b = 7 * (r0 + kRows);
    int lo = 7 * t, hi = 7 * t + 7; // relative to a when MODE < 1
    if (MODE >= 1) {
      if (MODE >= 3) {
        __syncthreads();
        s_rp[t] = rowptr[r0 + t];
        if (t == 0)
          s_rp[kRows] = rowptr[r0 + kRows];
        __syncthreads();
        a = s_rp[0];
        b = s_rp[kRows];
        lo = s_rp[t];
        hi = s_rp[t + 1];
      } else {
        a = rowptr[r0];
        b = rowptr[r0 + kRows];
      }
    }
    if (MODE == 9) {
      // whole row block in ONE round: 4 x (16-B value + 8-B column) loads per
      // lane issued up front, then 8 gathers, one LDS exchange
      __shared__ double s_big[4 * kTile];
      const double* vb = values + (a & ~1L); // uniform bases, 32-bit offsets
      const int* cb = colind + (a & ~1L);
      const int span = (int)(b - (a & ~1L));
      f64x2 v[4];
      i32x2 c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int off = u * kTile + 2 * t;
        off = off < span - 2 ? off : span - 2;
        v[u] = sld<NT>(reinterpret_cast<const f64x2*>(vb + off));
        c[u] = sld<NT>(reinterpret_cast<const i32x2*>(cb + off));
      }
      double xg[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xg[2 * u] = x[c[u].x];
        xg[2 * u + 1] = x[c[u].y];
      }
      __syncthreads(); // previous block done with s_big
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int off = u * kTile + 2 * t;
        s_big[off] = off < span ? v[u].x * xg[2 * u] : 0.0;
        s_big[off + 1] = off + 1 < span ? v[u].y * xg[2 * u + 1] : 0.0;
      }
      __syncthreads();
      double acc9 = 0.0;
      const int rl = (int)(a - (a & ~1L)) + 7 * t;
#pragma unroll
      for (int k = 0; k < 7; ++k)
        acc9 += s_big[rl + k];
      y[r0 + t] = acc9;
      continue;
    }
    if (MODE >= 8) {
      // stage the five x windows of this row block (coalesced 16-B loads)
      const long N = nrows, n2l = (long)n * n;
      long starts[5] = {r0 - n2l, r0 - n, r0 - 2, r0 + n, r0 + n2l};
      __syncthreads(); // previous block done with s_x
      if (MODE >= 10) { // row pointer through memory and LDS, as the library does
        s_rp[t] = rowptr[r0 + t];
        if (t == 0)
          s_rp[kRows] = rowptr[r0 + kRows];
        if (MODE >= 11 && t < 5) // window table of this block from memory
          s_win[t] = g_wtab[(long)rb * 8 + t];
        __syncthreads();
        a = s_rp[0];
        b = s_rp[kRows];
        if (MODE >= 11)
          for (int w = 0; w < 5; ++w)
            starts[w] = s_win[w];
      }
      for (int f = t; f < kStaged / 2; f += 256) {
        int w = f < 128 ? 0 : f < 256 ? 1 : f < 386 ? 2 : f < 514 ? 3 : 4;
        const int e = f - win_off(w) / 2;
        const int len = w == 2 ? 260 : 256;
        long st = starts[w];
        st = st < 0 ? 0 : (st + len > N ? N - len : st); // probe: clamp
        const f64x2 v = *reinterpret_cast<const f64x2*>(x + st + 2 * e);
        *reinterpret_cast<f64x2*>(&s_x[win_off(w) + 2 * e]) = v;
      }
      __syncthreads();
      double acc8 = 0.0;
      for (long base = a & ~1L; base < b; base += kTile) {
        const long j = base + 2 * t;
        const long jl = j < b - 2 ? j : b - 2;
        const f64x2 v = sld<NT>(reinterpret_cast<const f64x2*>(values + jl));
        const unsigned int li
            = sld<NT>(reinterpret_cast<const unsigned int*>(lidx + jl));
        const double x0 = s_x[li & 0xffffu], x1 = s_x[li >> 16];
        const double p0 = j < b ? v.x * x0 : 0.0, p1 = j + 1 < b ? v.y * x1 : 0.0;
        if (base != (a & ~1L))
          __syncthreads();
        s_prod[0][2 * t] = p0;
        s_prod[0][2 * t + 1] = p1;
        __syncthreads();
        const long rlo = MODE >= 10 ? (long)s_rp[t] : a + 7L * t;
        const long rhi = MODE >= 10 ? (long)s_rp[t + 1] : rlo + 7;
        const long klo = (rlo > base ? rlo : base) - base;
        const long khi = (rhi < base + kTile ? rhi : base + kTile) - base;
        for (long k = klo; k < khi; ++k)
          acc8 += s_prod[0][k];
      }
      y[r0 + t] = acc8;
      continue;
    }
    double acc = 0.0;
    int buf = 0;
    for (long base = a & ~1L; base < b; base += kTile) {
      const long j = base + 2 * t;
      const long jl = j < b - 2 ? j : b - 2;
      const f64x2 v = sld<NT>(reinterpret_cast<const f64x2*>(values + jl));
      const i32x2 c = sld<NT>(reinterpret_cast<const i32x2*>(colind + jl));
      if (MODE <= 1) {
        acc += v.x + v.y + (double)(c.x ^ c.y);
      } else {
        const double x0 = x[c.x], x1 = x[c.y];
        const double p0 = j < b ? v.x * x0 : 0.0, p1 = j + 1 < b ? v.y * x1 : 0.0;
        if (MODE == 2) {
          acc += p0 + p1;
        } else if (MODE == 5 || MODE == 6) {
          // 5: LDS round trip of the own products behind ONE barrier
          // 6: the same with the second barrier of the real kernel
          if (MODE == 6 && base != (a & ~1L))
            __syncthreads();
          s_prod[buf][2 * t] = p0;
          s_prod[buf][2 * t + 1] = p1;
          __syncthreads();
          acc += s_prod[buf][(2 * t + 64) & (kTile - 1)]
                 + s_prod[buf][(2 * t + 65) & (kTile - 1)];
          if (MODE == 5)
            buf ^= 1;
        } else if (MODE == 7) {
          // per-row loop over the tile with bounds from arithmetic (no s_rp)
          if (base != (a & ~1L))
            __syncthreads();
          s_prod[buf][2 * t] = p0;
          s_prod[buf][2 * t + 1] = p1;
          __syncthreads();
          const long rlo = a + 7L * t, rhi = rlo + 7;
          const long klo = (rlo > base ? rlo : base) - base;
          const long khi = (rhi < base + kTile ? rhi : base + kTile) - base;
          for (long k = klo; k < khi; ++k)
            acc += s_prod[buf][k];
        } else {
          if (MODE == 3 && base != (a & ~1L))
            __syncthreads();
          s_prod[buf][2 * t] = p0;
          s_prod[buf][2 * t + 1] = p1;
          __syncthreads();
          const long klo = (lo > base ? lo : base) - base;
          const long khi = (hi < base + kTile ? hi : base + kTile) - base;
          for (long k = klo; k < khi; ++k)
            acc += s_prod[buf][k];
          if (MODE == 4)
            buf ^= 1;
        }
      }
    }
    y[r0 + t] = acc;
  }
}

int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const long N = (long)n * n * n, nnz = 7 * N;
  if (N % (kRows * 128) != 0) {
    fprintf(stderr, "n^3 must be a multiple of %d\n", kRows * 128);
    return 1;
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int *rowptr, *colind;
  double *values, *x, *y;
  CK(hipMalloc(&rowptr, (N + 1) * 4));
  CK(hipMalloc(&colind, nnz * 4));
  CK(hipMalloc(&values, nnz * 8));
  CK(hipMalloc(&x, N * 8));
  CK(hipMalloc(&y, N * 8));
  unsigned short *lidx, *lidx64;
  CK(hipMalloc(&lidx, nnz * 2));
  CK(hipMalloc(&lidx64, nnz * 2));
  gen<<<cus * 8, 256>>>(n, N, rowptr, colind, values, x, lidx, lidx64);
  CK(hipDeviceSynchronize());
  // SCHED 3: every XCD sweeps its own band of `yc` grid lines through all z
  // (row blocks of one XCD = virtual indices q with q % 8 == xcd)
  const int yc = argc > 3 ? atoi(argv[3]) : 64;
  {
    const int lines_per_plane = n, rb_per_line = n / kRows; // n % 256 == 0 here
    std::vector<std::vector<int>> lists(8);
    const int nbands = (lines_per_plane + yc - 1) / yc;
    for (int band = 0; band < nbands; ++band)
      for (int z = 0; z < n; ++z)
        for (int yy = band * yc; yy < (band + 1) * yc && yy < lines_per_plane; ++yy)
          for (int h = 0; h < (rb_per_line > 0 ? rb_per_line : 1); ++h)
            lists[band % 8].push_back(((z * lines_per_plane + yy) * (long)n) / kRows + h);
    size_t longest = 0;
    for (auto& l : lists)
      longest = l.size() > longest ? l.size() : longest;
    std::vector<int> order(8 * longest, -1);
    for (int k = 0; k < 8; ++k)
      for (size_t i = 0; i < lists[k].size(); ++i)
        order[8 * i + k] = lists[k][i];
    int* d_order;
    CK(hipMalloc(&d_order, order.size() * 4));
    CK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    const int len = (int)order.size();
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_order), &d_order, sizeof(d_order)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_order_len), &len, sizeof(len)));
  }
  {
    const long nrbl = N / kRows, n2l = (long)n * n;
    std::vector<long> wt((size_t)nrbl * 8, 0);
    for (long k = 0; k < nrbl; ++k) {
      const long r0 = k * kRows;
      const long st[5] = {r0 - n2l, r0 - n, r0 - 2, r0 + n, r0 + n2l};
      for (int w = 0; w < 5; ++w)
        wt[(size_t)k * 8 + w] = st[w];
    }
    long* d_wt;
    CK(hipMalloc(&d_wt, wt.size() * 8));
    CK(hipMemcpy(d_wt, wt.data(), wt.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wtab), &d_wt, sizeof(d_wt)));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = nnz * 12.0 + (N + 1) * 4.0 + N * 16.0;
  auto run = [&](const char* name, int wpc, const std::function<void(int)>& f) {
    const int grid = cus * wpc;
    f(grid);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int round = 0; round < 3; ++round) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r)
        f(grid);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / reps < best)
        best = ms / reps;
    }
    printf("{\"test\": \"%s\", \"n\": %d, \"wg_per_cu\": %d, \"ms\": %.4f, "
           "\"GB/s\": %.1f}\n", name, n, wpc, best, bytes / best / 1e6);
    fflush(stdout);
  };
#define P(MODE, NT, SCHED)                                                     \
  run("mode" #MODE "_nt" #NT "_sched" #SCHED, wpc, [&](int grid) {             \
    probe<MODE, NT, SCHED><<<grid, 256>>>((int)N, rowptr, colind, values, x, y, \
                                          lidx, n, lidx64);                      \
  })
  for (int wpc : {8}) {
    P(0, false, 0); P(0, true, 0); P(0, false, 1); P(0, true, 1); P(0, false, 2); P(0, true, 2);
    P(1, false, 1); P(1, true, 1); P(1, true, 2);
    P(2, false, 0); P(2, false, 1); P(2, true, 1); P(2, false, 2); P(2, true, 2);
    P(3, false, 0); P(3, false, 1); P(3, true, 1); P(3, false, 2); P(3, true, 2);
    P(4, false, 1); P(4, true, 1); P(4, true, 2);
    P(12, false, 0); P(12, true, 0);
    P(10, false, 1); P(10, true, 1); P(11, false, 1); P(11, true, 1);
    P(9, false, 0); P(9, false, 1); P(9, true, 1);
    P(8, false, 0); P(8, false, 1); P(8, true, 1); P(8, false, 3); P(8, true, 3);
    P(5, true, 3); P(6, true, 3); P(7, true, 3); P(5, false, 1); P(6, false, 1); P(7, false, 1);
    P(2, true, 4); P(3, false, 4); P(3, true, 4); P(4, true, 4); P(7, true, 4);
    P(0, false, 3); P(2, false, 3); P(2, true, 3); P(3, false, 3); P(3, true, 3);
    P(4, false, 3); P(4, true, 3);
  }
  return 0;
}
