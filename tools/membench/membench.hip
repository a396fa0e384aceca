// Streaming-pattern probe for MI355X: which access shape reaches the highest
// HBM rate?  Read-only, write-only and copy kernels in several shapes, timed
// with HIP events, interleaved in one process.  Build:
//   hipcc --offload-arch=gfx950 -O3 -o tools/membench/membench tools/membench/membench.hip
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

template <bool NT>
__device__ __forceinline__ f64x2 ld(const f64x2* p)
{
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st(f64x2* p, f64x2 v)
{
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// grid-stride, U independent 16-B loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_gs(const f64x2* __restrict__ a,
                                               size_t n2, double* sink)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    f64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += v[u].x + v[u].y;
  }
  for (; i < n2; i += stride) {
    f64x2 v = ld<NT>(a + i);
    acc += v.x + v.y;
  }
  if (acc == 123.456)
    *sink = acc;
}

// each workgroup owns one contiguous chunk and walks it 4 KiB at a time
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_chunk(const f64x2* __restrict__ a,
                                                  size_t n2, double* sink)
{
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t b = (size_t)blockIdx.x * per;
  const size_t e = b + per < n2 ? b + per : n2;
  double acc = 0.0;
  size_t i = b + threadIdx.x;
  for (; i + (U - 1) * 256 < e; i += U * 256) {
    f64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = ld<NT>(a + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += v[u].x + v[u].y;
  }
  for (; i < e; i += 256) {
    f64x2 v = ld<NT>(a + i);
    acc += v.x + v.y;
  }
  if (acc == 123.456)
    *sink = acc;
}

// 8-byte loads
__global__ __launch_bounds__(256) void read_gs8(const double* __restrict__ a,
                                                size_t n, double* sink)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    acc += a[i];
  if (acc == 123.456)
    *sink = acc;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void write_gs(f64x2* __restrict__ a, size_t n2,
                                                double val)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const f64x2 v = {val, val};
  for (; i + (U - 1) * stride < n2; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      st<NT>(a + i + u * stride, v);
  }
  for (; i < n2; i += stride)
    st<NT>(a + i, v);
}

__global__ __launch_bounds__(256) void write_gs8(double* __restrict__ a, size_t n,
                                                 double val)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    a[i] = val;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_gs(const f64x2* __restrict__ a,
                                               f64x2* __restrict__ b, size_t n2)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n2; i += U * stride) {
    f64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u)
      st<NT>(b + i + u * stride, v[u]);
  }
  for (; i < n2; i += stride)
    st<NT>(b + i, ld<NT>(a + i));
}

// two read streams + one write stream (the shape of r -= alpha*Ap)
template <bool NT>
__global__ __launch_bounds__(256) void triad_gs(const f64x2* __restrict__ a,
                                                f64x2* __restrict__ b, size_t n2,
                                                double s)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
       i += stride) {
    f64x2 x = ld<NT>(a + i), y = ld<NT>(b + i);
    y.x += s * x.x;
    y.y += s * x.y;
    st<NT>(b + i, y);
  }
}

// ---- chunked shapes: every workgroup owns one contiguous range -------------
template <int U, bool NT>
__global__ __launch_bounds__(256) void write_chunk(f64x2* __restrict__ a,
                                                   size_t n2, double val)
{
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t b = (size_t)blockIdx.x * per;
  const size_t e = b + per < n2 ? b + per : n2;
  const f64x2 v = {val, val};
  size_t i = b + threadIdx.x;
  for (; i + (U - 1) * 256 < e; i += U * 256) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      st<NT>(a + i + u * 256, v);
  }
  for (; i < e; i += 256)
    st<NT>(a + i, v);
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_chunk(const f64x2* __restrict__ a,
                                                  f64x2* __restrict__ b, size_t n2)
{
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t e = lo + per < n2 ? lo + per : n2;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < e; i += U * 256) {
    f64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = ld<NTL>(a + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u)
      st<NTS>(b + i + u * 256, v[u]);
  }
  for (; i < e; i += 256)
    st<NTS>(b + i, ld<NTL>(a + i));
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void triad_chunk(const f64x2* __restrict__ a,
                                                   f64x2* __restrict__ b,
                                                   size_t n2, double s)
{
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t e = lo + per < n2 ? lo + per : n2;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < e; i += U * 256) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = ld<NTL>(a + i + u * 256);
      y[u] = ld<NTL>(b + i + u * 256);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      y[u].x += s * x[u].x;
      y[u].y += s * x[u].y;
      st<NTS>(b + i + u * 256, y[u]);
    }
  }
  for (; i < e; i += 256) {
    f64x2 x = ld<NTL>(a + i), y = ld<NTL>(b + i);
    y.x += s * x.x;
    y.y += s * x.y;
    st<NTS>(b + i, y);
  }
}

// two read streams (dot product shape)
template <int U, bool NT>
__global__ __launch_bounds__(256) void dot_chunk(const f64x2* __restrict__ a,
                                                 const f64x2* __restrict__ b,
                                                 size_t n2, double* sink)
{
  const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per;
  const size_t e = lo + per < n2 ? lo + per : n2;
  double acc = 0.0;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < e; i += U * 256) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = ld<NT>(a + i + u * 256);
      y[u] = ld<NT>(b + i + u * 256);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += x[u].x * y[u].x + x[u].y * y[u].y;
  }
  for (; i < e; i += 256) {
    f64x2 x = ld<NT>(a + i), y = ld<NT>(b + i);
    acc += x.x * y.x + x.y * y.y;
  }
  if (acc == 123.456)
    *sink = acc;
}

// ---- grid-stride over UNITS of U*4 KiB: persistent grid, sliding window ------
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void triad_gsblk(const f64x2* __restrict__ a,
                                                   f64x2* __restrict__ b,
                                                   size_t n2, double s)
{
  const size_t unit = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * unit; base < n2;
       base += (size_t)gridDim.x * unit) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      if (i < n2) {
        x[u] = ld<NTL>(a + i);
        y[u] = ld<NTL>(b + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      if (i < n2) {
        y[u].x += s * x[u].x;
        y[u].y += s * x[u].y;
        st<NTS>(b + i, y[u]);
      }
    }
  }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void dot_gsblk(const f64x2* __restrict__ a,
                                                 const f64x2* __restrict__ b,
                                                 size_t n2, double* sink)
{
  const size_t unit = (size_t)U * 256;
  double acc = 0.0;
  for (size_t base = (size_t)blockIdx.x * unit; base < n2;
       base += (size_t)gridDim.x * unit) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      if (i < n2) {
        x[u] = ld<NT>(a + i);
        y[u] = ld<NT>(b + i);
      } else {
        x[u] = y[u] = f64x2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc += x[u].x * y[u].x + x[u].y * y[u].y;
  }
  if (acc == 123.456)
    *sink = acc;
}

// x += alpha p ; p = beta p + r  (three reads, two writes: cg_update_xp shape)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void xp_gsblk(const f64x2* __restrict__ r,
                                                f64x2* __restrict__ x,
                                                f64x2* __restrict__ p, size_t n2,
                                                double al, double be)
{
  const size_t unit = (size_t)U * 256;
  for (size_t base = (size_t)blockIdx.x * unit; base < n2;
       base += (size_t)gridDim.x * unit) {
    f64x2 rv[U], xv[U], pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      if (i < n2) {
        rv[u] = ld<NTL>(r + i);
        xv[u] = ld<NTL>(x + i);
        pv[u] = ld<NTL>(p + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + u * 256 + threadIdx.x;
      if (i < n2) {
        xv[u].x += al * pv[u].x;
        xv[u].y += al * pv[u].y;
        pv[u].x = be * pv[u].x + rv[u].x;
        pv[u].y = be * pv[u].y + rv[u].y;
        st<NTS>(x + i, xv[u]);
        st<NTS>(p + i, pv[u]);
      }
    }
  }
}

struct Timer {
  hipEvent_t e0, e1;
  Timer()
  {
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
  }
  double run(const std::function<void()>& f, int reps)
  {
    f();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int round = 0; round < 3; ++round) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r)
        f();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / reps < best)
        best = ms / reps;
    }
    return best;
  }
};

int main(int argc, char** argv)
{
  const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t bytes = mib << 20, n = bytes / 8, n2 = n / 2;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double *a, *b, *c, *sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&c, bytes));
  CK(hipMemset(c, 0, bytes));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  Timer t;
  auto report = [&](const std::string& name, int wgs_per_cu, double ms,
                    double nbytes) {
    printf("{\"test\": \"%s\", \"mib\": %zu, \"wg_per_cu\": %d, \"ms\": %.5f, "
           "\"GB/s\": %.1f}\n",
           name.c_str(), mib, wgs_per_cu, ms, nbytes / ms / 1e6);
    fflush(stdout);
  };
  const f64x2* a2 = reinterpret_cast<const f64x2*>(a);
  f64x2* b2 = reinterpret_cast<f64x2*>(b);
  f64x2* a2w = reinterpret_cast<f64x2*>(a);
  f64x2* c2 = reinterpret_cast<f64x2*>(c);

  for (int wpc : {4, 8, 16, 32, 64, 0}) {
    // 0 = one workgroup per 16 KiB (non-persistent)
    const int grid = wpc ? cus * wpc : (int)((bytes + 16383) / 16384);
#define RUN(name, nbytes, ...)                                                 \
  report(name, wpc, t.run([&] { __VA_ARGS__; }, reps), (double)(nbytes))
    RUN("read_gs_u1", bytes, (read_gs<1, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_gs_u2", bytes, (read_gs<2, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_gs_u4", bytes, (read_gs<4, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_gs_u4_nt", bytes, (read_gs<4, true><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_gs_u8", bytes, (read_gs<8, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_chunk_u1", bytes, (read_chunk<1, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_chunk_u4", bytes, (read_chunk<4, false><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_chunk_u4_nt", bytes, (read_chunk<4, true><<<grid, 256>>>(a2, n2, sink)));
    RUN("read_gs_8B", bytes, (read_gs8<<<grid, 256>>>(a, n, sink)));
    RUN("write_gs_8B", bytes, (write_gs8<<<grid, 256>>>(a, n, 1.5)));
    RUN("write_gs_u1", bytes, (write_gs<1, false><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("write_gs_u4", bytes, (write_gs<4, false><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("write_gs_u4_nt", bytes, (write_gs<4, true><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("copy_gs_u1", 2 * bytes, (copy_gs<1, false><<<grid, 256>>>(a2, b2, n2)));
    RUN("copy_gs_u4", 2 * bytes, (copy_gs<4, false><<<grid, 256>>>(a2, b2, n2)));
    RUN("copy_gs_u4_nt", 2 * bytes, (copy_gs<4, true><<<grid, 256>>>(a2, b2, n2)));
    RUN("triad", 3 * bytes, (triad_gs<false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_nt", 3 * bytes, (triad_gs<true><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("write_chunk_u1", bytes, (write_chunk<1, false><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("write_chunk_u4", bytes, (write_chunk<4, false><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("write_chunk_u4_nt", bytes, (write_chunk<4, true><<<grid, 256>>>(a2w, n2, 1.5)));
    RUN("copy_chunk_u4", 2 * bytes, (copy_chunk<4, false, false><<<grid, 256>>>(a2, b2, n2)));
    RUN("copy_chunk_u4_ntl", 2 * bytes, (copy_chunk<4, true, false><<<grid, 256>>>(a2, b2, n2)));
    RUN("copy_chunk_u4_ntls", 2 * bytes, (copy_chunk<4, true, true><<<grid, 256>>>(a2, b2, n2)));
    RUN("triad_chunk_u1", 3 * bytes, (triad_chunk<1, false, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_chunk_u2_ntl", 3 * bytes, (triad_chunk<2, true, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_chunk_u2_ntls", 3 * bytes, (triad_chunk<2, true, true><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_chunk_u4_ntl", 3 * bytes, (triad_chunk<4, true, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_chunk_u4_ntls", 3 * bytes, (triad_chunk<4, true, true><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_gsblk_u1", 3 * bytes, (triad_gsblk<1, false, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_gsblk_u2_ntls", 3 * bytes, (triad_gsblk<2, true, true><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_gsblk_u4", 3 * bytes, (triad_gsblk<4, false, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_gsblk_u4_ntl", 3 * bytes, (triad_gsblk<4, true, false><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("triad_gsblk_u4_ntls", 3 * bytes, (triad_gsblk<4, true, true><<<grid, 256>>>(a2, b2, n2, 0.5)));
    RUN("dot_gsblk_u2_nt", 2 * bytes, (dot_gsblk<2, true><<<grid, 256>>>(a2, b2, n2, sink)));
    RUN("dot_gsblk_u4_nt", 2 * bytes, (dot_gsblk<4, true><<<grid, 256>>>(a2, b2, n2, sink)));
    RUN("dot_gsblk_u4", 2 * bytes, (dot_gsblk<4, false><<<grid, 256>>>(a2, b2, n2, sink)));
    RUN("xp_gsblk_u1", 5 * bytes, (xp_gsblk<1, false, false><<<grid, 256>>>(a2, b2, c2, n2, 0.5, 0.25)));
    RUN("xp_gsblk_u2_ntls", 5 * bytes, (xp_gsblk<2, true, true><<<grid, 256>>>(a2, b2, c2, n2, 0.5, 0.25)));
    RUN("xp_gsblk_u4_ntls", 5 * bytes, (xp_gsblk<4, true, true><<<grid, 256>>>(a2, b2, c2, n2, 0.5, 0.25)));
    RUN("xp_gsblk_u2_ntl", 5 * bytes, (xp_gsblk<2, true, false><<<grid, 256>>>(a2, b2, c2, n2, 0.5, 0.25)));
    RUN("dot_chunk_u1", 2 * bytes, (dot_chunk<1, false><<<grid, 256>>>(a2, b2, n2, sink)));
    RUN("dot_chunk_u2_nt", 2 * bytes, (dot_chunk<2, true><<<grid, 256>>>(a2, b2, n2, sink)));
    RUN("dot_chunk_u4_nt", 2 * bytes, (dot_chunk<4, true><<<grid, 256>>>(a2, b2, n2, sink)));
  }
  const int wpc = -1;
  RUN("hipMemsetAsync_0", bytes, CK(hipMemsetAsync(a, 0, bytes, 0)));
  RUN("hipMemsetD32Async_nonzero", bytes,
      CK(hipMemsetD32Async((hipDeviceptr_t)a, 0x3fc00000, bytes / 4, 0)));
  RUN("hipMemcpyAsync_d2d", 2 * bytes,
      CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)));
  return 0;
}
