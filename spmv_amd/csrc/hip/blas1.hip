// Ghost pack kernel and the CG loop's fused BLAS-1 kernels (gfx950).
//
// gather: DeviceExecutor::gather_ghosts_run (spmv/device_executor.h:123-126,
//         spmv/reference_executor.cpp:150-164).
// CG:     spmv::cg (spmv/cg.cpp:21-98).  Seven BLAS calls + two host
//         reductions per iteration become two streaming kernels
//           K2: x += alpha p ; r -= alpha Ap ; partial r.r      (cg.cpp:66-73)
//           K3: converged? ; p = beta p + r                     (cg.cpp:77-85)
//         plus single-workgroup reducers.  alpha, beta and the stopping test
//         are evaluated on the device from the scalar history rr[], pAp[];
//         precedent for device-resident scalars: cuda/cg.cuda.cu:14-38,73-84.
//
// Built with -ffp-contract=off: axpy/scal round like unfused BLAS-1.
//
// Streaming shape (measured with tools/membench on MI355X, 1-4 GiB vectors):
// a persistent grid walks UNITS of kU x 4 KiB, every lane keeps kU 16-byte
// loads per stream in flight, and vectors that cannot stay in the 256 MiB
// Infinity Cache anyway are read and written non-temporally.  Against plain
// 16-byte grid-stride loops this gave 4.6-4.9 -> 5.8-5.9 TB/s for the
// two-read-one-write shape (r -= alpha Ap), 4.7-5.0 -> 5.6-5.7 TB/s for the
// three-read-two-write shape (x, p update) and 6.3 -> 7.1 TB/s for dots.
#include "common.h"

#include <cmath>
#include <new>

namespace
{

__device__ __forceinline__ double block_sum(double v, double* s_red)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    s_red[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w)
      r += s_red[w];
  }
  return r; // valid in thread 0
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int kU = 4;                           // 16-B loads in flight per stream
constexpr int64_t kUnit = (int64_t)kU * kBlock; // double2 elements per step

template <bool NT>
__device__ __forceinline__ f64x2 vload(const double* p, int64_t i2)
{
  const f64x2* q = reinterpret_cast<const f64x2*>(p) + i2;
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void vstore(double* p, int64_t i2, f64x2 v)
{
  f64x2* q = reinterpret_cast<f64x2*>(p) + i2;
  if (NT)
    __builtin_nontemporal_store(v, q);
  else
    *q = v;
}

// for (unit of this workgroup) { load phase ; compute + store phase }
#define SPMV_FOR_UNITS(n2)                                                     \
  for (int64_t base = (int64_t)blockIdx.x * kUnit; base < (n2);               \
       base += (int64_t)gridDim.x * kUnit)
#define SPMV_FOR_LANE_ELEMS(i, n2)                                             \
  _Pragma("unroll") for (int u = 0; u < kU; ++u)                               \
    if (const int64_t i = base + u * kBlock + threadIdx.x; i < (n2))

// sum over the double2 elements [0, n2) of x . y : this thread's share
template <bool NT>
__device__ __forceinline__ double stream_dot(int64_t n2, const double* x,
                                             const double* y)
{
  double acc = 0.0;
  SPMV_FOR_UNITS(n2)
  {
    f64x2 a[kU], b[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      a[u] = vload<NT>(x, i);
      b[u] = vload<NT>(y, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      acc += a[u].x * b[u].x;
      acc += a[u].y * b[u].y;
    }
  }
  return acc;
}

// r += nalpha * Ap (cg.cpp:70), returns this thread's share of r.r (:73)
template <bool NT>
__device__ __forceinline__ double stream_update_r(int64_t n2, double nalpha,
                                                  const double* Ap, double* r)
{
  double acc = 0.0;
  SPMV_FOR_UNITS(n2)
  {
    f64x2 av[kU], rv[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      av[u] = vload<NT>(Ap, i);
      rv[u] = vload<NT>(r, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      rv[u].x += nalpha * av[u].x;
      rv[u].y += nalpha * av[u].y;
      vstore<NT>(r, i, rv[u]);
      acc += rv[u].x * rv[u].x;
      acc += rv[u].y * rv[u].y;
    }
  }
  return acc;
}

// x += alpha p (cg.cpp:69) ; r += nalpha Ap (:70) ; share of r.r (:73)
template <bool NT>
__device__ __forceinline__ double stream_update_xr(int64_t n2, double alpha,
                                                   double nalpha, const double* p,
                                                   const double* Ap, double* x,
                                                   double* r)
{
  double acc = 0.0;
  SPMV_FOR_UNITS(n2)
  {
    f64x2 pv[kU], av[kU], xv[kU], rv[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      pv[u] = vload<NT>(p, i);
      av[u] = vload<NT>(Ap, i);
      xv[u] = vload<NT>(x, i);
      rv[u] = vload<NT>(r, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      xv[u].x += alpha * pv[u].x;
      xv[u].y += alpha * pv[u].y;
      rv[u].x += nalpha * av[u].x;
      rv[u].y += nalpha * av[u].y;
      vstore<NT>(x, i, xv[u]);
      vstore<NT>(r, i, rv[u]);
      acc += rv[u].x * rv[u].x;
      acc += rv[u].y * rv[u].y;
    }
  }
  return acc;
}

// p = beta p + r (cg.cpp:84-85)
template <bool NT>
__device__ __forceinline__ void stream_update_p(int64_t n2, double beta,
                                                const double* r, double* p)
{
  SPMV_FOR_UNITS(n2)
  {
    f64x2 pv[kU], rv[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      pv[u] = vload<NT>(p, i);
      rv[u] = vload<NT>(r, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      pv[u].x = beta * pv[u].x;
      pv[u].y = beta * pv[u].y;
      pv[u].x += rv[u].x;
      pv[u].y += rv[u].y;
      vstore<NT>(p, i, pv[u]);
    }
  }
}

// x += alpha p (cg.cpp:69)
template <bool NT>
__device__ __forceinline__ void stream_axpy(int64_t n2, double alpha,
                                            const double* p, double* x)
{
  SPMV_FOR_UNITS(n2)
  {
    f64x2 pv[kU], xv[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      pv[u] = vload<NT>(p, i);
      xv[u] = vload<NT>(x, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      xv[u].x += alpha * pv[u].x;
      xv[u].y += alpha * pv[u].y;
      vstore<NT>(x, i, xv[u]);
    }
  }
}

// x += alpha p (cg.cpp:69) ; p = beta p + r (:84-85)
template <bool NT>
__device__ __forceinline__ void stream_update_xp(int64_t n2, double alpha,
                                                 double beta, const double* r,
                                                 double* x, double* p)
{
  SPMV_FOR_UNITS(n2)
  {
    f64x2 pv[kU], xv[kU], rv[kU];
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      pv[u] = vload<NT>(p, i);
      xv[u] = vload<NT>(x, i);
      rv[u] = vload<NT>(r, i);
    }
    SPMV_FOR_LANE_ELEMS(i, n2)
    {
      xv[u].x += alpha * pv[u].x;
      xv[u].y += alpha * pv[u].y;
      vstore<NT>(x, i, xv[u]);
      pv[u].x = beta * pv[u].x;
      pv[u].y = beta * pv[u].y;
      pv[u].x += rv[u].x;
      pv[u].y += rv[u].y;
      vstore<NT>(p, i, pv[u]);
    }
  }
}

__device__ __forceinline__ void clear_partials_tail(double* partials, int len)
{
  for (int i = gridDim.x + blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += gridDim.x * blockDim.x)
    partials[i] = 0.0;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_kernel(
    int n, const int32_t* __restrict__ indices, const T* __restrict__ in,
    T* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    out[i] = in[indices[i]];
}

// reverse halo accumulate (spmv/L2GMap.cpp:921-922,947-948) for ONE neighbour's
// segment: indices are distinct inside a segment, so no atomics are needed
template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_add_kernel(
    int n, const int32_t* __restrict__ indices, const T* __restrict__ in,
    T* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    out[indices[i]] += in[i];
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void dot_partial_kernel(
    int64_t n, const double* __restrict__ x, const double* __restrict__ y,
    DotOut dot)
{
  __shared__ double s_red[kBlock / 64];
  double acc = stream_dot<NT>(n >> 1, x, y);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    acc += x[n - 1] * y[n - 1];
  spmv_dot_epilogue(dot, acc, s_red);
}

// Sum `len` partials in a fixed order with one workgroup.
__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(
    const double* __restrict__ partials, int len, double* __restrict__ result,
    const int32_t* __restrict__ done)
{
  __shared__ double s_red[kBlock / 64];
  if (done && *done)
    return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < len; i += kBlock)
    acc += partials[i];
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    *result = s;
}

// ---- CG -----------------------------------------------------------------
struct CgScalars {
  double rtol;
  int32_t done;
  int32_t kstop;
};

// x += alpha p ; r += (-alpha) Ap ; partial r.r
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_xr_kernel(
    int64_t n, const double* __restrict__ rr_prev,
    const double* __restrict__ pAp, const CgScalars* __restrict__ sc,
    const double* __restrict__ p, const double* __restrict__ Ap,
    double* __restrict__ x, double* __restrict__ r,
    double* __restrict__ partials, int len)
{
  __shared__ double s_red[kBlock / 64];
  if (sc->done)
    return;
  const double rnorm_old = sqrt(*rr_prev);             // cg.cpp:50,76
  const double alpha = (rnorm_old * rnorm_old) / *pAp; // cg.cpp:66
  const double nalpha = -alpha;
  double acc = stream_update_xr<NT>(n >> 1, alpha, nalpha, p, Ap, x, r);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    double rv = r[i] + nalpha * Ap[i];
    r[i] = rv;
    acc += rv * rv;
  }
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = s;
  clear_partials_tail(partials, len);
}

// stopping test on rr[k], then p = beta p + r
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_p_kernel(
    int64_t n, int k, const double* __restrict__ rr0,
    const double* __restrict__ rr_prev, const double* __restrict__ rr_new,
    CgScalars* __restrict__ sc, const double* __restrict__ r,
    double* __restrict__ p)
{
  if (sc->done)
    return;
  const double rnorm0 = sqrt(*rr0);
  const double rnorm_old = sqrt(*rr_prev);
  const double rnorm_new = sqrt(*rr_new);                           // cg.cpp:76
  const double beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old); // :77
  if (rnorm_new / rnorm0 < sc->rtol)                                // :80
    return; // p is left untouched (:81); cg_reduce_pAp_kernel raises `done`
  stream_update_p<NT>(n >> 1, beta, r, p);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    p[i] = beta * p[i] + r[i];
  }
}

// ---- the same two updates, regrouped so that p is read once ---------------
// K2': r += (-alpha) Ap ; partial r.r                  (cg.cpp:66,70,73)
// K3': x += alpha p ; stop test ; p = beta p + r        (cg.cpp:69,77-85)
// Element-wise arithmetic and its order per element are unchanged; only the
// kernel an update lives in differs (8 instead of 9 vector passes).  x is
// still updated in the iteration that converges and p is not (cg.cpp:80-81).
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_r_kernel(
    int64_t n, const double* __restrict__ rr_prev,
    const double* __restrict__ pAp, const CgScalars* __restrict__ sc,
    const double* __restrict__ Ap, double* __restrict__ r,
    double* __restrict__ partials, int len)
{
  __shared__ double s_red[kBlock / 64];
  if (sc->done)
    return;
  const double rnorm_old = sqrt(*rr_prev);
  const double alpha = (rnorm_old * rnorm_old) / *pAp; // cg.cpp:66
  const double nalpha = -alpha;
  double acc = stream_update_r<NT>(n >> 1, nalpha, Ap, r);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double rv = r[i] + nalpha * Ap[i];
    r[i] = rv;
    acc += rv * rv;
  }
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = s;
  clear_partials_tail(partials, len);
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_xp_kernel(
    int64_t n, int k, const double* __restrict__ rr0,
    const double* __restrict__ rr_prev, const double* __restrict__ rr_new,
    const double* __restrict__ pAp, CgScalars* __restrict__ sc,
    const double* __restrict__ r, double* __restrict__ x,
    double* __restrict__ p)
{
  if (sc->done)
    return;
  const double rnorm0 = sqrt(*rr0);
  const double rnorm_old = sqrt(*rr_prev);
  const double rnorm_new = sqrt(*rr_new);                                // :76
  const double alpha = (rnorm_old * rnorm_old) / *pAp;                   // :66
  const double beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old); // :77
  const bool converged = rnorm_new / rnorm0 < sc->rtol;                  // :80
  if (converged) { // x takes this iteration's update, p stays (:80-81)
    stream_axpy<NT>(n >> 1, alpha, p, x);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
      x[n - 1] += alpha * p[n - 1];
    return;
  }
  stream_update_xp<NT>(n >> 1, alpha, beta, r, x, p);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    p[i] = beta * p[i] + r[i];
  }
}

// Reduces the p.Ap partials of iteration k.  It is the first single-workgroup
// kernel after the p-update of iteration k-1, so it also raises `done` when
// rr[k-1] met the tolerance (cg.cpp:80-81): every later cg_* kernel then
// returns at once and x, r, p keep their iteration-(k-1) values.  In-order
// stream execution makes the flag visible to the following launches.
__global__ __launch_bounds__(kBlock) void cg_reduce_pAp_kernel(
    const double* __restrict__ partials, const double* __restrict__ partials2,
    int len, int k,
    const double* __restrict__ rr, double* __restrict__ pAp,
    CgScalars* __restrict__ sc)
{
  __shared__ double s_red[kBlock / 64];
  if (sc->done)
    return;
  if (k >= 2) {
    const double rnorm0 = sqrt(rr[0]);
    const double rnorm_prev = sqrt(rr[k - 1]);
    if (rnorm_prev / rnorm0 < sc->rtol) { // uniform across the workgroup
      if (threadIdx.x == 0) {
        sc->kstop = k - 1;
        sc->done = 1;
      }
      return;
    }
  }
  double acc = 0.0;
  for (int i = threadIdx.x; i < len; i += kBlock)
    acc += partials[i];
  if (partials2) // the remote block's share of p.Ap
    for (int i = threadIdx.x; i < len; i += kBlock)
      acc += partials2[i];
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    pAp[k] = s;
}

// ---- consumer-side reductions (one rank) -----------------------------------
// Instead of a single-workgroup reducer launch between producer and consumer,
// EVERY workgroup of the consuming kernel adds the <= 2048 partials itself, in
// the reducers' order (same loop, same tree => the same bits), and workgroup 0
// records the value in the history.  16 KB of L2-resident reads per workgroup
// against two kernel launches per iteration: what a small problem spends most
// of its iteration on.  Needs the scalar on this rank only, so it is used when
// the communicator has one rank.
__device__ __forceinline__ double consume_partials(
    const double* __restrict__ partials, const double* __restrict__ partials2,
    int len, double* s_red, double* s_bcast)
{
  double acc = 0.0;
  for (int i = threadIdx.x; i < len; i += kBlock)
    acc += partials[i];
  if (partials2)
    for (int i = threadIdx.x; i < len; i += kBlock)
      acc += partials2[i];
  const double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    *s_bcast = s;
  __syncthreads();
  return *s_bcast;
}

// cg_reduce_pAp_kernel + cg_update_r_kernel in one launch
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_r_cs_kernel(
    int64_t n, int k, const double* __restrict__ rr, double* __restrict__ pAp,
    CgScalars* __restrict__ sc, const double* __restrict__ pap_partials,
    const double* __restrict__ pap_partials2, int len,
    const double* __restrict__ Ap, double* __restrict__ r,
    double* __restrict__ rr_partials)
{
  __shared__ double s_red[kBlock / 64];
  __shared__ double s_bcast;
  if (sc->done)
    return;
  if (k >= 2) { // cg.cpp:80-81 of iteration k-1, as cg_reduce_pAp_kernel
    const double rnorm0 = sqrt(rr[0]);
    const double rnorm_prev = sqrt(rr[k - 1]);
    if (rnorm_prev / rnorm0 < sc->rtol) { // uniform across the grid
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->kstop = k - 1;
        sc->done = 1;
      }
      return;
    }
  }
  const double pap
      = consume_partials(pap_partials, pap_partials2, len, s_red, &s_bcast);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    pAp[k] = pap;
  const double rnorm_old = sqrt(rr[k - 1]);
  const double alpha = (rnorm_old * rnorm_old) / pap; // cg.cpp:66
  const double nalpha = -alpha;
  double acc = stream_update_r<NT>(n >> 1, nalpha, Ap, r);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double rv = r[i] + nalpha * Ap[i];
    r[i] = rv;
    acc += rv * rv;
  }
  __syncthreads(); // s_red is reused
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    rr_partials[blockIdx.x] = s;
  clear_partials_tail(rr_partials, len);
}

// reduce_partials_kernel (r.r) + cg_update_xp_kernel in one launch
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_xp_cs_kernel(
    int64_t n, int k, double* __restrict__ rr, const double* __restrict__ pAp,
    CgScalars* __restrict__ sc, const double* __restrict__ rr_partials, int len,
    const double* __restrict__ r, double* __restrict__ x, double* __restrict__ p)
{
  __shared__ double s_red[kBlock / 64];
  __shared__ double s_bcast;
  if (sc->done)
    return;
  const double rr_new
      = consume_partials(rr_partials, nullptr, len, s_red, &s_bcast);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    rr[k] = rr_new;
  const double rnorm0 = sqrt(rr[0]);
  const double rnorm_old = sqrt(rr[k - 1]);
  const double rnorm_new = sqrt(rr_new);                                 // :76
  const double alpha = (rnorm_old * rnorm_old) / pAp[k];                 // :66
  const double beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old); // :77
  const bool converged = rnorm_new / rnorm0 < sc->rtol;                  // :80
  if (converged) { // x takes this iteration's update, p stays (:80-81)
    stream_axpy<NT>(n >> 1, alpha, p, x);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
      x[n - 1] += alpha * p[n - 1];
    return;
  }
  stream_update_xp<NT>(n >> 1, alpha, beta, r, x, p);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    p[i] = beta * p[i] + r[i];
  }
}

// CG start (cg.cpp:39-50) in one pass over b: r = p = b, x0 = 0 (defined here
// instead of relying on fresh pages) and the partials of r.r.
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_init_kernel(
    int64_t n, const double* __restrict__ b, double* __restrict__ r,
    double* __restrict__ p, double* __restrict__ x,
    double* __restrict__ partials, int len)
{
  __shared__ double s_red[kBlock / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double v = b[i];
    if constexpr (NT) {
      __builtin_nontemporal_store(v, &r[i]);
      __builtin_nontemporal_store(v, &p[i]);
      __builtin_nontemporal_store(0.0, &x[i]);
    } else {
      r[i] = v;
      p[i] = v;
      x[i] = 0.0;
    }
    acc += v * v;
  }
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = s;
  clear_partials_tail(partials, len);
}

// ---- mixed-precision CG support -----------------------------
// out = (float) in
__global__ __launch_bounds__(kBlock) void convert_f64_f32_kernel(
    int64_t n, const double* __restrict__ in, float* __restrict__ out)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

// y += a x
__global__ __launch_bounds__(kBlock) void axpy_kernel(
    int64_t n, double a, const double* __restrict__ x, double* __restrict__ y)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}

// residual replacement: r = b - Ax (Ax given), partials of r.r.  Inside the
// CG loop it is a no-op once `done` is raised, like every other cg_* kernel;
// the closing check after the loop passes sc = nullptr.
__global__ __launch_bounds__(kBlock) void cg_residual_kernel(
    int64_t n, const CgScalars* __restrict__ sc, const double* __restrict__ b,
    const double* __restrict__ Ax, double* __restrict__ r,
    double* __restrict__ partials, int len)
{
  __shared__ double s_red[kBlock / 64];
  if (sc && sc->done)
    return;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double rv = b[i] - Ax[i];
    r[i] = rv;
    acc += rv * rv;
  }
  double s = block_sum(acc, s_red);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = s;
  clear_partials_tail(partials, len);
}

__global__ void cg_reset_kernel(CgScalars* sc, double rtol, double* rr,
                                double* pAp, int kmax)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    sc->rtol = rtol;
    sc->done = 0;
    sc->kstop = -1;
  }
  if (i <= kmax) {
    rr[i] = 0.0;
    pAp[i] = 0.0;
  }
}

__global__ __launch_bounds__(kBlock) void fill_gaussian_kernel(
    int64_t N, int64_t i_begin, int64_t count, double* __restrict__ x)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count;
       k += (int64_t)gridDim.x * blockDim.x) {
    const double z = (double)(k + i_begin) / (double)N; // demos/spmv.cpp:65
    const double u = 5 * (z - 0.5);
    x[k] = exp(-10 * (u * u)); // pow(u, 2.0) == u*u exactly
  }
}

__global__ __launch_bounds__(kBlock) void fill_const_kernel(
    int64_t count, double value, double* __restrict__ x)
{
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count;
       k += (int64_t)gridDim.x * blockDim.x)
    x[k] = value;
}

bool aligned16(const void* p)
{
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

} // namespace

struct spmv_hip_cg_ws {
  spmv_hip_ctx* ctx = nullptr;
  int kmax = 0;
  double* rr = nullptr;       // kmax + 1
  double* pAp = nullptr;      // kmax + 1
  double* partials = nullptr; // ctx->dot_blocks
  double* partials_rr = nullptr; // ctx->dot_blocks (consumer-side reductions)
  CgScalars* sc = nullptr;
};

// Vectors of at least ctx->blas1_nt_min_elems doubles stream past the caches
// (non-temporal loads and stores); shorter ones stay cached between kernels.
#define SPMV_LAUNCH_NT(ctx, n, kernel, grid, st, ...)                          \
  do {                                                                         \
    if ((int64_t)(n) >= (ctx)->blas1_nt_min_elems)                             \
      hipLaunchKernelGGL(kernel<true>, dim3(grid), dim3(kBlock), 0, st,        \
                         __VA_ARGS__);                                         \
    else                                                                       \
      hipLaunchKernelGGL(kernel<false>, dim3(grid), dim3(kBlock), 0, st,       \
                         __VA_ARGS__);                                         \
  } while (0)

extern "C" {

int spmv_hip_gather_f64(spmv_hip_ctx* ctx, int num_indices,
                        const int32_t* indices, const double* in, double* out,
                        void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_indices >= 0);
  if (num_indices == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(indices && in && out);
  const int grid = spmv_grid_for(ctx, num_indices, kBlock);
  hipLaunchKernelGGL((gather_kernel<double>), dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), num_indices, indices, in, out);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_scatter_add_f64(spmv_hip_ctx* ctx, int num_indices,
                             const int32_t* indices, const double* in,
                             double* out, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_indices >= 0);
  if (num_indices == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(indices && in && out);
  const int grid = spmv_grid_for(ctx, num_indices, kBlock);
  hipLaunchKernelGGL((scatter_add_kernel<double>), dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), num_indices, indices, in, out);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_scatter_add_f32(spmv_hip_ctx* ctx, int num_indices,
                             const int32_t* indices, const float* in,
                             float* out, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_indices >= 0);
  if (num_indices == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(indices && in && out);
  const int grid = spmv_grid_for(ctx, num_indices, kBlock);
  hipLaunchKernelGGL((scatter_add_kernel<float>), dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), num_indices, indices, in, out);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_gather_f32(spmv_hip_ctx* ctx, int num_indices,
                        const int32_t* indices, const float* in, float* out,
                        void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(num_indices >= 0);
  if (num_indices == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(indices && in && out);
  const int grid = spmv_grid_for(ctx, num_indices, kBlock);
  hipLaunchKernelGGL((gather_kernel<float>), dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), num_indices, indices, in, out);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_dot_partials_len(const spmv_hip_ctx* ctx, int* len)
{
  SPMV_REQUIRE(ctx && len);
  *len = ctx->dot_blocks;
  return SPMV_HIP_OK;
}

int spmv_hip_dot_partial_f64(spmv_hip_ctx* ctx, int64_t n, const double* x,
                             const double* y, double* partials, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(n >= 0 && partials && (n == 0 || (x && y)));
  SPMV_REQUIRE(aligned16(x) && aligned16(y));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  DotOut dot;
  dot.partials = partials;
  dot.len = ctx->dot_blocks;
  SPMV_LAUNCH_NT(ctx, n, dot_partial_kernel, grid, spmv_stream(ctx, stream), n, x, y, dot);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_init_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int64_t n,
                         const double* b, double* r, double* p, double* x,
                         void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && n >= 0 && (n == 0 || (b && r && p && x)));
  int grid = spmv_grid_for(ctx, n, kBlock);
  if (grid > ctx->dot_blocks)
    grid = ctx->dot_blocks;
  SPMV_LAUNCH_NT(ctx, n, cg_init_kernel, grid, spmv_stream(ctx, stream), n, b, r,
                 p, x, ws->partials, ctx->dot_blocks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_convert_f64_f32(spmv_hip_ctx* ctx, int64_t n, const double* in,
                             float* out, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(n >= 0 && (n == 0 || (in && out)));
  if (n == 0)
    return SPMV_HIP_OK;
  const int grid = spmv_grid_for(ctx, n, kBlock);
  hipLaunchKernelGGL(convert_f64_f32_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), n, in, out);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_axpy_f64(spmv_hip_ctx* ctx, int64_t n, double a, const double* x,
                      double* y, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(n >= 0 && (n == 0 || (x && y)));
  if (n == 0)
    return SPMV_HIP_OK;
  const int grid = spmv_grid_for(ctx, n, kBlock);
  hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), n, a, x, y);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_residual_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws,
                             int respect_done, int64_t n, const double* b,
                             const double* Ax, double* r, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && n >= 0 && (n == 0 || (b && Ax && r)));
  int grid = spmv_grid_for(ctx, n, kBlock);
  if (grid > ctx->dot_blocks)
    grid = ctx->dot_blocks;
  hipLaunchKernelGGL(cg_residual_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), n,
                     respect_done ? ws->sc : (const CgScalars*)nullptr, b, Ax, r,
                     ws->partials, ctx->dot_blocks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_reduce_partials_f64(spmv_hip_ctx* ctx, const double* partials,
                                 double* result, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(partials && result);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), partials, ctx->dot_blocks,
                     result, (const int32_t*)nullptr);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

// ---- CG workspace -------------------------------------------------------------
int spmv_hip_cg_ws_create(spmv_hip_ctx* ctx, int kmax, spmv_hip_cg_ws** out)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(out && kmax >= 0);
  spmv_hip_cg_ws* ws = new (std::nothrow) spmv_hip_cg_ws;
  if (!ws)
    return SPMV_HIP_ENOMEM;
  ws->ctx = ctx;
  ws->kmax = kmax;
  hipError_t e = hipMalloc(&ws->rr, sizeof(double) * (kmax + 1));
  if (e == hipSuccess)
    e = hipMalloc(&ws->pAp, sizeof(double) * (kmax + 1));
  if (e == hipSuccess)
    e = hipMalloc(&ws->partials, sizeof(double) * ctx->dot_blocks);
  if (e == hipSuccess)
    e = hipMalloc(&ws->partials_rr, sizeof(double) * ctx->dot_blocks);
  if (e == hipSuccess)
    e = hipMalloc(&ws->sc, sizeof(CgScalars));
  if (e != hipSuccess) {
    spmv_hip_cg_ws_destroy(ws);
    return static_cast<int>(e);
  }
  *out = ws;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_destroy(spmv_hip_cg_ws* ws)
{
  if (!ws)
    return SPMV_HIP_OK;
  (void)hipSetDevice(ws->ctx->device);
  (void)hipFree(ws->rr);
  (void)hipFree(ws->pAp);
  (void)hipFree(ws->partials);
  (void)hipFree(ws->partials_rr);
  (void)hipFree(ws->sc);
  delete ws;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_reset(spmv_hip_cg_ws* ws, double rtol, void* stream)
{
  SPMV_REQUIRE(ws);
  SPMV_SET_DEVICE(ws->ctx);
  const int n = ws->kmax + 1;
  hipLaunchKernelGGL(cg_reset_kernel, dim3((n + kBlock - 1) / kBlock),
                     dim3(kBlock), 0, spmv_stream(ws->ctx, stream), ws->sc,
                     rtol, ws->rr, ws->pAp, ws->kmax);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_rr(spmv_hip_cg_ws* ws, int k, double** slot)
{
  SPMV_REQUIRE(ws && slot && k >= 0 && k <= ws->kmax);
  *slot = ws->rr + k;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_pAp(spmv_hip_cg_ws* ws, int k, double** slot)
{
  SPMV_REQUIRE(ws && slot && k >= 0 && k <= ws->kmax);
  *slot = ws->pAp + k;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_partials(spmv_hip_cg_ws* ws, double** partials)
{
  SPMV_REQUIRE(ws && partials);
  *partials = ws->partials;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_done_flag(spmv_hip_cg_ws* ws, const int32_t** done)
{
  SPMV_REQUIRE(ws && done);
  *done = &ws->sc->done;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_capacity(const spmv_hip_cg_ws* ws, int* kmax)
{
  SPMV_REQUIRE(ws && kmax);
  *kmax = ws->kmax;
  return SPMV_HIP_OK;
}

int spmv_hip_cg_ws_read_async(spmv_hip_cg_ws* ws, int32_t* host_done_kstop,
                              double* host_rr, size_t host_rr_len, void* stream)
{
  SPMV_REQUIRE(ws);
  // checked before anything is enqueued: a short buffer gets nothing at all
  SPMV_REQUIRE(!host_rr || host_rr_len >= (size_t)ws->kmax + 1);
  SPMV_SET_DEVICE(ws->ctx);
  hipStream_t st = spmv_stream(ws->ctx, stream);
  if (host_done_kstop)
    SPMV_CHECK_HIP(hipMemcpyAsync(host_done_kstop, &ws->sc->done,
                                  2 * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  st));
  if (host_rr)
    SPMV_CHECK_HIP(hipMemcpyAsync(host_rr, ws->rr,
                                  sizeof(double) * (ws->kmax + 1),
                                  hipMemcpyDeviceToHost, st));
  return SPMV_HIP_OK;
}

int spmv_hip_cg_dot_rr_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int64_t n,
                           const double* r, void* stream)
{
  SPMV_REQUIRE(ws && ws->ctx == ctx);
  return spmv_hip_dot_partial_f64(ctx, n, r, r, ws->partials, stream);
}

int spmv_hip_cg_reduce_rr(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                          void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 0 && k <= ws->kmax);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), ws->partials, ctx->dot_blocks,
                     ws->rr + k, &ws->sc->done);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_reduce_pAp(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                           void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax);
  hipLaunchKernelGGL(cg_reduce_pAp_kernel, dim3(1), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), ws->partials,
                     (const double*)nullptr, ctx->dot_blocks, k, ws->rr,
                     ws->pAp, ws->sc);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_reduce_pAp2(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                            const double* partials2, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && partials2);
  hipLaunchKernelGGL(cg_reduce_pAp_kernel, dim3(1), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), ws->partials, partials2,
                     ctx->dot_blocks, k, ws->rr, ws->pAp, ws->sc);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_xr_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                              int64_t n, const double* p, const double* Ap,
                              double* x, double* r, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (p && Ap && x && r));
  SPMV_REQUIRE(aligned16(p) && aligned16(Ap) && aligned16(x) && aligned16(r));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_xr_kernel, grid, spmv_stream(ctx, stream), n, ws->rr + (k - 1),
                     ws->pAp + k, ws->sc, p, Ap, x, r, ws->partials,
                     ctx->dot_blocks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_p_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                             int64_t n, const double* r, double* p,
                             void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (r && p));
  SPMV_REQUIRE(aligned16(r) && aligned16(p));
  hipStream_t st = spmv_stream(ctx, stream);
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_p_kernel, grid, st, n, k,
                     ws->rr, ws->rr + (k - 1), ws->rr + k, ws->sc, r, p);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_r_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                             int64_t n, const double* Ap, double* r,
                             void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (Ap && r));
  SPMV_REQUIRE(aligned16(Ap) && aligned16(r));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_r_kernel, grid, spmv_stream(ctx, stream), n, ws->rr + (k - 1), ws->pAp + k,
                     ws->sc, Ap, r, ws->partials, ctx->dot_blocks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_xp_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                              int64_t n, const double* r, double* x, double* p,
                              void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (r && x && p));
  SPMV_REQUIRE(aligned16(r) && aligned16(x) && aligned16(p));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_xp_kernel, grid, spmv_stream(ctx, stream), n, k, ws->rr, ws->rr + (k - 1),
                     ws->rr + k, ws->pAp + k, ws->sc, r, x, p);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_r_cs_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                                int64_t n, const double* Ap, double* r,
                                const double* pap_partials2, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (Ap && r));
  SPMV_REQUIRE(aligned16(Ap) && aligned16(r));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_r_cs_kernel, grid, spmv_stream(ctx, stream),
                 n, k, ws->rr, ws->pAp, ws->sc, ws->partials, pap_partials2,
                 ctx->dot_blocks, Ap, r, ws->partials_rr);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_cg_update_xp_cs_f64(spmv_hip_ctx* ctx, spmv_hip_cg_ws* ws, int k,
                                 int64_t n, const double* r, double* x,
                                 double* p, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ws && ws->ctx == ctx && k >= 1 && k <= ws->kmax && n >= 0);
  SPMV_REQUIRE(n == 0 || (r && x && p));
  SPMV_REQUIRE(aligned16(r) && aligned16(x) && aligned16(p));
  const int grid = spmv_grid_for(ctx, n / 2, (int)kUnit);
  SPMV_LAUNCH_NT(ctx, n, cg_update_xp_cs_kernel, grid, spmv_stream(ctx, stream),
                 n, k, ws->rr, ws->pAp, ws->sc, ws->partials_rr,
                 ctx->dot_blocks, r, x, p);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_fill_gaussian_f64(spmv_hip_ctx* ctx, int64_t N, int64_t i_begin,
                               int64_t count, double* x, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(N > 0 && i_begin >= 0 && count >= 0);
  if (count == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(x);
  const int grid = spmv_grid_for(ctx, count, kBlock);
  hipLaunchKernelGGL(fill_gaussian_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), N, i_begin, count, x);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_fill_const_f64(spmv_hip_ctx* ctx, int64_t count, double value,
                            double* x, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(count >= 0);
  if (count == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(x);
  const int grid = spmv_grid_for(ctx, count, kBlock);
  hipLaunchKernelGGL(fill_const_kernel, dim3(grid), dim3(kBlock), 0,
                     spmv_stream(ctx, stream), count, value, x);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

} // extern "C"
