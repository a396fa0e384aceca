// Are vectors backed by their OWN physical allocation (hipMemCreate, one handle
// per vector) free of the same-"class" penalty that hipMalloc'ed vectors show
// (timing modes)?  y += a x over 1-GiB vectors, every
// ordered pair, for K hipMalloc buffers and K VMM buffers.
//   hipcc --offload-arch=gfx950 -O3 -o vmm_probe vmm_probe.hip && ./vmm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("%s -> %s\n", #x, hipGetErrorString(e_));                    \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

__global__ __launch_bounds__(256) void axpy(size_t n2, double a,
                                            const double2* __restrict__ x,
                                            double2* __restrict__ y)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
       i += (size_t)gridDim.x * blockDim.x) {
    double2 xv = x[i], yv = y[i];
    yv.x += a * xv.x;
    yv.y += a * xv.y;
    y[i] = yv;
  }
}

static float time_pair(double* x, double* y, size_t n)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(axpy, dim3(2048), dim3(256), 0, 0, n / 2, 0.5,
                       (const double2*)x, (double2*)y);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best)
      best = ms;
  }
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return best;
}

static double* vmm_alloc(size_t bytes)
{
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop,
                                    hipMemAllocationGranularityRecommended));
  bytes = (bytes + gran - 1) / gran * gran;
  hipMemGenericAllocationHandle_t h;
  CK(hipMemCreate(&h, bytes, &prop, 0));
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, bytes, gran, nullptr, 0));
  CK(hipMemMap(va, bytes, 0, h, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, bytes, &acc, 1));
  return (double*)va;
}

int main(int argc, char** argv)
{
  const int K = argc > 1 ? std::atoi(argv[1]) : 6;
  const size_t n = (size_t)1 << 27; // 1 GiB of doubles
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<double*> b(K);
    for (int i = 0; i < K; ++i) {
      if (mode == 0)
        CK(hipMalloc(&b[i], n * 8));
      else
        b[i] = vmm_alloc(n * 8);
      CK(hipMemset(b[i], 0, n * 8));
    }
    std::printf("%s\n", mode == 0 ? "hipMalloc" : "hipMemCreate (one handle per vector)");
    for (int i = 0; i < K; ++i) {
      for (int j = 0; j < K; ++j)
        std::printf(" %5.3f", i == j ? 0.f : time_pair(b[i], b[j], n));
      std::printf("\n");
    }
    std::fflush(stdout);
  }
  return 0;
}
