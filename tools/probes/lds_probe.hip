// How much LDS may ONE workgroup of 1024 threads take on this device, and does
// a launch with 143 KB of dynamic LDS run?  (csr_box27_half_kernel's footprint.)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void touch(double* out, int words)
{
  extern __shared__ double s[];
  for (int i = threadIdx.x; i < words; i += blockDim.x)
    s[i] = (double)i;
  __syncthreads();
  double a = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x)
    a += s[words - 1 - i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main()
{
  int v = 0;
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0);
  printf("MaxSharedMemoryPerBlock %d\n", v);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n",
         p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
  double* d = nullptr;
  hipMalloc(&d, 256 * 1024 * sizeof(double));
  const int sizes[] = {64 * 1024, 96 * 1024, 143408, 160 * 1024};
  for (int bytes : sizes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(touch),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    printf("bytes %d setattr %s", bytes, hipGetErrorName(e));
    (void)hipGetLastError();
    hipLaunchKernelGGL(touch, dim3(256), dim3(1024), bytes, 0, d, bytes / 8);
    e = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    printf(" launch %s sync %s\n", hipGetErrorName(e), hipGetErrorName(e2));
    (void)hipGetLastError();
  }
  return 0;
}
