// Shared internals of libspmv_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "spmv_hip.h"

struct spmv_hip_ctx {
  int device = 0;
  int num_cus = 0;
  hipStream_t stream = nullptr;  // current stream (set_stream), null = default
  int dot_blocks = 0;            // length of every dot-partials array
};

#define SPMV_CHECK_HIP(expr)                                                   \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
      return static_cast<int>(_e);                                             \
  } while (0)

#define SPMV_REQUIRE(cond)                                                     \
  do {                                                                         \
    if (!(cond))                                                               \
      return SPMV_HIP_EINVAL;                                                  \
  } while (0)

// Launch-error check that does not synchronise.
#define SPMV_CHECK_LAUNCH() SPMV_CHECK_HIP(hipGetLastError())

static inline hipStream_t spmv_stream(const spmv_hip_ctx* ctx, void* stream)
{
  return stream ? static_cast<hipStream_t>(stream) : ctx->stream;
}

// Every entry point runs on the context's device.
#define SPMV_SET_DEVICE(ctx)                                                   \
  do {                                                                         \
    SPMV_REQUIRE((ctx) != nullptr);                                            \
    SPMV_CHECK_HIP(hipSetDevice((ctx)->device));                               \
  } while (0)

// Threads per workgroup of the streaming kernels: 4 waves of 64.
constexpr int kBlock = 256;
// Workgroups per CU a grid-stride launch asks for.
constexpr int kBlocksPerCU = 8;

static inline int spmv_grid_for(const spmv_hip_ctx* ctx, int64_t work_items,
                                int items_per_block)
{
  int64_t need = (work_items + items_per_block - 1) / items_per_block;
  int64_t cap = static_cast<int64_t>(ctx->num_cus) * kBlocksPerCU;
  if (need < 1)
    need = 1;
  return static_cast<int>(need < cap ? need : cap);
}
