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
  // BLAS-1 kernels read/write vectors of at least this many doubles
  // non-temporally (spmv_hip_ctx_set_option "blas1_nt_min_elems")
  int64_t blas1_nt_min_elems = (int64_t)1 << 24;
  // plans build the LX form (LDS-staged x windows, 16-bit column offsets)
  // for general matrices with at least this many entries ("lx_min_nnz")
  int64_t lx_min_nnz = (int64_t)1 << 20;
  // ... and only while x (num_cols * 8 bytes) is at most this large
  // ("lx_max_x_bytes"; no limit by default)
  int64_t lx_max_x_bytes = INT64_MAX;
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

// ---------------------------------------------------------------------------
// Dot-product plumbing shared by the SpMV and BLAS-1 kernels.
//
// Every producing workgroup leaves one partial sum; either a separate
// single-workgroup kernel adds them (spmv_hip_reduce_partials_f64), or -- when
// `result` is set -- the workgroup that finishes LAST adds them itself, in
// index order, so the sum is deterministic and no extra launch is needed
// ("single device-side reduction").  No workgroup ever waits for another one:
// the last finisher is identified by an arrival ticket.  Visibility follows
// cdna_hip_programming.md Guideline 16 (8-byte agent-scope atomics on both
// sides, ticket taken after the partial's atomic has returned).
// ---------------------------------------------------------------------------
constexpr int kDotShards = 32; // ticket words = kDotShards + 1 (C ABI:
                               // SPMV_HIP_DOT_COUNTER_WORDS)

struct DotOut {
  double* partials = nullptr;  // one slot per workgroup (>= gridDim.x, len total)
  int len = 0;                 // length of the partial array
  double* result = nullptr;    // optional fused final sum
  unsigned int* counter = nullptr; // kDotShards+1 ticket words, zero before
                                   // the launch, reset by the kernel
  int accumulate = 0;          // result += sum instead of result = sum
};

#ifdef __HIPCC__
// Block-wide sum of one double per thread (fixed tree => deterministic).
__device__ __forceinline__ double spmv_block_sum(double v, double* s_red)
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

// Epilogue of every dot-producing kernel.  `acc` = this thread's share.
// s_red: kBlock/64 doubles of LDS; s_flag: one int of LDS.
__device__ __forceinline__ void spmv_dot_epilogue(const DotOut& d, double acc,
                                                  double* s_red, int* s_flag)
{
  const double s = spmv_block_sum(acc, s_red);
  if (d.result == nullptr) {
    // two-stage form: leave the partial, clear the unused tail of the array
    if (threadIdx.x == 0)
      d.partials[blockIdx.x] = s;
    for (int i = gridDim.x + blockIdx.x * blockDim.x + threadIdx.x; i < d.len;
         i += gridDim.x * blockDim.x)
      d.partials[i] = 0.0;
    return;
  }
  // Hand-off with 8-byte agent-scope atomics on BOTH sides (Guideline 16,
  // "valid forms"): atomics execute at the memory side, so neither a release
  // fence on the producers (it would write back the whole XCD L2 once per
  // workgroup) nor an acquire on the consumer is needed.  The partial's
  // exchange returns before the ticket is taken (drained vmcnt), and only the
  // workgroup whose ticket is the last one reads the partials.
  // The arrival ticket is two-level (kDotShards shard words + one top word):
  // a persistent grid finishes almost at once, and ~1800 arrivals on ONE word
  // cost ~20 us (one word takes ~88 atomics/us); sharded, each word sees <= 56.
  if (threadIdx.x == 0) {
    (void)__hip_atomic_exchange(&d.partials[blockIdx.x], s, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int shard = blockIdx.x % kDotShards;
    const unsigned int in_shard
        = (gridDim.x - shard + kDotShards - 1) / kDotShards;
    const unsigned int nshards
        = gridDim.x < (unsigned)kDotShards ? gridDim.x : (unsigned)kDotShards;
    int last = 0;
    const unsigned int t1 = __hip_atomic_fetch_add(
        &d.counter[shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t1 == in_shard - 1) { // last of this shard: reset it, arrive at the top
      __hip_atomic_store(&d.counter[shard], 0u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int t2 = __hip_atomic_fetch_add(
          &d.counter[kDotShards], 1u, __ATOMIC_RELAXED,
          __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == nshards - 1) {
        __hip_atomic_store(&d.counter[kDotShards], 0u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        last = 1;
      }
    }
    *s_flag = last;
  }
  __syncthreads();
  if (*s_flag == 0)
    return;
  // last workgroup: add all partials in index order (atomic reads)
  // eight independent reads in flight per lane, then a fixed-order sum
  double a = 0.0;
  for (int base = 0; base < (int)gridDim.x; base += 8 * kBlock) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * kBlock + threadIdx.x;
      v[j] = (i < (int)gridDim.x)
                 ? __hip_atomic_fetch_add(&d.partials[i], 0.0, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT)
                 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      a += v[j];
  }
  __syncthreads(); // s_red is reused
  const double total = spmv_block_sum(a, s_red);
  if (threadIdx.x == 0) // the ticket words were reset by their last arrivers
    *d.result = d.accumulate ? (*d.result + total) : total;
}
#endif
