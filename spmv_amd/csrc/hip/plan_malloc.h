// Where a plan's set-up time goes, part two: the memory calls.  Included LAST by
// the plan-building translation units; from here on their hipMalloc / hipFree
// go through a wrapper that adds the call's wall time to a per-thread counter.
// spmv_hip_csr_plan_create / _bake_values_* / _values_changed read the counter
// around their work and file the difference under the plan ("plan_mem_us"):
// plan_us then splits into analysis (kernels, copies, host logic) and memory
// (allocation of the plan's own arrays, release of the scratch).  Why: one
// bench record of round 5 showed 2.3 s and 2.8 s for the two multi-GB baking
// forms where every other run shows 12-18 ms (profiles/
// r05_bench_n1_default_detail.json) -- with this split a recurrence says
// whether the device driver's allocator stalled or the analysis did.
#pragma once

#include <chrono>
#include <cstdint>

#include <hip/hip_runtime.h>

extern thread_local int64_t spmv_plan_mem_ns; // spmv_csr_plan.hip

template <typename P>
static inline hipError_t spmv_timed_malloc(P** p, size_t bytes)
{
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMalloc(p, bytes);
  spmv_plan_mem_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::steady_clock::now() - t0)
                          .count();
  return e;
}
static inline hipError_t spmv_timed_free(void* p)
{
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipFree(p);
  spmv_plan_mem_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::steady_clock::now() - t0)
                          .count();
  return e;
}
#define hipMalloc(...) spmv_timed_malloc(__VA_ARGS__)
#define hipFree(...) spmv_timed_free(__VA_ARGS__)
