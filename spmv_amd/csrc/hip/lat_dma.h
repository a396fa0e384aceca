// LDS-DMA helpers shared by the lattice kernels (spmv_lat.hip, spmv_symlat.hip).
#pragma once

#include "csr_plan.h"

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to
// lds_dst + lane*16 (lds_dst wave-uniform, passed in M0).  Written as inline
// assembly on purpose: with the builtin the compiler puts `s_waitcnt vmcnt(0)`
// in front of every later LDS read (it cannot tell the slot being filled from
// the slot being read) and the prefetch would be drained at once.  An
// instruction the compiler does not count can only make its own vmcnt waits
// stricter, never too weak (the counter retires in issue order); this file
// waits for the pieces itself, with vmcnt(0) before the barrier that
// publishes a slot.  M0 is saved and restored inside the statement.
template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst)
{
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Entries [base, b) of `values` -> LDS slot, base 16-byte aligned.  One DMA
// piece = one wave-instruction = 1 KiB.  Lanes past the span re-read its last
// 16-byte chunk (one cached line) instead of streaming the next block's data.
template <typename T, bool NT>
__device__ __forceinline__ void lat_issue_dma(const T* __restrict__ values,
                                              int64_t nnz, int64_t base,
                                              int64_t b, T* s_slot, int t)
{
  constexpr int V = 16 / (int)sizeof(T); // entries per 16-byte chunk
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int64_t jclamp = (b - 1) & ~(int64_t)(V - 1);
  const int pieces = (int)(((b - base) * (int64_t)sizeof(T) + 1023) >> 10);
  if (jclamp + V <= nnz) {
    // LDS byte address of the slot (uniform)
    const unsigned lds0 = (unsigned)(uintptr_t)(
        (__attribute__((address_space(3))) void*)s_slot);
    for (int q = wave; q < pieces; q += kBlock / 64) {
      int64_t j = base + (int64_t)(q * 64 + lane) * V;
      j = j < jclamp ? j : jclamp;
      glds16<NT>(values + j, lds0 + (unsigned)q * 1024u);
    }
  } else {
    // the last row block of the array: a 16-byte chunk would end past
    // values[nnz) -- element-wise, in bounds
    for (int64_t j = base + t; j < b; j += kBlock)
      s_slot[j - base] = values[j];
  }
}

