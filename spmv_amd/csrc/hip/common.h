// Shared internals of libspmv_hip.so (gfx950 only).
#pragma once

#include <mutex>
#include <vector>

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
  // ... in the layout of the LDS-DMA kernel (spmv_lxw.hip: values, offsets and
  // x windows by LDS-DMA one row block ahead; "lx_dma", 0 = the register-staged
  // kernel's layout)
  int lx_dma = 1;
  // the host mirror's CSRMatrix frees its device copies of colind / values
  // once a plan holds the matrix in its own format ("release_csr";
  // spmv_hip_csr_plan_owns_matrix); read by the mirror, nothing here acts on it
  int release_csr = 0;
  // plans that keep the caller's CSR arrays as they are (no lattice, LX or
  // sliced jagged form) stage the x windows of every row block in LDS from
  // this many entries on ("xw_min_nnz": the XW kernel of spmv_lxw.hip; the
  // plan adds 144 B per row block, the CSR arrays are streamed untouched)
  int64_t xw_min_nnz = (int64_t)1 << 20;
  // ... and only when x (num_cols * 8 bytes) is at least this large
  // ("xw_min_x_bytes"): while x fits the Infinity Cache the gather kernel is the
  // faster of the two (216^3: 0.200 against 0.211 ms), beyond it the staged
  // windows are (512^3: 2.49 against 2.57 ms on one box, 13.8 instead of
  // 15.7 GB across the fabric)
  int64_t xw_min_x_bytes = (int64_t)128 << 20;
  // ... and lets its first four launches decide between that kernel and the
  // gather kernel ("xw_probe"; 0: XW whenever its records exist)
  int xw_probe = 1;
  // plans keep the caller's CSR arrays of a matrix without lattice structure
  // as they are -- no LX form (2 B per entry of plan memory), no sliced jagged
  // copy (10 B per entry): the XW / gather kernels ("csr_in_place"; default 0)
  int csr_in_place = 0;
  // plans try the lattice form (constant column offsets per row block, no
  // index stream) for general matrices with at least this many entries
  // ("lat_min_nnz")
  int64_t lat_min_nnz = (int64_t)1 << 20;
  // plan_bake_values on a GENERAL plan looks for a symmetric matrix and keeps
  // its lower half by offset ("bake_general"; 0: always SPMV_HIP_ENOTSUP)
  int bake_general = 1;
  // ... and the wide diagonal form keeps only the offsets <= 0 of a matrix it
  // finds symmetric bit for bit ("wdia_half"; 0: always all offsets)
  int wdia_half = 1;
  // the diagonal forms keep ONE number per diagonal and no copy of the values
  // when every diagonal of the matrix is constant, bit for bit
  // ("const_diagonals"; 0: always stream the values)
  int const_diagonals = 1;
  // ... with this many lattice lines per lane where the lattice is 3-D
  // ("const_tile": 1, 2 or 4)
  int const_tile = 4;
  // plans build the sliced jagged form (spmv_sjds.hip) for general matrices
  // with at least this many entries that take neither the lattice nor the LX
  // form ("sj_min_nnz")
  int64_t sj_min_nnz = (int64_t)1 << 20;
  // ... staging at most this many 16-column chunks of x per block
  // ("sj_max_chunks", <= 432: 54 KiB of fp64)
  int sj_max_chunks = 432;
  // ... with this many slices per block ("sj_wpb": 4, 8, 16; 0 = choose)
  int sj_wpb = 0;
  // ... and this many entries per lane and step ("sj_unit": 1, 2, 4; 0 = choose)
  int sj_unit = 0;
  // blocks of 16 slices sorted by length across the block, two slices per wave
  // ("sj_sigma"; 0: every slice sorted for itself, one slice per wave)
  int sj_sigma = 1;
  // symmetric storage takes the sliced jagged form only while the long
  // COLUMNS of the stored lower block (rows of the merged matrix: they stay
  // inside the slices) hold at most this share of the entries
  // ("sym_sj_long_permille")
  int sym_sj_long_permille = 50;
  // ... its long ROWS go to the long-row kernels ("sym_sj_long_rows"; 0: they
  // stay inside the slices too and count against the share above)
  int sym_sj_long_rows = 1;
  // one-sided halo: how long a put kernel polls for its neighbour before the
  // exchange fails with SPMV_HIP_EPEER ("put_timeout_ms")
  int put_timeout_ms = 60000;
  // error words of the put / reduction windows (see below): a growable list
  // under a mutex -- a context may carry any number of one-sided maps, and
  // threads of an application may build them side by side
  std::vector<const int32_t*> watched;
  std::mutex watched_mutex;
  // the device Poisson generator's non-symmetric variant ("poisson_skew_ppm":
  // lower neighbours -1 - s, upper -1 + s, s = value * 1e-6; 0 = the Poisson
  // matrix).  For measurements of kernels on matrices that are not symmetric.
  int poisson_skew_ppm = 0;
  // ... and its stencil ("poisson_stencil": 7, or 27 = all neighbours with
  // |dx|, |dy|, |dz| <= 1, diagonal 26, off-diagonal -1: HPCG's operator)
  int poisson_stencil = 7;
};

// What a bounded wait that times out leaves behind: a record of kPeerErrWords
// int32 in pinned host memory whose FIRST word is the error flag the context
// watches.  The first waiter to fail claims the record and describes itself --
// which wait, on whom, the epoch it saw against the epoch it wanted -- so that
// SPMV_HIP_EPEER names the kernel that never arrived instead of one bit:
//   [0] error flag      [1] claim          [2] which wait (PeerWait)
//   [3] peer: neighbour slot (put) / rank (reduce)
//   [4] workgroup of the waiter            [5] my rank (label, -1 = unknown)
//   [6..7] epoch seen   [8..9] epoch wanted
//   [kPeerErrLabels + k] rank behind neighbour slot k (put; -1 = unknown)
enum PeerWait : int32_t {
  kWaitNone = 0,
  kWaitPutFree = 1,   // put kernel, step (b): the neighbour's "your segment in
                      // my window is free" -- its put kernel of this epoch has
                      // not STARTED
  kWaitPutData = 2,   // put kernel, step (d): the neighbour's data flag -- its
                      // put kernel started but has not finished its stores
  kWaitReduceSlot = 3 // reduction kernel: the peer's slot in my window -- its
                      // reduction kernel of this epoch has not run
};
constexpr int kPeerErrLabels = 12;
constexpr int kPeerErrWords = kPeerErrLabels + 16; // SPMV_HIP_PUT_MAX_PEERS

// error words (pinned host memory, written by kernels) the context looks at
// whenever the host synchronises: a non-zero one = SPMV_HIP_EPEER
// (false: no memory for the entry -- the window must not be used unwatched)
bool spmv_ctx_watch(spmv_hip_ctx* ctx, const int32_t* word, bool add);
int spmv_ctx_check_watched(const spmv_hip_ctx* ctx);

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
// Every producing workgroup leaves one partial sum; a single workgroup adds
// them in index order afterwards -- a reducer kernel, or the consuming CG
// update kernel itself (blas1.hip) -- so every dot product is deterministic.
// ---------------------------------------------------------------------------
struct DotOut {
  double* partials = nullptr; // one slot per workgroup (>= gridDim.x, len total)
  int len = 0;                // length of the partial array
};

#ifdef __HIPCC__
// a timed-out wait describes itself (first failure wins the record), then
// raises the flag
__device__ __forceinline__ void spmv_peer_fail(int32_t* err, int32_t which,
                                               int32_t peer, uint64_t seen,
                                               uint64_t wanted)
{
  if (atomicCAS_system(reinterpret_cast<int*>(err + 1), 0, 1) == 0) {
    err[2] = which;
    err[3] = peer;
    err[4] = (int32_t)blockIdx.x;
    err[6] = (int32_t)(seen & 0xffffffffu);
    err[7] = (int32_t)(seen >> 32);
    err[8] = (int32_t)(wanted & 0xffffffffu);
    err[9] = (int32_t)(wanted >> 32);
    __threadfence_system();
  }
  __hip_atomic_store(err, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

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
// s_red: kBlock/64 doubles of LDS.  Leaves the workgroup's partial and clears
// the unused tail of the array.
__device__ __forceinline__ void spmv_dot_epilogue(const DotOut& d, double acc,
                                                  double* s_red)
{
  const double s = spmv_block_sum(acc, s_red);
  if (threadIdx.x == 0)
    d.partials[blockIdx.x] = s;
  for (int i = gridDim.x + blockIdx.x * blockDim.x + threadIdx.x; i < d.len;
       i += gridDim.x * blockDim.x)
    d.partials[i] = 0.0;
}
#endif
