// One-sided halo: peer stores over xGMI instead of RCCL send/recv.
//
// The reference's onesided_put_* communication models move the ghosts with
// MPI_Put into a window of the neighbour's memory (spmv/L2GMap.cpp:645-682).
// The equivalent on one node of MI355X: every rank owns a WINDOW -- a staging
// buffer for its ghost tail plus two flag words per neighbour -- exported as a
// HIP IPC memory handle (one process per GPU) or shared by address (ranks that
// are threads of one process); a neighbour maps it once, at plan time, and
// from then on an exchange is ONE kernel launch per rank:
//
//   workgroups (k, 0..G) serve neighbour k
//   (a) tell k "my staging segment for you is free again"      flag store
//   (b) wait for k's "your segment in my window is free"        flag poll
//   (c) store my send segment into k's window                   peer stores
//       system-scope fence; the last of the G workgroups raises
//       k's "data of this epoch has landed" flag                flag store
//   (d) wait for k's data flag, copy my staging segment into    flag poll,
//       the ghost tail of the vector                            local copy
//
// Flags carry the EPOCH (a counter of exchanges, monotonic), so nothing is ever
// reset and a late reader of an old epoch cannot be confused.  Every wait is
// bounded (ctx option "put_timeout_ms", 60 s by default): a peer that died
// makes the exchange fail -- the ghost segment is filled with NaN, the error
// word in pinned host memory makes the context's next synchronisation point
// (and every later exchange) return SPMV_HIP_EPEER -- not hang.
// The staging hop costs one local copy of the ghost tail (2-4 MB at 512^3 over
// 8 ranks) and buys a registration that happens once per L2GMap instead of
// once per vector: the window never moves, whatever vector is exchanged.
//
// Validated on ONE device only (1-GPU boxes): ranks as processes that share
// GPU 0 through IPC handles (tests/mp_gpu_worker.py) and ranks as threads of
// one process (tests/thread_world.py).  For peers on OTHER devices the window
// is fine-grained memory (a peer's stores become visible to the owner's running
// kernel), flags and staged data are read with system-scope loads that bypass
// the caches, data is published with a system-scope fence before the flag.
#include "common.h"

#include <cstring>
#include <new>
#include <unistd.h>

struct PutPeer {
  char* dst;            // the neighbour's staging buffer (mapped)
  uint64_t* peer_flags; // ... and its flag words
  int32_t dst_off;      // where my data goes there (elements)
  int32_t slot_at_peer; // which neighbour slot I am there
  int32_t send_off, send_count; // my segment in the send buffer (elements)
  int32_t recv_off, recv_count; // the neighbour's segment in MY staging buffer
};

struct spmv_hip_put {
  spmv_hip_ctx* ctx = nullptr;
  char* window = nullptr; // staging | data flags | free flags | counters
  size_t stage_bytes = 0;
  int num_peers = 0;
  PutPeer host_tab[SPMV_HIP_PUT_MAX_PEERS];
  void* mapped[SPMV_HIP_PUT_MAX_PEERS]; // IPC mappings to close (or null)
  PutPeer* dev_tab = nullptr;
  int32_t* host_err = nullptr; // pinned, device-visible
  uint64_t epoch = 0;
  int fine_grained = 0; // the window is fine-grained (uncached) device memory
};

namespace
{

constexpr int kPutGroup = 8; // workgroups per neighbour

__device__ __forceinline__ uint64_t* put_flags(char* window, size_t stage_bytes)
{
  return reinterpret_cast<uint64_t*>(window + stage_bytes);
}

// thread 0 polls, the workgroup follows; false = timed out
__device__ __forceinline__ bool put_wait(const uint64_t* flag, uint64_t epoch,
                                         int32_t* err,
                                         unsigned long long timeout_ticks,
                                         int32_t which, int32_t slot)
{
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    uint64_t seen;
    while ((seen = __hip_atomic_load(flag, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM))
           < epoch) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > timeout_ticks) {
        ok = 0;
        spmv_peer_fail(err, which, slot, seen, epoch);
        break;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  __syncthreads(); // s_ok may be written again by the next wait
  // what the flag announced was written before it: order this workgroup's
  // later loads behind the poll
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return ok;
}

// WORD = uint64_t (fp64) or uint32_t (fp32)
template <typename WORD>
__global__ __launch_bounds__(kBlock) void put_exchange_kernel(
    const PutPeer* __restrict__ tab, const char* __restrict__ send_buf,
    char* __restrict__ ghost_tail, char* window, size_t stage_bytes,
    uint64_t epoch, int32_t* err, unsigned long long timeout_ticks)
{
  const int k = blockIdx.x / kPutGroup, g = blockIdx.x % kPutGroup;
  const PutPeer p = tab[k];
  uint64_t* my_flags = put_flags(window, stage_bytes);
  uint64_t* my_data_flag = my_flags + k;
  uint64_t* my_free_flag = my_flags + SPMV_HIP_PUT_MAX_PEERS + k;
  unsigned* my_counter
      = reinterpret_cast<unsigned*>(my_flags + 2 * SPMV_HIP_PUT_MAX_PEERS) + k;
  const int t = threadIdx.x;
  // (a) the previous exchange's launch copied my staging segment out (stream
  // order): the neighbour may fill it again
  if (g == 0 && t == 0)
    __hip_atomic_store(p.peer_flags + SPMV_HIP_PUT_MAX_PEERS + p.slot_at_peer,
                       epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // A wait that times out leaves this neighbour's ghost segment filled with
  // NaN: a failed exchange must not look like a valid one to the SpMV that
  // consumes the ghosts (the host reports SPMV_HIP_EPEER at its next
  // synchronisation point).
  auto poison = [&]() {
    WORD* dst = reinterpret_cast<WORD*>(ghost_tail) + p.recv_off;
    for (int64_t i = (int64_t)g * kBlock + t; i < p.recv_count;
         i += (int64_t)kPutGroup * kBlock)
      dst[i] = ~WORD(0); // all ones: a NaN in either width
  };
  // (b) ... and so may I, once it says the same
  if (!put_wait(my_free_flag, epoch, err, timeout_ticks, kWaitPutFree, k)) {
    poison();
    return;
  }
  // (c) my segment -> the neighbour's window
  {
    const WORD* src = reinterpret_cast<const WORD*>(send_buf) + p.send_off;
    WORD* dst = reinterpret_cast<WORD*>(p.dst) + p.dst_off;
    for (int64_t i = (int64_t)g * kBlock + t; i < p.send_count;
         i += (int64_t)kPutGroup * kBlock)
      dst[i] = src[i];
  }
  __threadfence_system(); // my stores are out before anybody sees the flag
  __syncthreads();
  if (t == 0) {
    const unsigned done = atomicAdd(my_counter, 1u) + 1u;
    if (done % kPutGroup == 0) // the last of this neighbour's workgroups
      __hip_atomic_store(p.peer_flags + p.slot_at_peer, epoch, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (d) the neighbour's data -> the ghost tail (loads that bypass the caches:
  // the lines were written by another agent)
  if (!put_wait(my_data_flag, epoch, err, timeout_ticks, kWaitPutData, k)) {
    poison();
    return;
  }
  {
    const WORD* src = reinterpret_cast<const WORD*>(window) + p.recv_off;
    WORD* dst = reinterpret_cast<WORD*>(ghost_tail) + p.recv_off;
    for (int64_t i = (int64_t)g * kBlock + t; i < p.recv_count;
         i += (int64_t)kPutGroup * kBlock)
      dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

size_t put_window_bytes(size_t stage_bytes)
{
  return stage_bytes + sizeof(uint64_t) * 3 * SPMV_HIP_PUT_MAX_PEERS;
}

} // namespace

extern "C" {

int spmv_hip_put_create(spmv_hip_ctx* ctx, size_t stage_bytes,
                        spmv_hip_put** put, void* ipc_handle,
                        uint64_t* raw_address, int64_t* process_id)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(put && ipc_handle && raw_address && process_id);
  static_assert(sizeof(hipIpcMemHandle_t) <= SPMV_HIP_IPC_HANDLE_BYTES,
                "handle size");
  spmv_hip_put* p = new (std::nothrow) spmv_hip_put;
  if (!p)
    return SPMV_HIP_ENOMEM;
  p->ctx = ctx;
  p->stage_bytes = (stage_bytes + 255) & ~(size_t)255;
  for (int k = 0; k < SPMV_HIP_PUT_MAX_PEERS; ++k)
    p->mapped[k] = nullptr;
  const size_t bytes = put_window_bytes(p->stage_bytes);
  // The window is written by PEER devices while a kernel of the owner polls
  // it: that is only defined for fine-grained (uncached) memory -- plain
  // hipMalloc memory is coarse-grained, the owner's L2 may keep serving a
  // stale flag or stale staged data (RCCL allocates its peer-written buffers
  // the same way).  Where the runtime cannot export such memory over IPC the
  // window falls back to hipMalloc and is marked: it then serves ranks on the
  // SAME device only (spmv_hip_put_connect refuses anything else).
  hipIpcMemHandle_t h;
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&p->window), bytes,
                                       hipDeviceMallocFinegrained);
  if (e == hipSuccess)
    e = hipIpcGetMemHandle(&h, p->window);
  if (e == hipSuccess) {
    p->fine_grained = 1;
  } else {
    (void)hipGetLastError();
    (void)hipFree(p->window);
    p->window = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&p->window), bytes);
    if (e == hipSuccess)
      e = hipIpcGetMemHandle(&h, p->window);
  }
  if (e == hipSuccess)
    e = hipMemset(p->window, 0, bytes);
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void**>(&p->host_err),
                      sizeof(int32_t) * kPeerErrWords, hipHostMallocMapped);
  bool watched = false;
  if (e == hipSuccess) {
    memset(p->host_err, 0, sizeof(int32_t) * kPeerErrWords);
    for (int k = 5; k < kPeerErrWords; k = k == 5 ? kPeerErrLabels : k + 1)
      p->host_err[k] = -1; // labels unknown until spmv_hip_put_label
    watched = spmv_ctx_watch(ctx, p->host_err, true);
  }
  if (e != hipSuccess || !watched) { // (never a window whose failures go unseen)
    (void)hipFree(p->window);
    (void)hipHostFree(p->host_err);
    delete p;
    return e != hipSuccess ? static_cast<int>(e) : SPMV_HIP_ENOMEM;
  }
  memset(ipc_handle, 0, SPMV_HIP_IPC_HANDLE_BYTES);
  memcpy(ipc_handle, &h, sizeof(h));
  *raw_address = reinterpret_cast<uint64_t>(p->window);
  *process_id = (int64_t)getpid();
  *put = p;
  return SPMV_HIP_OK;
}

int spmv_hip_put_connect(spmv_hip_put* put, int k, const void* peer_ipc_handle,
                         uint64_t peer_raw_address, int64_t peer_process_id,
                         size_t peer_stage_bytes, int32_t dst_offset,
                         int32_t slot_at_peer, int32_t send_offset,
                         int32_t send_count, int32_t recv_offset,
                         int32_t recv_count, int peer_fine_grained)
{
  SPMV_REQUIRE(put && k >= 0 && k < SPMV_HIP_PUT_MAX_PEERS && peer_ipc_handle
               && slot_at_peer >= 0 && slot_at_peer < SPMV_HIP_PUT_MAX_PEERS
               && dst_offset >= 0 && send_offset >= 0 && send_count >= 0
               && recv_offset >= 0 && recv_count >= 0);
  SPMV_SET_DEVICE(put->ctx);
  // my segments stay inside the two staging buffers (8-byte elements at most)
  SPMV_REQUIRE(((size_t)dst_offset + (size_t)send_count) * 8
                   <= ((peer_stage_bytes + 255) & ~(size_t)255)
               && ((size_t)recv_offset + (size_t)recv_count) * 8 <= put->stage_bytes);
  const bool peer_fine = peer_fine_grained != 0;
  char* base = nullptr;
  if (peer_process_id == (int64_t)getpid()) {
    // a rank of this process (threads): one address space.  On ANOTHER device
    // the window must be fine-grained and this device needs peer access to it.
    base = reinterpret_cast<char*>(peer_raw_address);
    hipPointerAttribute_t attr;
    SPMV_CHECK_HIP(hipPointerGetAttributes(&attr, base));
    if (attr.device != put->ctx->device) {
      if (!peer_fine || !put->fine_grained)
        return SPMV_HIP_ENOTSUP;
      int can = 0;
      SPMV_CHECK_HIP(hipDeviceCanAccessPeer(&can, put->ctx->device, attr.device));
      if (!can)
        return SPMV_HIP_ENOTSUP;
      const hipError_t ep = hipDeviceEnablePeerAccess(attr.device, 0);
      if (ep != hipSuccess && ep != hipErrorPeerAccessAlreadyEnabled)
        return static_cast<int>(ep);
      (void)hipGetLastError();
    }
  } else {
    hipIpcMemHandle_t h;
    memcpy(&h, peer_ipc_handle, sizeof(h));
    void* m = nullptr;
    SPMV_CHECK_HIP(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
    put->mapped[k] = m;
    base = static_cast<char*>(m);
    // a coarse-grained window of another process is only safe on this device
    if (!peer_fine || !put->fine_grained) {
      hipPointerAttribute_t attr;
      SPMV_CHECK_HIP(hipPointerGetAttributes(&attr, m));
      if (attr.device != put->ctx->device)
        return SPMV_HIP_ENOTSUP;
    }
  }
  const size_t peer_stage = (peer_stage_bytes + 255) & ~(size_t)255;
  PutPeer& pp = put->host_tab[k];
  pp.dst = base;
  pp.peer_flags = reinterpret_cast<uint64_t*>(base + peer_stage);
  pp.dst_off = dst_offset;
  pp.slot_at_peer = slot_at_peer;
  pp.send_off = send_offset;
  pp.send_count = send_count;
  pp.recv_off = recv_offset;
  pp.recv_count = recv_count;
  if (k + 1 > put->num_peers)
    put->num_peers = k + 1;
  return SPMV_HIP_OK;
}

int spmv_hip_put_finish(spmv_hip_put* put)
{
  SPMV_REQUIRE(put && put->num_peers > 0);
  SPMV_SET_DEVICE(put->ctx);
  (void)hipFree(put->dev_tab);
  put->dev_tab = nullptr;
  SPMV_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&put->dev_tab),
                           sizeof(PutPeer) * put->num_peers));
  SPMV_CHECK_HIP(hipMemcpy(put->dev_tab, put->host_tab,
                           sizeof(PutPeer) * put->num_peers,
                           hipMemcpyHostToDevice));
  return SPMV_HIP_OK;
}

int spmv_hip_put_exchange(spmv_hip_ctx* ctx, spmv_hip_put* put, size_t elem_bytes,
                          const void* send_buf, void* ghost_tail, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(put && put->ctx == ctx && put->dev_tab && send_buf && ghost_tail
               && (elem_bytes == 4 || elem_bytes == 8));
  if (*put->host_err) // an earlier exchange timed out: the ghosts are not valid
    return SPMV_HIP_EPEER;
  hipStream_t st = spmv_stream(ctx, stream);
  const uint64_t epoch = ++put->epoch;
  const dim3 grid(put->num_peers * kPutGroup), block(kBlock);
  // wall_clock64 ticks at 100 MHz
  const unsigned long long ticks
      = (unsigned long long)ctx->put_timeout_ms * 100000ull;
  int32_t* dev_err = nullptr;
  SPMV_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_err),
                                         put->host_err, 0));
  if (elem_bytes == 8)
    hipLaunchKernelGGL(put_exchange_kernel<uint64_t>, grid, block, 0, st,
                       put->dev_tab, static_cast<const char*>(send_buf),
                       static_cast<char*>(ghost_tail), put->window,
                       put->stage_bytes, epoch, dev_err, ticks);
  else
    hipLaunchKernelGGL(put_exchange_kernel<uint32_t>, grid, block, 0, st,
                       put->dev_tab, static_cast<const char*>(send_buf),
                       static_cast<char*>(ghost_tail), put->window,
                       put->stage_bytes, epoch, dev_err, ticks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_put_label(spmv_hip_put* put, int my_rank, int k, int peer_rank)
{
  SPMV_REQUIRE(put && k >= -1 && k < SPMV_HIP_PUT_MAX_PEERS);
  put->host_err[5] = my_rank;
  if (k >= 0)
    put->host_err[kPeerErrLabels + k] = peer_rank;
  return SPMV_HIP_OK;
}

int spmv_hip_put_fine_grained(const spmv_hip_put* put, int* fine_grained)
{
  SPMV_REQUIRE(put && fine_grained);
  *fine_grained = put->fine_grained;
  return SPMV_HIP_OK;
}

int spmv_hip_put_status(const spmv_hip_put* put, int* failed)
{
  SPMV_REQUIRE(put && failed);
  *failed = *put->host_err != 0;
  return SPMV_HIP_OK;
}

int spmv_hip_put_destroy(spmv_hip_put* put)
{
  if (!put)
    return SPMV_HIP_OK;
  (void)hipSetDevice(put->ctx->device);
  (void)hipDeviceSynchronize(); // no exchange still runs
  for (int k = 0; k < SPMV_HIP_PUT_MAX_PEERS; ++k)
    if (put->mapped[k])
      (void)hipIpcCloseMemHandle(put->mapped[k]);
  spmv_ctx_watch(put->ctx, put->host_err, false);
  (void)hipFree(put->dev_tab);
  (void)hipFree(put->window);
  (void)hipHostFree(put->host_err);
  delete put;
  return SPMV_HIP_OK;
}


// ---------------------------------------------------------------------------
// Deterministic peer reduction of the CG scalars (include/spmv_hip.h)
// ---------------------------------------------------------------------------
} // extern "C"

struct ReduceSlot { // one rank's contribution to one reduction (64 bytes)
  uint64_t epoch;
  double v[SPMV_HIP_REDUCE_MAX_COUNT];
  uint64_t pad[7 - SPMV_HIP_REDUCE_MAX_COUNT];
};
static_assert(sizeof(ReduceSlot) == 64, "slot size");

struct spmv_hip_reduce {
  spmv_hip_ctx* ctx = nullptr;
  int nranks = 0, rank = 0;
  ReduceSlot* window = nullptr; // [2 parities][MAX_RANKS]
  ReduceSlot* host_peers[SPMV_HIP_REDUCE_MAX_RANKS]; // every rank's window
  void* mapped[SPMV_HIP_REDUCE_MAX_RANKS];
  ReduceSlot** dev_peers = nullptr;
  bool tab_dirty = true;
  int32_t* host_err = nullptr;
  int32_t* dev_err = nullptr; // its device address
  uint64_t epoch = 0;
  int fine_grained = 0;
};

namespace
{

__global__ __launch_bounds__(64) void peer_reduce_kernel(
    ReduceSlot* const* __restrict__ peers, int nranks, int me, uint64_t epoch,
    double* __restrict__ inout, int count, int32_t* err,
    unsigned long long timeout_ticks)
{
  __shared__ double s_v[SPMV_HIP_REDUCE_MAX_RANKS][SPMV_HIP_REDUCE_MAX_COUNT];
  __shared__ int s_fail;
  const int t = threadIdx.x;
  const int par = (int)(epoch & 1u) * SPMV_HIP_REDUCE_MAX_RANKS;
  if (t == 0)
    s_fail = 0;
  __syncthreads();
  if (t < nranks) {
    // my contribution into my slot of rank t's window (t == me: my own)
    ReduceSlot* dst = peers[t] + par + me;
    for (int c = 0; c < count; ++c)
      __hip_atomic_store(&dst->v[c], inout[c], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); // the values before the epoch
    __hip_atomic_store(&dst->epoch, epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    // rank t's contribution in MY window
    const ReduceSlot* src = peers[me] + par + t;
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    uint64_t seen;
    while ((seen = __hip_atomic_load(&src->epoch, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM))
           < epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > timeout_ticks) {
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (ok) {
      for (int c = 0; c < count; ++c)
        s_v[t][c] = __hip_atomic_load(&src->v[c], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      s_fail = 1;
      spmv_peer_fail(err, kWaitReduceSlot, t, seen, epoch);
    }
  }
  __syncthreads();
  if (t < count) {
    double sum = 0.0; // rank order: the same bits on every rank
    for (int r = 0; r < nranks; ++r)
      sum += s_v[r][t];
    inout[t] = s_fail ? __builtin_nan("") : sum;
  }
}

} // namespace

extern "C" {

int spmv_hip_reduce_create(spmv_hip_ctx* ctx, int nranks, int rank,
                           spmv_hip_reduce** reduce, void* ipc_handle,
                           uint64_t* raw_address, int64_t* process_id,
                           int* fine_grained)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(reduce && ipc_handle && raw_address && process_id && fine_grained
               && nranks >= 1 && nranks <= SPMV_HIP_REDUCE_MAX_RANKS && rank >= 0
               && rank < nranks);
  spmv_hip_reduce* r = new (std::nothrow) spmv_hip_reduce;
  if (!r)
    return SPMV_HIP_ENOMEM;
  r->ctx = ctx;
  r->nranks = nranks;
  r->rank = rank;
  for (int k = 0; k < SPMV_HIP_REDUCE_MAX_RANKS; ++k) {
    r->host_peers[k] = nullptr;
    r->mapped[k] = nullptr;
  }
  const size_t bytes = sizeof(ReduceSlot) * 2 * SPMV_HIP_REDUCE_MAX_RANKS;
  // (fine-grained where the runtime exports that over IPC: see put_create)
  hipIpcMemHandle_t h;
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&r->window), bytes,
                                       hipDeviceMallocFinegrained);
  if (e == hipSuccess)
    e = hipIpcGetMemHandle(&h, r->window);
  if (e == hipSuccess) {
    r->fine_grained = 1;
  } else {
    (void)hipGetLastError();
    (void)hipFree(r->window);
    r->window = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&r->window), bytes);
    if (e == hipSuccess)
      e = hipIpcGetMemHandle(&h, r->window);
  }
  if (e == hipSuccess)
    e = hipMemset(r->window, 0, bytes);
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void**>(&r->dev_peers),
                  sizeof(ReduceSlot*) * SPMV_HIP_REDUCE_MAX_RANKS);
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void**>(&r->host_err),
                      sizeof(int32_t) * kPeerErrWords, hipHostMallocMapped);
  if (e != hipSuccess) {
    (void)hipFree(r->window);
    (void)hipFree(r->dev_peers);
    delete r;
    return static_cast<int>(e);
  }
  memset(r->host_err, 0, sizeof(int32_t) * kPeerErrWords);
  r->host_err[5] = rank;
  if (!spmv_ctx_watch(ctx, r->host_err, true)) {
    (void)hipFree(r->window);
    (void)hipFree(r->dev_peers);
    (void)hipHostFree(r->host_err);
    delete r;
    return SPMV_HIP_ENOMEM;
  }
  (void)hipHostGetDevicePointer(reinterpret_cast<void**>(&r->dev_err), r->host_err, 0);
  r->host_peers[rank] = r->window;
  if (nranks == 1) { // (nobody to connect)
    (void)hipMemcpy(r->dev_peers, r->host_peers, sizeof(ReduceSlot*),
                    hipMemcpyHostToDevice);
    r->tab_dirty = false;
  }
  memset(ipc_handle, 0, SPMV_HIP_IPC_HANDLE_BYTES);
  memcpy(ipc_handle, &h, sizeof(h));
  *raw_address = reinterpret_cast<uint64_t>(r->window);
  *process_id = (int64_t)getpid();
  *fine_grained = r->fine_grained;
  *reduce = r;
  return SPMV_HIP_OK;
}

int spmv_hip_reduce_connect(spmv_hip_reduce* reduce, int peer_rank,
                            const void* peer_ipc_handle, uint64_t peer_raw_address,
                            int64_t peer_process_id, int peer_fine_grained)
{
  SPMV_REQUIRE(reduce && peer_ipc_handle && peer_rank >= 0
               && peer_rank < reduce->nranks && peer_rank != reduce->rank
               && !reduce->host_peers[peer_rank]);
  SPMV_SET_DEVICE(reduce->ctx);
  const bool fine = peer_fine_grained != 0 && reduce->fine_grained != 0;
  ReduceSlot* base = nullptr;
  if (peer_process_id == (int64_t)getpid()) { // a thread of this process
    base = reinterpret_cast<ReduceSlot*>(peer_raw_address);
    hipPointerAttribute_t attr;
    SPMV_CHECK_HIP(hipPointerGetAttributes(&attr, base));
    if (attr.device != reduce->ctx->device) {
      if (!fine)
        return SPMV_HIP_ENOTSUP;
      int can = 0;
      SPMV_CHECK_HIP(hipDeviceCanAccessPeer(&can, reduce->ctx->device, attr.device));
      if (!can)
        return SPMV_HIP_ENOTSUP;
      const hipError_t ep = hipDeviceEnablePeerAccess(attr.device, 0);
      if (ep != hipSuccess && ep != hipErrorPeerAccessAlreadyEnabled)
        return static_cast<int>(ep);
      (void)hipGetLastError();
    }
  } else {
    hipIpcMemHandle_t h;
    memcpy(&h, peer_ipc_handle, sizeof(h));
    void* m = nullptr;
    SPMV_CHECK_HIP(hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess));
    reduce->mapped[peer_rank] = m;
    base = static_cast<ReduceSlot*>(m);
    if (!fine) { // a coarse-grained window is only safe on this device
      hipPointerAttribute_t attr;
      SPMV_CHECK_HIP(hipPointerGetAttributes(&attr, m));
      if (attr.device != reduce->ctx->device)
        return SPMV_HIP_ENOTSUP;
    }
  }
  reduce->host_peers[peer_rank] = base;
  reduce->tab_dirty = true;
  // the table goes to the device HERE, at set-up time, once it is complete: a
  // synchronous copy inside the first reduction would wait for every stream of
  // the process -- with ranks as threads, for a peer's reduction kernel that
  // in turn waits for this rank's
  bool complete = true;
  for (int k = 0; k < reduce->nranks; ++k)
    complete = complete && reduce->host_peers[k] != nullptr;
  if (complete) {
    SPMV_CHECK_HIP(hipMemcpy(reduce->dev_peers, reduce->host_peers,
                             sizeof(ReduceSlot*) * reduce->nranks,
                             hipMemcpyHostToDevice));
    reduce->tab_dirty = false;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_reduce_sum_f64(spmv_hip_ctx* ctx, spmv_hip_reduce* reduce,
                            double* inout, int count, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(reduce && reduce->ctx == ctx && inout && count >= 1
               && count <= SPMV_HIP_REDUCE_MAX_COUNT);
  for (int k = 0; k < reduce->nranks; ++k)
    SPMV_REQUIRE(reduce->host_peers[k] != nullptr); // every rank connected
  if (*reduce->host_err)
    return SPMV_HIP_EPEER;
  hipStream_t st = spmv_stream(ctx, stream);
  SPMV_REQUIRE(!reduce->tab_dirty);
  const uint64_t epoch = ++reduce->epoch;
  const unsigned long long ticks
      = (unsigned long long)ctx->put_timeout_ms * 100000ull;
  hipLaunchKernelGGL(peer_reduce_kernel, dim3(1), dim3(64), 0, st,
                     reduce->dev_peers, reduce->nranks, reduce->rank, epoch, inout,
                     count, reduce->dev_err, ticks);
  SPMV_CHECK_LAUNCH();
  return SPMV_HIP_OK;
}

int spmv_hip_reduce_destroy(spmv_hip_reduce* reduce)
{
  if (!reduce)
    return SPMV_HIP_OK;
  (void)hipSetDevice(reduce->ctx->device);
  (void)hipDeviceSynchronize(); // no reduction of mine still runs
  for (int k = 0; k < SPMV_HIP_REDUCE_MAX_RANKS; ++k)
    if (reduce->mapped[k])
      (void)hipIpcCloseMemHandle(reduce->mapped[k]);
  spmv_ctx_watch(reduce->ctx, reduce->host_err, false);
  (void)hipFree(reduce->dev_peers);
  (void)hipFree(reduce->window);
  (void)hipHostFree(reduce->host_err);
  delete reduce;
  return SPMV_HIP_OK;
}

} // extern "C"
