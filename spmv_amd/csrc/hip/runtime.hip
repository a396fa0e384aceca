// Context, memory, stream and event plumbing of libspmv_hip.so.
// Stands behind DeviceExecutor's byte-level virtuals
// (spmv/device_executor.h:129-139) and CudaExecutor's stream accessors
// (spmv/cuda/cuda_executor.h:72-76).
#include "common.h"

#include <cstdio>
#include <cstring>
#include <new>

// The put windows' error words (put.hip): a put kernel that gave up waiting for
// its neighbour leaves NaN ghosts and sets its word; the host learns of it at
// the next point where it waits for the device anyway.
bool spmv_ctx_watch(spmv_hip_ctx* ctx, const int32_t* word, bool add)
{
  std::lock_guard<std::mutex> lock(ctx->watched_mutex);
  auto& ws = ctx->watched;
  if (add) {
    try {
      ws.push_back(word);
    } catch (...) {
      return false;
    }
    return true;
  }
  for (size_t i = 0; i < ws.size(); ++i)
    if (ws[i] == word) {
      ws[i] = ws.back();
      ws.pop_back();
      break;
    }
  return true;
}

int spmv_ctx_check_watched(const spmv_hip_ctx* ctx)
{
  std::lock_guard<std::mutex> lock(const_cast<spmv_hip_ctx*>(ctx)->watched_mutex);
  for (const int32_t* w : ctx->watched)
    if (*reinterpret_cast<const volatile int32_t*>(w))
      return SPMV_HIP_EPEER;
  return SPMV_HIP_OK;
}

extern "C" {

int spmv_hip_peer_error_detail(const spmv_hip_ctx* ctx, char* buf, size_t len)
{
  SPMV_REQUIRE(ctx && buf && len > 0);
  buf[0] = 0;
  std::lock_guard<std::mutex> lock(const_cast<spmv_hip_ctx*>(ctx)->watched_mutex);
  for (const int32_t* w : ctx->watched) {
    const volatile int32_t* v = reinterpret_cast<const volatile int32_t*>(w);
    if (!v[0])
      continue;
    if (!v[1]) { // (flag without a claimed record: cannot happen after a sync)
      snprintf(buf, len, "a bounded wait timed out (no record)");
      return SPMV_HIP_OK;
    }
    const int which = v[2], peer = v[3];
    const unsigned long long seen
        = (unsigned long long)(uint32_t)v[6] | ((unsigned long long)(uint32_t)v[7] << 32);
    const unsigned long long wanted
        = (unsigned long long)(uint32_t)v[8] | ((unsigned long long)(uint32_t)v[9] << 32);
    if (which == kWaitReduceSlot)
      snprintf(buf, len,
               "rank %d: reduction kernel of epoch %llu timed out waiting for rank "
               "%d's slot in its window (slot shows epoch %llu: that rank's "
               "reduction kernel of this epoch never ran)",
               (int)v[5], wanted, peer, seen);
    else
      snprintf(buf, len,
               "rank %d: put kernel of epoch %llu (workgroup %d) timed out waiting "
               "for the %s flag of neighbour slot %d (rank %d); the flag shows "
               "epoch %llu: %s",
               (int)v[5], wanted, (int)v[4],
               which == kWaitPutFree ? "FREE" : "DATA", peer,
               peer >= 0 && peer < 16 ? (int)v[kPeerErrLabels + peer] : -1, seen,
               which == kWaitPutFree
                   ? "that neighbour's put kernel of this epoch never started"
                   : "that neighbour's put kernel started (or not) but its stores "
                     "never completed");
    return SPMV_HIP_OK;
  }
  return SPMV_HIP_OK;
}

int spmv_hip_abi_version(void) { return SPMV_HIP_ABI_VERSION; }

const char* spmv_hip_error_string(int code)
{
  if (code == SPMV_HIP_OK)
    return "success";
  if (code == SPMV_HIP_EINVAL)
    return "spmv_hip: invalid argument";
  if (code == SPMV_HIP_ENOMEM)
    return "spmv_hip: host allocation failed";
  if (code == SPMV_HIP_ENOTSUP)
    return "spmv_hip: not supported in this build";
  if (code == SPMV_HIP_ERANGE)
    return "spmv_hip: size exceeds 32-bit index range";
  if (code == SPMV_HIP_EPEER)
    return "spmv_hip: one-sided halo: a neighbour did not answer in time";
  if (code >= 10000)
    return "spmv_hip: RCCL error (code - 10000 = ncclResult_t)";
  if (code > 0)
    return hipGetErrorString(static_cast<hipError_t>(code));
  return "spmv_hip: unknown error";
}

int spmv_hip_device_count(int* count)
{
  SPMV_REQUIRE(count != nullptr);
  SPMV_CHECK_HIP(hipGetDeviceCount(count));
  return SPMV_HIP_OK;
}

int spmv_hip_ctx_create(int device_id, spmv_hip_ctx** out)
{
  SPMV_REQUIRE(out != nullptr && device_id >= 0);
  int count = 0;
  SPMV_CHECK_HIP(hipGetDeviceCount(&count));
  SPMV_REQUIRE(device_id < count);
  SPMV_CHECK_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  SPMV_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
  spmv_hip_ctx* ctx = new (std::nothrow) spmv_hip_ctx;
  if (!ctx)
    return SPMV_HIP_ENOMEM;
  ctx->device = device_id;
  ctx->num_cus = prop.multiProcessorCount;
  ctx->dot_blocks = ctx->num_cus * kBlocksPerCU;
  *out = ctx;
  return SPMV_HIP_OK;
}

int spmv_hip_ctx_destroy(spmv_hip_ctx* ctx)
{
  delete ctx;
  return SPMV_HIP_OK;
}

int spmv_hip_ctx_device(const spmv_hip_ctx* ctx, int* device_id)
{
  SPMV_REQUIRE(ctx && device_id);
  *device_id = ctx->device;
  return SPMV_HIP_OK;
}

int spmv_hip_num_cus(const spmv_hip_ctx* ctx, int* num_cus)
{
  SPMV_REQUIRE(ctx && num_cus);
  *num_cus = ctx->num_cus;
  return SPMV_HIP_OK;
}

int spmv_hip_ctx_get_option(const spmv_hip_ctx* ctx, const char* key, int64_t* value)
{
  SPMV_REQUIRE(ctx && key && value);
  if (!strcmp(key, "release_csr"))
    *value = ctx->release_csr;
  else if (!strcmp(key, "put_timeout_ms"))
    *value = ctx->put_timeout_ms;
  else if (!strcmp(key, "xw_min_nnz"))
    *value = ctx->xw_min_nnz;
  else if (!strcmp(key, "xw_min_x_bytes"))
    *value = ctx->xw_min_x_bytes;
  else if (!strcmp(key, "xw_probe"))
    *value = ctx->xw_probe;
  else if (!strcmp(key, "csr_in_place"))
    *value = ctx->csr_in_place;
  else if (!strcmp(key, "lx_min_nnz"))
    *value = ctx->lx_min_nnz;
  else if (!strcmp(key, "lat_min_nnz"))
    *value = ctx->lat_min_nnz;
  else if (!strcmp(key, "sj_min_nnz"))
    *value = ctx->sj_min_nnz;
  else
    return SPMV_HIP_EINVAL;
  return SPMV_HIP_OK;
}

int spmv_hip_ctx_set_option(spmv_hip_ctx* ctx, const char* key, int64_t value)
{
  SPMV_REQUIRE(ctx && key);
  if (!strcmp(key, "blas1_nt_min_elems")) {
    SPMV_REQUIRE(value >= 0);
    ctx->blas1_nt_min_elems = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "lx_max_x_bytes")) {
    SPMV_REQUIRE(value >= 0);
    ctx->lx_max_x_bytes = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "lx_min_nnz")) {
    SPMV_REQUIRE(value >= 0);
    ctx->lx_min_nnz = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "release_csr")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->release_csr = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "xw_min_x_bytes")) {
    SPMV_REQUIRE(value >= 0);
    ctx->xw_min_x_bytes = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "xw_min_nnz")) {
    SPMV_REQUIRE(value >= 0);
    ctx->xw_min_nnz = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "xw_probe")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->xw_probe = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "csr_in_place")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->csr_in_place = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "lat_min_nnz")) {
    SPMV_REQUIRE(value >= 0);
    ctx->lat_min_nnz = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "poisson_skew_ppm")) {
    SPMV_REQUIRE(value >= 0 && value < 1000000);
    ctx->poisson_skew_ppm = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "lx_dma")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->lx_dma = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "poisson_stencil")) {
    SPMV_REQUIRE(value == 7 || value == 27);
    ctx->poisson_stencil = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "const_diagonals")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->const_diagonals = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "const_tile")) {
    SPMV_REQUIRE(value == 1 || value == 2 || value == 4);
    ctx->const_tile = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "wdia_half")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->wdia_half = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sj_min_nnz")) {
    SPMV_REQUIRE(value >= 0);
    ctx->sj_min_nnz = value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sj_max_chunks")) {
    SPMV_REQUIRE(value >= 8 && value <= 432);
    ctx->sj_max_chunks = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sj_wpb")) {
    SPMV_REQUIRE(value == 0 || value == 4 || value == 8 || value == 16);
    ctx->sj_wpb = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sj_unit")) {
    SPMV_REQUIRE(value == 0 || value == 1 || value == 2 || value == 4);
    ctx->sj_unit = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sj_sigma")) {
    SPMV_REQUIRE(value >= 0 && value <= 2); // 2: whatever the rows' average length
    ctx->sj_sigma = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sym_sj_long_rows")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->sym_sj_long_rows = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "sym_sj_long_permille")) {
    SPMV_REQUIRE(value >= 0 && value <= 1000);
    ctx->sym_sj_long_permille = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "put_timeout_ms")) {
    SPMV_REQUIRE(value >= 1 && value <= 3600000);
    ctx->put_timeout_ms = (int)value;
    return SPMV_HIP_OK;
  }
  if (!strcmp(key, "bake_general")) {
    SPMV_REQUIRE(value == 0 || value == 1);
    ctx->bake_general = (int)value;
    return SPMV_HIP_OK;
  }
  return SPMV_HIP_EINVAL;
}

int spmv_hip_synchronize(spmv_hip_ctx* ctx)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_CHECK_HIP(hipDeviceSynchronize());
  return spmv_ctx_check_watched(ctx);
}

// ---- streams / events ------------------------------------------------------
int spmv_hip_stream_create(spmv_hip_ctx* ctx, void** stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(stream);
  hipStream_t s;
  SPMV_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return SPMV_HIP_OK;
}

int spmv_hip_stream_create_priority(spmv_hip_ctx* ctx, int high_priority,
                                    void** stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(stream);
  int least = 0, greatest = 0; // numerically lower = higher priority
  SPMV_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t s;
  SPMV_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking,
                                             high_priority ? greatest : least));
  *stream = s;
  return SPMV_HIP_OK;
}

int spmv_hip_stream_destroy(spmv_hip_ctx* ctx, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  if (stream) {
    if (ctx->stream == stream)
      ctx->stream = nullptr;
    SPMV_CHECK_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  }
  return SPMV_HIP_OK;
}

int spmv_hip_stream_synchronize(spmv_hip_ctx* ctx, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_CHECK_HIP(hipStreamSynchronize(spmv_stream(ctx, stream)));
  return spmv_ctx_check_watched(ctx);
}

int spmv_hip_set_stream(spmv_hip_ctx* ctx, void* stream)
{
  SPMV_REQUIRE(ctx);
  ctx->stream = static_cast<hipStream_t>(stream);
  return SPMV_HIP_OK;
}

int spmv_hip_get_stream(const spmv_hip_ctx* ctx, void** stream)
{
  SPMV_REQUIRE(ctx && stream);
  *stream = ctx->stream;
  return SPMV_HIP_OK;
}

int spmv_hip_event_create(spmv_hip_ctx* ctx, int timing, void** event)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(event);
  hipEvent_t e;
  SPMV_CHECK_HIP(hipEventCreateWithFlags(
      &e, timing ? hipEventDefault : hipEventDisableTiming));
  *event = e;
  return SPMV_HIP_OK;
}

int spmv_hip_event_destroy(spmv_hip_ctx* ctx, void* event)
{
  SPMV_SET_DEVICE(ctx);
  if (event)
    SPMV_CHECK_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return SPMV_HIP_OK;
}

int spmv_hip_event_record(spmv_hip_ctx* ctx, void* event, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(event);
  SPMV_CHECK_HIP(hipEventRecord(static_cast<hipEvent_t>(event),
                                spmv_stream(ctx, stream)));
  return SPMV_HIP_OK;
}

int spmv_hip_event_synchronize(spmv_hip_ctx* ctx, void* event)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(event);
  SPMV_CHECK_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return spmv_ctx_check_watched(ctx);
}

int spmv_hip_stream_wait_event(spmv_hip_ctx* ctx, void* stream, void* event)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(event);
  SPMV_CHECK_HIP(hipStreamWaitEvent(spmv_stream(ctx, stream),
                                    static_cast<hipEvent_t>(event), 0));
  return SPMV_HIP_OK;
}

int spmv_hip_event_elapsed_ms(spmv_hip_ctx* ctx, void* start, void* stop,
                              float* ms)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(start && stop && ms);
  SPMV_CHECK_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start),
                                     static_cast<hipEvent_t>(stop)));
  return SPMV_HIP_OK;
}

// ---- memory ------------------------------------------------------------------
int spmv_hip_alloc(spmv_hip_ctx* ctx, size_t num_bytes, void** ptr)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(ptr);
  *ptr = nullptr;
  if (num_bytes == 0)
    return SPMV_HIP_OK;
  SPMV_CHECK_HIP(hipMalloc(ptr, num_bytes));
  return SPMV_HIP_OK;
}

int spmv_hip_free(spmv_hip_ctx* ctx, void* ptr)
{
  SPMV_SET_DEVICE(ctx);
  if (ptr)
    SPMV_CHECK_HIP(hipFree(ptr));
  return SPMV_HIP_OK;
}

int spmv_hip_host_alloc(spmv_hip_ctx* ctx, size_t num_bytes, void** host_ptr)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(host_ptr);
  *host_ptr = nullptr;
  if (num_bytes == 0)
    return SPMV_HIP_OK;
  SPMV_CHECK_HIP(hipHostMalloc(host_ptr, num_bytes, hipHostMallocDefault));
  return SPMV_HIP_OK;
}

int spmv_hip_host_free(spmv_hip_ctx* ctx, void* host_ptr)
{
  SPMV_SET_DEVICE(ctx);
  if (host_ptr)
    SPMV_CHECK_HIP(hipHostFree(host_ptr));
  return SPMV_HIP_OK;
}

int spmv_hip_memset_async(spmv_hip_ctx* ctx, void* ptr, int value,
                          size_t num_bytes, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  if (num_bytes == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(ptr);
  SPMV_CHECK_HIP(hipMemsetAsync(ptr, value, num_bytes,
                                spmv_stream(ctx, stream)));
  return SPMV_HIP_OK;
}

static int copy_async(spmv_hip_ctx* ctx, void* dst, const void* src,
                      size_t num_bytes, hipMemcpyKind kind, void* stream)
{
  SPMV_SET_DEVICE(ctx);
  if (num_bytes == 0 || dst == src)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(dst && src);
  SPMV_CHECK_HIP(hipMemcpyAsync(dst, src, num_bytes, kind,
                                spmv_stream(ctx, stream)));
  return SPMV_HIP_OK;
}

int spmv_hip_copy_d2d_async(spmv_hip_ctx* ctx, void* dst, const void* src,
                            size_t num_bytes, void* stream)
{
  return copy_async(ctx, dst, src, num_bytes, hipMemcpyDeviceToDevice, stream);
}

int spmv_hip_copy_h2d_async(spmv_hip_ctx* ctx, void* dst,
                            const void* host_src, size_t num_bytes,
                            void* stream)
{
  return copy_async(ctx, dst, host_src, num_bytes, hipMemcpyHostToDevice,
                    stream);
}

int spmv_hip_copy_d2h_async(spmv_hip_ctx* ctx, void* host_dst, const void* src,
                            size_t num_bytes, void* stream)
{
  return copy_async(ctx, host_dst, src, num_bytes, hipMemcpyDeviceToHost,
                    stream);
}

int spmv_hip_copy_peer_async(spmv_hip_ctx* dst_ctx, void* dst,
                             spmv_hip_ctx* src_ctx, const void* src,
                             size_t num_bytes, void* stream)
{
  SPMV_REQUIRE(dst_ctx && src_ctx);
  SPMV_SET_DEVICE(dst_ctx);
  if (num_bytes == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(dst && src);
  SPMV_CHECK_HIP(hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device,
                                    num_bytes, spmv_stream(dst_ctx, stream)));
  return SPMV_HIP_OK;
}

} // extern "C"
