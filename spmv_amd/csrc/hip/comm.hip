// RCCL transport over xGMI: one process per GPU, one communicator per process.
//
// Stands behind the MPI calls of the reference's hot path:
//   - L2GMap::update p2p models (spmv/L2GMap.cpp:564-642): MPI_Irecv straight
//     into the ghost tail + MPI_Isend from the packed buffer become ONE
//     grouped ncclSend/ncclRecv on the caller's (side) stream; completion is
//     an event on that stream instead of MPI_Waitall -- the host never waits.
//   - MPI_Allreduce(1 x double) in cg (spmv/cg.cpp:49,65,75): in-stream
//     ncclAllReduce on the device-resident scalar.
//   - the small host-side collectives of plan construction
//     (L2GMap.cpp:353-354,387-388,444-447) are staged through device memory.
// xGMI is point-to-point; a slab partition talks to <= 2 neighbours, each on
// its own link, so the exchange is latency- not bandwidth-bound.
#include "common.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <new>

static_assert(sizeof(ncclUniqueId) <= SPMV_HIP_UNIQUE_ID_BYTES,
              "unique id does not fit the ABI's byte buffer");

// Two communicators per process.  The grouped halo send/recv runs on the
// map's side stream while the CG loop issues its one-double all-reduces on the
// compute stream; RCCL serialises the operations of ONE communicator and ties
// their streams together, which would take the overlap of the halo with the
// local SpMV away and make correctness depend on an identical issue order on
// every rank.  So the reductions (and the plan-time all-gathers) get their own
// communicator, split off the first one.
struct spmv_hip_comm {
  spmv_hip_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr; // halo exchange (neighbour send/recv)
  ncclComm_t red = nullptr;  // all-reduce / all-gather; == comm if the split
                             // is not available
  int nranks = 1;
  int rank = 0;
};

#define SPMV_CHECK_NCCL(expr)                                                  \
  do {                                                                         \
    ncclResult_t _r = (expr);                                                  \
    if (_r != ncclSuccess)                                                     \
      return 10000 + static_cast<int>(_r);                                     \
  } while (0)

extern "C" {

int spmv_hip_comm_rank(const spmv_hip_comm* comm, int* rank, int* nranks)
{
  SPMV_REQUIRE(comm);
  if (rank)
    *rank = comm->rank;
  if (nranks)
    *nranks = comm->nranks;
  return SPMV_HIP_OK;
}

int spmv_hip_comm_unique_id(void* host_id_bytes)
{
  SPMV_REQUIRE(host_id_bytes);
  ncclUniqueId id;
  SPMV_CHECK_NCCL(ncclGetUniqueId(&id));
  memset(host_id_bytes, 0, SPMV_HIP_UNIQUE_ID_BYTES);
  memcpy(host_id_bytes, &id, sizeof(id));
  return SPMV_HIP_OK;
}

int spmv_hip_comm_create(spmv_hip_ctx* ctx, int nranks, int rank,
                         const void* host_id_bytes, spmv_hip_comm** out)
{
  SPMV_SET_DEVICE(ctx);
  SPMV_REQUIRE(out && host_id_bytes && nranks >= 1 && rank >= 0
               && rank < nranks);
  spmv_hip_comm* c = new (std::nothrow) spmv_hip_comm;
  if (!c)
    return SPMV_HIP_ENOMEM;
  c->ctx = ctx;
  c->nranks = nranks;
  c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, host_id_bytes, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return 10000 + static_cast<int>(r);
  }
  c->red = c->comm;
  if (nranks > 1) {
    // A second communicator for the scalar all-reduces (RCCL serialises the
    // operations of ONE communicator, which would tie the halo stream to the
    // compute stream).  The split is collective over the parent: every rank is
    // in this call right now.  Its OUTCOME must be collective too -- ranks that
    // fell back to the parent while others use the split would issue their
    // all-reduces on different communicators and hang -- so the ranks agree
    // on it with a MIN all-reduce over the parent; if that fails the parent
    // itself is unusable and creation fails.
    ncclComm_t red = nullptr;
    const ncclResult_t rs = ncclCommSplit(c->comm, 0, rank, &red, nullptr);
    int32_t ok = (rs == ncclSuccess && red) ? 1 : 0;
    int32_t* d_ok = nullptr;
    hipError_t e = hipMalloc(&d_ok, sizeof(int32_t));
    if (e == hipSuccess)
      e = hipMemcpy(d_ok, &ok, sizeof(int32_t), hipMemcpyHostToDevice);
    ncclResult_t ra = ncclSuccess;
    if (e == hipSuccess) {
      ra = ncclAllReduce(d_ok, d_ok, 1, ncclInt32, ncclMin, c->comm, nullptr);
      if (ra == ncclSuccess)
        e = hipStreamSynchronize(nullptr);
      if (ra == ncclSuccess && e == hipSuccess)
        e = hipMemcpy(&ok, d_ok, sizeof(int32_t), hipMemcpyDeviceToHost);
    }
    ncclResult_t async = ncclSuccess;
    if (ra == ncclSuccess)
      ra = ncclCommGetAsyncError(c->comm, &async);
    (void)hipFree(d_ok);
    if (e != hipSuccess || ra != ncclSuccess || async != ncclSuccess) {
      if (red)
        (void)ncclCommDestroy(red);
      (void)ncclCommDestroy(c->comm);
      delete c;
      return e != hipSuccess
                 ? static_cast<int>(e)
                 : 10000 + static_cast<int>(ra != ncclSuccess ? ra : async);
    }
    if (ok)
      c->red = red; // on every rank
    else if (red)
      (void)ncclCommDestroy(red); // on every rank: the parent serves both
  }
  *out = c;
  return SPMV_HIP_OK;
}

int spmv_hip_comm_info(const spmv_hip_comm* comm, int* nranks, int* rank,
                       int* rccl_version, int* separate_reduction_comm,
                       char* lib_path, int lib_path_len)
{
  SPMV_REQUIRE(comm);
  if (nranks || rank) {
    // what RCCL itself says, not what the caller passed in
    int n = 0, r = 0;
    SPMV_CHECK_NCCL(ncclCommCount(comm->comm, &n));
    SPMV_CHECK_NCCL(ncclCommUserRank(comm->comm, &r));
    if (nranks)
      *nranks = n;
    if (rank)
      *rank = r;
  }
  if (rccl_version) {
    int v = 0;
    SPMV_CHECK_NCCL(ncclGetVersion(&v));
    *rccl_version = v;
  }
  if (separate_reduction_comm)
    *separate_reduction_comm = comm->red != comm->comm ? 1 : 0;
  if (lib_path && lib_path_len > 0) {
    lib_path[0] = 0;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&ncclGetVersion), &info)
        && info.dli_fname) {
      strncpy(lib_path, info.dli_fname, (size_t)lib_path_len - 1);
      lib_path[lib_path_len - 1] = 0;
    }
  }
  return SPMV_HIP_OK;
}

int spmv_hip_comm_destroy(spmv_hip_comm* comm)
{
  if (!comm)
    return SPMV_HIP_OK;
  (void)hipSetDevice(comm->ctx->device);
  if (comm->red && comm->red != comm->comm)
    (void)ncclCommDestroy(comm->red);
  if (comm->comm)
    (void)ncclCommDestroy(comm->comm);
  delete comm;
  return SPMV_HIP_OK;
}

static int neighbor_exchange(spmv_hip_comm* comm, size_t elem, int num_neighbours,
                             const int32_t* host_neighbours, const char* send_buf,
                             const int32_t* host_send_counts,
                             const int32_t* host_send_offsets, char* recv_base,
                             const int32_t* host_recv_counts,
                             const int32_t* host_recv_offsets, void* stream)
{
  SPMV_REQUIRE(comm && num_neighbours >= 0);
  if (num_neighbours == 0)
    return SPMV_HIP_OK;
  SPMV_REQUIRE(host_neighbours && host_send_counts && host_send_offsets
               && host_recv_counts && host_recv_offsets);
  SPMV_SET_DEVICE(comm->ctx);
  hipStream_t st = spmv_stream(comm->ctx, stream);
  for (int i = 0; i < num_neighbours; ++i) {
    // (a neighbour is another rank -- except on a ONE-rank communicator, where
    // rank 0 may exchange with itself: a loopback through RCCL's own send /
    // recv pair, which lets a 1-GPU box execute this grouped call for real;
    // L2GMap never asks for it)
    SPMV_REQUIRE(host_neighbours[i] >= 0 && host_neighbours[i] < comm->nranks
                 && (host_neighbours[i] != comm->rank || comm->nranks == 1));
    SPMV_REQUIRE(host_send_counts[i] >= 0 && host_recv_counts[i] >= 0);
    SPMV_REQUIRE(host_send_counts[i] == 0 || send_buf);
    SPMV_REQUIRE(host_recv_counts[i] == 0 || recv_base);
  }
  SPMV_CHECK_NCCL(ncclGroupStart());
  ncclResult_t r = ncclSuccess;
  for (int i = 0; i < num_neighbours && r == ncclSuccess; ++i) {
    if (host_recv_counts[i] > 0) // L2GMap.cpp:624-628
      r = ncclRecv(recv_base + (size_t)host_recv_offsets[i] * elem,
                   (size_t)host_recv_counts[i] * elem, ncclChar,
                   host_neighbours[i], comm->comm, st);
    if (r == ncclSuccess && host_send_counts[i] > 0) // L2GMap.cpp:630-634
      r = ncclSend(send_buf + (size_t)host_send_offsets[i] * elem,
                   (size_t)host_send_counts[i] * elem, ncclChar,
                   host_neighbours[i], comm->comm, st);
  }
  ncclResult_t r2 = ncclGroupEnd();
  if (r != ncclSuccess)
    return 10000 + static_cast<int>(r);
  SPMV_CHECK_NCCL(r2);
  return SPMV_HIP_OK;
}

int spmv_hip_comm_neighbor_exchange_f64(
    spmv_hip_comm* comm, int num_neighbours, const int32_t* host_neighbours,
    const double* send_buf, const int32_t* host_send_counts,
    const int32_t* host_send_offsets, double* recv_base,
    const int32_t* host_recv_counts, const int32_t* host_recv_offsets,
    void* stream)
{
  return neighbor_exchange(comm, sizeof(double), num_neighbours,
                           host_neighbours, (const char*)send_buf,
                           host_send_counts, host_send_offsets,
                           (char*)recv_base, host_recv_counts,
                           host_recv_offsets, stream);
}

int spmv_hip_comm_neighbor_exchange_f32(
    spmv_hip_comm* comm, int num_neighbours, const int32_t* host_neighbours,
    const float* send_buf, const int32_t* host_send_counts,
    const int32_t* host_send_offsets, float* recv_base,
    const int32_t* host_recv_counts, const int32_t* host_recv_offsets,
    void* stream)
{
  return neighbor_exchange(comm, sizeof(float), num_neighbours, host_neighbours,
                           (const char*)send_buf, host_send_counts,
                           host_send_offsets, (char*)recv_base,
                           host_recv_counts, host_recv_offsets, stream);
}

int spmv_hip_comm_allreduce_sum_f64(spmv_hip_comm* comm, double* inout,
                                    size_t count, void* stream)
{
  SPMV_REQUIRE(comm && (count == 0 || inout));
  if (count == 0)
    return SPMV_HIP_OK;
  // (a one-rank communicator goes through RCCL too: the call is then a copy
  // onto itself, and a 1-GPU box has executed the very call the N-rank run
  // makes; one-rank RUNS use SelfComm, which has no transport at all)
  SPMV_SET_DEVICE(comm->ctx);
  SPMV_CHECK_NCCL(ncclAllReduce(inout, inout, count, ncclDouble, ncclSum,
                                comm->red, spmv_stream(comm->ctx, stream)));
  return SPMV_HIP_OK;
}

int spmv_hip_comm_allgather_host(spmv_hip_comm* comm, const void* host_send,
                                 void* host_recv, size_t bytes_per_rank)
{
  SPMV_REQUIRE(comm && (bytes_per_rank == 0 || (host_send && host_recv)));
  if (bytes_per_rank == 0)
    return SPMV_HIP_OK;
  if (comm->nranks == 1) {
    memcpy(host_recv, host_send, bytes_per_rank);
    return SPMV_HIP_OK;
  }
  SPMV_SET_DEVICE(comm->ctx);
  hipStream_t st = comm->ctx->stream;
  char *d_send = nullptr, *d_recv = nullptr;
  SPMV_CHECK_HIP(hipMalloc(&d_send, bytes_per_rank));
  hipError_t e = hipMalloc(&d_recv, bytes_per_rank * comm->nranks);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_send, host_send, bytes_per_rank,
                       hipMemcpyHostToDevice, st);
  if (e == hipSuccess)
    r = ncclAllGather(d_send, d_recv, bytes_per_rank, ncclChar, comm->red, st);
  if (e == hipSuccess && r == ncclSuccess)
    e = hipMemcpyAsync(host_recv, d_recv, bytes_per_rank * comm->nranks,
                       hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && r == ncclSuccess)
    e = hipStreamSynchronize(st);
  (void)hipFree(d_send);
  (void)hipFree(d_recv);
  if (r != ncclSuccess)
    return 10000 + static_cast<int>(r);
  if (e != hipSuccess)
    return static_cast<int>(e);
  return SPMV_HIP_OK;
}

} // extern "C"
