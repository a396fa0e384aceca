// Communicator abstraction replacing MPI_Comm in the mirrored signatures
// (L2GMap, Matrix::create_matrix, cg).  MPI may be absent on a GPU box and
// the hot path runs on RCCL over xGMI, one process per GPU (SURVEY section
// 8b "MPI in signatures").  Argument order of the mirrored functions is
// otherwise identical to the reference.
//
//   host side  : allgather() is the only primitive; plan construction builds
//                the reference's Allgather / Alltoall / Neighbor_alltoallv
//                (L2GMap.cpp:353-354,387-388,444-447) on top of it.  Setup
//                only, untimed.
//   device side: neighbor_exchange() = the p2p halo (L2GMap.cpp:564-642),
//                allreduce_sum() = MPI_Allreduce(1 x double) of cg
//                (cg.cpp:49,65,75); both are stream-ordered, the host never
//                waits.
//
// Implementations: SelfComm (1 rank, no transport), RcclComm (libspmv_hip.so's
// RCCL wrappers), CallbackComm (function pointers supplied by the embedding
// program -- e.g. MPI in the reference's own drivers, or torch.distributed in
// this repo's tests).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct spmv_hip_comm;
struct spmv_hip_reduce;
struct spmv_hip_ctx;

namespace spmv
{

class HipExecutor;

// same enumerators, same order as spmv/mpi_utils.h:43-52
enum class CommunicationModel {
  p2p_blocking,
  p2p_nonblocking,
  collective_blocking,
  collective_nonblocking,
  onesided_put_active,
  onesided_put_passive,
  shmem,
  shmem_nodup
};

class Comm
{
public:
  virtual ~Comm() = default;
  virtual int rank() const = 0;
  virtual int size() const = 0;

  // recv[r*bytes_per_rank ...] = rank r's send buffer (host memory)
  virtual void allgather(const void* send, void* recv,
                         size_t bytes_per_rank) const = 0;

  // Grouped neighbour exchange on DEVICE memory, enqueued on `stream`:
  // for each neighbour i send send_counts[i] elements of `elem_bytes` from
  // send_buf + send_offsets[i], receive recv_counts[i] into
  // recv_base + recv_offsets[i] (offsets in elements).
  virtual void neighbor_exchange(size_t elem_bytes,
                                 const std::vector<int>& neighbours,
                                 const void* send_buf,
                                 const std::vector<int32_t>& send_counts,
                                 const std::vector<int32_t>& send_offsets,
                                 void* recv_base,
                                 const std::vector<int32_t>& recv_counts,
                                 const std::vector<int32_t>& recv_offsets,
                                 void* stream) const = 0;

  // In-place sum of `count` device doubles over all ranks, on `stream`.
  virtual void allreduce_sum(double* device_inout, size_t count,
                             void* stream) const = 0;

  // ---- deterministic peer reduction of the CG scalars (opt-in) -------------
  // cg() reduces through reduce_sum(): the transport's allreduce_sum() unless
  // enable_peer_reduce() has set up the peer windows (spmv_hip_reduce_*: one
  // single-wave kernel per rank and reduction, values added in rank order --
  // the same bits on every rank and in every run, no RCCL launch).  Both calls
  // are COLLECTIVE over the communicator; close_peer_reduce() before the
  // communicator (or the executor) goes away -- derived destructors do.
  // Returns false (and stays on allreduce_sum) when some rank cannot reach
  // some other rank's window -- or when an L2GMap with the ONE-SIDED halo
  // lives on this communicator AND two ranks of it share a process (ranks as
  // threads): that pair is refused by the library, on every rank alike
  // (conversely an L2GMap built after enable_peer_reduce() then falls back to
  // the two-sided exchange).  Why: DESIGN.md section 6 -- with both in use
  // the exchanges of 2 to 8 thread ranks time out whatever the host's
  // run-ahead, while one process per rank (the production topology) runs the
  // pair.  The executor closes the reduction of a communicator that outlives
  // it.  reduce_sum() relies on stream order between consecutive reductions
  // (double-buffered slots): a call on another stream than the previous one
  // drains that one first.
  bool enable_peer_reduce(const HipExecutor& exec) const;
  void close_peer_reduce() const;
  bool peer_reduce() const { return _reduce != nullptr; }
  void reduce_sum(double* device_inout, size_t count, void* stream) const;
  // L2GMap's bookkeeping of one-sided maps on this communicator (collective
  // by construction: every rank builds and destroys the same maps)
  void note_onesided_map(int delta) const { _onesided_maps += delta; }
  int onesided_maps() const { return _onesided_maps; }
  // SPMV_ALLOW_PUT_WITH_PEER_REDUCE=1 lifts the refusal (probes only)
  static bool pair_allowed();
  // two ranks of this communicator live in one process (collective on its
  // first call: the process ids are exchanged once)
  bool ranks_share_a_process() const;

  // ---- helpers built on allgather (host, setup only) ----------------------
  template <typename T>
  std::vector<T> allgather_value(const T& v) const
  {
    std::vector<T> out(size());
    allgather(&v, out.data(), sizeof(T));
    return out;
  }
  // every rank contributes a vector of arbitrary length; returns all of them
  std::vector<std::vector<int32_t>>
  allgatherv(const std::vector<int32_t>& mine) const;
  // byte-level variant: out[r] = rank r's buffer
  std::vector<std::vector<unsigned char>>
  allgatherv_bytes(const void* mine, size_t num_bytes) const;

private:
  mutable spmv_hip_reduce* _reduce = nullptr;
  mutable spmv_hip_ctx* _reduce_ctx = nullptr;
  mutable const HipExecutor* _reduce_exec = nullptr;
  mutable void* _reduce_stream = nullptr;
  mutable bool _reduce_stream_set = false;
  mutable int _onesided_maps = 0;
  mutable int _shared_process = -1; // unknown until asked
};

class SelfComm final : public Comm
{
public:
  ~SelfComm() override
  {
    try {
      close_peer_reduce();
    } catch (...) {
    }
  }
  int rank() const override { return 0; }
  int size() const override { return 1; }
  void allgather(const void* send, void* recv, size_t bytes) const override;
  void neighbor_exchange(size_t, const std::vector<int>&, const void*,
                         const std::vector<int32_t>&,
                         const std::vector<int32_t>&, void*,
                         const std::vector<int32_t>&,
                         const std::vector<int32_t>&, void*) const override;
  void allreduce_sum(double*, size_t, void*) const override {}
};

class RcclComm final : public Comm
{
public:
  // `unique_id` = SPMV_HIP_UNIQUE_ID_BYTES bytes produced by
  // RcclComm::unique_id() on one rank and handed to every rank by the
  // launcher (torch.distributed store, MPI_Bcast, a file, ...).
  RcclComm(const HipExecutor& exec, int nranks, int rank,
           const void* unique_id);
  ~RcclComm() override;
  static std::vector<unsigned char> unique_id();

  int rank() const override { return _rank; }
  int size() const override { return _size; }
  void allgather(const void* send, void* recv, size_t bytes) const override;
  void neighbor_exchange(size_t elem_bytes, const std::vector<int>& neighbours,
                         const void* send_buf,
                         const std::vector<int32_t>& send_counts,
                         const std::vector<int32_t>& send_offsets,
                         void* recv_base,
                         const std::vector<int32_t>& recv_counts,
                         const std::vector<int32_t>& recv_offsets,
                         void* stream) const override;
  void allreduce_sum(double* device_inout, size_t count,
                     void* stream) const override;

  // what the transport really is (spmv_hip_comm_info): for benchmark records
  struct Info {
    int nranks = 0, rank = 0, rccl_version = 0, separate_reduction_comm = 0;
    std::string lib_path;
  };
  Info info() const;

private:
  spmv_hip_comm* _comm = nullptr;
  int _rank = 0, _size = 1;
};

// Transport supplied as C callbacks (all must return 0 on success).
struct CommCallbacks {
  void* user = nullptr;
  int (*allgather)(void* user, const void* send, void* recv,
                   size_t bytes_per_rank) = nullptr;
  // optional: device transport; when null the device calls throw
  int (*neighbor_exchange)(void* user, size_t elem_bytes, int num_neighbours,
                           const int* neighbours, const void* send_buf,
                           const int32_t* send_counts,
                           const int32_t* send_offsets, void* recv_base,
                           const int32_t* recv_counts,
                           const int32_t* recv_offsets, void* stream) = nullptr;
  int (*allreduce_sum)(void* user, double* device_inout, size_t count,
                       void* stream) = nullptr;
};

class CallbackComm final : public Comm
{
public:
  CallbackComm(int rank, int size, CommCallbacks cb)
      : _rank(rank), _size(size), _cb(cb)
  {
  }
  ~CallbackComm() override;
  int rank() const override { return _rank; }
  int size() const override { return _size; }
  void allgather(const void* send, void* recv, size_t bytes) const override;
  void neighbor_exchange(size_t elem_bytes, const std::vector<int>& neighbours,
                         const void* send_buf,
                         const std::vector<int32_t>& send_counts,
                         const std::vector<int32_t>& send_offsets,
                         void* recv_base,
                         const std::vector<int32_t>& recv_counts,
                         const std::vector<int32_t>& recv_offsets,
                         void* stream) const override;
  void allreduce_sum(double* device_inout, size_t count,
                     void* stream) const override;

private:
  int _rank, _size;
  CommCallbacks _cb;
};

} // namespace spmv
