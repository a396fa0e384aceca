// Local-to-global map + halo exchange, mirroring spmv/L2GMap.{h,cpp}
// (public interface L2GMap.h:24-175, default-branch plan L2GMap.cpp:346-479,
// p2p update paths :564-642, dispatch :868-905).
//
// What changes on MI355X
//   * transport: one grouped RCCL send/recv over xGMI on a side HIP stream
//     owned by the map; completion is an event the compute stream waits on.
//     The host never blocks in update()/update_finalise().
//   * "blocking" models (p2p_blocking, collective_blocking) make the compute
//     stream wait inside update(); "non-blocking" models (p2p_nonblocking,
//     collective_nonblocking) defer that wait to update_finalise(), so the
//     local-block SpMV overlaps the exchange (Matrix.cpp:498-511).
//   * the one-sided and shmem models (CPU-MPI research variants) are accepted
//     and behave like the blocking p2p model: same data movement, same
//     completion point.
//   * when the indices a neighbour wants are one contiguous run (slab
//     partitions of stencil matrices), the data is sent straight out of the
//     vector and the pack kernel is skipped.
// Plan ordering (neighbour order, offsets, index buffer) is the reference's.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <vector>

#include "comm.h"
#include "executor.h"

struct spmv_hip_put;

namespace spmv
{

class L2GMap
{
public:
  L2GMap(std::shared_ptr<const Comm> comm, std::int64_t local_size,
         const std::vector<std::int64_t>& ghosts,
         std::shared_ptr<DeviceExecutor> exec,
         CommunicationModel cm = CommunicationModel::collective_blocking);
  ~L2GMap();
  L2GMap(const L2GMap&) = delete;
  L2GMap& operator=(const L2GMap&) = delete;

  std::int32_t local_size() const;
  std::int32_t num_ghosts() const { return static_cast<int32_t>(_ghosts.size()); }
  std::int64_t global_size() const { return _ranges.back(); }
  std::int64_t global_offset() const { return _ranges[_rank]; }
  std::int32_t global_to_local(std::int64_t i) const;
  bool overlapping() const;
  const Comm& global_comm() const { return *_comm; }
  std::shared_ptr<const Comm> comm() const { return _comm; }
  int rank() const { return _rank; }
  const std::vector<std::int64_t>& ghosts() const { return _ghosts; }

  // Forward halo: afterwards vec[local_size()+k] holds the owner's value of
  // ghosts()[k]; vec is a DEVICE pointer of local_size()+num_ghosts() elems.
  template <typename T>
  void update(T* vec_data) const;
  template <typename T>
  void update_finalise(T* vec_data) const;
  // Reverse halo (L2GMap.cpp:907-959): every ghost-tail value is sent to its
  // owner and ADDED to the owner's entry; the ghost tail itself is unchanged.
  // Stream-ordered on the executor's stream; the host does not wait.
  template <typename T>
  void reverse_update(T* vec_data) const;

  // Plan inspection (tests compare these with the oracle's restatement).
  const std::vector<int>& neighbours() const { return _neighbours; }
  const std::vector<std::int32_t>& send_count() const { return _send_count; }
  const std::vector<std::int32_t>& recv_count() const { return _recv_count; }
  const std::vector<std::int32_t>& send_offset() const { return _send_offset; }
  const std::vector<std::int32_t>& recv_offset() const { return _recv_offset; }
  const std::vector<std::int32_t>& indexbuf() const { return _indexbuf_host; }
  bool packs() const { return !_direct_send; }
  // the halo moves by peer stores into the neighbours' windows (the
  // onesided_put_* models, when every rank could set its window up)
  bool onesided() const { return _put != nullptr; }

private:
  std::shared_ptr<const Comm> _comm;
  std::shared_ptr<DeviceExecutor> _exec;
  HipExecutor* _hip = nullptr; // non-owning view of _exec when it is a GPU
  CommunicationModel _cm;
  int _rank = 0;

  std::vector<std::int64_t> _ranges;
  std::map<std::int64_t, std::int32_t> _global_to_local;
  std::vector<std::int64_t> _ghosts;

  // NB the reference's naming (L2GMap.cpp:583): _send_* describe what this
  // rank RECEIVES into its ghost tail, _recv_* what it SENDS.
  std::vector<int> _neighbours;
  std::vector<std::int32_t> _send_count, _recv_count;
  std::vector<std::int32_t> _send_offset, _recv_offset;
  std::vector<std::int32_t> _indexbuf_host;
  int _num_indices = 0;
  std::int32_t* _indexbuf = nullptr; // device copy (L2GMap.cpp:473-478)

  // contiguous-run fast path: send straight from the vector
  bool _direct_send = false;
  std::vector<std::int32_t> _direct_offset;

  // argument lists of the grouped exchange (one entry per neighbour), built
  // with the plan: what this rank sends (counts, offsets into the vector or
  // the packed buffer) and receives (counts, offsets into the ghost tail)
  std::vector<std::int32_t> _x_send_count, _x_send_offset, _x_packed_offset;
  std::vector<std::int32_t> _x_recv_count, _x_recv_offset;

  // one-sided models (L2GMap.cpp:645-682): this rank's window and its
  // connections to the neighbours' (libspmv_hip.so, put.hip); null = the
  // two-sided exchange
  spmv_hip_put* _put = nullptr;
  bool _put_agreed = false; // every rank set its window up (same on all ranks)
  void setup_put(std::int64_t local_size);

  mutable void* _send_buf = nullptr; // lazily allocated (L2GMap.cpp:607-614)
  mutable size_t _send_buf_bytes = 0;
  void* _comm_stream = nullptr;
  void* _ev_ready = nullptr; // vec produced on the compute stream
  void* _ev_done = nullptr;  // exchange finished on the comm stream

  template <typename T>
  void start_exchange(T* vec_data) const;
};

} // namespace spmv
