// L2GMap: see l2gmap.h.  Plan construction follows spmv/L2GMap.cpp:346-479.
#include "l2gmap.h"

#include "spmv_hip.h"

#include <algorithm>
#include <stdexcept>

namespace spmv
{

L2GMap::L2GMap(std::shared_ptr<const Comm> comm, std::int64_t local_size,
               const std::vector<std::int64_t>& ghosts,
               std::shared_ptr<DeviceExecutor> exec, CommunicationModel cm)
    : _comm(std::move(comm)), _exec(std::move(exec)), _cm(cm), _ghosts(ghosts)
{
  // All eight models of the reference are accepted.  They differ only in the
  // MPI mechanism that moves the ghosts (L2GMap.cpp:868-896); here every one
  // of them is the same grouped RCCL send/recv.  Only the two *_nonblocking
  // models defer completion to update_finalise() (overlapping(), :975-981);
  // the one-sided and shmem models complete inside update() like the
  // reference's.
  _hip = dynamic_cast<HipExecutor*>(_exec.get());
  const int P = _comm->size();
  _rank = _comm->rank();

  // ownership ranges (L2GMap.cpp:351-356)
  std::vector<std::int64_t> sizes = _comm->allgather_value<std::int64_t>(local_size);
  _ranges.assign(P + 1, 0);
  for (int r = 0; r < P; ++r)
    _ranges[r + 1] = _ranges[r] + sizes[r];
  const std::int64_t r0 = _ranges[_rank], r1 = _ranges[_rank + 1];

  if (!std::is_sorted(_ghosts.begin(), _ghosts.end()))
    throw std::runtime_error("Ghosts must be sorted"); // :362-363

  // owner and owner-local index of every ghost (:366-381)
  std::vector<std::int32_t> ghost_count(P, 0), ghost_local;
  ghost_local.reserve(_ghosts.size());
  for (std::size_t i = 0; i < _ghosts.size(); ++i) {
    const std::int64_t idx = _ghosts[i];
    if (idx >= r0 && idx < r1)
      throw std::runtime_error("Ghost index in local range"); // :371-372
    if (idx < 0 || idx >= _ranges[P])
      throw std::runtime_error("Ghost index outside the global range");
    _global_to_local.insert({idx, static_cast<std::int32_t>(local_size + i)});
    const int p = static_cast<int>(
        std::upper_bound(_ranges.begin(), _ranges.end(), idx) - _ranges.begin()
        - 1);
    ++ghost_count[p];
    ghost_local.push_back(static_cast<std::int32_t>(idx - _ranges[p]));
  }

  // who needs what from whom: the reference's Alltoall (:386-388) as an
  // allgather of every rank's per-owner counts
  std::vector<std::int32_t> all_counts(static_cast<std::size_t>(P) * P);
  _comm->allgather(ghost_count.data(), all_counts.data(),
                   sizeof(std::int32_t) * P);
  auto wants = [&](int from, int owner) { return all_counts[from * P + owner]; };

  // symmetric neighbour list in rank order (:390-412)
  for (int p = 0; p < P; ++p) {
    const std::int32_t c = ghost_count[p], rc = wants(p, _rank);
    if (c > 0 || rc > 0) {
      _neighbours.push_back(p);
      _send_count.push_back(c);
      _recv_count.push_back(rc);
    }
  }
  if (_neighbours.empty()) { // :421-425 keeps one zero entry
    _send_count = {0};
    _recv_count = {0};
  }
  _send_offset = {0};
  for (std::int32_t c : _send_count)
    _send_offset.push_back(_send_offset.back() + c);
  _recv_offset = {0};
  for (std::int32_t c : _recv_count)
    _recv_offset.push_back(_recv_offset.back() + c);
  _num_indices = _recv_offset.back();

  // index buffer: the owner-local indices each neighbour asks me for, in the
  // neighbour's ghost order (Neighbor_alltoallv, :444-447)
  _indexbuf_host.assign(_num_indices, 0);
  if (P > 1) {
    std::vector<std::vector<std::int32_t>> lists = _comm->allgatherv(ghost_local);
    for (std::size_t i = 0; i < _neighbours.size(); ++i) {
      const int p = _neighbours[i];
      std::int64_t off = 0; // p's ghosts are sorted, so grouped by owner
      for (int q = 0; q < _rank; ++q)
        off += wants(p, q);
      std::copy_n(lists[p].begin() + off, _recv_count[i],
                  _indexbuf_host.begin() + _recv_offset[i]);
    }
  }
  for (std::int32_t idx : _indexbuf_host)
    if (idx < 0 || idx >= local_size)
      throw std::runtime_error("L2GMap: neighbour requested a non-local index");

  // ghosts land after the owned entries (:460-461)
  for (std::int32_t& s : _send_offset)
    s += static_cast<std::int32_t>(local_size);

  // contiguous-run detection: one run per neighbour => no pack kernel
  _direct_send = !_neighbours.empty();
  _direct_offset.assign(_neighbours.size(), 0);
  for (std::size_t i = 0; i < _neighbours.size() && _direct_send; ++i) {
    const std::int32_t* seg = _indexbuf_host.data() + _recv_offset[i];
    for (std::int32_t k = 1; k < _recv_count[i]; ++k)
      if (seg[k] != seg[0] + k) {
        _direct_send = false;
        break;
      }
    _direct_offset[i] = _recv_count[i] > 0 ? seg[0] : 0;
  }

  // the argument lists of the grouped exchange, once: update() runs inside
  // every CG iteration and must not build vectors there
  {
    const auto nn = static_cast<std::ptrdiff_t>(_neighbours.size());
    _x_send_count.assign(_recv_count.begin(), _recv_count.begin() + nn);
    _x_send_offset = _direct_send
                         ? _direct_offset
                         : std::vector<std::int32_t>(_recv_offset.begin(),
                                                     _recv_offset.begin() + nn);
    _x_recv_count.assign(_send_count.begin(), _send_count.begin() + nn);
    _x_recv_offset.assign(_send_offset.begin(), _send_offset.begin() + nn);
    _x_packed_offset.assign(_recv_offset.begin(), _recv_offset.begin() + nn);
  }

  if (_num_indices > 0) { // :473-478
    _indexbuf = _exec->alloc<std::int32_t>(_num_indices);
    _exec->copy_from<std::int32_t>(_indexbuf, _exec->get_host(),
                                   _indexbuf_host.data(), _num_indices);
  }
  if (_hip && !_neighbours.empty()) {
    // high priority: the (small) RCCL kernel must be placed before the local
    // SpMV, whose persistent workgroups would otherwise fill every CU first
    _comm_stream = _hip->create_stream(/*high_priority=*/true);
    _ev_ready = _hip->create_event();
    _ev_done = _hip->create_event();
  }
  if (_hip && P > 1
      && (_cm == CommunicationModel::onesided_put_active
          || _cm == CommunicationModel::onesided_put_passive))
    setup_put(local_size);
}

// The one-sided models: every rank creates its window, the ranks exchange
// {process, address, IPC handle, neighbour list, where each neighbour's data
// goes in the ghost tail}, and each connects its neighbours.  Collective, like
// the rest of the plan; the put path is used only if EVERY rank could set it
// up (a rank that fell back alone would wait for stores that never come).
void L2GMap::setup_put(std::int64_t local_size)
{
  struct Record {
    std::int64_t pid = 0;
    std::uint64_t raw = 0;
    unsigned char handle[SPMV_HIP_IPC_HANDLE_BYTES] = {};
    std::int64_t stage_bytes = 0;
    std::int32_t ok = 0, nn = 0, fine = 0;
    std::int32_t nbr[SPMV_HIP_PUT_MAX_PEERS] = {};
    std::int32_t ghost_off[SPMV_HIP_PUT_MAX_PEERS] = {};
  };
  Record mine;
  const std::size_t nn = _neighbours.size();
  spmv_hip_put* put = nullptr;
  mine.stage_bytes = 8 * static_cast<std::int64_t>(_ghosts.size() > 0 ? _ghosts.size() : 1);
  // not beside the peer reduction of the CG scalars where ranks share a
  // process (comm.h): such a communicator keeps the two-sided exchange -- on
  // every rank, `ok` travels
  const bool refused = _comm->peer_reduce() && _comm->ranks_share_a_process()
                       && !Comm::pair_allowed();
  if (!refused && nn <= SPMV_HIP_PUT_MAX_PEERS
      && spmv_hip_put_create(_hip->context(), (size_t)mine.stage_bytes, &put,
                             mine.handle, &mine.raw, &mine.pid)
             == SPMV_HIP_OK) {
    mine.ok = 1;
    int fine = 0;
    (void)spmv_hip_put_fine_grained(put, &fine);
    mine.fine = fine;
    mine.nn = static_cast<std::int32_t>(nn);
    for (std::size_t i = 0; i < nn; ++i) {
      mine.nbr[i] = _neighbours[i];
      mine.ghost_off[i]
          = _x_recv_offset[i] - static_cast<std::int32_t>(local_size);
    }
  }
  const int P = _comm->size();
  std::vector<Record> all(P);
  _comm->allgather(&mine, all.data(), sizeof(Record));
  bool everybody = true;
  for (const Record& r : all)
    everybody = everybody && r.ok;
  int rc = everybody ? SPMV_HIP_OK : SPMV_HIP_ENOTSUP;
  for (std::size_t i = 0; i < nn && rc == SPMV_HIP_OK; ++i) {
    const Record& peer = all[_neighbours[i]];
    int slot = -1;
    for (int k = 0; k < peer.nn; ++k)
      if (peer.nbr[k] == _rank)
        slot = k;
    if (slot < 0) { // the neighbour relation is symmetric (:390-412)
      rc = SPMV_HIP_EINVAL;
      break;
    }
    rc = spmv_hip_put_connect(put, static_cast<int>(i), peer.handle, peer.raw,
                              peer.pid, (size_t)peer.stage_bytes,
                              peer.ghost_off[slot], slot, _x_send_offset[i],
                              _x_send_count[i],
                              _x_recv_offset[i]
                                  - static_cast<std::int32_t>(local_size),
                              _x_recv_count[i], peer.fine);
    if (rc == SPMV_HIP_OK) // (names in a timed-out wait's diagnosis)
      rc = spmv_hip_put_label(put, _rank, static_cast<int>(i), _neighbours[i]);
  }
  if (rc == SPMV_HIP_OK && nn > 0)
    rc = spmv_hip_put_finish(put);
  // agree on the outcome: one rank that could not connect sends everybody
  // back to the two-sided exchange
  const std::int32_t my_ok = rc == SPMV_HIP_OK ? 1 : 0;
  std::vector<std::int32_t> oks = _comm->allgather_value<std::int32_t>(my_ok);
  bool all_ok = true;
  for (std::int32_t o : oks)
    all_ok = all_ok && o;
  // what every rank knows to be the same everywhere: the destructor's barrier
  // is keyed on it, not on this rank's own window (a rank without neighbours
  // has none, yet must take part)
  _put_agreed = all_ok;
  if (all_ok)
    _comm->note_onesided_map(+1);
  if (all_ok && nn > 0) {
    _put = put;
  } else {
    spmv_hip_put_destroy(put);
    _put = nullptr;
  }
}

L2GMap::~L2GMap()
{
  if (_put_agreed) {
    // a neighbour may still be storing into this rank's window (its side of
    // the last exchange): every rank arrives here before anybody frees --
    // EVERY rank of the communicator, also one that has no window of its own
    try {
      if (_comm_stream && _put)
        _hip->synchronize_stream(_comm_stream);
    } catch (...) { // (a timed-out exchange: the barrier is still owed)
    }
    try {
      (void)_comm->allgather_value<std::int32_t>(0);
    } catch (...) {
    }
    if (_put)
      spmv_hip_put_destroy(_put);
    _put = nullptr;
    _comm->note_onesided_map(-1);
  }
  try {
    if (_hip) {
      if (_comm_stream)
        _hip->synchronize_stream(_comm_stream);
      _hip->destroy_event(_ev_ready);
      _hip->destroy_event(_ev_done);
      _hip->destroy_stream(_comm_stream);
    }
    _exec->free(_send_buf);
    _exec->free(_indexbuf); // the reference leaks this
  } catch (...) {
  }
}

std::int32_t L2GMap::local_size() const
{
  return static_cast<std::int32_t>(_ranges[_rank + 1] - _ranges[_rank]);
}

std::int32_t L2GMap::global_to_local(std::int64_t i) const // :961-973
{
  const std::int64_t r0 = _ranges[_rank], r1 = _ranges[_rank + 1];
  if (i >= r0 && i < r1)
    return static_cast<std::int32_t>(i - r0);
  auto it = _global_to_local.find(i);
  if (it == _global_to_local.end())
    throw std::runtime_error("L2GMap::global_to_local: index is not a ghost");
  return it->second;
}

bool L2GMap::overlapping() const // :975-981
{
  return _cm == CommunicationModel::p2p_nonblocking
         || _cm == CommunicationModel::collective_nonblocking;
}

template <typename T>
void L2GMap::start_exchange(T* vec) const
{
  if (!_hip)
    throw std::runtime_error(
        "L2GMap::update: the halo exchange needs a HipExecutor");
  void* compute = _hip->get_stream();
  // the exchange may only start once the kernels producing vec are done
  _hip->record_event(_ev_ready, compute);
  _hip->stream_wait_event(_comm_stream, _ev_ready);

  const void* send_base = vec;
  if (!_direct_send) {
    const size_t need = sizeof(T) * static_cast<size_t>(_num_indices > 0 ? _num_indices : 1);
    if (_send_buf == nullptr || _send_buf_bytes < need) { // :607-614
      _exec->free(_send_buf);
      _send_buf = _exec->alloc<char>(need);
      _send_buf_bytes = need;
    }
    // pack on the comm stream through the executor interface (:618)
    _hip->set_stream(_comm_stream);
    try {
      _exec->gather_ghosts_run(_num_indices, _indexbuf, vec,
                               static_cast<T*>(_send_buf));
    } catch (...) {
      _hip->set_stream(compute);
      throw;
    }
    _hip->set_stream(compute);
    send_base = _send_buf;
  }
  // receive straight into the ghost tail (:624-628), send packed or direct
  // data (:630-634); one grouped call, ordered on the comm stream; the
  // argument lists were built with the plan
  if (_put) // one-sided models: peer stores, one launch
    throw_on_error(spmv_hip_put_exchange(_hip->context(), _put, sizeof(T),
                                         send_base, vec + local_size(),
                                         _comm_stream),
                   "spmv_hip_put_exchange");
  else
    _comm->neighbor_exchange(sizeof(T), _neighbours, send_base, _x_send_count,
                             _x_send_offset, vec, _x_recv_count, _x_recv_offset,
                             _comm_stream);
  _hip->record_event(_ev_done, _comm_stream);
}

template <typename T>
void L2GMap::update(T* vec) const // :868-896
{
  if (_neighbours.empty())
    return;
  start_exchange(vec);
  if (!overlapping()) // blocking models: later compute-stream work sees ghosts
    _hip->stream_wait_event(_hip->get_stream(), _ev_done);
}

template <typename T>
void L2GMap::update_finalise(T*) const // :899-905
{
  if (_neighbours.empty() || !overlapping())
    return;
  _hip->stream_wait_event(_hip->get_stream(), _ev_done); // no host wait
}

// Ghost tail -> owners, accumulating (L2GMap.cpp:907-959).  The transfer is the
// forward exchange with the roles swapped: each neighbour's slice of the ghost
// tail is sent as it lies (contiguous, no pack), the owner receives into the
// staging buffer in index-buffer order and adds segment by segment, neighbour
// by neighbour -- the ascending-i order of the reference's loop (:921,:947),
// so a value wanted by two neighbours is accumulated in the same order.
// The reference only acts for the two blocking models (:953-959) and silently
// does nothing for the rest; here every model performs the (blocking) update.
template <typename T>
void L2GMap::reverse_update(T* vec) const
{
  if (_neighbours.empty())
    return;
  if (!_hip)
    throw std::runtime_error(
        "L2GMap::reverse_update: the halo exchange needs a HipExecutor");
  void* compute = _hip->get_stream();
  const size_t need
      = sizeof(T) * static_cast<size_t>(_num_indices > 0 ? _num_indices : 1);
  if (_send_buf == nullptr || _send_buf_bytes < need) {
    _hip->synchronize_stream(_comm_stream); // nobody still reads the old one
    _exec->free(_send_buf);
    _send_buf = _exec->alloc<char>(need);
    _send_buf_bytes = need;
  }
  _hip->record_event(_ev_ready, compute);
  _hip->stream_wait_event(_comm_stream, _ev_ready);
  const size_t nn = _neighbours.size();
  // the forward lists with the roles swapped
  _comm->neighbor_exchange(sizeof(T), _neighbours, vec, _x_recv_count,
                           _x_recv_offset, _send_buf, _x_send_count,
                           _x_packed_offset, _comm_stream);
  _hip->record_event(_ev_done, _comm_stream);
  _hip->stream_wait_event(compute, _ev_done);
  const T* staged = static_cast<const T*>(_send_buf);
  for (size_t i = 0; i < nn; ++i)
    _hip->scatter_add_run(_recv_count[i], _indexbuf + _recv_offset[i],
                          staged + _recv_offset[i], vec);
}

template void L2GMap::update<float>(float*) const;
template void L2GMap::update<double>(double*) const;
template void L2GMap::update_finalise<float>(float*) const;
template void L2GMap::update_finalise<double>(double*) const;
template void L2GMap::reverse_update<float>(float*) const;
template void L2GMap::reverse_update<double>(double*) const;

} // namespace spmv
