// Communicators: see comm.h.
#include "comm.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <random>
#include <stdexcept>

#include <unistd.h>

#include "executor.h"
#include "spmv_hip.h"

namespace spmv
{

std::vector<std::vector<int32_t>>
Comm::allgatherv(const std::vector<int32_t>& mine) const
{
  const int P = size();
  const int64_t my_len = static_cast<int64_t>(mine.size());
  std::vector<int64_t> lens = allgather_value<int64_t>(my_len);
  const int64_t max_len = *std::max_element(lens.begin(), lens.end());
  std::vector<std::vector<int32_t>> out(P);
  if (max_len == 0)
    return out;
  // pad to the longest contribution; setup-only traffic
  std::vector<int32_t> send(max_len, 0), recv(max_len * P);
  std::copy(mine.begin(), mine.end(), send.begin());
  allgather(send.data(), recv.data(), sizeof(int32_t) * max_len);
  for (int r = 0; r < P; ++r)
    out[r].assign(recv.begin() + r * max_len,
                  recv.begin() + r * max_len + lens[r]);
  return out;
}

std::vector<std::vector<unsigned char>>
Comm::allgatherv_bytes(const void* mine, size_t num_bytes) const
{
  const int P = size();
  std::vector<int64_t> lens
      = allgather_value<int64_t>(static_cast<int64_t>(num_bytes));
  const int64_t max_len = *std::max_element(lens.begin(), lens.end());
  std::vector<std::vector<unsigned char>> out(P);
  if (max_len == 0)
    return out;
  std::vector<unsigned char> send(max_len, 0), recv(max_len * P);
  if (num_bytes)
    std::memcpy(send.data(), mine, num_bytes);
  allgather(send.data(), recv.data(), static_cast<size_t>(max_len));
  for (int r = 0; r < P; ++r)
    out[r].assign(recv.begin() + r * max_len,
                  recv.begin() + r * max_len + lens[r]);
  return out;
}

// ---- deterministic peer reduction ------------------------------------------------
namespace
{
struct ReduceCard { // what a rank tells the others about its window
  unsigned char handle[SPMV_HIP_IPC_HANDLE_BYTES];
  uint64_t address;
  int64_t pid;
  int32_t fine, ok;
};
} // namespace

bool Comm::pair_allowed()
{
  const char* e = std::getenv("SPMV_ALLOW_PUT_WITH_PEER_REDUCE");
  return e && e[0] == '1';
}

namespace
{
// What identifies THIS process among the ranks of a communicator: not its pid
// alone -- ranks in separate pid namespaces (one container per GPU: every rank
// is pid 1 or 7) or on different hosts can share one -- but the pid together
// with a 64-bit token drawn once per process.
struct ProcessToken {
  int64_t pid;
  uint64_t nonce;
};
ProcessToken process_token()
{
  static const uint64_t nonce = [] {
    std::random_device rd;
    uint64_t v = (static_cast<uint64_t>(rd()) << 32) ^ rd();
    v ^= static_cast<uint64_t>(
        std::chrono::steady_clock::now().time_since_epoch().count());
    return v ^ reinterpret_cast<uintptr_t>(&v);
  }();
  return ProcessToken{static_cast<int64_t>(getpid()), nonce};
}
} // namespace

bool Comm::ranks_share_a_process() const
{
  if (_shared_process < 0) {
    const std::vector<ProcessToken> ids
        = allgather_value<ProcessToken>(process_token());
    int shared = 0;
    for (size_t a = 0; a < ids.size(); ++a)
      for (size_t b = a + 1; b < ids.size(); ++b)
        shared = shared
                 || (ids[a].pid == ids[b].pid && ids[a].nonce == ids[b].nonce);
    _shared_process = shared;
  }
  return _shared_process != 0;
}

bool Comm::enable_peer_reduce(const HipExecutor& exec) const
{
  if (_reduce)
    return true;
  const int P = size(), me = rank();
  if (P == 1)
    return false; // nothing to reduce
  ReduceCard mine;
  std::memset(&mine, 0, sizeof(mine));
  spmv_hip_reduce* r = nullptr;
  int fine = 0;
  // not beside a one-sided halo where ranks share a process (comm.h); every
  // rank takes the same decision: the counts are collective by construction
  // and the process ids are exchanged
  const bool refused
      = _onesided_maps > 0 && ranks_share_a_process() && !pair_allowed();
  if (refused && me == 0) // (said once per attempt: the fast path is off, the
    // results are not affected -- ADVICE r05)
    std::fprintf(stderr, "spmv: peer reduction not enabled: a one-sided halo lives on "
                         "this communicator and two of its ranks share a process "
                         "(comm.h); cg() keeps the transport's all-reduce\n");
  int rc = (P <= SPMV_HIP_REDUCE_MAX_RANKS && !refused)
               ? spmv_hip_reduce_create(exec.context(), P, me, &r, mine.handle,
                                        &mine.address, &mine.pid, &fine)
               : SPMV_HIP_ENOTSUP;
  mine.fine = fine;
  mine.ok = rc == SPMV_HIP_OK ? 1 : 0;
  // (every rank takes part in both collectives whatever happened to it)
  std::vector<ReduceCard> cards = allgather_value<ReduceCard>(mine);
  int32_t good = mine.ok;
  for (int p = 0; p < P && good; ++p)
    good = cards[p].ok;
  for (int p = 0; p < P && good; ++p) {
    if (p == me)
      continue;
    rc = spmv_hip_reduce_connect(r, p, cards[p].handle, cards[p].address,
                                 cards[p].pid, cards[p].fine);
    if (rc != SPMV_HIP_OK)
      good = 0;
  }
  std::vector<int32_t> all = allgather_value<int32_t>(good);
  for (int p = 0; p < P; ++p)
    good = good && all[p];
  if (!good) {
    if (r)
      spmv_hip_reduce_destroy(r);
    return false;
  }
  _reduce = r;
  _reduce_ctx = exec.context();
  _reduce_exec = &exec;
  _reduce_stream_set = false;
  exec.attach_reduce_owner(this); // ~HipExecutor closes it if we are still here
  return true;
}

void Comm::close_peer_reduce() const
{
  if (!_reduce)
    return;
  // nobody still stores into a window that is about to go: every rank's last
  // reduction has completed when its kernel has (stream order), and a rank
  // arrives here after synchronising -- the allgather is the barrier
  // (the local teardown happens whether or not the barrier throws -- a callback
  // transport whose peer is already gone: a second close, from the
  // communicator's destructor after the executor's, must find nothing left
  // to synchronise or detach)
  struct Teardown {
    const Comm* c;
    ~Teardown()
    {
      spmv_hip_reduce_destroy(c->_reduce);
      c->_reduce = nullptr;
      c->_reduce_ctx = nullptr;
      if (c->_reduce_exec)
        c->_reduce_exec->detach_reduce_owner(c);
      c->_reduce_exec = nullptr;
    }
  } teardown{this};
  spmv_hip_synchronize(_reduce_ctx);
  int32_t token = 1;
  (void)allgather_value<int32_t>(token);
}

void Comm::reduce_sum(double* device_inout, size_t count, void* stream) const
{
  if (_reduce && count <= SPMV_HIP_REDUCE_MAX_COUNT) {
    // the double-buffered slots rely on stream order (epoch k+1's store into
    // a slot follows epoch k-1's read of it): when the caller moves to another
    // stream, the previous one is drained first (rare: cg() stays on one)
    void* resolved = stream ? stream : _reduce_exec->get_stream();
    if (_reduce_stream_set && _reduce_stream != resolved)
      _reduce_exec->synchronize_stream(_reduce_stream);
    _reduce_stream = resolved;
    _reduce_stream_set = true;
    throw_on_error(spmv_hip_reduce_sum_f64(_reduce_ctx, _reduce, device_inout,
                                           static_cast<int>(count), stream),
                   "spmv_hip_reduce_sum_f64");
    return;
  }
  allreduce_sum(device_inout, count, stream);
}

// ---- SelfComm ----------------------------------------------------------------
void SelfComm::allgather(const void* send, void* recv, size_t bytes) const
{
  if (bytes)
    std::memcpy(recv, send, bytes);
}

void SelfComm::neighbor_exchange(size_t, const std::vector<int>& neighbours,
                                 const void*, const std::vector<int32_t>&,
                                 const std::vector<int32_t>&, void*,
                                 const std::vector<int32_t>&,
                                 const std::vector<int32_t>&, void*) const
{
  if (!neighbours.empty())
    throw std::runtime_error("SelfComm: a single rank has no neighbours");
}

// ---- RcclComm ----------------------------------------------------------------
std::vector<unsigned char> RcclComm::unique_id()
{
  std::vector<unsigned char> id(SPMV_HIP_UNIQUE_ID_BYTES);
  throw_on_error(spmv_hip_comm_unique_id(id.data()), "spmv_hip_comm_unique_id");
  return id;
}

RcclComm::RcclComm(const HipExecutor& exec, int nranks, int rank,
                   const void* unique_id)
    : _rank(rank), _size(nranks)
{
  throw_on_error(spmv_hip_comm_create(exec.context(), nranks, rank, unique_id,
                                      &_comm),
                 "spmv_hip_comm_create");
}

RcclComm::~RcclComm()
{
  try { // (collective, like the destruction of the communicator itself)
    close_peer_reduce();
  } catch (...) {
  }
  spmv_hip_comm_destroy(_comm);
}

void RcclComm::allgather(const void* send, void* recv, size_t bytes) const
{
  throw_on_error(spmv_hip_comm_allgather_host(_comm, send, recv, bytes),
                 "spmv_hip_comm_allgather_host");
}

void RcclComm::neighbor_exchange(size_t elem_bytes,
                                 const std::vector<int>& neighbours,
                                 const void* send_buf,
                                 const std::vector<int32_t>& send_counts,
                                 const std::vector<int32_t>& send_offsets,
                                 void* recv_base,
                                 const std::vector<int32_t>& recv_counts,
                                 const std::vector<int32_t>& recv_offsets,
                                 void* stream) const
{
  const int n = static_cast<int>(neighbours.size());
  std::vector<int32_t> nb(neighbours.begin(), neighbours.end());
  int rc;
  if (elem_bytes == sizeof(double))
    rc = spmv_hip_comm_neighbor_exchange_f64(
        _comm, n, nb.data(), static_cast<const double*>(send_buf),
        send_counts.data(), send_offsets.data(),
        static_cast<double*>(recv_base), recv_counts.data(),
        recv_offsets.data(), stream);
  else if (elem_bytes == sizeof(float))
    rc = spmv_hip_comm_neighbor_exchange_f32(
        _comm, n, nb.data(), static_cast<const float*>(send_buf),
        send_counts.data(), send_offsets.data(), static_cast<float*>(recv_base),
        recv_counts.data(), recv_offsets.data(), stream);
  else
    throw std::runtime_error("RcclComm: unsupported element size");
  throw_on_error(rc, "spmv_hip_comm_neighbor_exchange");
}

RcclComm::Info RcclComm::info() const
{
  Info out;
  char path[512];
  throw_on_error(spmv_hip_comm_info(_comm, &out.nranks, &out.rank,
                                    &out.rccl_version,
                                    &out.separate_reduction_comm, path,
                                    (int)sizeof(path)),
                 "spmv_hip_comm_info");
  out.lib_path = path;
  return out;
}

void RcclComm::allreduce_sum(double* device_inout, size_t count,
                             void* stream) const
{
  throw_on_error(spmv_hip_comm_allreduce_sum_f64(_comm, device_inout, count,
                                                 stream),
                 "spmv_hip_comm_allreduce_sum_f64");
}

// ---- CallbackComm ------------------------------------------------------------
CallbackComm::~CallbackComm()
{
  try { // (collective: every rank destroys its communicator)
    close_peer_reduce();
  } catch (...) {
  }
}

void CallbackComm::allgather(const void* send, void* recv, size_t bytes) const
{
  if (!_cb.allgather || _cb.allgather(_cb.user, send, recv, bytes) != 0)
    throw std::runtime_error("CallbackComm: allgather callback failed");
}

void CallbackComm::neighbor_exchange(size_t elem_bytes,
                                     const std::vector<int>& neighbours,
                                     const void* send_buf,
                                     const std::vector<int32_t>& send_counts,
                                     const std::vector<int32_t>& send_offsets,
                                     void* recv_base,
                                     const std::vector<int32_t>& recv_counts,
                                     const std::vector<int32_t>& recv_offsets,
                                     void* stream) const
{
  if (!_cb.neighbor_exchange)
    throw std::runtime_error("CallbackComm: no device transport was supplied");
  if (_cb.neighbor_exchange(_cb.user, elem_bytes,
                            static_cast<int>(neighbours.size()),
                            neighbours.data(), send_buf, send_counts.data(),
                            send_offsets.data(), recv_base, recv_counts.data(),
                            recv_offsets.data(), stream)
      != 0)
    throw std::runtime_error("CallbackComm: neighbor_exchange callback failed");
}

void CallbackComm::allreduce_sum(double* device_inout, size_t count,
                                 void* stream) const
{
  if (_size == 1)
    return;
  if (!_cb.allreduce_sum)
    throw std::runtime_error("CallbackComm: no device transport was supplied");
  if (_cb.allreduce_sum(_cb.user, device_inout, count, stream) != 0)
    throw std::runtime_error("CallbackComm: allreduce_sum callback failed");
}

} // namespace spmv
