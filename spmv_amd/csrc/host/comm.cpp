// Communicators: see comm.h.
#include "comm.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>

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

RcclComm::~RcclComm() { spmv_hip_comm_destroy(_comm); }

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
