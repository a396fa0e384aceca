"""torch.distributed (gloo) transport for spmv::CallbackComm: the shape of the
callbacks an MPI application hands to Comm::callback,
written over torch.distributed.  Used by `bench.py --transport gloo` (the
rehearsal of the N-rank run on a 1-GPU box -- never a benchmark result) and by
the multi-process tests.  On a GPU box the device callbacks stage through host
memory, so several ranks can share ONE GPU; this exercises every line of the
C++ multi-rank logic (plan, pack / direct send, local+remote split, stream
events, CG reductions) without RCCL, which needs one GPU per rank.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist


def init_gloo():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world


def make_allgather(world):
    def allgather(user, send, recv, nbytes):
        try:
            mine = torch.frombuffer(bytearray(C.string_at(send, nbytes)),
                                    dtype=torch.uint8)
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, mine)
            for r, t in enumerate(outs):
                C.memmove(recv + r * nbytes, t.data_ptr(), nbytes)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allgather callback failed:", e, flush=True)
            return 1
    return allgather


def make_device_transport(ctx_handle):
    """neighbor_exchange / allreduce_sum on DEVICE pointers, staged through
    the host and gloo.  Blocking, but stream-correct: it first drains the
    stream it was asked to run on."""
    from . import _lib

    def d2h(ptr, nbytes, stream):
        buf = np.empty(nbytes, np.uint8)
        _lib.call("spmv_hip_copy_d2h_async", ctx_handle,
                  buf.ctypes.data_as(C.c_void_p), ptr, nbytes, stream)
        _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)
        return buf

    def h2d(ptr, buf, stream):
        _lib.call("spmv_hip_copy_h2d_async", ctx_handle, ptr,
                  buf.ctypes.data_as(C.c_void_p), buf.nbytes, stream)
        _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)

    def exchange(user, elem, nn, nbrs, send_buf, scnt, soff, recv_base, rcnt,
                 roff, stream):
        try:
            _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)
            reqs, recvs = [], []
            for i in range(nn):
                if rcnt[i] > 0:
                    t = torch.empty(rcnt[i] * elem, dtype=torch.uint8)
                    reqs.append(dist.irecv(t, src=nbrs[i]))
                    recvs.append((i, t))
            for i in range(nn):
                if scnt[i] > 0:
                    buf = d2h(send_buf + soff[i] * elem, scnt[i] * elem, stream)
                    reqs.append(dist.isend(torch.from_numpy(buf), dst=nbrs[i]))
            for r in reqs:
                r.wait()
            for i, t in recvs:
                h2d(recv_base + roff[i] * elem, t.numpy(), stream)
            return 0
        except Exception as e:
            print("exchange callback failed:", e, flush=True)
            return 1

    def allreduce(user, dev, count, stream):
        try:
            buf = d2h(dev, count * 8, stream).view(np.float64)
            # sum in rank order (what oracle.dist_cg pins)
            outs = [torch.empty(count, dtype=torch.float64)
                    for _ in range(dist.get_world_size())]
            dist.all_gather(outs, torch.from_numpy(buf.copy()))
            s = np.zeros(count)
            for t in outs:
                s += t.numpy()
            h2d(dev, s.view(np.uint8), stream)
            return 0
        except Exception as e:
            print("allreduce callback failed:", e, flush=True)
            return 1

    return exchange, allreduce


def gather_concat(arr):
    """All ranks' float64 arrays concatenated in rank order (on every rank)."""
    world = dist.get_world_size()
    outs = [None] * world
    dist.all_gather_object(outs, np.ascontiguousarray(arr))
    return np.concatenate(outs)
