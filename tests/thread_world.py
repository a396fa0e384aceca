"""P ranks as P Python threads of ONE process, for host-logic tests with more
ranks than the gloo tests spawn.  Each rank owns a spmv::CallbackComm whose
allgather meets the other ranks at a barrier (ctypes drops the GIL around the
C++ call and the callback re-takes it, so the ranks really interleave)."""
import ctypes as C
import queue
import threading

import numpy as np

from spmv_amd import host


class ThreadWorld:
    def __init__(self, nranks, timeout=60.0):
        self.P = nranks
        self.timeout = timeout
        self.bar = threading.Barrier(nranks, timeout=timeout)
        self.slots = [b""] * nranks
        # device transport: one mailbox per ordered pair of ranks
        self.mail = {(s, d): queue.Queue() for s in range(nranks)
                     for d in range(nranks)}
        self.acks = {(s, d): queue.Queue() for s in range(nranks)
                     for d in range(nranks)}
        self.red = [None] * nranks

    # -- device transport (all ranks are threads of this process and share the
    # GPU, so a neighbour's send pointer is directly usable: the halo is a
    # device-to-device copy issued by the RECEIVER on its own stream) --------
    def device_transport(self, rank, ctx_handle):
        from spmv_amd import _lib

        def sync(stream):
            _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)

        def exchange(user, elem, nn, nbrs, send_buf, scnt, soff, recv_base,
                     rcnt, roff, stream):
            try:
                sync(stream)  # my outgoing data is complete
                for i in range(nn):
                    if scnt[i] > 0:
                        self.mail[(rank, nbrs[i])].put(
                            ((send_buf or 0) + soff[i] * elem, scnt[i] * elem))
                for i in range(nn):
                    if rcnt[i] > 0:
                        src, nbytes = self.mail[(nbrs[i], rank)].get(
                            timeout=self.timeout)
                        assert nbytes == rcnt[i] * elem, (nbytes, rcnt[i], elem)
                        _lib.call("spmv_hip_copy_d2d_async", ctx_handle,
                                  (recv_base or 0) + roff[i] * elem, src, nbytes,
                                  stream)
                sync(stream)  # copies landed: the senders may reuse their buffers
                for i in range(nn):
                    if rcnt[i] > 0:
                        self.acks[(rank, nbrs[i])].put(True)
                for i in range(nn):
                    if scnt[i] > 0:
                        self.acks[(nbrs[i], rank)].get(timeout=self.timeout)
                return 0
            except Exception as e:
                print("thread exchange failed:", repr(e), flush=True)
                return 1

        def allreduce(user, dev, count, stream):
            try:
                buf = np.empty(count, np.float64)
                _lib.call("spmv_hip_copy_d2h_async", ctx_handle,
                          buf.ctypes.data_as(C.c_void_p), dev, count * 8, stream)
                sync(stream)
                self.red[rank] = buf
                self.bar.wait()
                total = np.zeros(count)
                for r in range(self.P):  # rank order, as oracle.dist_cg pins it
                    total += self.red[r]
                self.bar.wait()
                _lib.call("spmv_hip_copy_h2d_async", ctx_handle, dev,
                          total.ctypes.data_as(C.c_void_p), count * 8, stream)
                sync(stream)
                return 0
            except Exception as e:
                print("thread allreduce failed:", repr(e), flush=True)
                return 1

        return exchange, allreduce

    def _allgather(self, rank):
        def allgather(user, send, recv, nbytes):
            try:
                self.slots[rank] = C.string_at(send, nbytes)
                self.bar.wait()
                data = b"".join(self.slots)
                assert len(data) == nbytes * self.P
                C.memmove(recv, data, len(data))
                self.bar.wait()  # nobody overwrites a slot still being read
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("thread allgather failed:", repr(e), flush=True)
                return 1
        return allgather

    def gather(self, rank, arr):
        """All ranks' arrays concatenated in rank order (collective)."""
        self.red[rank] = np.ascontiguousarray(arr)
        self.bar.wait()
        out = np.concatenate(self.red)
        self.bar.wait()
        return out

    def run(self, fn, gpu=False):
        """fn(rank, comm) on every rank -- or fn(rank, comm, exec_) with
        gpu=True, every rank holding its own HipExecutor on device 0;
        re-raises the first failure."""
        errors = [None] * self.P

        def body(rank):
            comm = exec_ = None
            try:
                if gpu:
                    exec_ = host.HipExecutor(0)
                    ex, ar = self.device_transport(rank, exec_.context)
                    comm = host.Comm.callback(rank, self.P,
                                              self._allgather(rank), ex, ar)
                    fn(rank, comm, exec_)
                else:
                    comm = host.Comm.callback(rank, self.P,
                                              self._allgather(rank))
                    fn(rank, comm)
            except BaseException as e:  # noqa: BLE001 (reported below)
                errors[rank] = e
                self.bar.abort()
            finally:
                if comm is not None:
                    comm.close()
                if exec_ is not None:
                    exec_.close()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        # a rank that fails breaks the barrier for the others: report the
        # root cause, not the follow-up "allgather callback failed" errors
        failed = [e for e in errors if e is not None]
        root = [e for e in failed
                if not isinstance(e, threading.BrokenBarrierError)
                and "callback failed" not in str(e)]
        if root:
            raise root[0]
        if failed:
            raise failed[0]
