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
        self.events = {}      # rank -> [(ctx, event)] of the async transport
        self.scratch = []     # its filler buffers
        self.delay_launches = 40

    # -- device transport (all ranks are threads of this process and share the
    # GPU, so a neighbour's send pointer is directly usable: the halo is a
    # device-to-device copy issued by the RECEIVER on its own stream) --------
    def device_transport(self, rank, ctx_handle, asynchronous=False):
        from spmv_amd import _lib

        def sync(stream):
            _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)

        if asynchronous:
            return (self._async_exchange(rank, ctx_handle),
                    self._allreduce(rank, ctx_handle))

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

        return exchange, self._allreduce(rank, ctx_handle)

    def _allreduce(self, rank, ctx_handle):
        from spmv_amd import _lib

        def sync(stream):
            _lib.call("spmv_hip_stream_synchronize", ctx_handle, stream)

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

        return allreduce

    # -- ASYNCHRONOUS device transport: like RCCL, the exchange is only
    # ENQUEUED on the given stream and the host returns at once; the data moves
    # late (a few ms of filler kernels sit in front of the copy).  Ordering
    # between ranks is carried by events alone: the receiver's stream waits for
    # the sender's "data ready" event, the sender's stream waits for the
    # receiver's "copied" event before anything later may overwrite the send
    # buffer.  Under this transport a missing stream_wait_event in L2GMap /
    # Matrix / cg shows up as stale (NaN-poisoned) ghosts or a clobbered send
    # buffer -- the blocking transports above drain the stream and hide it.
    def _async_exchange(self, rank, ctx_handle):
        from spmv_amd import _lib
        events = self.events.setdefault(rank, [])
        scratch = {}

        def new_event():
            ev = C.c_void_p()
            _lib.call("spmv_hip_event_create", ctx_handle, 0, C.byref(ev))
            events.append((ctx_handle, ev))
            return ev

        def delay(stream):
            if "buf" not in scratch:
                p = C.c_void_p()
                _lib.call("spmv_hip_alloc", ctx_handle, 8 << 22, C.byref(p))
                scratch["buf"] = p
                self.scratch.append((ctx_handle, p))
            for _ in range(self.delay_launches):
                _lib.call("spmv_hip_fill_const_f64", ctx_handle, 1 << 22, 1.0,
                          scratch["buf"], stream)

        def exchange(user, elem, nn, nbrs, send_buf, scnt, soff, recv_base,
                     rcnt, roff, stream):
            try:
                ready = new_event()
                _lib.call("spmv_hip_event_record", ctx_handle, ready, stream)
                for i in range(nn):
                    if scnt[i] > 0:
                        self.mail[(rank, nbrs[i])].put(
                            ((send_buf or 0) + soff[i] * elem, scnt[i] * elem,
                             ready))
                for i in range(nn):
                    if rcnt[i] > 0:
                        src, nbytes, ev = self.mail[(nbrs[i], rank)].get(
                            timeout=self.timeout)
                        assert nbytes == rcnt[i] * elem, (nbytes, rcnt[i], elem)
                        _lib.call("spmv_hip_stream_wait_event", ctx_handle,
                                  stream, ev)
                        delay(stream)
                        _lib.call("spmv_hip_copy_d2d_async", ctx_handle,
                                  (recv_base or 0) + roff[i] * elem, src, nbytes,
                                  stream)
                        done = new_event()
                        _lib.call("spmv_hip_event_record", ctx_handle, done,
                                  stream)
                        self.acks[(rank, nbrs[i])].put(done)
                for i in range(nn):
                    if scnt[i] > 0:
                        done = self.acks[(nbrs[i], rank)].get(
                            timeout=self.timeout)
                        _lib.call("spmv_hip_stream_wait_event", ctx_handle,
                                  stream, done)
                return 0  # nothing was waited for on the host
            except Exception as e:
                print("async exchange failed:", repr(e), flush=True)
                return 1

        return exchange

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

    def run(self, fn, gpu=False, asynchronous=False):
        """fn(rank, comm) on every rank -- or fn(rank, comm, exec_) with
        gpu=True, every rank holding its own HipExecutor on device 0;
        re-raises the first failure."""
        errors = [None] * self.P

        def body(rank):
            comm = exec_ = None
            try:
                if gpu:
                    exec_ = host.HipExecutor(0)
                    ex, ar = self.device_transport(rank, exec_.context,
                                                   asynchronous)
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
                    from spmv_amd import _lib
                    try:
                        exec_.synchronize()
                        self.bar.wait()  # nobody still waits on my events
                    except Exception:  # noqa: BLE001 (a failed rank broke it)
                        pass
                    for c, ev in self.events.pop(rank, []):
                        _lib.call("spmv_hip_event_destroy", c, ev)
                    for c, ptr in [s for s in self.scratch
                                   if s[0] is exec_.context]:
                        _lib.call("spmv_hip_free", c, ptr)
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
