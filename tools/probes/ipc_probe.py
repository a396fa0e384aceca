"""Feasibility probe for a one-sided (peer-store) halo between PROCESSES: HIP
IPC memory handles + interprocess events on one device.  Process A owns a
buffer and an event; process B opens both, copies into A's buffer on its own
stream and records the event; A makes its stream wait for the event (no host
wait) and reads the data.  Prints what works.
"""
import ctypes as C
import multiprocessing as mp
import os
import sys


class MemH(C.Structure):
    _fields_ = [("reserved", C.c_ubyte * 64)]


def hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), MemH, C.c_uint]
    lib.hipIpcOpenEventHandle.argtypes = [C.POINTER(C.c_void_p), MemH]
    lib.hipIpcGetMemHandle.argtypes = [C.POINTER(MemH), C.c_void_p]
    lib.hipIpcGetEventHandle.argtypes = [C.POINTER(MemH), C.c_void_p]
    return lib


def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> hipError {rc}")


def owner(q_to_peer, q_from_peer):
    h = hip()
    chk(h.hipSetDevice(0), "set")
    n = 1 << 20
    buf = C.c_void_p()
    chk(h.hipMalloc(C.byref(buf), 8 * n), "malloc")
    chk(h.hipMemset(buf, 0, 8 * n), "memset")
    mh = MemH()
    chk(h.hipIpcGetMemHandle(C.byref(mh), buf), "hipIpcGetMemHandle")
    ev = C.c_void_p()
    # hipEventDisableTiming | hipEventInterprocess
    chk(h.hipEventCreateWithFlags(C.byref(ev), 0x2 | 0x4), "event create ipc")
    eh = MemH()
    chk(h.hipIpcGetEventHandle(C.byref(eh), ev), "hipIpcGetEventHandle")
    q_to_peer.put((bytes(bytearray(mh.reserved)), bytes(bytearray(eh.reserved))))
    assert q_from_peer.get(timeout=30) == "recorded"
    st = C.c_void_p()
    chk(h.hipStreamCreate(C.byref(st)), "stream")
    chk(h.hipStreamWaitEvent(st, ev, 0), "hipStreamWaitEvent on own ipc event")
    out = (C.c_double * 4)()
    chk(h.hipMemcpyAsync(out, buf, 32, 2, st), "d2h")
    chk(h.hipStreamSynchronize(st), "sync")
    print("owner reads", list(out), flush=True)
    q_to_peer.put("done")
    assert list(out) == [1.5, 2.5, 3.5, 4.5]
    print("IPC PROBE OK", flush=True)


def peer(q_from_owner, q_to_owner):
    h = hip()
    chk(h.hipSetDevice(0), "set")
    mhb, ehb = q_from_owner.get(timeout=30)
    mh, eh = MemH(), MemH()
    C.memmove(C.byref(mh), mhb, 64)
    C.memmove(C.byref(eh), ehb, 64)
    rbuf = C.c_void_p()
    chk(h.hipIpcOpenMemHandle(C.byref(rbuf), mh, 1), "hipIpcOpenMemHandle")
    ev = C.c_void_p()
    chk(h.hipIpcOpenEventHandle(C.byref(ev), eh), "hipIpcOpenEventHandle")
    st = C.c_void_p()
    chk(h.hipStreamCreate(C.byref(st)), "stream")
    src = (C.c_double * 4)(1.5, 2.5, 3.5, 4.5)
    dsrc = C.c_void_p()
    chk(h.hipMalloc(C.byref(dsrc), 32), "malloc")
    chk(h.hipMemcpy(dsrc, src, 32, 1), "h2d")
    chk(h.hipMemcpyAsync(rbuf, dsrc, 32, 3, st), "d2d into the peer's buffer")
    chk(h.hipEventRecord(ev, st), "record opened ipc event")
    chk(h.hipStreamSynchronize(st), "sync")
    q_to_owner.put("recorded")
    q_from_owner.get(timeout=30)
    chk(h.hipIpcCloseMemHandle(rbuf), "close")


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.set_start_method("spawn")
    a, b = mp.Queue(), mp.Queue()
    po = mp.Process(target=owner, args=(a, b))
    pp = mp.Process(target=peer, args=(a, b))
    po.start(), pp.start()
    po.join(60), pp.join(60)
    for p in (po, pp):
        if p.is_alive():
            p.kill()
    print("exit codes", po.exitcode, pp.exitcode)
    sys.exit(0 if (po.exitcode == 0 and pp.exitcode == 0) else 1)
