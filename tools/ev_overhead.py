"""What the per-iteration HIP event records of CgOptions::time_spmv cost: the
same CG solve with and without them, alternating, in one process."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import _lib, host
exec_ = host.HipExecutor(0)
comm = host.Comm.self_comm()
n = 512; N = n**3
A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_NONBLOCKING)
d_b, d_x = exec_.alloc(N), exec_.alloc(N)
_lib.call("spmv_hip_fill_gaussian_f64", exec_.context, N, 0, N, d_b, None)
ws = host.CgWorkspace(exec_)
ws.reserve_timing(200)
host.cg_ex(comm, exec_, A, d_b, d_x, 5, 0.0, ws)
for K in (20, 100):
    for timed in (False, True, False, True):
        exec_.synchronize(); t0 = time.perf_counter()
        k, _, ms, nl = host.cg_ex(comm, exec_, A, d_b, d_x, K, 0.0, ws, time_spmv=timed)
        exec_.synchronize(); el = time.perf_counter() - t0
        print(K, "events" if timed else "no events", round(el / K * 1e3, 4), "ms/step", round(ms / max(nl, 1), 4))
