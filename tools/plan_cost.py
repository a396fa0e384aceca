"""Plan creation cost at 512^3, repeated in one process (first creation pays
for fresh device memory, later ones find it in the runtime's pool).

    python tools/plan_cost.py [--n 512] [--reps 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--skew", type=int, default=0,
                    help="poisson_skew_ppm: a matrix that is NOT symmetric")
    args = ap.parse_args()
    exec_ = host.HipExecutor(0)
    comm = host.Comm.self_comm()
    if args.skew:
        from spmv_amd import _lib
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"poisson_skew_ppm",
                  args.skew)
    for symmetric in ((False,) if args.skew else (False, True)):
        for rep in range(args.reps):
            exec_.synchronize()
            t0 = time.perf_counter()
            A = host.Matrix.create_poisson3d(comm, exec_, args.n, symmetric,
                                             host.P2P_NONBLOCKING)
            exec_.synchronize()
            t1 = time.perf_counter()
            print(dict(symmetric=symmetric, rep=rep,
                       create_ms=round((t1 - t0) * 1e3, 1),
                       plan_ms=A.plan_get("plan_us") / 1e3,
                       plan_mib=A.plan_get("plan_kib") // 1024), flush=True)
            A.close()
    comm.close()
    exec_.close()


if __name__ == "__main__":
    main()
