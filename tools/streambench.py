"""Device streaming ceilings measured with the library's own BLAS-1 kernels:
read-only (dot of two vectors), write-only (fill), copy (d2d memcpy).  Puts the
SpMV GB/s figures in context: the 8 TB/s roofline is the HBM3E spec, these are
what a trivially coalesced kernel reaches on the same box in the same run.

    python tools/streambench.py --mib 256 1024 4096 --out gpurun_out/stream.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spmv_amd import hip  # noqa: E402
from spmv_amd.hip import call  # noqa: E402
from kbench import time_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, nargs="+", default=[256, 1024, 4096])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ctx = hip.Context(0)
    rows = []
    for mib in args.mib:
        n = mib * (1 << 20) // 8
        a, b = ctx.empty(n, np.float64), ctx.empty(n, np.float64)
        ctx.fill_const(n, 1.0, a.ptr)
        ctx.fill_const(n, 2.0, b.ptr)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        tests = {
            "read_dot": (lambda: call("spmv_hip_dot_partial_f64", ctx.h, n,
                                      a.ptr, b.ptr, part.ptr, None), 16 * n),
            "write_fill": (lambda: ctx.fill_const(n, 3.0, a.ptr), 8 * n),
            "copy_d2d": (lambda: ctx.copy(b.ptr, a.ptr, 8 * n), 16 * n),
        }
        for name, (fn, nbytes) in tests.items():
            best, med = time_ms(ctx, fn, args.reps)
            row = dict(test=name, mib_per_vector=mib, bytes=nbytes,
                       ms_best=round(best, 5), ms_median=round(med, 5),
                       gbs_best=round(nbytes / best / 1e6, 1),
                       gbs_median=round(nbytes / med / 1e6, 1))
            rows.append(row)
            print(json.dumps(row), flush=True)
        for buf in (a, b, part):
            buf.free()
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
