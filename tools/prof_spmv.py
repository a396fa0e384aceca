"""Runs a fixed number of SpMV launches of ONE configuration so that a
rocprofv3 pass over this script profiles exactly that kernel.

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_spmv.py --n 512
    rocprofv3 --pmc FETCH_SIZE -d out_f -- python3 tools/prof_spmv.py --n 512
    rocprofv3 --pmc WRITE_SIZE -d out_w -- python3 tools/prof_spmv.py --n 512
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--dot", action="store_true")
    ap.add_argument("--set", nargs="*", default=[], help="knob=value ...")
    ap.add_argument("--ctx", nargs="*", default=[],
                    help="context option=value ... (e.g. const_diagonals=0)")
    ap.add_argument("--no-lat", action="store_true",
                    help="no lattice form: the plan takes the LX form")
    ap.add_argument("--asym", action="store_true",
                    help="general storage, made non-symmetric: the full diagonal form")
    ap.add_argument("--no-bake", action="store_true",
                    help="no baked copy: the CSR-order lattice kernels")
    ap.add_argument("--no-lx", action="store_true",
                    help="neither: the plain gather kernel")
    args = ap.parse_args()
    ctx = hip.Context(0)
    for kv in args.ctx:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.no_lat or args.no_lx:
        ctx.set_option("lat_min_nnz", 1 << 62)
    if args.no_lx:
        ctx.set_option("lx_min_nnz", 1 << 62)
    n, N = args.n, args.n ** 3
    part = hip.PART_LOCAL_LOWER if args.symmetric else hip.PART_ALL
    blk = hip.poisson3d_block(ctx, n, 0, N, part, with_diagonal=args.symmetric)
    if args.asym:  # three entries changed: the matrix is no longer symmetric
        ctx.copy_h2d(blk.values.ptr + 8 * 12345, np.array([-1.25, -1.5, -1.75]))
    if not args.no_bake and not (args.no_lat or args.no_lx):
        blk.bake()  # diagonal form (general: the matrix is found symmetric)
    for kv in args.set:
        k, v = kv.split("=")
        blk.set(k, int(v))
    x, y = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
    ctx.fill_gaussian(N, 0, N, x.ptr)
    partials = ctx.empty(ctx.dot_partials_len, np.float64) if args.dot else None
    # calibration launches with known byte counts: a streaming dot product
    # (reads 2*N*8 B, 16 B per lane) and a device memset (writes N*8 B)
    cal = ctx.empty(ctx.dot_partials_len, np.float64)
    for _ in range(3):
        hip.call("spmv_hip_dot_partial_f64", ctx.h, N, x.ptr, y.ptr, cal.ptr,
                 None)
        ctx.fill_const(N, 1.0, y.ptr)
    for _ in range(args.reps):
        blk.mult(1.0, x.ptr, 0.0, y.ptr,
                 dot_partials=partials.ptr if partials else None)
    ctx.synchronize()
    print("done", blk.nnz)
    blk.free()
    ctx.close()


if __name__ == "__main__":
    main()
