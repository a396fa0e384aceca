"""Ad-hoc knob sweep of the general row-block SpMV on the device-generated
Poisson matrix: every combination of the given knob values, interleaved.

    python tools/ksweep.py --n 512 --reps 8 --knob zwalk=0,1 --knob lat_blocks_per_cu=2,4
"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spmv_amd import hip, poisson  # noqa: E402
from kbench import time_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--knob", action="append", default=[],
                    help="name=v1,v2,... (applied in the order given)")
    ap.add_argument("--dot", action="store_true", help="fused p.Ap partials")
    ap.add_argument("--symmetric", action="store_true",
                    help="symmetric storage (lower block + diagonal)")
    ap.add_argument("--bake", action="store_true",
                    help="general storage: let the plan find the matrix symmetric")
    ap.add_argument("--asym", action="store_true",
                    help="general storage, made non-symmetric: the FULL diagonal form")
    ap.add_argument("--no-lat", action="store_true",
                    help="no lattice analysis: the LX form")
    ap.add_argument("--no-lx", action="store_true",
                    help="... nor the LX form: the plain row-block kernel")
    ap.add_argument("--stencil27", action="store_true",
                    help="the 27-point operator instead of the 7-point one")
    ap.add_argument("--set", nargs="*", default=[], help="ctx option=value ...")
    ap.add_argument("--get", nargs="*", default=[],
                    help="plan keys to print once the plan is built and baked")
    ap.add_argument("--out", default=None)
    ap.add_argument("--calib", action="store_true",
                    help="also time plain streaming kernels on this box")
    args = ap.parse_args()
    ctx = hip.Context(0)
    ctx.set_option("lx_max_x_bytes", 1 << 62)  # build the LX form at any size
    if args.no_lat or args.no_lx:
        ctx.set_option("lat_min_nnz", 1 << 62)
    if args.no_lx:
        ctx.set_option("lx_min_nnz", 1 << 62)
    for kv in args.set:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    n, N = args.n, args.n ** 3
    if args.stencil27:
        ctx.set_option("poisson_stencil", 27)
    if args.symmetric:
        blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_LOCAL_LOWER,
                                  with_diagonal=True)
    else:
        blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
    if args.asym:  # one entry changed: the matrix is no longer symmetric
        ctx.copy_h2d(blk.values.ptr + 8 * 12345, np.array([-1.25, -1.5, -1.75]))
    if args.symmetric or args.bake or args.asym:
        blk.bake()  # the diagonal form; knob sdia=0 runs the CSR-order kernel
    if args.get:
        print(json.dumps(dict(plan={k: blk.get(k) for k in args.get})), flush=True)
    x, y = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
    ctx.fill_gaussian(N, 0, N, x.ptr)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    nbytes = (poisson.sym_csr_bytes(N, blk.nnz) if args.symmetric
              else poisson.csr_bytes(N, N, blk.nnz))
    if args.calib:
        # what this box streams: a dot product over two N-vectors (reads) and a
        # fill (writes), both through the library's BLAS-1 kernels
        y2 = ctx.empty(N, np.float64)
        ctx.fill_const(N, 1.0, y2.ptr)
        b_r, _ = time_ms(ctx, lambda: hip.call("spmv_hip_dot_partial_f64", ctx.h,
                                               N, x.ptr, y2.ptr, part.ptr, None),
                         args.reps)
        b_w, _ = time_ms(ctx, lambda: ctx.fill_const(N, 1.0, y2.ptr), args.reps)
        print(json.dumps(dict(calib=dict(read_gbs=round(16 * N / b_r / 1e6, 1),
                                         write_gbs=round(8 * N / b_w / 1e6, 1)))),
              flush=True)
        y2.free()
    names = [k.split("=")[0] for k in args.knob]
    values = [[int(v) for v in k.split("=")[1].split(",")] for k in args.knob]
    rows = []
    for combo in itertools.product(*values):
        try:
            for k, v in zip(names, combo):
                blk.set(k, v)
        except Exception as e:  # an invalid combination: report and go on
            print(json.dumps(dict(knobs=dict(zip(names, combo)), error=str(e))))
            continue
        fn = (lambda: blk.mult(1.0, x.ptr, 0.0, y.ptr, dot_partials=part.ptr)) \
            if args.dot else (lambda: blk.mult(1.0, x.ptr, 0.0, y.ptr))
        best, med = time_ms(ctx, fn, args.reps)
        row = dict(n=n, knobs=dict(zip(names, combo)), ms=round(best, 4),
                   ms_med=round(med, 4), gbs=round(nbytes / best / 1e6, 1))
        rows.append(row)
        print(json.dumps(row), flush=True)
    rows.sort(key=lambda r: r["ms"])
    print("best:", json.dumps(rows[:5]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
