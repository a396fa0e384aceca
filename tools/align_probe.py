"""Does the SpMV time depend on where x and y sit relative to the plan's
arrays (HBM channel interleave)?  One process, one plan; x and y are views at
varying byte offsets into oversized buffers.

    python tools/align_probe.py --n 512 [--symmetric]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spmv_amd import hip  # noqa: E402
from kbench import time_ms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--symmetric", action="store_true")
    args = ap.parse_args()
    ctx = hip.Context(0)
    n, N = args.n, args.n ** 3
    part = hip.PART_LOCAL_LOWER if args.symmetric else hip.PART_ALL
    blk = hip.poisson3d_block(ctx, n, 0, N, part, with_diagonal=args.symmetric)
    blk.bake()
    pad = 1 << 22  # bytes of slack per buffer
    xb = ctx.empty(N + pad // 8, np.float64)
    yb = ctx.empty(N + pad // 8, np.float64)
    ctx.fill_gaussian(N + pad // 8, 0, N + pad // 8, xb.ptr)
    part_buf = ctx.empty(ctx.dot_partials_len, np.float64)
    offs = [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 3 << 19]
    for ox in offs:
        for oy in offs:
            fn = lambda: blk.mult(1.0, xb.ptr + ox, 0.0, yb.ptr + oy,  # noqa: E731
                                  dot_partials=part_buf.ptr)
            best, med = time_ms(ctx, fn, args.reps)
            print(json.dumps(dict(ox=ox, oy=oy, ms=round(best, 4),
                                  ms_med=round(med, 4))), flush=True)


if __name__ == "__main__":
    main()
