"""Inside ONE large allocation: y += a x with x at a fixed offset and y at
offset d -- where are the boundaries of the 'classes' of device memory (pairs of
streams inside one class run slower than pairs across two)?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spmv_amd import hip  # noqa: E402
from kbench import time_ms  # noqa: E402

ctx = hip.Context(0)
GiB = 1 << 30
total_gib = int(sys.argv[1]) if len(sys.argv) > 1 else 48
total = total_gib * GiB
big = ctx.empty(total // 8, np.float64)
ctx.fill_const(total // 8, 1.0, big.ptr)
print(json.dumps(dict(ptr=hex(big.ptr), gib=total_gib)))
N = (1 << 30) // 8
for x_off in (0, 20 * GiB, 40 * GiB):
    row = []
    for g in range(0, total_gib - 1, 2):
        y_off = g * GiB
        if abs(y_off - x_off) < GiB:
            row.append(0)
            continue
        t, _ = time_ms(ctx, lambda: hip.call("spmv_hip_axpy_f64", ctx.h, N, 0.5,
                                             big.ptr + x_off, big.ptr + y_off,
                                             None), 4)
        row.append(int(round(3 * N * 8 / t / 1e6)))
    print(json.dumps(dict(x_off_gib=x_off // GiB, gbs_by_y_at_2gib_steps=row)),
          flush=True)
