"""Map the 'regions' of device memory: K buffers of S MiB from separate
hipMalloc calls; y += a x for every pair; pairs that are slow together share
whatever resource the region stands for."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spmv_amd import hip  # noqa: E402
from kbench import time_ms  # noqa: E402

ctx = hip.Context(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = S * (1 << 20) // 8
bufs = [ctx.empty(N, np.float64) for _ in range(K)]
for b in bufs:
    ctx.fill_const(N, 1.0, b.ptr)
print(json.dumps(dict(ptrs=[hex(b.ptr) for b in bufs])))
for i in range(K):
    row = []
    for j in range(K):
        if i == j:
            row.append(0)
            continue
        t, _ = time_ms(ctx, lambda: hip.call("spmv_hip_axpy_f64", ctx.h, N, 0.5,
                                             bufs[i].ptr, bufs[j].ptr, None), 5)
        row.append(int(round(3 * N * 8 / t / 1e6)))  # GB/s
    print(json.dumps(dict(x=i, gbs_by_y=row)), flush=True)
