import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from spmv_amd import hip
from kbench import time_ms
ctx = hip.Context(0)
n = 512; N = n ** 3
x = ctx.empty(N, np.float64); y = ctx.empty(N, np.float64)
ctx.fill_gaussian(N, 0, N, x.ptr)
part = ctx.empty(ctx.dot_partials_len, np.float64)
junk = []
for trial in range(8):
    blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
    blk.bake()
    best, med = time_ms(ctx, lambda: blk.mult(1.0, x.ptr, 0.0, y.ptr, dot_partials=part.ptr), 6)
    print(json.dumps(dict(trial=trial, ms=round(best, 4), med=round(med, 4), values_ptr=hex(blk.values.ptr))), flush=True)
    blk.free()
    if trial % 2 == 0:  # perturb the allocator between trials
        junk.append(ctx.empty((trial + 1) * 12345678, np.float64))
