"""Helpers shared by the GPU kernel tests (tests/test_gpu_*.py): one SpMV through
the C ABI with a NaN-poisoned output, and the seeded test matrices several
kernel families use."""
import os

import numpy as np

from spmv_amd import hip

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
EXACT_ALGOS = [hip.ALGO_ROWBLOCK, hip.ALGO_SCALAR]


def run_spmv(ctx, rowptr, colind, values, x, nrows, ncols, alpha=1.0,
             beta=0.0, y0=None, algo=hip.ALGO_AUTO, diagonal=None,
             symmetric=False, knobs=None, dtype=np.float64):
    blk = hip.CsrBlock(ctx, nrows, ncols, rowptr, colind, values, diagonal,
                       symmetric, algo, dtype)
    for k, v in (knobs or {}).items():
        blk.set(k, v)
    dx = ctx.upload(x, dtype)
    # NaN-poisoned output when beta == 0: the kernel must not read it
    init = np.full(nrows, np.nan, dtype) if y0 is None else y0
    dy = ctx.upload(init, dtype)
    blk.mult(alpha, dx.ptr, beta, dy.ptr)
    y = dy.numpy()
    for b in (dx, dy):
        b.free()
    blk.free()
    return y


def banded_mixed(rng, n):
    """Rows near the diagonal plus a few far blocks, and a stretch of rows with
    scattered columns (those row blocks cannot be staged)."""
    rows, cols = [], []
    for i in range(n):
        near = i + rng.integers(-40, 41, 5)
        far = (i + 2000 + rng.integers(0, 30, 2)) % n
        c = np.concatenate([near, far])
        if 1024 <= i < 1536:  # two row blocks of scattered columns
            c = rng.integers(0, n, 9)
        c = np.clip(c, 0, n - 1)
        rows += [i] * len(c)
        cols += list(c)
    order = np.lexsort((np.arange(len(rows)), rows))
    rows, cols = np.array(rows)[order], np.array(cols, np.int32)[order]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp).astype(np.int32), cols, rng.uniform(-1, 1, len(cols))


def stencil_csr(rng, N, offsets, drop=0.0, dtype=np.float64):
    """Rows i with entries in columns i + d for d in `offsets` (ascending),
    kept when in range and, with probability `drop`, removed at random."""
    rows, cols = [], []
    for d in sorted(offsets):
        i = np.arange(max(0, -d), min(N, N - d))
        keep = rng.random(len(i)) >= drop
        rows.append(i[keep])
        cols.append(i[keep] + d)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rp = np.zeros(N + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return (np.cumsum(rp).astype(np.int32), cols.astype(np.int32),
            rng.uniform(-1, 1, len(cols)).astype(dtype))
