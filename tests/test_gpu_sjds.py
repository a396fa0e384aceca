"""GPU parity tests of the sliced jagged form (spmv_sjds*.hip): general and
symmetric storage, long rows, plan memory, coefficients updated in place, the
FEM-like matrices -- bit-exact against the CPU oracle (split from
test_gpu_kernels.py in round 6; bars: that file's header)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from gpu_helpers import EXACT_ALGOS, GOLDEN, banded_mixed as _banded_mixed, run_spmv, \
    stencil_csr as _stencil_csr
from spmv_amd import hip, poisson
from util import U, abs_bound, lower_split, random_csr

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------
# Sliced jagged form (spmv_sjds.hip): ragged / long rows, x staged in LDS
# ---------------------------------------------------------------------------
@pytest.fixture()
def sj_ctx():
    c = hip.Context(0)
    c.set_option("sj_min_nnz", 0)       # build the form for small matrices too
    c.set_option("lx_min_nnz", 1 << 62)  # ... instead of the LX form
    c.set_option("lat_min_nnz", 1 << 62)
    yield c
    c.close()


def _sj_cases():
    rng = np.random.default_rng(0x5EED0042)
    cases = {}
    # random ragged matrices: unsorted, repeated columns, empty rows, rows far
    # longer than a slice is wide (the wave takes them over), rectangular
    for name, (nr, nc, avg, nlong, llen) in dict(
            tiny=(64, 64, 3, 0, 0), ragged=(1500, 1500, 9, 2, 700),
            long_rows=(700, 5000, 20, 3, 5000), wide=(4000, 900, 40, 1, 100),
            dense_rows=(300, 300, 120, 0, 0), odd=(1027, 3001, 6, 5, 130)).items():
        cases[name] = random_csr(rng, nr, nc, avg, long_rows=nlong, long_len=llen)
    # FEM-like: clusters around the diagonal (everything staged, 16-bit codes)
    cases["fem"] = poisson.fem_like_csr(6000, jitter=64, layer=400)
    cases["fem_tail"] = poisson.fem_like_csr(9000, jitter=64, layer=500,
                                             tail_permille=20, tail_min=100,
                                             tail_max=900, tail_stride=4)
    # columns spread over 3 M: beyond the plan's bitmap span -> far entries
    rp, ci, va = random_csr(rng, 2000, 3_000_000, 8)
    near = rng.random(len(ci)) < 0.7  # ... the rest near enough to be staged
    ci[near] = rng.integers(0, 4000, int(near.sum())).astype(np.int32)
    cases["far"] = (rp, ci, va)
    return cases


@pytest.mark.parametrize("wpb,unit,sigma", [(4, 1, 1), (8, 2, 1), (16, 4, 1), (4, 4, 1),
                                            (16, 1, 1), (16, 2, 0), (0, 0, 1)])
def test_sliced_jagged_form_bit_exact(sj_ctx, wpb, unit, sigma):
    """csr_sjds_kernel against oracle.csr_spmv (csr_kernels.cpp:41-51), every
    element identical: blocks of 4 / 8 / 16 slices, 1 / 2 / 4 entries per lane
    and step (0 = the plan's choice), staged and far entries (a chunk budget of
    8 forces most entries far), long rows (the 8-lanes-per-row phase) and rows
    the wave takes over inside a slice, alpha / beta, fused dot, fp32; other
    value arrays than the baked one and a dropped copy take the CSR-order
    kernels."""
    ctx = sj_ctx
    ctx.set_option("sj_wpb", wpb)
    ctx.set_option("sj_unit", unit)
    # blocks of 16 slices: sorted by length across the block, two slices per wave
    # (the sigma layout) -- or every slice sorted for itself
    ctx.set_option("sj_sigma", sigma)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for name, (rp, ci, va) in _sj_cases().items():
        nr = len(rp) - 1
        nc = int(ci.max()) + 1 if len(ci) else 1
        nc = {"long_rows": 5000, "wide": 900, "odd": 3001,
              "far": 3_000_000}.get(name, max(nc, nr))
        rng = np.random.default_rng(len(ci))
        x = rng.uniform(-1, 1, nc)
        y0 = rng.uniform(-1, 1, nr)
        # (left to choose, the plan does not build the form when most entries
        # would be far)
        for budget in ((432, 8) if wpb else (432,)):
            ctx.set_option("sj_max_chunks", budget)
            blk = hip.CsrBlock(ctx, nr, nc, rp, ci, va, None, False)
            # (the structure is built with the values: plan_bake_values, once the
            # diagonal forms have refused the matrix)
            assert blk.get("sj_built") == 0 and blk.get("sjds") == 0, name
            blk.bake()
            assert blk.get("sj_built") == 1 and blk.get("sjds") == 1, name
            assert blk.get("lx") == 0
            if wpb:
                assert blk.get("sj_wpb") == wpb and blk.get("sj_unit") == unit
                # (the sigma layout: left only for rows that are long AND alike)
                assert blk.get("sj_sigma") == (1 if wpb == 16 and sigma else 0), name
            if name == "far" or (budget == 8 and nc > 1000):
                assert blk.get("sj_far_permille") > 0 and blk.get("sj_wide") == 1
            if name == "fem" and budget == 432:
                assert blk.get("sj_far_permille") == 0 and blk.get("sj_wide") == 0
            if name == "fem_tail" and budget == 432:
                # the long rows stay out of the slices (one wave each): the
                # short rows are staged entirely
                assert blk.get("sj_far_permille") == 0 and blk.get("sj_wide") == 0
                assert blk.get("sj_long_rows") > 50
                assert blk.get("sj_long_panels") == 1  # ascending columns
            if name in ("ragged", "long_rows", "odd"):
                assert blk.get("sj_long_rows") >= 2
                assert blk.get("sj_long_panels") == 0  # ... not here: gathered
            dx = ctx.upload(x)
            for alpha, beta in ((1.0, 0.0), (-0.75, 0.0), (2.5, -0.5)):
                y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
                dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
                dot = beta == 0 and nr == nc
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                assert np.array_equal(dy.numpy(), y_ref), (name, budget, alpha)
                if dot:
                    want = float(np.dot(x, alpha * oracle.csr_spmv(rp, ci, va, x)))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-11 * (np.abs(x) @ np.abs(y_ref) + 1)
                dy.free()
            if name == "fem_tail":  # the same rows gathered instead
                blk.set("sj_long_panels", 0)
                dy = ctx.upload(np.full(nr, np.nan))
                blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x))
                dy.free()
                blk.set("sj_long_panels", 1)
            # the plan's copy is tied to the array it was made from
            other = ctx.upload(2.0 * va)
            keep, blk.values = blk.values, other
            dy = ctx.upload(np.full(nr, np.nan))
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, 2.0 * va, x))
            blk.values = keep
            blk.bake(drop=True)
            assert blk.get("sjds") == 0 and blk.get("sj_built") == 1
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x))
            for b in (dx, dy, other):
                b.free()
            blk.free()
        ctx.set_option("sj_max_chunks", 432)
    part.free()
    # fp32
    rp, ci, va = poisson.fem_like_csr(5000, jitter=64, layer=300, tail_permille=30,
                                      tail_min=70, tail_max=400, tail_stride=2)
    va32 = va.astype(np.float32)
    x32 = np.random.default_rng(3).uniform(-1, 1, 5000).astype(np.float32)
    blk = hip.CsrBlock(ctx, 5000, 5000, rp, ci, va32, None, False,
                       dtype=np.float32)
    blk.bake()
    assert blk.get("sjds") == 1
    dx, dy = ctx.upload(x32), ctx.upload(np.full(5000, np.nan, np.float32))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va32, x32))
    for b in (dx, dy):
        b.free()
    blk.free()


def _long_row_matrix(rng, nr, nc, long_rows, short_avg=6):
    """Short random rows plus the given long rows (row -> sorted, strictly
    ascending column array)."""
    lens = rng.integers(1, 2 * short_avg, nr)
    for r, cols in long_rows.items():
        lens[r] = len(cols)
    rp = np.zeros(nr + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.empty(rp[-1], np.int32)
    for r in range(nr):
        if r in long_rows:
            ci[rp[r]:rp[r + 1]] = long_rows[r]
        else:
            lo = max(0, min(nc - 64, r - 32))
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(
                np.arange(lo, min(nc, lo + 64)), lens[r], replace=False))
    va = rng.uniform(-1, 1, rp[-1])
    return rp.astype(np.int32), ci, va


def test_long_rows_table_kernel_bit_exact(sj_ctx):
    """csr_sjds_longt_kernel (long rows with ascending columns, marched through
    LDS panels of x by the plan's table of panel crossings) against
    oracle.csr_spmv (csr_kernels.cpp:41-51), every element identical; the same
    rows by the older panel kernel (sj_long_table = 0) and gathered
    (sj_long_panels = 0).  Cases: rows spanning several panels in supergroups
    of neighbours (the benchmark's tail, scaled down), a partial last
    supergroup, fewer long rows than one group, rows that are NOT neighbours
    (more than 64 panels: the rows go one by one), an odd number of columns
    with long rows that end at the last one, ranges that end a panel exactly at
    its boundary, alpha / beta, the fused dot, fp32."""
    ctx = sj_ctx
    rng = np.random.default_rng(0x10C6)
    cases = {}
    # the benchmark's tail, scaled down: 1200 long rows of 100 ... 2000 entries,
    # one per 16 columns (up to 32,000 columns: four panels of 8192)
    cases["tail"] = poisson.fem_like_csr(40_000, jitter=64, layer=1200,
                                         tail_permille=30, tail_min=100,
                                         tail_max=2000, tail_stride=16)
    # 210 long rows: three full supergroups of 64 and a partial one
    cases["tail_partial"] = poisson.fem_like_csr(20_001, jitter=64, layer=700,
                                                 tail_permille=10, tail_min=300,
                                                 tail_max=1200, tail_stride=16)
    # three long rows only; one of them ends at the last (odd) column, one has
    # an entry at every column of a panel boundary's neighbourhood
    nc = 30_001
    cases["few"] = _long_row_matrix(rng, 3000, nc, {
        5: np.arange(0, 20_000, 7, dtype=np.int32),
        1500: np.arange(nc - 2500, nc, dtype=np.int32),
        2900: np.concatenate([np.arange(8192 - 200, 8192 + 200),
                              np.arange(16_384 - 1, 16_384 + 130)]).astype(np.int32)})
    # long rows that are not neighbours in x: columns over 3 M (> 64 panels)
    nc_far = 3_000_001
    far = {r: np.sort(rng.choice(nc_far, 400 + 37 * k, replace=False)).astype(np.int32)
           for k, r in enumerate(range(100, 2000, 190))}
    far[1990] = np.arange(nc_far - 300, nc_far, dtype=np.int32)
    cases["not_neighbours"] = _long_row_matrix(rng, 2000, nc_far, far)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for name, (rp, ci, va) in cases.items():
        nr = len(rp) - 1
        ncols = {"few": nc, "not_neighbours": nc_far}.get(name, nr)
        x = rng.uniform(-1, 1, ncols)
        y0 = rng.uniform(-1, 1, nr)
        blk = hip.CsrBlock(ctx, nr, ncols, rp, ci, va, None, False)
        blk.bake()
        assert blk.get("sjds") == 1, name
        assert blk.get("sj_long_rows") >= 3, name
        assert blk.get("sj_long_panels") == 1 and blk.get("sj_long_table") == 1, name
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (-0.75, 0.0), (2.5, -0.5)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(sj_long_table=1), dict(sj_long_table=0),
                          dict(sj_long_panels=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
                dot = beta == 0 and nr == ncols
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                assert np.array_equal(dy.numpy(), y_ref), (name, knobs, alpha)
                if dot:
                    want = float(np.dot(x, alpha * oracle.csr_spmv(rp, ci, va, x)))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-11 * (np.abs(x) @ np.abs(y_ref) + 1)
                dy.free()
                blk.set("sj_long_panels", 1)
                blk.set("sj_long_table", 1)
        dx.free()
        blk.free()
    part.free()
    # fp32
    rp, ci, va = cases["tail_partial"]
    nr = len(rp) - 1
    va32 = va.astype(np.float32)
    x32 = rng.uniform(-1, 1, nr).astype(np.float32)
    blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va32, None, False, dtype=np.float32)
    blk.bake()
    assert blk.get("sjds") == 1 and blk.get("sj_long_table") == 1
    dx, dy = ctx.upload(x32), ctx.upload(np.full(nr, np.nan, np.float32))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va32, x32))
    for b in (dx, dy):
        b.free()
    blk.free()


def _sym_lower_cases():
    """(rowptr, colind, values, diagonal) of strictly lower blocks without lattice
    structure: the lower part of the FEM-like matrix, the same with a few long
    rows and one long COLUMN (a long row of the transposed block), random
    ragged rows with unsorted columns and empty rows."""
    rng = np.random.default_rng(0x51A3)
    cases = {}
    for name, kw in (("fem", dict()),
                     ("fem_tail", dict(tail_permille=5, tail_min=150, tail_max=600,
                                       tail_stride=3))):
        rp, ci, va = poisson.fem_like_csr(7000, jitter=64, layer=400, **kw)
        cases[name] = lower_split(rp, ci, va)
    nr = 5000
    lens = rng.integers(0, 14, nr)
    lens[0] = 0
    lens[rng.integers(1, nr, 200)] = 0
    lens = np.minimum(lens, np.arange(nr))
    rp = np.zeros(nr + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.empty(rp[-1], np.int32)
    for r in range(nr):
        if lens[r]:
            ci[rp[r]:rp[r + 1]] = rng.permutation(
                rng.choice(r, lens[r], replace=False))  # unsorted, below the diagonal
    # a long column: every row from 1000 on has an entry in column 7
    add = np.arange(1000, nr)
    rows = np.concatenate([np.repeat(np.arange(nr), lens), add])
    cols = np.concatenate([ci, np.full(len(add), 7, np.int32)])
    order = np.argsort(rows, kind="stable")
    rows, cols = rows[order], cols[order]
    rp2 = np.zeros(nr + 1, np.int64)
    np.add.at(rp2, rows + 1, 1)
    rp2 = np.cumsum(rp2)
    va = rng.uniform(-1, 1, len(cols))
    cases["ragged_long_column"] = (rp2.astype(np.int32), cols.astype(np.int32), va,
                                   rng.uniform(1, 2, nr))
    # an "arrow": ragged short rows, 40 LONG rows of 200-900 entries with
    # UNSORTED columns (the gathered long-row kernel), and a dense LAST row
    # (within the arrays' last entries: it stays inside the slices)
    nr = 6000
    lens = np.minimum(rng.integers(0, 12, nr), np.arange(nr))
    long_rows = rng.choice(np.arange(1500, nr - 1), 40, replace=False)
    lens[long_rows] = rng.integers(200, 900, 40)
    lens[nr - 1] = 700
    rp = np.zeros(nr + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    ci = np.empty(rp[-1], np.int32)
    for r in range(nr):
        if lens[r]:
            ci[rp[r]:rp[r + 1]] = rng.permutation(rng.choice(r, lens[r], replace=False))
    cases["arrow_unsorted"] = (rp.astype(np.int32), ci, rng.uniform(-1, 1, len(ci)),
                               rng.uniform(1, 2, nr))
    return cases


@pytest.mark.parametrize("wpb", [0, 8, 16])
def test_symmetric_storage_sliced_jagged_bit_exact(sj_ctx, wpb):
    """Symmetric storage of matrices WITHOUT lattice structure: the reference's
    loop (csr_kernels.cpp:26-40) seen from the row, in the sliced jagged form of
    the MERGED matrix -- a row's stored lower entries (sum starts at d_i x_i),
    then the entries of its column in ascending (r, j), where the sum turns into
    y_i = fl(alpha sum + beta y0_i) and every product into fl(fl(alpha v) x_r)
    -- against oracle.csr_spmv_sym, every element identical; any alpha / beta,
    the fused dot, fp32, coefficients rewritten in place, and the
    transposed-map kernel (sjds = 0) on the same plan."""
    ctx = sj_ctx
    ctx.set_option("sj_wpb", wpb)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    rng = np.random.default_rng(77)
    for name, (rp, ci, va, dg) in _sym_lower_cases().items():
        nr = len(rp) - 1
        x = rng.uniform(-1, 1, nr)
        y0 = rng.uniform(-1, 1, nr)
        blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va, dg, True)
        assert blk.get("sym_sj") == 0
        if name == "ragged_long_column":
            # a long COLUMN holds more than 5 % of the entries (it would stay
            # inside the slices): left to itself the plan keeps the
            # transposed-map kernel ...
            with pytest.raises(Exception):  # SPMV_HIP_ENOTSUP: no form applies
                blk.bake()
            assert blk.get("sym_sj") == 0 and blk.get("sjds") == 0, name
            dx, dy = ctx.upload(x), ctx.upload(np.full(nr, np.nan))
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(), oracle.csr_spmv_sym(rp, ci, va, dg, x))
            dx.free(), dy.free()
            ctx.set_option("sym_sj_long_permille", 1000)  # ... here: take the form
        blk.bake()
        ctx.set_option("sym_sj_long_permille", 50)
        assert blk.get("sym_sj") == 1 and blk.get("sjds") == 1, name
        if wpb:
            assert blk.get("sj_wpb") == wpb
        # LONG rows of the stored block: their lower part by the long-row
        # kernels on the caller's arrays (table-driven where the columns
        # ascend, gathered where not), launched before the slices' kernel
        nlong = blk.get("sj_long_rows")
        if name == "fem_tail":
            assert nlong > 10 and blk.get("sj_long_sorted") == 1
        elif name == "arrow_unsorted":
            assert nlong == 40 and blk.get("sj_long_sorted") == 0
        else:
            assert nlong == 0, name
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (-0.75, 0.0), (2.5, -0.5)):
            y_ref = oracle.csr_spmv_sym(rp, ci, va, dg, x, alpha, beta, y0)
            for sjds in (1, 0):  # ... and the transposed-map kernel
                blk.set("sjds", sjds)
                dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if beta == 0 else None)
                assert np.array_equal(dy.numpy(), y_ref), (name, alpha, sjds)
                if beta == 0:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-11 * (np.abs(x) @ np.abs(y_ref) + 1)
                dy.free()
            blk.set("sjds", 1)
        # coefficients rewritten in place
        for scale in (-0.5, 3.0):
            va2, dg2 = scale * va + 0.25, dg * scale
            blk.values.write(va2)
            blk.diagonal.write(dg2)
            blk.values_changed()
            assert blk.get("sym_sj") == 1
            dy = ctx.upload(np.full(nr, np.nan))
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(),
                                  oracle.csr_spmv_sym(rp, ci, va2, dg2, x)), (name, scale)
            dy.free()
        # the SIGN of a zero: positive coefficients, x = -0.0 and
        # y0 = -0.0 everywhere -- every product and every sum is -0.0, and so is
        # the reference's y = fl(1 sum) + fl(1 y0); a kernel that pads a row's
        # sum with +0.0 (the long-row kernels did, before the sums of symmetric
        # storage could start at d_i x_i = -0.0) returns +0.0.  Compared as BITS:
        # np.array_equal takes the two zeros for equal.
        va3, dg3 = np.abs(va) + 0.125, np.abs(dg) + 1.0
        blk.values.write(va3)
        blk.diagonal.write(dg3)
        blk.values_changed()
        xz, yz = np.full(nr, -0.0), np.full(nr, -0.0)
        yz_ref = oracle.csr_spmv_sym(rp, ci, va3, dg3, xz, 1.0, 1.0, yz)
        assert np.all(np.signbit(yz_ref)) and not yz_ref.any()
        dxz = ctx.upload(xz)
        for sjds in (1, 0):
            blk.set("sjds", sjds)
            dy = ctx.upload(yz)
            blk.mult(1.0, dxz.ptr, 1.0, dy.ptr)
            assert np.array_equal(dy.numpy().view(np.uint64),
                                  yz_ref.view(np.uint64)), (name, sjds)
            dy.free()
        blk.set("sjds", 1)
        dxz.free()
        blk.values.write(va2)  # (back to the last coefficients of the loop above)
        blk.diagonal.write(dg2)
        blk.values_changed()
        # a dropped copy: the transposed-map kernel again
        blk.bake(drop=True)
        assert blk.get("sym_sj") == 0 and blk.get("sjds") == 0
        dy = ctx.upload(np.full(nr, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv_sym(rp, ci, va2, dg2, x))
        for b in (dx, dy):
            b.free()
        blk.free()
    part.free()
    # long rows kept INSIDE the slices (context option sym_sj_long_rows = 0: they
    # then count against the 5 %, lifted here) -- the round-4 form of the same
    ctx.set_option("sym_sj_long_rows", 0)
    ctx.set_option("sym_sj_long_permille", 1000)
    for name in ("fem_tail", "arrow_unsorted"):
        rp, ci, va, dg = _sym_lower_cases()[name]
        nr = len(rp) - 1
        x = rng.uniform(-1, 1, nr)
        y0 = rng.uniform(-1, 1, nr)
        blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va, dg, True)
        blk.bake()
        assert blk.get("sym_sj") == 1 and blk.get("sj_long_rows") == 0
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (2.5, -0.5)):
            dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr)
            assert np.array_equal(dy.numpy(), oracle.csr_spmv_sym(rp, ci, va, dg, x,
                                                                  alpha, beta, y0)), name
            dy.free()
        dx.free()
        blk.free()
    ctx.set_option("sym_sj_long_rows", 1)
    ctx.set_option("sym_sj_long_permille", 50)
    # fp32
    rp, ci, va, dg = _sym_lower_cases()["fem"]
    nr = len(rp) - 1
    va32, dg32 = va.astype(np.float32), dg.astype(np.float32)
    x32 = rng.uniform(-1, 1, nr).astype(np.float32)
    blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va32, dg32, True, dtype=np.float32)
    blk.bake()
    assert blk.get("sym_sj") == 1
    dx, dy = ctx.upload(x32), ctx.upload(np.full(nr, np.nan, np.float32))
    blk.mult(-1.5, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv_sym(rp, ci, va32, dg32, x32, -1.5))
    for b in (dx, dy):
        b.free()
    blk.free()
    rp, ci, va, dg = _sym_lower_cases()["fem_tail"]  # ... with long rows
    nr = len(rp) - 1
    va32, dg32 = va.astype(np.float32), dg.astype(np.float32)
    x32 = rng.uniform(-1, 1, nr).astype(np.float32)
    blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va32, dg32, True, dtype=np.float32)
    blk.bake()
    assert blk.get("sym_sj") == 1 and blk.get("sj_long_rows") > 10
    dx, dy = ctx.upload(x32), ctx.upload(np.full(nr, np.nan, np.float32))
    for alpha in (1.0, -1.5):
        blk.mult(alpha, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(),
                              oracle.csr_spmv_sym(rp, ci, va32, dg32, x32, alpha))
    for b in (dx, dy):
        b.free()
    blk.free()


def test_sliced_jagged_rows_too_long_for_the_sigma_word(sj_ctx):
    """a row that stays in the slices shares a 32-bit word with its
    position -- 21 bits of length beside the sigma layout's 10.  A bordered
    matrix's dense LAST row (never taken out as long: it ends with the arrays)
    with more than 2^21 entries makes the plan leave the sigma layout; the same
    as a dense COLUMN of symmetric storage (a row of the merged matrix)."""
    ctx = sj_ctx
    rng = np.random.default_rng(0x2021)
    n = (1 << 21) + 70_000
    # general storage: tridiagonal + a dense last row
    i = np.arange(n - 1)
    rows = np.concatenate([i, i[1:], i[:-1], np.full(n, n - 1)])
    cols = np.concatenate([i, i[1:] - 1, i[:-1] + 1, np.arange(n)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order].astype(np.int32)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    va = rng.uniform(-1, 1, len(cols))
    x = rng.uniform(-1, 1, n)
    blk = hip.CsrBlock(ctx, n, n, rp, cols, va, None, False)
    blk.bake()
    assert blk.get("sjds") == 1 and blk.get("sj_sigma") == 0
    assert blk.get("sj_long_rows") == 0  # the dense row is inside a slice
    dx, dy = ctx.upload(x), ctx.upload(np.full(n, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, cols, va, x))
    dy.free()
    blk.free()
    # symmetric storage: a subdiagonal + a dense first column
    rows = np.concatenate([np.arange(1, n), np.arange(2, n)])
    cols = np.concatenate([np.zeros(n - 1, np.int64), np.arange(1, n - 1)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order].astype(np.int32)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    va = rng.uniform(-1, 1, len(cols))
    dg = rng.uniform(1, 2, n)
    ctx.set_option("sym_sj_long_permille", 1000)  # (the column is half the entries)
    blk = hip.CsrBlock(ctx, n, n, rp, cols, va, dg, True)
    blk.bake()
    ctx.set_option("sym_sj_long_permille", 50)
    assert blk.get("sym_sj") == 1 and blk.get("sj_sigma") == 0
    for alpha, beta in ((1.0, 0.0), (-0.5, 0.0)):
        dy = ctx.upload(np.full(n, np.nan))
        blk.mult(alpha, dx.ptr, beta, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv_sym(rp, cols, va, dg, x, alpha))
        dy.free()
    dx.free()
    blk.free()


def test_plan_owns_the_matrix_and_the_caller_releases_it(sj_ctx):
    """PLAN MEMORY (ABI 4).  A general plan in the sliced jagged form without
    long rows, or in a diagonal form, reports that it no longer reads colind /
    values (spmv_hip_csr_plan_owns_matrix = 3); after
    spmv_hip_csr_plan_release_matrix the caller frees them: launches with the
    same (now dangling) pointers return the same bits, and whatever would read
    the arrays is refused cleanly -- plan_values_changed, a re-bake, a knob that
    selects a CSR-order kernel, a launch with other pointers.  Symmetric storage
    in the merged form likewise (its transposed map goes too).  Plans that still
    stream the caller's arrays (long rows, no baked copy) own nothing."""
    ctx = sj_ctx
    rng = np.random.default_rng(0x0A4)
    # (1) sliced jagged, ragged rows, no long ones
    rp, ci, va = poisson.fem_like_csr(9000, jitter=64, layer=500)
    nr = len(rp) - 1
    x = rng.uniform(-1, 1, nr)
    y0 = rng.uniform(-1, 1, nr)
    blk = hip.CsrBlock(ctx, nr, nr, rp, ci, va, None, False)
    assert blk.owns_matrix() == 0  # nothing baked yet
    blk.bake()
    assert blk.get("sjds") == 1 and blk.get("sj_long_rows") == 0
    assert blk.owns_matrix() == 3
    dx = ctx.upload(x)
    refs = {}
    for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
        refs[(alpha, beta)] = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
    assert blk.release_matrix() == 3
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for (alpha, beta), y_ref in refs.items():
        dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
        blk.mult(alpha, dx.ptr, beta, dy.ptr,
                 dot_partials=part.ptr if beta == 0 else None)
        assert np.array_equal(dy.numpy(), y_ref), (alpha, beta)
        dy.free()
    dy = ctx.upload(np.zeros(nr))
    with pytest.raises(Exception):
        blk.values_changed()  # the arrays it would re-read are gone
    with pytest.raises(Exception):
        blk.bake()
    with pytest.raises(Exception):
        blk.bake(drop=True)
    with pytest.raises(Exception):
        blk.set("sjds", 0)  # the CSR-order kernels would read freed memory
    with pytest.raises(Exception):
        blk.set("algo", hip.ALGO_SCALAR)
    blk.set("sj_blocks_per_cu", 1)  # (a knob of the form itself: fine)
    # another values pointer (a live allocation, so surely another address):
    # there is no CSR-order fallback any more
    # (the allocator may hand the freed address out again: the second half of a
    # double-length buffer cannot be it)
    other = ctx.upload(np.concatenate([va, va]))
    with pytest.raises(Exception):
        hip.call("spmv_hip_csr_spmv_f64", ctx.h, blk.plan, nr, nr, len(va),
                 blk.rowptr.ptr, blk.colind.ptr, other.ptr + 8 * len(va), None, 1.0,
                 dx.ptr, 0.0, dy.ptr, None, None)
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), refs[(1.0, 0.0)])
    for b in (other, dy, part):
        b.free()
    blk.free()
    # (2) long rows are streamed from the caller's arrays: nothing to release
    rp2, ci2, va2 = poisson.fem_like_csr(30_000, jitter=64, layer=900, tail_permille=20,
                                         tail_min=100, tail_max=1500, tail_stride=16)
    blk = hip.CsrBlock(ctx, 30_000, 30_000, rp2, ci2, va2, None, False)
    blk.bake()
    assert blk.get("sj_long_rows") > 0 and blk.owns_matrix() == 0
    assert blk.release_matrix() == 0
    with pytest.raises(Exception):  # not a subset of what the plan owns
        hip.call("spmv_hip_csr_plan_release_matrix", blk.plan, 3)
    blk.values_changed()  # still allowed
    blk.free()
    # (3) symmetric storage in the merged form, no long rows: the kernel reads
    # the merged copy, the row pointer and the diagonal -- colind and values go,
    # and with them the plan's transposed map and value positions (16 B per
    # stored entry, needed by the refused paths only)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    blk = hip.CsrBlock(ctx, nr, nr, lrp, lci, lva, dg, True)
    blk.bake()
    assert blk.get("sym_sj") == 1 and blk.owns_matrix() == 3
    kib0 = blk.get("plan_kib")
    srefs = {ab: oracle.csr_spmv_sym(lrp, lci, lva, dg, x, ab[0], ab[1], y0)
             for ab in ((1.0, 0.0), (-0.5, 0.75), (2.0, 0.0))}
    assert blk.release_matrix() == 3
    assert kib0 - blk.get("plan_kib") >= 16 * len(lva) // 1024 - 1
    for (alpha, beta), y_ref in srefs.items():
        dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
        blk.mult(alpha, dx.ptr, beta, dy.ptr)
        assert np.array_equal(dy.numpy(), y_ref), ("symmetric", alpha, beta)
        dy.free()
    for bad in (lambda: blk.values_changed(), lambda: blk.set("sjds", 0),
                lambda: blk.set("sym_det", 0), lambda: blk.bake()):
        with pytest.raises(Exception):
            bad()
    blk.free()
    # ... with long rows (streamed from the caller's arrays): nothing
    trp, tci, tva, tdg = lower_split(rp2, ci2, va2)
    blk = hip.CsrBlock(ctx, 30_000, 30_000, trp, tci, tva, tdg, True)
    blk.bake()
    assert blk.get("sym_sj") == 1 and blk.get("sj_long_rows") > 0
    assert blk.owns_matrix() == 0
    blk.free()
    dx.free()
    # (4) a diagonal form (27-point stencil, values by offset)
    c2 = hip.Context(0)
    c2.set_option("lat_min_nnz", 0)
    c2.set_option("lx_min_nnz", 0)
    c2.set_option("const_diagonals", 0)
    n = 20
    rp, ci, va = poisson.stencil27_csr(n)
    ci = ci.astype(np.int32)
    va = va * rng.uniform(0.5, 1.5, len(va))  # not symmetric: the full form
    N = n ** 3
    x = rng.uniform(-1, 1, N)
    blk = hip.CsrBlock(c2, N, N, rp, ci, va, None, False)
    blk.bake()
    assert blk.get("wdia") == 1 and blk.owns_matrix() == 3
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    assert blk.release_matrix() == 3
    dx, dy = c2.upload(x), c2.upload(np.full(N, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), y_ref)
    with pytest.raises(Exception):
        blk.set("wdia", 0)
    with pytest.raises(Exception):
        blk.values_changed()
    for b in (dx, dy):
        b.free()
    blk.free()
    c2.close()


def test_mixed_precision_sliced_jagged_bit_exact(sj_ctx):
    """plan_bake_values_f32f64 on a plan in the sliced jagged form: the fp32 twin
    of the jagged copy (and, for the long rows, the caller's fp32 CSR values).
    spmv_f32f64 with the baked pointer = the reference loop
    (csr_kernels.cpp:41-51) on the fp32-rounded values in fp64, bit for bit:
    slices, long rows by panels (table-driven) and gathered, far entries, fused
    dot; another pointer takes the CSR-order kernels; both copies follow an
    update in place."""
    ctx = sj_ctx
    rng = np.random.default_rng(0x3264)
    cases = {"fem": poisson.fem_like_csr(6000, jitter=64, layer=400),
             "fem_tail": poisson.fem_like_csr(30_000, jitter=64, layer=900,
                                              tail_permille=20, tail_min=100,
                                              tail_max=1500, tail_stride=16),
             "ragged": random_csr(rng, 1500, 1500, 9, long_rows=2, long_len=700)}
    rp, ci, va = random_csr(rng, 2000, 3_000_000, 8)
    near = rng.random(len(ci)) < 0.7
    ci[near] = rng.integers(0, 4000, int(near.sum())).astype(np.int32)
    cases["far"] = (rp, ci, va)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for name, (rp, ci, va) in cases.items():
        nr = len(rp) - 1
        nc = 3_000_000 if name == "far" else nr
        va32 = va.astype(np.float32)
        x = rng.uniform(-1, 1, nc)
        y0 = rng.uniform(-1, 1, nr)
        blk = hip.CsrBlock(ctx, nr, nc, rp, ci, va, None, False)
        d32 = ctx.upload(va32, np.float32)
        blk.bake()
        assert blk.get("sjds") == 1 and blk.get("sj_mixed") == 0, name
        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan, d32.ptr, None)
        assert blk.get("sj_mixed") == 1, name
        if name == "fem_tail":
            assert blk.get("sj_long_rows") > 100 and blk.get("sj_long_table") == 1
        dx = ctx.upload(x)
        other = ctx.upload(va32, np.float32)

        def mixed(vals, alpha, beta, dot):
            dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
            hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, nr, nc, blk.nnz,
                     blk.rowptr.ptr, blk.colind.ptr, vals.ptr, float(alpha), dx.ptr,
                     float(beta), dy.ptr, part.ptr if dot else None, None)
            y = dy.numpy()
            dy.free()
            return y
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, alpha, beta, y0)
            for vals, knobs in ((d32, dict()), (d32, dict(sj_long_table=0)),
                                (d32, dict(sj_long_panels=0)), (other, dict())):
                for k, v in knobs.items():
                    blk.set(k, v)
                dot = beta == 0 and nr == nc
                assert np.array_equal(mixed(vals, alpha, beta, dot), y_ref), \
                    (name, alpha, knobs, vals is other)
                if dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-11 * (np.abs(x) @ np.abs(y_ref) + 1)
                blk.set("sj_long_table", 1)
                blk.set("sj_long_panels", 1)
        # the fp64 SpMV of the same plan is untouched
        dy = ctx.upload(np.full(nr, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x)), name
        # both copies follow an update in place
        va2 = rng.uniform(-1, 1, len(va))
        blk.values.write(va2)
        d32.write(va2.astype(np.float32))
        blk.values_changed()
        assert blk.get("sj_mixed") == 1
        assert np.array_equal(
            mixed(d32, 1.0, 0.0, False),
            oracle.csr_spmv(rp, ci, va2.astype(np.float32).astype(np.float64), x))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va2, x)), name
        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan, None, None)
        assert blk.get("sj_mixed") == 0
        assert np.array_equal(
            mixed(d32, 1.0, 0.0, False),
            oracle.csr_spmv(rp, ci, va2.astype(np.float32).astype(np.float64), x))
        for b in (dx, dy, d32, other):
            b.free()
        blk.free()
    part.free()


def test_plan_values_changed_after_updates_in_place():
    """spmv_hip_csr_plan_values_changed: a caller that keeps the sparsity and
    rewrites the coefficients IN PLACE (time stepping) -- three updates on every
    form that keeps its own copy of the values, each followed by the call, each
    product identical to the oracle's on the new values; without the call the
    copy is stale by contract (the old product), and a plan without a copy needs
    no call.  Forms: sliced jagged (ragged rows), half / full / constant
    diagonal form behind the general SpMV, symmetric storage, wide diagonal
    form; values that change the form on the way (constant -> varying ->
    constant, symmetric -> not symmetric)."""
    ctx = hip.Context(0)
    for k in ("sj_min_nnz", "lat_min_nnz", "lx_min_nnz"):
        ctx.set_option(k, 0)
    rng = np.random.default_rng(0xC0EFF)
    n = 12
    N = n ** 3
    prp, pci, pva = poisson.poisson3d_csr(n)
    pci = pci.astype(np.int32)

    def sym_values(scale):  # symmetric, varying coefficients on the 7-point lattice
        rows = np.repeat(np.arange(N), np.diff(prp))
        lo, hi = np.minimum(rows, pci), np.maximum(rows, pci)
        h = (lo * 1000003 + hi * 7919) % 1021
        return np.where(rows == pci, 6.0 * scale, -(1.0 + h / 1021.0) * scale)

    cases = []
    # (name, rowptr, colind, [values per step], symmetric storage?, expect form)
    frp, fci, fva = poisson.fem_like_csr(5000, jitter=64, layer=300,
                                         tail_permille=20, tail_min=100,
                                         tail_max=400, tail_stride=2)
    cases.append(("sjds", frp, fci, [fva, -0.5 * fva, rng.uniform(-1, 1, len(fva)),
                                     fva * 3.0], False, dict(sjds=1)))
    cases.append(("half_diagonal", prp, pci,
                  [sym_values(1.0), sym_values(0.25), sym_values(-2.0),
                   sym_values(7.0)], False, dict(sdia=1)))
    cases.append(("const_to_varying_and_back", prp, pci,
                  [pva, sym_values(1.0), 2.0 * pva,
                   rng.uniform(-1, 1, len(pva))], False, dict(sdia=1)))
    o27 = sorted(a * n * n + b * n + c for a in (-1, 0, 1) for b in (-1, 0, 1)
                 for c in (-1, 0, 1))
    wrp, wci, wva = _stencil_csr(rng, N, o27, drop=0.1)
    cases.append(("wide_diagonal", wrp, wci,
                  [wva, 0.5 * wva, rng.uniform(-1, 1, len(wva)), -wva], False,
                  dict(wdia=1)))
    lrp, lci, lva0, ldg0 = lower_split(prp, pci, sym_values(1.0))
    cases.append(("symmetric_storage", lrp, lci,
                  [(lower_split(prp, pci, sym_values(s))[2],
                    lower_split(prp, pci, sym_values(s))[3])
                   for s in (1.0, 0.5, -3.0, 2.0)], True, dict(sdia=1)))
    for name, rp, ci, steps, symmetric, form in cases:
        nr = len(rp) - 1
        x = rng.uniform(-1, 1, nr)
        v0 = steps[0]
        blk = hip.CsrBlock(ctx, nr, nr, rp, ci, v0[0] if symmetric else v0,
                           v0[1] if symmetric else None, symmetric)
        blk.bake()
        for k, v in form.items():
            assert blk.get(k) == v, (name, k)
        dx, dy = ctx.upload(x), ctx.upload(np.full(nr, np.nan))

        def ref(v):
            if symmetric:
                return oracle.csr_spmv_sym(rp, ci, v[0], v[1], x)
            return oracle.csr_spmv(rp, ci, v, x)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), ref(v0)), name
        for step, v in enumerate(steps[1:], 1):
            if symmetric:
                blk.values.write(v[0])
                blk.diagonal.write(v[1])
            else:
                blk.values.write(v)
            if step == 1 and name != "sjds":
                # stale by contract: the plan's own copy is the old one (the
                # sliced jagged form reads its long rows from the caller's
                # arrays: neither product until the call)
                blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), ref(v0)), name
            blk.values_changed()
            assert blk.get("values_changed_us") > 0
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(), ref(v)), (name, step)
        if name == "const_to_varying_and_back":
            assert blk.get("sdia") == 1 and blk.get("sdia_general") == 2
        for b in (dx, dy):
            b.free()
        blk.free()
    # a plan without a copy: the call does nothing, launches read the caller's
    # arrays as they are
    ctx.set_option("sj_min_nnz", 1 << 62)
    rp, ci, va = random_csr(rng, 700, 700, 9)
    blk = hip.CsrBlock(ctx, 700, 700, rp, ci, va, None, False)
    x = rng.uniform(-1, 1, 700)
    dx, dy = ctx.upload(x), ctx.upload(np.full(700, np.nan))
    blk.values.write(2.0 * va)
    blk.values_changed()
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, 2.0 * va, x))
    for b in (dx, dy):
        b.free()
    blk.free()
    ctx.close()


FEM_KINDS = {"fem": dict(), "fem_tail": dict(tail_permille=10),
             "fem81": dict(min_len=81, max_len=81),
             "fem_odd": dict(min_len=1, max_len=9, jitter=8, layer=50,
                             tail_permille=200, tail_min=30, tail_max=90,
                             tail_stride=3, seed=7)}


@pytest.mark.parametrize("kind", list(FEM_KINDS))
def test_fem_like_generator_matches_numpy_twin(ctx, kind):
    """spmv_hip_fem_count / spmv_hip_fem_fill_f64 (ragged rows, optional tail of
    very long rows, bandwidth-reducing order) against
    spmv_amd.poisson.fem_like_csr: same arrays; columns strictly ascending and
    the diagonal in every row."""
    from spmv_amd.host import FemParams
    for N in ((300, 4097) if kind == "fem_odd" else (40_000, 300_000)):
        kw = FEM_KINDS[kind]
        rp, ci, va = poisson.fem_like_csr(N, **kw)
        prm = FemParams(**poisson.fem_params(N, **kw))
        d_rp = ctx.empty(N + 1, np.int32)
        nnz = C.c_int64()
        hip.call("spmv_hip_fem_count", ctx.h, C.byref(prm), d_rp.ptr,
                 C.byref(nnz), None)
        assert nnz.value == len(ci) and np.array_equal(d_rp.numpy(), rp)
        d_ci, d_va = ctx.empty(nnz.value, np.int32), ctx.empty(nnz.value, np.float64)
        hip.call("spmv_hip_fem_fill_f64", ctx.h, C.byref(prm), nnz.value,
                 d_rp.ptr, d_ci.ptr, d_va.ptr, None)
        assert np.array_equal(d_ci.numpy(), ci)
        assert np.array_equal(d_va.numpy(), va)
        inner = np.ones(len(ci), bool)
        inner[rp[:-1][np.diff(rp) > 0]] = False   # first entry of each row
        assert np.all(np.diff(ci.astype(np.int64))[inner[1:]] > 0)
        rows = np.repeat(np.arange(N), np.diff(rp))
        assert np.array_equal(np.bincount(rows[ci == rows], minlength=N),
                              np.ones(N, np.int64))
        for b in (d_rp, d_ci, d_va):
            b.free()
    bad = FemParams(**poisson.fem_params(1000, max_len=40, jitter=8, layer=16))
    with pytest.raises(Exception):  # a cluster window narrower than its entries
        hip.call("spmv_hip_fem_count", ctx.h, C.byref(bad), 1, C.byref(nnz), None)


@pytest.mark.parametrize("kind", ["fem", "fem_tail"])
def test_fem_ten_million_rows_against_the_oracle_itself(kind):
    """The benchmark's ragged records at THEIR size, compared with the oracle
    (not kernel against kernel): the 10 M-row FEM-like matrix from the device
    generator (the numpy twin's arrays, test above; the twin itself takes three
    minutes at this size) copied to the host -- general storage against
    oracle.omp_spmv (csr_kernels.cpp:41-51: rows are summed left to right on any
    thread count), its symmetric storage (device-side lower split,
    Matrix.cpp:337-349) against the sequential oracle.csr_spmv_sym
    (csr_kernels.cpp:26-40) -- the sliced jagged form, the long rows' kernels,
    the merged symmetric form.  Every element identical."""
    from spmv_amd.host import FemParams
    avail = 0.0
    try:
        with open("/proc/meminfo") as f:
            avail = next(int(ln.split()[1]) for ln in f
                         if ln.startswith("MemAvailable")) / 2 ** 20
    except (OSError, StopIteration):
        pass
    if avail < 24:
        pytest.skip(f"MemAvailable is {avail:.0f} GB: the host copies of the 10 M-row "
                    "matrix (3.6 GB) and the oracle's vectors need 24 GB")
    ctx = hip.Context(0)
    N = 10_000_000
    kw = FEM_KINDS[kind]
    prm = FemParams(**poisson.fem_params(N, **kw))
    d_rp = ctx.empty(N + 1, np.int32)
    nnz = C.c_int64()
    hip.call("spmv_hip_fem_count", ctx.h, C.byref(prm), d_rp.ptr, C.byref(nnz), None)
    d_ci, d_va = ctx.empty(nnz.value, np.int32), ctx.empty(nnz.value, np.float64)
    hip.call("spmv_hip_fem_fill_f64", ctx.h, C.byref(prm), nnz.value, d_rp.ptr,
             d_ci.ptr, d_va.ptr, None)
    rp, ci, va = d_rp.numpy(), d_ci.numpy(), d_va.numpy()
    x = oracle.gaussian_x_fast(N) + 0.25
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    y_ref = oracle.omp_spmv(rp, ci, va, x, num_threads=threads)
    # general storage: the plan on the generator's own device arrays
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False)
    blk.bake()
    assert blk.get("sjds") == 1
    assert (blk.get("sj_long_rows") > 0) == (kind == "fem_tail")
    dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), y_ref), "general storage"
    blk.free()
    del y_ref
    # symmetric storage: lower part + diagonal, split on the device
    o_rp = ctx.empty(N + 1, np.int32)
    lnnz = C.c_int64()
    hip.call("spmv_hip_csr_lower_split_count", ctx.h, N, d_rp.ptr, d_ci.ptr, o_rp.ptr,
             C.byref(lnnz), None)
    o_ci, o_va = ctx.empty(lnnz.value, np.int32), ctx.empty(lnnz.value, np.float64)
    o_dg = ctx.empty(N, np.float64)
    hip.call("spmv_hip_csr_lower_split_fill_f64", ctx.h, N, d_rp.ptr, d_ci.ptr,
             d_va.ptr, o_rp.ptr, o_ci.ptr, o_va.ptr, o_dg.ptr, None)
    lrp, lci, lva, ldg = o_rp.numpy(), o_ci.numpy(), o_va.numpy(), o_dg.numpy()
    for b in (d_rp, d_ci, d_va, o_rp, o_ci, o_va, o_dg):
        b.free()
    del rp, ci, va
    ys_ref = oracle.csr_spmv_sym(lrp, lci, lva, ldg, x)
    blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, ldg, True)
    blk.bake()
    assert blk.get("sym_sj") == 1
    assert (blk.get("sj_long_rows") > 0) == (kind == "fem_tail")
    for alpha, beta in ((1.0, 0.0), (-0.5, 0.0)):
        dy.write(np.full(N, np.nan))
        blk.mult(alpha, dx.ptr, beta, dy.ptr)
        ref = ys_ref if alpha == 1.0 else oracle.csr_spmv_sym(lrp, lci, lva, ldg, x,
                                                              alpha)
        assert np.array_equal(dy.numpy(), ref), ("symmetric storage", alpha)
    blk.free()
    for b in (dx, dy):
        b.free()
    ctx.close()


def test_sliced_jagged_random_stress(sj_ctx):
    """Seeded random FEM-like matrices (row lengths, jitter, level-set width,
    share / length / stride of the long rows all drawn; SPMV_FUZZ_SEED,
    SPMV_FUZZ_TRIALS) through the default plans of both storages -- the slices
    with y handed over in LDS, the long rows' kernels, the merged symmetric form
    with its long rows, alpha / beta, the fused dot -- against the oracle."""
    ctx = sj_ctx
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", str(0x5A5A))))
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for trial in range(int(os.environ.get("SPMV_FUZZ_TRIALS", "6"))):
        nr = int(rng.choice([3000, 20_000, 70_001, 150_000]))
        jitter = int(rng.choice([32, 64, 256]))
        lo = int(rng.integers(1, 12))
        hi = int(min(2 * jitter, lo + rng.integers(0, 60)))
        layer = int(max(2 * jitter, rng.integers(2 * jitter, max(2 * jitter + 1, nr // 8))))
        kw = dict(min_len=lo, max_len=hi, jitter=jitter, layer=layer)
        stride = int(rng.choice([1, 3, 16]))
        tmax = int(min(nr // (2 * stride), rng.integers(150, 1500)))
        if rng.random() < 0.6 and tmax >= 120:
            kw.update(tail_permille=int(rng.choice([2, 10, 40])),
                      tail_min=int(min(tmax, max(100, tmax // 4))), tail_max=tmax,
                      tail_stride=stride)
        rp, ci, va = poisson.fem_like_csr(nr, **kw)
        x = rng.uniform(-1, 1, nr)
        y0 = rng.uniform(-1, 1, nr)
        dx = ctx.upload(x)
        lrp, lci, lva, dg = lower_split(rp, ci, va)
        for sym in (False, True):
            blk = (hip.CsrBlock(ctx, nr, nr, lrp, lci, lva, dg, True) if sym
                   else hip.CsrBlock(ctx, nr, nr, rp, ci, va, None, False))
            try:
                blk.bake()
            except Exception:  # (a long column over 5 %: the transposed map stays)
                assert sym
            for alpha, beta in ((1.0, 0.0), (-1.25, 0.5)):
                ref = (oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0) if sym
                       else oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0))
                dy = ctx.upload(np.full(nr, np.nan) if beta == 0 else y0)
                dot = beta == 0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                info = (trial, kw, sym, alpha, blk.get("sjds"), blk.get("sj_sigma"),
                        blk.get("sj_long_rows"))
                assert np.array_equal(dy.numpy(), ref), info
                if dot:
                    want = float(np.dot(x, ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-10 * (np.abs(x) @ np.abs(ref) + 1), info
                dy.free()
            blk.free()
        dx.free()
    part.free()


def test_lower_split_on_the_device_matches_the_host_rule(ctx):
    """spmv_hip_csr_lower_split_count / _fill_f64 (symmetric storage from a
    general block: entries below the diagonal kept in order, diagonal entries
    summed, the rest dropped -- spmv/Matrix.cpp:337-349) against the numpy
    restatement tests/util.py:lower_split: same arrays.  Ragged rows, repeated
    diagonal entries, unsorted columns, rows longer than a wave, empty rows."""
    rng = np.random.default_rng(0x10E5)
    cases = [poisson.fem_like_csr(5000, jitter=64, layer=300, tail_permille=20,
                                  tail_min=100, tail_max=400, tail_stride=2),
             random_csr(rng, 1500, 1500, 9, long_rows=3, long_len=700)]
    rp, ci, va = random_csr(rng, 800, 800, 12)
    rows = np.repeat(np.arange(800), np.diff(rp))
    ci = ci.copy()
    hit = rng.random(len(ci)) < 0.15   # repeated entries ON the diagonal
    ci[hit] = rows[hit]
    cases.append((rp, ci.astype(np.int32), va))
    for rp, ci, va in cases:
        n = len(rp) - 1
        lrp, lci, lva, ldg = lower_split(rp, ci, va)
        d_rp, d_ci, d_va = ctx.upload(rp, np.int32), ctx.upload(ci, np.int32), ctx.upload(va)
        o_rp = ctx.empty(n + 1, np.int32)
        nnz = C.c_int64()
        hip.call("spmv_hip_csr_lower_split_count", ctx.h, n, d_rp.ptr, d_ci.ptr,
                 o_rp.ptr, C.byref(nnz), None)
        assert nnz.value == len(lci) and np.array_equal(o_rp.numpy(), lrp)
        o_ci = ctx.empty(max(nnz.value, 1), np.int32)
        o_va, o_dg = ctx.empty(max(nnz.value, 1), np.float64), ctx.empty(n, np.float64)
        hip.call("spmv_hip_csr_lower_split_fill_f64", ctx.h, n, d_rp.ptr, d_ci.ptr,
                 d_va.ptr, o_rp.ptr, o_ci.ptr, o_va.ptr, o_dg.ptr, None)
        assert np.array_equal(o_ci.numpy()[:nnz.value], lci)
        assert np.array_equal(o_va.numpy()[:nnz.value], lva)
        assert np.array_equal(o_dg.numpy(), ldg)
        for b in (d_rp, d_ci, d_va, o_rp, o_ci, o_va, o_dg):
            b.free()
