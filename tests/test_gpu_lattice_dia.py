"""GPU parity tests of the lattice, diagonal, constant-diagonal and wide diagonal
forms (spmv_lat.hip, spmv_symlat.hip, spmv_symdia.hip, spmv_wdia.hip) and the
mixed-precision SpMV on them -- bit-exact against the CPU oracle (split from
test_gpu_kernels.py in round 6)."""
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
# Lattice form (spmv_lat.hip): constant column offsets per row block, values by
# LDS-DMA one row block ahead, no index stream.  Same bits as the oracle.
# ---------------------------------------------------------------------------
@pytest.fixture(params=["values", "const"])
def lat_ctx(request):
    """Every lattice / diagonal-form test runs twice: with the value-streaming
    kernels only ("values": ctx option const_diagonals = 0) and with the
    constant-diagonal kernels allowed ("const", the default) -- the Poisson
    cases then take them, the random-valued ones cannot."""
    c = hip.Context(0)
    c.set_option("lat_min_nnz", 0)  # try the form on small test matrices too
    c.set_option("lx_min_nnz", 0)
    c.set_option("const_diagonals", 1 if request.param == "const" else 0)
    c.const_mode = request.param == "const"
    yield c
    c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lattice_form_bit_exact(lat_ctx, dtype):
    ctx = lat_ctx
    rng = np.random.default_rng(91)
    cases = []
    for n in (4, 9, 16, 33):  # 64 rows (one partial block) ... 35,937 rows
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", rp, ci.astype(np.int32), va, n ** 3))
    rp, ci, va = oracle.tridiag_csr(70001)
    cases.append(("tridiag", rp, ci, va, 70001))
    # eight offsets, a third of the entries missing at random, empty rows
    cases.append(("eight", *_stencil_csr(rng, 9001, [-700, -33, -2, -1, 0, 1, 40, 900],
                                         drop=0.33), 9001))
    # odd entry count at the end of the array: the last 16-byte chunk of
    # `values` would end past it (element-wise path of the last row block)
    rp, ci, va = _stencil_csr(rng, 1025, [-1, 0, 5])
    assert len(va) % 2 == 1
    cases.append(("odd_tail", rp, ci, va, 1025))
    for name, rp, ci, va, N in cases:
        va = va.astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False,
                           hip.ALGO_ROWBLOCK, dtype)
        assert blk.get("lat") == 1, name
        assert blk.get("lat_blocks") == (N + 255) // 256
        assert blk.get("lx") == 0  # not built when the lattice form was taken
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(lat=1, nontemporal=1), dict(lat=1, nontemporal=0),
                          dict(lat_blocks_per_cu=1, lat_xcd_group=3),
                          dict(lat_blocks_per_cu=2, lat_xcd_group=16),
                          dict(lat=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0,
                                dtype)
                dot = dtype == np.float64 and alpha == 1.0 and beta == 0.0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                y = dy.numpy()
                dy.free()
                assert np.array_equal(y, y_ref), (name, alpha, beta, knobs)
                if dot:
                    want = float(np.dot(x.astype(np.float64), y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * max(abs(want), 1.0), (name, knobs)
            for k, v in dict(lat_blocks_per_cu=4, lat_xcd_group=0).items():
                blk.set(k, v)
        dx.free(), part.free()
        blk.free()


@pytest.mark.parametrize("n", [16, 32, 33])
def test_lattice_plane_walk_and_chain_bit_exact(lat_ctx, n):
    """The plane-walk order table (forced: small lattices never build one) and
    the plane chain of the general lattice kernel -- x of the plane ahead
    handed to the next step in registers (n = 16, 32: planes a whole number of
    row blocks apart) -- give the bits of the plain order; runs along the plane
    axis break the chain at their ends."""
    ctx = lat_ctx
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    N = n ** 3
    rng = np.random.default_rng(98)
    va = rng.uniform(-1, 1, len(va))  # not just -1 / 6
    x = rng.uniform(-1, 1, N)
    y0 = rng.uniform(-1, 1, N)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("lat") == 1 and blk.get("lattice_d2") == n * n
    assert blk.get("zwalk") == 0 and blk.get("lat_chain") == 1
    dx = ctx.upload(x)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    for alpha, beta in ((1.0, 0.0), (2.0, -0.5)):
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        for knobs in (dict(zwalk_segments=0), dict(zwalk_segments=1),
                      dict(lat_blocks_per_cu=1, zwalk_segments=3),
                      dict(lat_chain=0), dict(lat_chain=1, lat_blocks_per_cu=2),
                      dict(zwalk=0), dict(zwalk=1, lat_xcd_group=5)):
            for k, v in knobs.items():
                blk.set(k, v)
            dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr,
                     dot_partials=part.ptr if beta == 0 else None)
            assert np.array_equal(dy.numpy(), y_ref), (alpha, beta, knobs)
            if beta == 0:
                want = float(np.dot(x, y_ref))
                assert abs(float(np.sum(part.numpy())) - want) <= 1e-12 * max(abs(want), 1)
            dy.free()
        assert blk.get("zwalk_grid") > 0
    dx.free(), part.free()
    blk.free()


def test_lattice_form_is_refused_when_it_does_not_apply(lat_ctx):
    """Nine offsets, unsorted or repeated columns, scattered columns: the plan
    falls back (LX form or gather) and the results stay exact."""
    ctx = lat_ctx
    rng = np.random.default_rng(92)
    N = 6000
    cases = []
    cases.append(("nine", *_stencil_csr(rng, N, [-900, -40, -3, -2, -1, 0, 1, 2, 77])))
    rp, ci, va = _stencil_csr(rng, N, [-5, -1, 0, 1, 9])
    ci2 = ci.copy()  # swap the first two columns of one row: no longer ascending
    r = 3000
    ci2[rp[r]], ci2[rp[r] + 1] = ci[rp[r] + 1], ci[rp[r]]
    cases.append(("unsorted_row", rp, ci2, va))
    ci3 = ci.copy()  # a repeated column
    ci3[rp[r] + 1] = ci3[rp[r]]
    cases.append(("repeat", rp, ci3, va))
    cases.append(("random", *random_csr(rng, N, N, 6)))
    for name, rp, ci, va in cases:
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        assert blk.get("lat") == 0, name
        assert blk.get("lat_blocks") < (N + 255) // 256
        with pytest.raises(Exception):
            blk.set("lat", 1)
        x = rng.uniform(-1, 1, N)
        dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x)), name
        dx.free(), dy.free()
        blk.free()


def test_lattice_form_rectangular_and_empty_rows(lat_ctx):
    """Columns beyond the row count (a block with a ghost tail), rows without
    entries, a matrix of a single row."""
    ctx = lat_ctx
    rng = np.random.default_rng(93)
    N, ncols = 3000, 3500
    rows = np.arange(N)
    keep = rng.random(N) > 0.2
    rp = np.zeros(N + 1, np.int64)
    rp[1:] = np.cumsum(np.where(keep, 2, 0))
    ci = np.stack([rows[keep], rows[keep] + 500], 1).reshape(-1).astype(np.int32)
    va = rng.uniform(-1, 1, len(ci))
    x = rng.uniform(-1, 1, ncols)
    blk = hip.CsrBlock(ctx, N, ncols, rp.astype(np.int32), ci, va, None, False,
                       hip.ALGO_ROWBLOCK)
    assert blk.get("lat") == 1
    dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp.astype(np.int32), ci, va, x))
    dx.free(), dy.free()
    blk.free()
    one = hip.CsrBlock(ctx, 1, 1, np.array([0, 1], np.int32), np.array([0], np.int32),
                       np.array([3.0]), None, False, hip.ALGO_ROWBLOCK)
    assert one.get("lat") == 1
    dx, dy = ctx.upload(np.array([2.0])), ctx.upload(np.array([np.nan]))
    one.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert dy.numpy()[0] == 6.0
    dx.free(), dy.free()
    one.free()


# ---------------------------------------------------------------------------
# Symmetric lattice form (spmv_symlat.hip): symmetric storage with <= 3 constant
# lower offsets -- atomic-free, the reference's order, the reference's bits
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_symmetric_lattice_form_bit_exact(lat_ctx, dtype):
    ctx = lat_ctx
    rng = np.random.default_rng(95)
    cases = []
    for n in (4, 9, 16, 33):  # offsets merged into the own window / separate
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", *lower_split(rp, ci.astype(np.int32), va), n ** 3))
    rp, ci, va = oracle.tridiag_csr(70001)
    cases.append(("tridiag", *lower_split(rp, ci, va), 70001))
    # three far offsets with a third of the entries missing, ragged tail
    N = 9001
    rp, ci, va = _stencil_csr(rng, N, [-2000, -300, -1], drop=0.33)
    cases.append(("far3", rp, ci, va, rng.uniform(1, 2, N), N))
    for name, lrp, lci, lva, dg, N in cases:
        lva, dg = lva.astype(dtype), np.asarray(dg).astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True, hip.ALGO_AUTO, dtype)
        assert blk.get("slat") == 1, name
        assert blk.get("sym_det") == 0  # the transposed map was not needed
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0)
            for knobs in (dict(), dict(nontemporal=0), dict(slat_blocks_per_cu=1),
                          dict(lat_xcd_group=3)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0, dtype)
                dot = dtype == np.float64 and beta == 0.0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                y = dy.numpy()
                dy.free()
                assert np.array_equal(y, y_ref), (name, alpha, beta, knobs)
                if dot:
                    want = float(np.dot(x.astype(np.float64), y_ref))
                    got = float(np.sum(part.numpy()))
                    scale = float(np.abs(x) @ np.abs(y_ref)) + 1e-300
                    assert abs(got - want) <= 1e-12 * scale, (name, knobs)
            for k, v in dict(slat_blocks_per_cu=8, lat_xcd_group=0, nontemporal=1).items():
                blk.set(k, v)
        # the atomic kernels on the same plan (tolerance) -- the form can be
        # switched off
        blk.set("slat", 0)
        dy = ctx.upload(np.zeros(N, dtype), dtype)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x)
        tol = (2.0 ** -24 if dtype == np.float32 else U) * 64 * (np.abs(y_ref).max() + 12)
        assert np.all(np.abs(dy.numpy() - y_ref) <= tol), name
        dy.free(), dx.free(), part.free()
        blk.free()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_symmetric_diagonal_form_bit_exact(lat_ctx, dtype):
    """spmv_hip_csr_plan_bake_values_*: values re-laid out by offset.  Same bits
    as the reference for every offset geometry (merged / separate / misaligned
    windows), missing entries, ragged tails; a launch with other pointers, or
    with the form switched off, takes the CSR-order kernel; baking again picks
    up rewritten values."""
    ctx = lat_ctx
    rng = np.random.default_rng(97)
    cases = []
    # n = 16, 32: planes a whole number of row blocks apart -> the plane chain
    # (offset-0 windows and x handed from block to block) under the forced
    # plane-walk orders below; the others take the plain slots
    for n in (4, 9, 16, 32, 33):
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", *lower_split(rp, ci.astype(np.int32), va), n ** 3))
    rp, ci, va = oracle.tridiag_csr(70001)
    cases.append(("tridiag", *lower_split(rp, ci, va), 70001))
    N = 9001
    rp, ci, va = _stencil_csr(rng, N, [-2000, -300, -1], drop=0.33)
    cases.append(("far3", rp, ci, va, rng.uniform(1, 2, N), N))
    N = 7013  # odd far offsets (misaligned windows), one merged offset of 255
    rp, ci, va = _stencil_csr(rng, N, [-1001, -257, -255], drop=0.2)
    cases.append(("odd", rp, ci, va, rng.uniform(1, 2, N), N))
    N = 700  # two offsets, fewer rows than the far offset reaches
    rp, ci, va = _stencil_csr(rng, N, [-650, -3], drop=0.1)
    cases.append(("short", rp, ci, va, rng.uniform(1, 2, N), N))
    for name, lrp, lci, lva, dg, N in cases:
        lva, dg = lva.astype(dtype), np.asarray(dg).astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True, hip.ALGO_AUTO, dtype)
        assert blk.get("slat") == 1 and blk.get("sdia") == 0, name
        with pytest.raises(Exception):
            blk.set("sdia", 1)  # nothing baked yet
        kib0 = blk.get("plan_kib")
        blk.bake()
        assert blk.get("sdia") == 1, name
        assert blk.get("plan_kib") >= kib0
        if not ctx.const_mode:
            assert blk.get("sdia_const") == 0 and blk.get("plan_kib") > kib0
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0)
            # zwalk_segments forces the plane-walk table (built on its own
            # only for large lattices): a permutation of the row blocks
            for knobs in (dict(), dict(slat_blocks_per_cu=1), dict(lat_xcd_group=3),
                          dict(sdia=0), dict(sdia=1, slat_blocks_per_cu=8,
                                             lat_xcd_group=0),
                          dict(zwalk_segments=0), dict(zwalk_segments=1),
                          dict(slat_blocks_per_cu=2, zwalk_segments=3),
                          dict(sdia_chain=0, sdia_nt=31),
                          dict(sdia_chain=1, zwalk_segments=2, sdia_nt=0),
                          dict(zwalk=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0, dtype)
                dot = dtype == np.float64 and beta == 0.0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                y = dy.numpy()
                dy.free()
                assert np.array_equal(y, y_ref), (name, alpha, beta, knobs)
                if "zwalk_segments" in knobs:
                    assert blk.get("zwalk") == 1 and blk.get("zwalk_grid") > 0
                    if knobs["zwalk_segments"]:
                        assert blk.get("zwalk_segments") <= knobs["zwalk_segments"]
                if dot:
                    want = float(np.dot(x.astype(np.float64), y_ref))
                    got = float(np.sum(part.numpy()))
                    scale = float(np.abs(x) @ np.abs(y_ref)) + 1e-300
                    assert abs(got - want) <= 1e-12 * scale, (name, knobs)
        # the baked copy is the plan's own: new values in place are seen only
        # after baking again; other pointers never use it
        lva2 = (lva * dtype(1.5)).astype(dtype)
        dg2 = (dg + dtype(1)).astype(dtype)
        y_old = oracle.csr_spmv_sym(lrp, lci, lva, dg, x)
        y_new = oracle.csr_spmv_sym(lrp, lci, lva2, dg2, x)
        ctx.copy_h2d(blk.values.ptr, lva2)
        ctx.copy_h2d(blk.diagonal.ptr, dg2)
        dy = ctx.upload(np.full(N, np.nan, dtype), dtype)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), y_old), name   # stale by contract
        blk.bake()
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), y_new), name
        other = ctx.upload(lva, dtype)                   # another values array
        keep = blk.values
        blk.values = other
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(),
                              oracle.csr_spmv_sym(lrp, lci, lva, dg2, x)), name
        blk.values = keep
        blk.bake(drop=True)
        assert blk.get("sdia") == 0
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), y_new), name
        other.free(), dy.free(), dx.free(), part.free()
        blk.free()
    # not in the symmetric lattice form: nothing to bake
    rp, ci, va = _stencil_csr(rng, 5000, [-700, -30, -2, -1])
    blk = hip.CsrBlock(ctx, 5000, 5000, rp, ci, va, rng.uniform(1, 2, 5000), True)
    with pytest.raises(Exception):
        blk.bake()
    blk.free()


def _symmetric_general_csr(rng, N, lower_offsets, drop=0.0, diag_drop=0.0,
                           dtype=np.float64):
    """A general CSR matrix that is symmetric entry for entry: random lower
    entries at the given offsets (some dropped), their mirrors, a diagonal
    (some rows without)."""
    import scipy.sparse as sp
    lrp, lci, lva = _stencil_csr(rng, N, lower_offsets, drop=drop, dtype=dtype)
    L = sp.csr_matrix((lva, lci, lrp), shape=(N, N))
    keep = rng.random(N) >= diag_drop
    D = sp.csr_matrix((rng.uniform(1, 2, int(keep.sum())).astype(dtype),
                       (np.nonzero(keep)[0], np.nonzero(keep)[0])), shape=(N, N))
    A = (L + L.T + D).tocsr()
    A.sort_indices()
    assert A.nnz == 2 * L.nnz + D.nnz
    return (A.indptr.astype(np.int32), A.indices.astype(np.int32),
            A.data.astype(dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_general_matrix_found_symmetric_takes_the_diagonal_form(lat_ctx, dtype):
    """plan_bake_values on a GENERAL plan: the device check finds the matrix
    symmetric bit for bit, the plan keeps the lower half by offset, and the
    general SpMV comes out with the bits of csr_kernels.cpp:41-51 (rows summed
    in ascending column order) -- every geometry, rows without a diagonal,
    missing entries, both orders, the plane chain."""
    ctx = lat_ctx
    rng = np.random.default_rng(99)
    cases = []
    for n in (9, 16, 32, 33):
        rp, ci, va = poisson.poisson3d_csr(n)
        lrp, lci, lva, _ = lower_split(rp, ci.astype(np.int32), va)
        N = n ** 3
        cases.append((f"poisson{n}",
                      *_symmetric_general_csr(rng, N, [-n * n, -n, -1], dtype=dtype), N))
    cases.append(("poisson_exact", *[a for a in poisson.poisson3d_csr(20)], 8000))
    cases.append(("tridiag", *_symmetric_general_csr(rng, 70001, [-1], dtype=dtype), 70001))
    cases.append(("far3", *_symmetric_general_csr(rng, 9001, [-2000, -300, -1],
                                                  drop=0.33, diag_drop=0.2,
                                                  dtype=dtype), 9001))
    cases.append(("odd", *_symmetric_general_csr(rng, 7013, [-1001, -257, -255],
                                                 drop=0.2, dtype=dtype), 7013))
    for name, rp, ci, va, N in cases:
        ci, va = ci.astype(np.int32), va.astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK,
                           dtype)
        assert blk.get("lat") == 1 and blk.get("sdia") == 0, name
        blk.bake()
        assert blk.get("sdia") == 1, name
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(), dict(slat_blocks_per_cu=1), dict(sdia=0),
                          dict(sdia=1, zwalk_segments=0),
                          dict(slat_blocks_per_cu=2, zwalk_segments=3),
                          dict(sdia_chain=0), dict(sdia_chain=1, zwalk=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0, dtype)
                dot = dtype == np.float64 and alpha == 1.0 and beta == 0.0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                y = dy.numpy()
                dy.free()
                assert np.array_equal(y, y_ref), (name, alpha, beta, knobs)
                if dot:
                    want = float(np.dot(x.astype(np.float64), y_ref))
                    got = float(np.sum(part.numpy()))
                    scale = float(np.abs(x) @ np.abs(y_ref)) + 1e-300
                    assert abs(got - want) <= 1e-12 * scale, (name, knobs)
        # other values through the same plan: the CSR-order kernel, not the copy
        va2 = (va * dtype(0.5)).astype(dtype)
        other = ctx.upload(va2, dtype)
        keep = blk.values
        blk.values = other
        dy = ctx.upload(np.full(N, np.nan, dtype), dtype)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va2, x)), name
        blk.values = keep
        blk.bake(drop=True)
        assert blk.get("sdia") == 0
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x)), name
        other.free(), dy.free(), dx.free(), part.free()
        blk.free()


def test_general_matrix_that_is_not_symmetric_takes_the_full_diagonal_form(lat_ctx):
    """One value off by an ulp, a sign of zero, a missing mirror entry, or
    values that are simply not symmetric: the device check refuses the HALF
    form and the plan keeps ALL values by offset (full form, arrays for the
    upper entries too) -- the bits of the general reference loop, all orders,
    mixed-precision copy included.  A fourth offset or a rectangular block: not
    diagonal form at all, the lattice kernel keeps running."""
    ctx = lat_ctx
    rng = np.random.default_rng(100)
    N = 6000
    rp, ci, va = _symmetric_general_csr(rng, N, [-700, -30, -1])
    cases = []
    v2 = va.copy()
    j = int(rp[3000])  # first entry of a middle row: a lower one
    v2[j] = np.nextafter(v2[j], 2.0)
    cases.append(("ulp", rp, ci, v2, N))
    v3 = va.copy()
    r = 2000
    jl = int(rp[r])          # entry (r, r - 700) and its mirror
    c = int(ci[jl])
    jm = int(rp[c]) + int(np.nonzero(ci[rp[c]:rp[c + 1]] == r)[0][0])
    v3[jl], v3[jm] = 0.0, -0.0
    cases.append(("signed_zero", rp, ci, v3, N))
    keep = np.ones(len(ci), bool)  # drop one upper entry: pattern not symmetric
    keep[jm] = False
    rp4 = np.concatenate([[0], np.cumsum(np.bincount(
        np.repeat(np.arange(N), np.diff(rp))[keep], minlength=N))]).astype(np.int32)
    cases.append(("missing_mirror", rp4, ci[keep], va[keep], N))
    # plain non-symmetric stencils: 3-D lattice (chain), far / odd offsets with
    # drops, one-sided (upwind) pattern
    for n in (16, 33):
        prp, pci, _ = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", prp, pci.astype(np.int32),
                      rng.uniform(-1, 1, len(pci)), n ** 3))
    cases.append(("far3", *_stencil_csr(rng, 9001, [-2000, -300, -1, 0, 1, 300, 2000],
                                        drop=0.3), 9001))
    cases.append(("upwind", *_stencil_csr(rng, 7013, [-1001, -257, -1, 0, 1]), 7013))
    for name, rp_, ci_, va_, n_ in cases:
        ci_ = ci_.astype(np.int32)
        x = rng.uniform(-1, 1, n_)
        y0 = rng.uniform(-1, 1, n_)
        blk = hip.CsrBlock(ctx, n_, n_, rp_, ci_, va_, None, False,
                           hip.ALGO_ROWBLOCK)
        blk.bake()
        assert blk.get("sdia") == 1 and blk.get("sdia_general") == 2, name
        va32 = va_.astype(np.float32)
        d32 = ctx.upload(va32, np.float32)
        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan, d32.ptr,
                 None)
        dx = ctx.upload(x)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-1.5, 0.5)):
            y_ref = oracle.csr_spmv(rp_, ci_, va_, x, alpha, beta, y0)
            y32_ref = oracle.csr_spmv(rp_, ci_, va32.astype(np.float64), x, alpha,
                                      beta, y0)
            for knobs in (dict(), dict(zwalk_segments=0), dict(sdia_chain=0),
                          dict(sdia_chain=1, slat_blocks_per_cu=2, zwalk_segments=3),
                          dict(sdia=0), dict(sdia=1, zwalk=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(n_, np.nan) if beta == 0 else y0)
                dot = beta == 0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                assert np.array_equal(dy.numpy(), y_ref), (name, alpha, knobs)
                if dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                dy.free()
                dy = ctx.upload(np.full(n_, np.nan) if beta == 0 else y0)
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, n_, n_,
                         blk.nnz, blk.rowptr.ptr, blk.colind.ptr, d32.ptr,
                         float(alpha), dx.ptr, float(beta), dy.ptr, None, None)
                assert np.array_equal(dy.numpy(), y32_ref), (name, "mixed", knobs)
                dy.free()
        for b_ in (d32, dx, part):
            b_.free()
        blk.free()
    # four distinct offsets, and a rectangular block: not the diagonal form
    # proper -- the WIDE diagonal form (spmv_wdia.hip) takes them
    x = rng.uniform(-1, 1, N)
    rp_, ci_, va_ = _symmetric_general_csr(rng, N, [-700, -30, -2, -1])
    blk = hip.CsrBlock(ctx, N, N, rp_, ci_, va_, None, False, hip.ALGO_ROWBLOCK)
    blk.bake()
    assert blk.get("sdia") == 0 and blk.get("wdia") == 1
    assert blk.get("wdia_offsets") == 9
    dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp_, ci_, va_, x))
    dx.free(), dy.free()
    blk.free()
    rpr, cir, var = _stencil_csr(rng, 3000, [-5, 0, 5])
    xr = rng.uniform(-1, 1, 3005)
    blk = hip.CsrBlock(ctx, 3000, 3005, rpr, cir.astype(np.int32), var, None, False,
                       hip.ALGO_ROWBLOCK)
    blk.bake()
    assert blk.get("sdia") == 0 and blk.get("wdia") == 1
    dx, dy = ctx.upload(xr), ctx.upload(np.full(3000, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rpr, cir.astype(np.int32),
                                                      var, xr))
    dx.free(), dy.free()
    blk.free()


def test_symmetric_lattice_form_is_refused_when_it_does_not_apply(lat_ctx):
    """Four lower offsets, unsorted rows, an entry on or above the diagonal:
    the plan falls back to the transposed map (or, not strictly lower, to the
    atomic kernels) and stays correct."""
    ctx = lat_ctx
    rng = np.random.default_rng(96)
    N = 5000
    cases = [("four", *_stencil_csr(rng, N, [-700, -30, -2, -1]), 1)]
    rp, ci, va = _stencil_csr(rng, N, [-40, -3, -1])
    ci2 = ci.copy()
    r = 2500
    ci2[rp[r]], ci2[rp[r] + 1] = ci[rp[r] + 1], ci[rp[r]]
    cases.append(("unsorted_row", rp, ci2, va, 1))
    for name, lrp, lci, lva, det in cases:
        dg = rng.uniform(1, 2, N)
        blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
        assert blk.get("slat") == 0 and blk.get("sym_det") == det, name
        with pytest.raises(Exception):
            blk.set("slat", 1)
        x = rng.uniform(-1, 1, N)
        dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
        blk.mult(1.5, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(),
                              oracle.csr_spmv_sym(lrp, lci, lva, dg, x, 1.5)), name
        dx.free(), dy.free()
        blk.free()


@pytest.mark.parametrize("n", [16, 33])
def test_plane_walk_order_is_a_permutation_of_the_work(lat_ctx, n):
    """The plane-walk table only permutes row blocks (and adds empty slots):
    any number of runs along the plane axis, on any grid, gives the bits of the
    plain order -- general lattice form and the CSR-order symmetric lattice
    form (the diagonal form has its own test).  Small grids never build a
    table on their own, so it is forced."""
    ctx = lat_ctx
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    N = n ** 3
    x = oracle.gaussian_x_fast(N)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    for sym in (False, True):
        if sym:
            blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
            y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, 0.5, 0.0)
            assert blk.get("slat") == 1 and blk.get("sdia") == 0
        else:
            blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
            y_ref = oracle.csr_spmv(rp, ci, va, x, 0.5, 0.0)
            assert blk.get("lat") == 1
        assert blk.get("lattice_d1") == n and blk.get("lattice_d2") == n * n
        assert blk.get("zwalk") == 0  # too small to need it
        dx = ctx.upload(x)
        for segs in (0, 1, 2, 5, n, 3 * n):
            blk.set("zwalk_segments", segs)
            assert blk.get("zwalk") == 1
            assert 1 <= blk.get("zwalk_segments") <= max(segs, n)
            for bpc in (1, 4):
                blk.set("slat_blocks_per_cu" if sym else "lat_blocks_per_cu", bpc)
                assert blk.get("zwalk") == 1  # rebuilt for the new grid
                dy = ctx.upload(np.full(N, np.nan))
                blk.mult(0.5, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), y_ref), (sym, segs, bpc)
                dy.free()
        blk.set("zwalk", 0)
        assert blk.get("zwalk") == 0
        dx.free()
        blk.free()
    # no 3-D lattice in a general matrix: no planes to walk
    rp, ci, va = oracle.tridiag_csr(100000)
    blk = hip.CsrBlock(ctx, 100000, 100000, rp, ci, va, None, False,
                       hip.ALGO_ROWBLOCK)
    assert blk.get("lattice_d2") == 0
    with pytest.raises(Exception):
        blk.set("zwalk_segments", 4)
    blk.free()


# ---------------------------------------------------------------------------
# Constant diagonals (spmv_symdia.hip, csr_const_dia_kernel): the bake keeps the
# mask byte per row and ONE number per diagonal, no copy of the values
# ---------------------------------------------------------------------------
def _const_diag_csr(rng, N, offsets, consts, drop=0.0, dtype=np.float64):
    """_stencil_csr with the value of an entry fixed by its diagonal."""
    rp, ci, _ = _stencil_csr(rng, N, offsets, drop=drop)
    rows = np.repeat(np.arange(N), np.diff(rp))
    lut = dict(zip(offsets, consts))
    va = np.array([lut[int(d)] for d in (ci.astype(np.int64) - rows)], dtype)
    return rp, ci, va


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_constant_diagonals_bit_exact(dtype):
    """Every diagonal constant, bit for bit (the Poisson operator; any
    constant-coefficient stencil, symmetric or not, entries missing anywhere):
    the plan keeps no values, the kernel multiplies by the constant -- the same
    products and sums in the same order, so the same bits as the oracle for
    general and symmetric storage, any alpha / beta, fused dot, every order
    knob.  One entry off by an ulp: the value-streaming form, as before."""
    ctx = hip.Context(0)
    ctx.set_option("lat_min_nnz", 0)
    ctx.set_option("lx_min_nnz", 0)
    rng = np.random.default_rng(1207)
    third = 1.0 / 3.0  # not representable: fp32 constants differ from fp64 ones
    cases = []
    # (n = 28: a plane is 196 / 392 work items of the tile kernel -- it then
    # works in blocks of 196 so that planes stay whole blocks)
    for n in (4, 9, 16, 28, 32, 33):
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", rp, ci.astype(np.int32), va, n ** 3, True))
    cases.append(("tridiag", *_const_diag_csr(rng, 70001, [-1, 0, 1],
                                              [0.1, 0.8, 0.1]), 70001, True))
    offs = [-2000, -300, -1, 0, 1, 300, 2000]
    cases.append(("far3_sym", *_const_diag_csr(
        rng, 9001, offs, [-third, 0.7, -1.1, 5.3, -1.1, 0.7, -third], drop=0.0),
        9001, True))
    cases.append(("far3_skew_holes", *_const_diag_csr(
        rng, 9001, offs, [-1.25, 0.7, -1.1, 5.3, -0.9, 0.6, -0.75], drop=0.3),
        9001, False))
    cases.append(("odd", *_const_diag_csr(
        rng, 7013, [-1001, -257, -255, 0, 255, 257, 1001],
        [third, 2.0, -3.0, 9.0, 4.0, -5.0, 6.0], drop=0.2), 7013, False))
    cases.append(("short", *_const_diag_csr(
        rng, 700, [-650, -3, 3, 650], [1.5, -2.5, -2.5, 1.5], drop=0.1), 700, False))
    for name, rp, ci, va, N, symmetric in cases:
        va = va.astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        storages = [("general", rp, ci, va, None)]
        if symmetric:
            lrp, lci, lva, dg = lower_split(rp, ci, va)
            storages.append(("symmetric", lrp, lci, lva.astype(dtype),
                             np.asarray(dg).astype(dtype)))
        for sname, srp, sci, sva, sdg in storages:
            sym = sdg is not None
            for variant in ("const", "ulp"):
                v2 = sva.copy()
                if variant == "ulp":
                    j = len(v2) // 2
                    v2[j] = np.nextafter(v2[j], dtype(100.0))
                    if not sym and symmetric:
                        continue  # (would only break the symmetry as well)
                blk = hip.CsrBlock(ctx, N, N, srp, sci, v2, sdg, sym,
                                   hip.ALGO_AUTO if sym else hip.ALGO_ROWBLOCK, dtype)
                kib0 = blk.get("plan_kib")
                blk.bake()
                tag = (name, sname, variant)
                assert blk.get("sdia") == 1, tag
                assert blk.get("sdia_const") == (1 if variant == "const" else 0), tag
                if variant == "const":  # the mask and the walk table, nothing else
                    assert blk.get("plan_kib") - kib0 <= N // 1024 + 2 \
                        + 4 * blk.get("zwalk_grid") + 64, tag
                ref = ((lambda a, b: oracle.csr_spmv_sym(srp, sci, v2, sdg, x, a, b, y0))
                       if sym else
                       (lambda a, b: oracle.csr_spmv(srp, sci, v2, x, a, b, y0)))
                dx = ctx.upload(x, dtype)
                part = ctx.empty(ctx.dot_partials_len, np.float64)
                for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
                    y_ref = ref(alpha, beta)
                    for knobs in (dict(), dict(slat_blocks_per_cu=1),
                                  dict(lat_xcd_group=3), dict(sdia=0),
                                  dict(sdia=1, slat_blocks_per_cu=8, lat_xcd_group=0),
                                  dict(zwalk_segments=0), dict(zwalk_segments=1),
                                  dict(slat_blocks_per_cu=2, zwalk_segments=3),
                                  dict(sdia_chain=0, sdia_nt=31),
                                  dict(sdia_chain=1, zwalk_segments=2, sdia_nt=0),
                                  dict(zwalk=0), dict(zwalk=1, sdia_tile=1),
                                  dict(sdia_tile=2), dict(sdia_tile=2,
                                                          sdia_tile_segments=3),
                                  dict(sdia_tile=4, sdia_tile_segments=0),
                                  dict(sdia_tile_blocks_per_cu=1, sdia_chain=0),
                                  dict(sdia_tile_blocks_per_cu=8, sdia_chain=1,
                                       sdia_nt=16)):
                        tile_knobs = any(k.startswith("sdia_tile") for k in knobs)
                        if tile_knobs and not (variant == "const"
                                               and blk.get("sdia_offsets") == 3):
                            continue  # the tile kernel: constant 3-D lattices
                        for k, v in knobs.items():
                            blk.set(k, v)
                        if "sdia_tile" in knobs:
                            assert blk.get("sdia_tile") == (
                                knobs["sdia_tile"] if knobs["sdia_tile"] > 1 else 0)
                        dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0,
                                        dtype)
                        dot = dtype == np.float64 and beta == 0.0
                        blk.mult(alpha, dx.ptr, beta, dy.ptr,
                                 dot_partials=part.ptr if dot else None)
                        y = dy.numpy()
                        dy.free()
                        assert np.array_equal(y, y_ref), (tag, alpha, beta, knobs)
                        if dot:
                            want = float(np.dot(x.astype(np.float64), y_ref))
                            got = float(np.sum(part.numpy()))
                            scale = float(np.abs(x) @ np.abs(y_ref)) + 1e-300
                            assert abs(got - want) <= 1e-12 * scale, (tag, knobs)
                if variant == "const" and not sym and dtype == np.float64:
                    # mixed precision: the fp32 array has constants of its own
                    va32 = v2.astype(np.float32)
                    d32 = ctx.upload(va32, np.float32)
                    hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                             d32.ptr, None)
                    assert blk.get("sdia_mixed") == 1, tag
                    y32 = oracle.csr_spmv(srp, sci, va32.astype(np.float64), x, -0.5,
                                          0.75, y0)
                    dy = ctx.upload(y0)
                    hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                             blk.nnz, blk.rowptr.ptr, blk.colind.ptr, d32.ptr, -0.5,
                             dx.ptr, 0.75, dy.ptr, None, None)
                    assert np.array_equal(dy.numpy(), y32), tag
                    bad = va32.copy()  # not constant: refused, CSR-order kernels
                    bad[len(bad) // 3] = np.nextafter(bad[len(bad) // 3],
                                                      np.float32(100.0))
                    dbad = ctx.upload(bad, np.float32)
                    with pytest.raises(Exception):
                        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h,
                                 blk.plan, dbad.ptr, None)
                    assert blk.get("sdia_mixed") == 0
                    dy2 = ctx.upload(y0)
                    hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                             blk.nnz, blk.rowptr.ptr, blk.colind.ptr, dbad.ptr, -0.5,
                             dx.ptr, 0.75, dy2.ptr, None, None)
                    assert np.array_equal(dy2.numpy(), oracle.csr_spmv(
                        srp, sci, bad.astype(np.float64), x, -0.5, 0.75, y0)), tag
                    for b in (d32, dbad, dy, dy2):
                        b.free()
                if variant == "const":
                    # stale by contract until baked again; other pointers never
                    # use the constants; the constants can be dropped
                    v3 = (v2 * dtype(1.5)).astype(dtype)
                    ctx.copy_h2d(blk.values.ptr, v3)
                    dy = ctx.upload(np.full(N, np.nan, dtype), dtype)
                    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                    old = (oracle.csr_spmv_sym(srp, sci, v2, sdg, x) if sym
                           else oracle.csr_spmv(srp, sci, v2, x))
                    new = (oracle.csr_spmv_sym(srp, sci, v3, sdg, x) if sym
                           else oracle.csr_spmv(srp, sci, v3, x))
                    assert np.array_equal(dy.numpy(), old), tag
                    blk.bake()
                    assert blk.get("sdia_const") == 1, tag
                    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                    assert np.array_equal(dy.numpy(), new), tag
                    other = ctx.upload(v2, dtype)
                    keep = blk.values
                    blk.values = other
                    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                    assert np.array_equal(dy.numpy(), old), tag
                    blk.values = keep
                    blk.bake(drop=True)
                    assert blk.get("sdia") == 0 and blk.get("sdia_const") == 0
                    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                    assert np.array_equal(dy.numpy(), new), tag
                    other.free(), dy.free()
                dx.free(), part.free()
                blk.free()
    ctx.close()


# ---------------------------------------------------------------------------
# Wide diagonal form (spmv_wdia.hip): general matrices on <= 32 diagonals
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wide_diagonal_form_bit_exact(lat_ctx, dtype):
    """plan_bake_values on a general matrix too wide for the diagonal form
    proper: 27-point and 2-D 9-point stencils, 19 random offsets with a third
    of the entries dropped, 32 offsets (the limit).  Same bits as the oracle's
    general loop, any alpha / beta, fused dot; other value pointers take the
    CSR-order kernels; the copy can be dropped."""
    ctx = lat_ctx
    rng = np.random.default_rng(271)
    cases = []
    for n in (7, 12):
        rp, ci, va = poisson.stencil27_csr(n)
        cases.append((f"stencil27_{n}", rp, ci.astype(np.int32), va, n ** 3, 27))
    m = 70  # 2-D 9-point on a 70 x 70 grid
    offs9 = [dy * m + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    rp, ci, va = _stencil_csr(rng, m * m, offs9)
    cases.append(("nine_point_2d", rp, ci, va, m * m, 9))
    offs19 = sorted(int(o) for o in rng.choice(np.arange(-1500, 1500), 19,
                                               replace=False))
    rp, ci, va = _stencil_csr(rng, 9001, offs19, drop=0.3)
    cases.append(("nineteen_random", rp, ci, va, 9001, 19))
    offs32 = list(range(-16, 16))
    rp, ci, va = _stencil_csr(rng, 2000, offs32, drop=0.1)
    cases.append(("thirty_two", rp, ci, va, 2000, 32))
    for name, rp, ci, va, N, K in cases:
        va = rng.uniform(-1, 1, len(ci)).astype(dtype)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK,
                           dtype)
        assert blk.get("lat") == 0, name  # more than 8 offsets per row block
        try:
            blk.bake()
        except Exception as e:
            raise AssertionError(name) from e
        assert blk.get("wdia") == 1 and blk.get("sdia") == 0, name
        assert blk.get("wdia_offsets") == K, name
        dx = ctx.upload(x, dtype)
        other = ctx.upload(va, dtype)  # same values, another array
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(), dict(wdia_xcd_group=4), dict(wdia=0),
                          dict(wdia=1, wdia_xcd_group=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0,
                                dtype)
                dot = beta == 0 and dtype == np.float64
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                assert np.array_equal(dy.numpy(), y_ref), (name, alpha, beta, knobs)
                if dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                dy.free()
        if dtype == np.float64:
            # mixed precision: the fp32 copy by offset (plan_bake_values_f32f64)
            va32 = va.astype(np.float32)
            d32 = ctx.upload(va32, np.float32)
            hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                     d32.ptr, None)
            assert blk.get("wdia_mixed") == 1, name
            y32_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, -0.5,
                                      0.75, y0)
            for vals in (d32, ctx.upload(va32, np.float32)):  # baked / another
                dy = ctx.upload(y0)
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                         blk.nnz, blk.rowptr.ptr, blk.colind.ptr, vals.ptr, -0.5,
                         dx.ptr, 0.75, dy.ptr, None, None)
                assert np.array_equal(dy.numpy(), y32_ref), name
                dy.free()
                if vals is not d32:
                    vals.free()
            hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                     None, None)
            assert blk.get("wdia_mixed") == 0
            d32.free()
        # another value array of the same shape: the CSR-order kernels
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        dy = ctx.upload(np.full(N, np.nan, dtype), dtype)
        name_fn = ("spmv_hip_csr_spmv_f64" if dtype == np.float64
                   else "spmv_hip_csr_spmv_f32")
        args = [ctx.h, blk.plan, N, N, blk.nnz, blk.rowptr.ptr, blk.colind.ptr,
                other.ptr, None, 1.0, dx.ptr, 0.0, dy.ptr]
        hip.call(name_fn, *(args + ([None, None] if dtype == np.float64
                                    else [None])))
        assert np.array_equal(dy.numpy(), y_ref), name
        blk.bake(drop=True)
        assert blk.get("wdia") == 0
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), y_ref), name
        for b in (dx, dy, other, part):
            b.free()
        blk.free()


def test_wide_diagonal_half_form_bit_exact(lat_ctx):
    """A general matrix the bake finds symmetric entry for entry, bit for bit,
    keeps only its diagonals <= 0 (the upper entry (i, i+d) is read as the
    lower entry of row i+d): same bits as the oracle's general loop.  One value
    off by an ulp, or one entry without its mirror, and the full form is kept.
    The fp32 copy of the mixed SpMV must be symmetric itself."""
    import scipy.sparse as sp
    ctx = lat_ctx
    rng = np.random.default_rng(273)
    cases = []
    for n in (7, 12):
        rp, ci, va = poisson.stencil27_csr(n)
        cases.append((f"stencil27_{n}", rp, ci.astype(np.int32), va, n ** 3, 27))
    for name, N, offs, drop in (("nine", 4900, [0, 1, 69, 70, 71], 0.0),
                                ("ragged", 9001, [0, 3, 17, 256, 700, 1499], 0.3),
                                ("sixteen_upper", 3000, list(range(0, 16)), 0.1)):
        rp, ci, va = _stencil_csr(rng, N, offs, drop=drop)
        A = sp.csr_matrix((rng.uniform(-1, 1, len(ci)), ci, rp), shape=(N, N))
        S = (A + A.T).tocsr()
        S.sort_indices()
        cases.append((name, S.indptr.astype(np.int32), S.indices.astype(np.int32),
                      S.data.copy(), N, 2 * len(offs) - 1))
    for name, rp, ci, va, N, K in cases:
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        for variant in ("symmetric", "ulp", "hole", "option_off"):
            rp_v, ci_v, va_v = rp, ci, va.copy()
            if variant == "ulp":
                j = int(rp[N // 2]) + 1
                j = j if ci[j] != N // 2 else j + 1
                va_v[j] = np.nextafter(va_v[j], 2.0)
            if variant == "hole":  # drop one off-diagonal entry, keep its mirror
                i = N // 3
                j = int(rp[i])
                assert ci[j] != i
                keep = np.ones(len(ci), bool)
                keep[j] = False
                cnt = np.diff(rp)
                cnt[i] -= 1
                rp_v = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
                ci_v, va_v = ci[keep], va_v[keep]
            ctx.set_option("wdia_half", 0 if variant == "option_off" else 1)
            blk = hip.CsrBlock(ctx, N, N, rp_v, ci_v, va_v, None, False,
                               hip.ALGO_ROWBLOCK)
            try:
                blk.bake()
            finally:
                ctx.set_option("wdia_half", 1)
            assert blk.get("wdia") == 1, (name, variant)
            assert blk.get("wdia_offsets") == K, (name, variant)
            const = ctx.const_mode and name.startswith("stencil27") \
                and variant != "ulp"  # (26 / -1 on every diagonal)
            assert blk.get("wdia_const") == (1 if const else 0), (name, variant)
            assert blk.get("wdia_half") == (1 if variant == "symmetric"
                                            and not const else 0), (name, variant)
            dx = ctx.upload(x)
            part = ctx.empty(ctx.dot_partials_len, np.float64)
            for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
                y_ref = oracle.csr_spmv(rp_v, ci_v, va_v, x, alpha, beta, y0)
                for grp in (4, 0):
                    blk.set("wdia_xcd_group", grp)
                    dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                    blk.mult(alpha, dx.ptr, beta, dy.ptr,
                             dot_partials=part.ptr if beta == 0 else None)
                    assert np.array_equal(dy.numpy(), y_ref), (name, variant,
                                                               alpha, beta, grp)
                    if beta == 0:
                        want = float(np.dot(x, y_ref))
                        got = float(np.sum(part.numpy()))
                        assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                    dy.free()
            if variant == "symmetric":
                va32 = va_v.astype(np.float32)  # rounding keeps the symmetry
                d32 = ctx.upload(va32, np.float32)
                hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                         d32.ptr, None)
                assert blk.get("wdia_mixed") == 1, name
                y32_ref = oracle.csr_spmv(rp_v, ci_v, va32.astype(np.float64), x,
                                          -0.5, 0.75, y0)
                dy = ctx.upload(y0)
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                         blk.nnz, blk.rowptr.ptr, blk.colind.ptr, d32.ptr, -0.5,
                         dx.ptr, 0.75, dy.ptr, None, None)
                assert np.array_equal(dy.numpy(), y32_ref), name
                # an fp32 array that is not symmetric: refused, the CSR-order
                # mixed kernels run on it
                bad = va32.copy()
                j = int(rp_v[N // 2])
                j = j if ci_v[j] != N // 2 else j + 1
                bad[j] = np.nextafter(bad[j], np.float32(2.0))
                dbad = ctx.upload(bad, np.float32)
                with pytest.raises(Exception):
                    hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h,
                             blk.plan, dbad.ptr, None)
                ybad_ref = oracle.csr_spmv(rp_v, ci_v, bad.astype(np.float64), x,
                                           -0.5, 0.75, y0)
                dy2 = ctx.upload(y0)
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                         blk.nnz, blk.rowptr.ptr, blk.colind.ptr, dbad.ptr, -0.5,
                         dx.ptr, 0.75, dy2.ptr, None, None)
                assert np.array_equal(dy2.numpy(), ybad_ref), name
                for b in (dy, dy2, d32, dbad):
                    b.free()
            dx.free(), part.free()
            blk.free()


def _sym_box_csr(rng, N, P, L, drop):
    """Symmetric matrix (bit for bit) on the 27 offsets a P + b L + c of a box
    stencil, N rows (not necessarily whole planes), a share `drop` of the
    mirrored pairs missing."""
    import scipy.sparse as sp
    up = sorted(a * P + b * L + c for a in (0, 1) for b in (-1, 0, 1)
                for c in (-1, 0, 1) if a * P + b * L + c >= 0)
    assert len(up) == 14
    rp, ci, va = _stencil_csr(rng, N, up, drop=drop)
    A = sp.csr_matrix((va, ci, rp), shape=(N, N))
    S = (A + A.T).tocsr()
    S.sort_indices()
    return (S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_box27_half_marched_kernel_bit_exact(dtype):
    """27-point box stencils with VARYING coefficients, symmetric bit for bit:
    the half form's marched kernel (csr_box27_half_kernel: tiles of 1024 rows
    walked down the planes, plane values handed on through LDS, x from a ring of
    plane windows).  Same bits as the oracle for whole boxes, planes that are
    not whole tiles, row counts that are not whole planes, missing entries, runs
    of planes of every length, alpha / beta, the fused dot, the fp32 copy of the
    mixed SpMV -- and as the general wide diagonal kernel on the same plan."""
    ctx = hip.Context(0)
    ctx.set_option("lat_min_nnz", 0)
    ctx.set_option("lx_min_nnz", 0)
    ctx.set_option("const_diagonals", 0)
    rng = np.random.default_rng(0xB0C5)
    #        name            P      L    rows               drop
    shapes = [("box_32x32x9", 1024, 32, 1024 * 9, 0.0),
              ("box_40x30x10", 1200, 40, 1200 * 10, 0.0),   # tiles of 1024 + 176
              ("box_64x50x8", 3200, 64, 3200 * 8, 0.15),    # holes
              ("ragged_end", 2048, 100, 2048 * 9 + 777, 0.05),
              ("long_lines", 5080, 508, 5080 * 8 + 3, 0.0)]  # the longest lines the LDS holds
    for name, P, L, N, drop in shapes:
        rp, ci, va = _sym_box_csr(rng, N, P, L, drop)
        va = va.astype(dtype)  # (rounding keeps the symmetry)
        x = rng.uniform(-1, 1, N).astype(dtype)
        y0 = rng.uniform(-1, 1, N).astype(dtype)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK,
                           dtype)
        blk.bake()
        assert blk.get("wdia") == 1 and blk.get("wdia_offsets") == 27, name
        assert blk.get("wdia_half") == 1 and blk.get("wdia_const") == 0, name
        assert blk.get("wdia_hbox") == 1, name
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        planes = -(-N // P)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(), dict(wdia_hbox_segs=1), dict(wdia_hbox_segs=2),
                          dict(wdia_hbox_segs=3), dict(wdia_hbox_segs=planes),
                          dict(wdia_hbox=0), dict(wdia_hbox=1, wdia_hbox_segs=0)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan, dtype) if beta == 0 else y0,
                                dtype)
                use_dot = beta == 0 and dtype == np.float64
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if use_dot else None)
                y = dy.numpy()
                assert np.array_equal(y, y_ref), (
                    name, alpha, beta, knobs, int(np.sum(y != y_ref)),
                    np.flatnonzero(y != y_ref)[:8])
                if use_dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref)), (
                        name, knobs)
                dy.free()
        assert blk.get("wdia_hbox") == 1
        if dtype == np.float64:
            va32 = va.astype(np.float32)
            d32 = ctx.upload(va32, np.float32)
            hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                     d32.ptr, None)
            assert blk.get("wdia_mixed") == 1, name
            y32_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, -0.5, 0.75,
                                      y0)
            dy = ctx.upload(y0)
            hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N, blk.nnz,
                     blk.rowptr.ptr, blk.colind.ptr, d32.ptr, -0.5, dx.ptr, 0.75,
                     dy.ptr, None, None)
            assert np.array_equal(dy.numpy(), y32_ref), name
            dy.free(), d32.free()
        dx.free(), part.free()
        blk.free()
    # what the marched kernel does not take: fewer than 8 planes, planes smaller
    # than a tile, lines longer than 511 rows -- the general kernel keeps them
    for name, P, L, N in (("few_planes", 1024, 32, 1024 * 7),
                          ("small_planes", 900, 30, 900 * 12),
                          ("long_lines", 5632, 512, 5632 * 8)):
        rp, ci, va = _sym_box_csr(rng, N, P, L, 0.0)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va.astype(dtype), None, False,
                           hip.ALGO_ROWBLOCK, dtype)
        blk.bake()
        assert blk.get("wdia") == 1 and blk.get("wdia_half") == 1, name
        assert blk.get("wdia_hbox") == 0, name
        blk.free()
    ctx.close()


def test_wide_diagonal_form_constant_diagonals_bit_exact():
    """More than three lower offsets, every diagonal constant (HPCG's 27-point
    operator, a 2-D 9-point stencil, 19 offsets with a third of the entries
    missing): the plan keeps the 32-bit mask per row and one number per
    diagonal; same bits as the oracle, every knob; one value off by an ulp and
    the values are streamed as before."""
    ctx = hip.Context(0)
    ctx.set_option("lat_min_nnz", 0)
    ctx.set_option("lx_min_nnz", 0)
    rng = np.random.default_rng(2707)
    third = 1.0 / 3.0
    cases = []
    # (n = 16: planes of whole line tuples; n = 32: whole row blocks too, so
    # the box kernel hands its planes on from step to step)
    # ... n = 28: planes of 196 work items -- blocks of 196 instead of 256)
    for n in (7, 12, 16, 28, 32, 33):
        rp, ci, va = poisson.stencil27_csr(n)
        cases.append((f"stencil27_{n}", rp, ci.astype(np.int32), va, n ** 3, 27))
    box = [a * 400 + b * 20 + c for a in (-1, 0, 1) for b in (-1, 0, 1)
           for c in (-1, 0, 1)]
    cases.append(("box_holes", *_const_diag_csr(
        rng, 20 * 20 * 23, box, list(rng.uniform(-2, 2, 27)), drop=0.25),
        20 * 20 * 23, 27))
    m = 70
    offs9 = [dy * m + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    cases.append(("nine_point_2d", *_const_diag_csr(
        rng, m * m, offs9, [third * (k + 1) for k in range(9)]), m * m, 9))
    offs19 = sorted(int(o) for o in rng.choice(np.arange(-1500, 1500), 19,
                                               replace=False))
    cases.append(("nineteen_holes", *_const_diag_csr(
        rng, 9001, offs19, list(rng.uniform(-2, 2, 19)), drop=0.3), 9001, 19))
    for name, rp, ci, va, N, K in cases:
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        for variant in ("const", "ulp"):
            v2 = va.copy()
            if variant == "ulp":
                v2[len(v2) // 2] = np.nextafter(v2[len(v2) // 2], 100.0)
            blk = hip.CsrBlock(ctx, N, N, rp, ci, v2, None, False,
                               hip.ALGO_ROWBLOCK)
            kib0 = blk.get("plan_kib")
            blk.bake()
            tag = (name, variant)
            assert blk.get("wdia") == 1 and blk.get("wdia_offsets") == K, tag
            assert blk.get("wdia_const") == (1 if variant == "const" else 0), tag
            if variant == "const":
                assert blk.get("plan_kib") - kib0 <= 4 * N // 1024 + 2 + 64 \
                    + 4 * blk.get("zwalk_grid") + 64, tag
            # the 27-point boxes take the box kernel (4 lines per lane)
            is_box = variant == "const" and K == 27
            assert blk.get("wdia_box") == (4 if is_box else 0), tag
            dx = ctx.upload(x)
            part = ctx.empty(ctx.dot_partials_len, np.float64)
            for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
                y_ref = oracle.csr_spmv(rp, ci, v2, x, alpha, beta, y0)
                for knobs in (dict(), dict(wdia_xcd_group=0), dict(wdia=0),
                              dict(wdia=1, wdia_xcd_group=4),
                              dict(wdia_zwalk_segments=0),
                              dict(wdia_zwalk_segments=3, wdia_blocks_per_cu=2),
                              dict(wdia_zwalk=0), dict(wdia_zwalk=1),
                              dict(wdia_box=0), dict(wdia_box=2),
                              dict(wdia_box=2, wdia_box_segments=3),
                              dict(wdia_box=4, wdia_box_segments=0),
                              dict(wdia_box=4, wdia_box_blocks_per_cu=1,
                                   wdia_zwalk=0),
                              dict(wdia_box=4, wdia_box_blocks_per_cu=8,
                                   wdia_zwalk=1)):
                    if any(k.startswith("wdia_box") for k in knobs) and not is_box:
                        continue
                    for k, v in knobs.items():
                        blk.set(k, v)
                    dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                    blk.mult(alpha, dx.ptr, beta, dy.ptr,
                             dot_partials=part.ptr if beta == 0 else None)
                    assert np.array_equal(dy.numpy(), y_ref), (tag, alpha, beta, knobs)
                    if beta == 0:
                        want = float(np.dot(x, y_ref))
                        got = float(np.sum(part.numpy()))
                        assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                    dy.free()
            if variant == "const":
                va32 = v2.astype(np.float32)
                d32 = ctx.upload(va32, np.float32)
                hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                         d32.ptr, None)
                assert blk.get("wdia_mixed") == 1, tag
                dy = ctx.upload(y0)
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N, blk.nnz,
                         blk.rowptr.ptr, blk.colind.ptr, d32.ptr, -0.5, dx.ptr,
                         0.75, dy.ptr, None, None)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(
                    rp, ci, va32.astype(np.float64), x, -0.5, 0.75, y0)), tag
                bad = va32.copy()
                bad[len(bad) // 3] = np.nextafter(bad[len(bad) // 3],
                                                  np.float32(100.0))
                dbad = ctx.upload(bad, np.float32)
                with pytest.raises(Exception):
                    hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h,
                             blk.plan, dbad.ptr, None)
                assert blk.get("wdia_mixed") == 0
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N, blk.nnz,
                         blk.rowptr.ptr, blk.colind.ptr, dbad.ptr, 1.0, dx.ptr,
                         0.0, dy.ptr, None, None)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(
                    rp, ci, bad.astype(np.float64), x)), tag
                # stale by contract; another pointer; dropped
                v3 = v2 * 1.5
                ctx.copy_h2d(blk.values.ptr, v3)
                blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, v2, x)), tag
                blk.bake()
                assert blk.get("wdia_const") == 1
                blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, v3, x)), tag
                blk.bake(drop=True)
                assert blk.get("wdia") == 0 and blk.get("wdia_const") == 0
                blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
                assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, v3, x)), tag
                for b in (d32, dbad, dy):
                    b.free()
            dx.free(), part.free()
            blk.free()
    ctx.close()


def test_wide_diagonal_form_fuzz(lat_ctx):
    """Random offset sets (4-32 offsets anywhere up to the matrix size), sizes
    around the row-block boundaries, rectangular blocks, random drops, empty
    rows: whatever the wide diagonal form accepts it must compute bit-exactly;
    what it refuses stays on the CSR-order kernels, bit-exact too."""
    ctx = lat_ctx
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", "3303")))
    sizes = [1, 2, 255, 256, 257, 511, 513, 1000, 4097, 20000]
    taken = 0
    for case in range(int(os.environ.get("SPMV_FUZZ_TRIALS", "40"))):
        N = int(rng.choice(sizes))
        K = int(rng.integers(4, 33))
        span = max(2, int(rng.choice([8, 40, N // 3 + 2, N])))
        offs = sorted(set(int(o) for o in rng.integers(-span, span + 1, K)))
        drop = float(rng.choice([0.0, 0.1, 0.4]))
        rp, ci, va = _stencil_csr(rng, N, offs, drop=drop)
        ncols = N + int(rng.choice([0, 0, 7]))  # sometimes a few spare columns
        if len(ci) == 0:
            continue
        if rng.random() < 0.3:  # a stretch of empty rows
            lo = int(rng.integers(0, N))
            hi = min(N, lo + int(rng.integers(1, 300)))
            keep = np.ones(len(ci), bool)
            keep[rp[lo]:rp[hi]] = False
            cnt = np.diff(rp)
            cnt[lo:hi] = 0
            rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            ci, va = ci[keep], va[keep]
            if len(ci) == 0:
                continue
        x = rng.uniform(-1, 1, ncols)
        y0 = rng.uniform(-1, 1, N)
        alpha, beta = float(rng.choice([1.0, -0.5])), float(rng.choice([0.0, 0.75]))
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        blk = hip.CsrBlock(ctx, N, ncols, rp, ci, va, None, False,
                           hip.ALGO_ROWBLOCK)
        try:
            blk.bake()
        except Exception:
            pass
        taken += blk.get("wdia")
        dx = ctx.upload(x)
        dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
        blk.mult(alpha, dx.ptr, beta, dy.ptr)
        assert np.array_equal(dy.numpy(), y_ref), (case, N, offs, drop,
                                                   blk.get("wdia"),
                                                   blk.get("sdia"))
        dx.free(), dy.free()
        blk.free()
    assert taken >= 5  # the form is exercised, not just refused


def test_wide_diagonal_form_is_refused_when_it_does_not_apply(lat_ctx):
    """33 diagonals, a row with a repeated column, a row whose columns do not
    ascend, arrays that would be mostly zeros: ENOTSUP, the CSR-order kernels
    keep running."""
    ctx = lat_ctx
    rng = np.random.default_rng(272)
    N = 3000
    cases = [("33", *_stencil_csr(rng, N, list(range(-16, 17))))]
    rp, ci, va = _stencil_csr(rng, N, list(range(-6, 7)))
    ci2 = ci.copy()
    j = int(rp[1500])
    ci2[j + 1] = ci2[j]  # a repeated column
    cases.append(("repeat", rp, ci2, va))
    ci3 = ci.copy()
    ci3[j], ci3[j + 1] = ci3[j + 1], ci3[j]  # not ascending
    cases.append(("unsorted", rp, ci3, va))
    cases.append(("sparse", *_stencil_csr(rng, N, list(range(-6, 7)), drop=0.7)))
    for name, rp, ci, va in cases:
        x = rng.uniform(-1, 1, N)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        with pytest.raises(Exception):
            blk.bake()
        assert blk.get("wdia") == 0 and blk.get("sdia") == 0, name
        dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x)), name
        dx.free(), dy.free()
        blk.free()


# ---------------------------------------------------------------------------
# Mixed precision: fp32 values, fp64 vectors and arithmetic
# ---------------------------------------------------------------------------
def test_mixed_precision_spmv_bit_exact(lat_ctx):
    """spmv_hip_csr_spmv_f32f64 = the reference loop on the fp32-rounded
    values, in fp64: lattice form, plain row blocks (aligned and not), row
    list; with the fused dot."""
    ctx = lat_ctx
    rng = np.random.default_rng(97)
    cases = []
    for n in (9, 20):
        rp, ci, _ = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", rp, ci.astype(np.int32), n ** 3, n ** 3))
    rp, ci, _ = random_csr(rng, 3000, 3500, 7, long_rows=2, long_len=900)
    cases.append(("ragged", rp, ci, 3000, 3500))
    rp, ci, _ = random_csr(rng, 5000, 5000, 0.05)  # mostly empty: row list
    cases.append(("rowlist", rp, ci, 5000, 5000))
    for name, rp, ci, nrows, ncols in cases:
        va = rng.uniform(-1, 1, len(ci))
        va32 = va.astype(np.float32)
        x = rng.uniform(-1, 1, ncols)
        y0 = rng.uniform(-1, 1, nrows)
        blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va)
        if name.startswith("poisson"):
            assert blk.get("lat") == 1
        d32 = ctx.upload(va32, np.float32)
        dx = ctx.upload(x)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, alpha,
                                    beta, y0)
            for off in (0, 1):  # 1: a view 4 bytes into the array (unaligned)
                if off and name != "ragged":
                    continue
                vals = d32
                if off:
                    vals = ctx.upload(np.concatenate([[0], va32]).astype(np.float32),
                                      np.float32)
                dy = ctx.upload(np.full(nrows, np.nan) if beta == 0 else y0)
                dot = beta == 0 and nrows == ncols
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, nrows,
                         ncols, blk.nnz, blk.rowptr.ptr, blk.colind.ptr,
                         vals.ptr + 4 * off, float(alpha), dx.ptr, float(beta),
                         dy.ptr, part.ptr if dot else None, None)
                y = dy.numpy()
                assert np.array_equal(y, y_ref), (name, alpha, beta, off)
                if dot:
                    want = float(np.dot(x[:nrows], y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x[:nrows]) @ np.abs(y_ref))
                dy.free()
                if off:
                    vals.free()
        for b in (d32, dx, part):
            b.free()
        blk.free()


def test_mixed_precision_vector_and_scalar_plans(ctx):
    """A general plan with long rows takes the VECTOR kernel (AUTO above 64
    entries per row); spmv_f32f64 must run on it -- and on SCALAR -- instead of
    returning ENOTSUP (CgOptions::mixed on e.g. a 3-D elasticity matrix)."""
    rng = np.random.default_rng(131)
    nrows, ncols = 2000, 2300
    rp, ci, _ = random_csr(rng, nrows, ncols, 90)
    va = rng.uniform(-1, 1, len(ci))
    va32 = va.astype(np.float32)
    x = rng.uniform(-1, 1, ncols)
    y0 = rng.uniform(-1, 1, nrows)
    for algo in (hip.ALGO_AUTO, hip.ALGO_VECTOR, hip.ALGO_SCALAR):
        blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va, None, False, algo)
        if algo == hip.ALGO_AUTO:
            assert blk.algo == hip.ALGO_VECTOR
        d32, dx = ctx.upload(va32, np.float32), ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, alpha,
                                    beta, y0)
            dy = ctx.upload(np.full(nrows, np.nan) if beta == 0 else y0)
            hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, nrows, ncols,
                     blk.nnz, blk.rowptr.ptr, blk.colind.ptr, d32.ptr,
                     float(alpha), dx.ptr, float(beta), dy.ptr, None, None)
            y = dy.numpy()
            if blk.algo == hip.ALGO_SCALAR:
                assert np.array_equal(y, y_ref), (algo, alpha, beta)
            else:  # another summation order
                bound = (16 + np.diff(rp)) * U * abs_bound(
                    rp, ci, va32.astype(np.float64), x, alpha, beta, y0)
                assert np.all(np.abs(y - y_ref) <= bound + 1e-300), (algo, alpha)
            dy.free()
        for b in (d32, dx):
            b.free()
        blk.free()


def test_bake_that_does_not_apply_leaves_the_plan_as_it_was(lat_ctx):
    """plan_bake_values on a plan that cannot take the diagonal form returns
    ENOTSUP and changes NOTHING -- in particular the plane-walk table that plan
    creation built for the CSR-order lattice kernel stays (it was dropped
    once); after a successful bake is dropped again, the table is the lattice
    kernel's again."""
    ctx = lat_ctx
    rng = np.random.default_rng(7)
    n = 40  # planes of 1600 rows; a table for so small a lattice needs forcing
    N = n ** 3
    # 8 offsets, 4 of them lower: lattice form yes, diagonal form no (> 3);
    # three fifths of the entries dropped: too sparse for the wide diagonal
    # form as well (it wants half of its slots filled)
    offs = [-n * n, -n, -2, -1, 0, 1, n, n * n]
    rp, ci, va = _stencil_csr(rng, N, offs, drop=0.6)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("lat") == 1
    blk.set("zwalk_segments", 2)  # force a table
    before = (blk.get("zwalk"), blk.get("zwalk_grid"), blk.get("zwalk_segments"))
    assert before[0] == 1 and before[1] > 0
    with pytest.raises(Exception):
        blk.bake()
    assert blk.get("sdia") == 0 and blk.get("wdia") == 0
    assert (blk.get("zwalk"), blk.get("zwalk_grid"),
            blk.get("zwalk_segments")) == before
    dx, dy = ctx.upload(x), ctx.upload(np.full(N, np.nan))
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), y_ref)
    blk.free()
    # a matrix that CAN be baked: bake, then drop -> the lattice kernel's table
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    blk.set("zwalk_segments", 2)
    lat_grid = blk.get("zwalk_grid")
    blk.bake()
    assert blk.get("sdia") == 1
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), y_ref)
    blk.bake(drop=True)
    assert blk.get("sdia") == 0
    # the restored table is the unforced choice for this small lattice (none)
    # or the lattice kernel's -- never the diagonal form's
    assert blk.get("zwalk_grid") in (0, lat_grid)
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), y_ref)
    for b in (dx, dy):
        b.free()
    blk.free()


def test_mixed_precision_on_the_diagonal_form_bit_exact(lat_ctx):
    """plan_bake_values_f32f64: the fp32 copy by offset of a general matrix
    found symmetric (fp64 copy baked first).  spmv_f32f64 with the baked fp32
    pointer = the general reference loop on the fp32 values in fp64, bit for
    bit; other pointers take the lattice kernel."""
    ctx = lat_ctx
    rng = np.random.default_rng(101)
    for name, N, offs, kw in (("poisson16", 16 ** 3, [-256, -16, -1], {}),
                              ("poisson33", 33 ** 3, [-1089, -33, -1], {}),
                              ("far3", 9001, [-2000, -300, -1],
                               dict(drop=0.3, diag_drop=0.2))):
        rp, ci, va = _symmetric_general_csr(rng, N, offs, **kw)
        va32 = va.astype(np.float32)
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        d32 = ctx.upload(va32, np.float32)
        with pytest.raises(Exception):  # the fp64 copy comes first
            hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                     d32.ptr, None)
        blk.bake()
        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan, d32.ptr,
                 None)
        assert blk.get("sdia_mixed") == 1, name
        dx = ctx.upload(x)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        other = ctx.upload(va32, np.float32)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va32.astype(np.float64), x, alpha,
                                    beta, y0)
            for vals, knobs in ((d32, dict()), (d32, dict(zwalk_segments=2)),
                                (d32, dict(sdia_chain=0)), (other, dict(sdia_chain=1))):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                dot = beta == 0
                hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N,
                         blk.nnz, blk.rowptr.ptr, blk.colind.ptr, vals.ptr,
                         float(alpha), dx.ptr, float(beta), dy.ptr,
                         part.ptr if dot else None, None)
                assert np.array_equal(dy.numpy(), y_ref), (name, alpha, beta, knobs)
                if dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                dy.free()
        # the fp64 SpMV of the same plan is untouched
        dy = ctx.upload(np.full(N, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x)), name
        hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan, None, None)
        assert blk.get("sdia_mixed") == 0
        # an fp32 array that is NOT symmetric cannot ride on the half form
        bad = va32.copy()
        bad[int(rp[N // 2])] *= np.float32(1.5)
        dbad = ctx.upload(bad, np.float32)
        with pytest.raises(Exception):
            hip.call("spmv_hip_csr_plan_bake_values_f32f64", ctx.h, blk.plan,
                     dbad.ptr, None)
        assert blk.get("sdia_mixed") == 0
        hip.call("spmv_hip_csr_spmv_f32f64", ctx.h, blk.plan, N, N, blk.nnz,
                 blk.rowptr.ptr, blk.colind.ptr, dbad.ptr, 1.0, dx.ptr, 0.0,
                 dy.ptr, None, None)
        assert np.array_equal(dy.numpy(), oracle.csr_spmv(
            rp, ci, bad.astype(np.float64), x)), name
        dbad.free()
        for b in (d32, dx, part, other, dy):
            b.free()
        blk.free()


def test_diagonal_form_fuzz(lat_ctx):
    """Random symmetric lattice matrices -- sizes around the row-block
    boundaries, 1-3 offsets anywhere between 1 and the matrix size (merged,
    separate, chained and misaligned windows), random drops, rows without a
    diagonal -- through BOTH storages of the diagonal form, forced plane-walk
    tables included: bit-exact against the oracle's general / symmetric loops."""
    ctx = lat_ctx
    # SPMV_FUZZ_SEED / SPMV_FUZZ_TRIALS: other seeds, longer runs (by hand)
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", "2026")))
    trials = int(os.environ.get("SPMV_FUZZ_TRIALS", "60"))
    sizes = [1, 2, 255, 256, 257, 511, 512, 513, 1000, 4096, 5000, 20000, 65536, 70001]
    done = 0
    for trial in range(trials):
        N = int(sizes[trial % len(sizes)] if trial < 28 else rng.integers(300, 60000))
        nd = int(rng.integers(1, 4))
        pool = [1, 2, 3, 63, 64, 65, 255, 256, 257, 512, 768, 1024, 2048, 4096]
        pool += [int(v) for v in rng.integers(1, max(2, N), 6)]
        offs = sorted({int(o) for o in rng.choice(pool, nd) if o < N}, reverse=True)
        if not offs:
            continue
        drop = float(rng.choice([0.0, 0.0, 0.3]))
        ddrop = float(rng.choice([0.0, 0.25]))
        rp, ci, va = _symmetric_general_csr(rng, N, [-o for o in offs], drop=drop,
                                            diag_drop=ddrop)
        if len(va) == 0:
            continue
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        alpha, beta = (1.0, 0.0) if trial % 2 else (-1.5, 0.5)
        # general storage: lattice form + device symmetry check
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        assert blk.get("lat") == 1, (trial, N, offs)
        # the offsets that really occur (drops and short matrices lose some)
        have = sorted(set(np.abs(ci - np.repeat(np.arange(N), np.diff(rp)))) - {0})
        if not have:  # diagonal only: nothing for the diagonal form to do
            with pytest.raises(Exception):
                blk.bake()
            blk.free()
            continue
        try:
            blk.bake()
        except Exception as e:
            raise AssertionError((trial, N, offs, have, drop, ddrop, str(e)))
        assert blk.get("sdia") == 1 and blk.get("sdia_offsets") == len(have), (
            trial, N, offs, have)
        dx = ctx.upload(x)
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        for knobs in (dict(), dict(zwalk_segments=int(rng.integers(0, 4))),
                      dict(slat_blocks_per_cu=int(rng.integers(1, 5)))):
            for k, v in knobs.items():
                blk.set(k, v)
            dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr)
            assert np.array_equal(dy.numpy(), y_ref), (trial, N, offs, drop, knobs)
            dy.free()
        blk.free()
        # the same pattern with values that are NOT symmetric: the full form
        va_ns = rng.uniform(-1, 1, len(va))
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va_ns, None, False, hip.ALGO_ROWBLOCK)
        blk.bake()
        offdiag = bool((ci != np.repeat(np.arange(N), np.diff(rp))).any())
        assert blk.get("sdia_general") == (2 if offdiag else 1), (trial, N, offs)
        y_ns = oracle.csr_spmv(rp, ci, va_ns, x, alpha, beta, y0)
        for knobs in (dict(), dict(zwalk_segments=int(rng.integers(0, 4)))):
            for k, v in knobs.items():
                blk.set(k, v)
            dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr)
            assert np.array_equal(dy.numpy(), y_ns), (trial, N, offs, "full", knobs)
            dy.free()
        blk.free()
        # symmetric storage of the same matrix (needs a full diagonal)
        if ddrop == 0.0:
            lrp, lci, lva, dg = lower_split(rp, ci, va)
            if len(lva):
                sb = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
                if sb.get("slat") == 1:
                    sb.bake()
                    ys = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0)
                    for knobs in (dict(), dict(zwalk_segments=int(rng.integers(0, 4)))):
                        for k, v in knobs.items():
                            sb.set(k, v)
                        dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                        sb.mult(alpha, dx.ptr, beta, dy.ptr)
                        assert np.array_equal(dy.numpy(), ys), (trial, N, offs, knobs)
                        dy.free()
                sb.free()
        dx.free()
        done += 1
    assert done >= 0.75 * trials


def test_constant_diagonals_fuzz():
    """Random offset sets (1-3 lower offsets anywhere between 1 and the matrix
    size: lines shorter than a wave, lines longer than the matrix, planes that
    are no whole number of lines), random holes, random constants, symmetric
    and skewed, sizes around the row-block boundaries -- through the
    constant-diagonal kernels with 1, 2 and 4 lines per lane, plain and forced
    plane-walk orders, both storages: bit-exact against the oracle's loops."""
    ctx = hip.Context(0)
    ctx.set_option("lat_min_nnz", 0)
    ctx.set_option("lx_min_nnz", 0)
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", "4711")))
    trials = int(os.environ.get("SPMV_FUZZ_TRIALS", "60"))
    sizes = [1, 2, 255, 256, 257, 511, 512, 513, 1000, 4096, 5000, 20000, 65536, 70001]
    done = tiled = 0
    for trial in range(trials):
        N = int(sizes[trial % len(sizes)] if trial < 28 else rng.integers(300, 60000))
        nd = int(rng.integers(1, 4)) if trial % 3 else 3
        pool = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 512, 768, 1024, 2048, 4096]
        pool += [int(v) for v in rng.integers(1, max(2, N), 6)]
        lows = sorted({int(o) for o in rng.choice(pool, nd) if o < N}, reverse=True)
        if not lows:
            continue
        symmetric = bool(trial % 2)
        offs = [-u for u in lows] + [0] + [u for u in reversed(lows)]
        cl = list(rng.uniform(-2, 2, len(lows)))
        cu = list(reversed(cl)) if symmetric else list(rng.uniform(-2, 2, len(lows)))
        consts = cl + [float(rng.uniform(3, 9))] + cu
        drop = float(rng.choice([0.0, 0.0, 0.3]))
        rp, ci, va = _const_diag_csr(rng, N, offs, consts, drop=drop)
        if len(va) == 0:
            continue
        rows = np.repeat(np.arange(N), np.diff(rp))
        have = sorted(set(np.abs(ci - rows)) - {0})
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        alpha, beta = (1.0, 0.0) if trial % 4 < 2 else (-1.5, 0.5)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        if not have:
            blk.free()
            continue
        blk.bake()
        tag = (trial, N, lows, drop, symmetric)
        assert blk.get("sdia") == 1 and blk.get("sdia_const") == 1, tag
        three = blk.get("sdia_offsets") == 3
        dx = ctx.upload(x)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        knob_sets = [dict(), dict(zwalk_segments=int(rng.integers(0, 4))),
                     dict(slat_blocks_per_cu=int(rng.integers(1, 5)))]
        if three:
            tiled += 1
            knob_sets += [dict(sdia_tile=1), dict(sdia_tile=2),
                          dict(sdia_tile=4, sdia_tile_blocks_per_cu=int(
                              rng.integers(1, 9))),
                          dict(sdia_tile=int(rng.choice([2, 4])), sdia_chain=int(
                              rng.integers(0, 2)))]
        for knobs in knob_sets:
            for k, v in knobs.items():
                blk.set(k, v)
            if three and "sdia_tile" in knobs and knobs["sdia_tile"] > 1 \
                    and rng.random() < 0.5:
                blk.set("sdia_tile_segments", int(rng.integers(0, 4)))
            dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr,
                     dot_partials=part.ptr if beta == 0 else None)
            assert np.array_equal(dy.numpy(), y_ref), (tag, knobs)
            if beta == 0:
                want = float(np.dot(x, y_ref))
                got = float(np.sum(part.numpy()))
                assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref) + 1e-300)
            dy.free()
        blk.free()
        # symmetric storage of the same matrix (needs the whole diagonal and
        # equal constants above and below)
        if symmetric and drop == 0.0:
            lrp, lci, lva, dg = lower_split(rp, ci, va)
            if len(lva):
                sb = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
                if sb.get("slat") == 1:
                    sb.bake()
                    assert sb.get("sdia_const") == 1, tag
                    ys = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0)
                    ks = [dict(), dict(zwalk_segments=int(rng.integers(0, 4)))]
                    if sb.get("sdia_offsets") == 3:
                        ks += [dict(sdia_tile=1), dict(sdia_tile=2), dict(sdia_tile=4)]
                    for knobs in ks:
                        for k, v in knobs.items():
                            sb.set(k, v)
                        dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                        sb.mult(alpha, dx.ptr, beta, dy.ptr)
                        assert np.array_equal(dy.numpy(), ys), (tag, "sym", knobs)
                        dy.free()
                sb.free()
        dx.free(), part.free()
        done += 1
    assert done >= 0.7 * trials and tiled >= 10
    ctx.close()
