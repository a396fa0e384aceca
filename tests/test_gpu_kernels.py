"""GPU parity tests of the HIP kernels (the basic general and symmetric
kernels, fp32, gather / dot, the device generators, the put window, the CG
kernels; the other kernel families: test_gpu_sjds.py, test_gpu_lx_xw.py,
test_gpu_lattice_dia.py), called through the C ABI
(include/spmv_hip.h) and checked against the CPU oracle.

Bars (BASELINE.json north_star "within a stated fp64
tolerance"):
  * general CSR, ROWBLOCK and SCALAR kernels: BIT-EXACT against
    oracle.csr_spmv (= spmv/csr_kernels.cpp:41-51): same left-to-right order,
    no FMA contraction.
  * symmetric storage, default (transposed map / symmetric lattice form):
    BIT-EXACT against oracle.csr_spmv_sym (= spmv/csr_kernels.cpp:26-40).
  * VECTOR kernel and the atomic symmetric kernels (plan_set sym_det = 0):
    element-wise |y - y_ref|_i <= 16 u (|alpha||A||x| + |beta||y0|)_i,
    u = 2^-53.
  * the reference's own check: ||y||_2 agrees with the KAT norm to 1 ulp
    relative (tests/test_spmv.cpp:20-23,159-160) for the exact kernels and to
    1e-14 relative for the others.
"""
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
# KAT (tests/test_spmv.cpp:56-80)
# ---------------------------------------------------------------------------
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        k = json.load(f)
    return (np.array(k["rowptr"], np.int32), np.array(k["colind"], np.int32),
            np.array(k["values"], np.float64), np.array(k["x"]),
            np.array(k["y"]), k["norm_y"])


@pytest.mark.parametrize("algo", EXACT_ALGOS)
def test_kat_general_exact(ctx, algo):
    rp, ci, va, x, y_ref, norm_ref = kat()
    y = run_spmv(ctx, rp, ci, va, x, 5, 5, algo=algo)
    assert np.array_equal(y, y_ref)
    norm = float(np.sqrt(np.sum(y * y)))
    assert abs(norm - norm_ref) <= min(abs(norm), abs(norm_ref)) * np.finfo(float).eps


def test_kat_vector_and_symmetric(ctx):
    rp, ci, va, x, y_ref, norm_ref = kat()
    bound = 16 * U * abs_bound(rp, ci, va, x)
    y = run_spmv(ctx, rp, ci, va, x, 5, 5, algo=hip.ALGO_VECTOR)
    assert np.all(np.abs(y - y_ref) <= bound)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    ys = run_spmv(ctx, lrp, lci, lva, x, 5, 5, diagonal=dg, symmetric=True)
    assert np.array_equal(ys, y_ref)  # reference: general == symmetric, bit for bit
    ys = run_spmv(ctx, lrp, lci, lva, x, 5, 5, diagonal=dg, symmetric=True,
                  knobs=dict(sym_det=0))
    assert np.all(np.abs(ys - y_ref) <= bound)
    assert abs(np.linalg.norm(ys) - norm_ref) <= 1e-14 * norm_ref


# ---------------------------------------------------------------------------
# Poisson + random ragged matrices, general kernels
# ---------------------------------------------------------------------------
KNOBS = [dict(), dict(chunks=1), dict(chunks=4), dict(nontemporal=0),
         dict(xcd_group=1), dict(xcd_group=16),
         dict(chunks=4, xcd_group=3, blocks_per_cu=2)]


@pytest.mark.parametrize("n", [4, 9, 16, 33])
@pytest.mark.parametrize("knobs", KNOBS)
def test_poisson_general_exact(ctx, n, knobs):
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    N = n ** 3
    x = oracle.gaussian_x_fast(N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    y = run_spmv(ctx, rp, ci, va, x, N, N, algo=hip.ALGO_ROWBLOCK, knobs=knobs)
    assert np.array_equal(y, y_ref)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("algo", EXACT_ALGOS + [hip.ALGO_VECTOR])
def test_random_ragged(ctx, seed, algo):
    rng = np.random.default_rng(0x5EED0001 + seed)
    nrows = int(rng.integers(1, 3000))
    ncols = int(rng.integers(1, 4000))
    avg = [0.5, 3, 7, 20, 70, 300][seed]
    rp, ci, va = random_csr(rng, nrows, ncols, avg, long_rows=seed % 3,
                            long_len=5000)
    x = rng.uniform(-1, 1, ncols)
    y0 = rng.uniform(-1, 1, nrows)
    for alpha, beta in [(1.0, 0.0), (-0.75, 0.0), (1.0, 1.0), (2.5, -0.5)]:
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        y = run_spmv(ctx, rp, ci, va, x, nrows, ncols, alpha, beta,
                     None if beta == 0 else y0, algo=algo)
        if algo in EXACT_ALGOS:
            assert np.array_equal(y, y_ref), (alpha, beta)
        else:
            # a different summation order: (len_i + 16) u per row
            bound = (16 + np.diff(rp)) * U * abs_bound(rp, ci, va, x, alpha,
                                                       beta, y0)
            assert np.all(np.abs(y - y_ref) <= bound + 1e-300)


@pytest.mark.parametrize("lpr", [4, 8, 16, 32, 64])
def test_vector_lanes_per_row(ctx, lpr):
    rng = np.random.default_rng(lpr)
    rp, ci, va = random_csr(rng, 777, 555, 40)
    x = rng.uniform(-1, 1, 555)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    y = run_spmv(ctx, rp, ci, va, x, 777, 555, algo=hip.ALGO_VECTOR,
                 knobs=dict(lanes_per_row=lpr))
    bound = (16 + np.diff(rp)) * U * abs_bound(rp, ci, va, x)
    assert np.all(np.abs(y - y_ref) <= bound + 1e-300)


def test_empty_and_degenerate(ctx):
    # all rows empty, nnz == 0, rowptr NULL (csr_matrix.cpp:34)
    y0 = np.arange(7, dtype=np.float64)
    y = run_spmv(ctx, None, None, None, np.ones(3), 7, 3, 2.0, 0.5, y0)
    assert np.array_equal(y, 0.5 * y0)
    y = run_spmv(ctx, None, None, None, np.ones(3), 7, 3, 2.0, 0.0)
    assert np.array_equal(y, np.zeros(7))
    # single entry, single row
    y = run_spmv(ctx, [0, 1], [0], [3.0], np.array([2.0]), 1, 1)
    assert np.array_equal(y, [6.0])
    # nnz not a multiple of the vector width, rows straddling tile edges
    rng = np.random.default_rng(7)
    for nnz_row in (1, 3, 5):
        nrows = 1031
        rp = (np.arange(nrows + 1) * nnz_row).astype(np.int32)
        ci = rng.integers(0, 50, nrows * nnz_row).astype(np.int32)
        va = rng.uniform(-1, 1, nrows * nnz_row)
        x = rng.uniform(-1, 1, 50)
        assert np.array_equal(run_spmv(ctx, rp, ci, va, x, nrows, 50,
                                       algo=hip.ALGO_ROWBLOCK),
                              oracle.csr_spmv(rp, ci, va, x))


def test_plan_mismatch_is_rejected(ctx):
    rp, ci, va = poisson.poisson3d_csr(4)
    blk = hip.CsrBlock(ctx, 64, 64, rp, ci.astype(np.int32), va)
    dx, dy = ctx.zeros(64, np.float64), ctx.zeros(64, np.float64)
    rc = hip._lib.hip.spmv_hip_csr_spmv_f64(
        ctx.h, blk.plan, 63, 64, blk.nnz, blk.rowptr.ptr, blk.colind.ptr,
        blk.values.ptr, None, 1.0, dx.ptr, 0.0, dy.ptr, None, None)
    assert rc == -1  # SPMV_HIP_EINVAL, nothing launched
    # a plain (gather) plan reads colind at run time: other index arrays of the
    # same shape are fine
    other = ctx.upload(ci.astype(np.int32), np.int32)
    rc = hip._lib.hip.spmv_hip_csr_spmv_f64(
        ctx.h, blk.plan, 64, 64, blk.nnz, blk.rowptr.ptr, other.ptr,
        blk.values.ptr, None, 1.0, dx.ptr, 0.0, dy.ptr, None, None)
    assert rc == 0
    blk.free()
    # a plan that baked the structure in (here: the lattice form) must get the
    # very arrays it analysed -- it would silently use the old structure
    c2 = hip.Context(0)
    c2.set_option("lat_min_nnz", 0)
    blk = hip.CsrBlock(c2, 64, 64, rp, ci.astype(np.int32), va)
    assert blk.get("lat") == 1
    other2 = c2.upload(ci.astype(np.int32), np.int32)
    dx2, dy2 = c2.zeros(64, np.float64), c2.zeros(64, np.float64)
    rc = hip._lib.hip.spmv_hip_csr_spmv_f64(
        c2.h, blk.plan, 64, 64, blk.nnz, blk.rowptr.ptr, other2.ptr,
        blk.values.ptr, None, 1.0, dx2.ptr, 0.0, dy2.ptr, None, None)
    assert rc == -1
    rc = hip._lib.hip.spmv_hip_csr_spmv_f64(
        c2.h, blk.plan, 64, 64, blk.nnz, blk.rowptr.ptr, blk.colind.ptr,
        blk.values.ptr, None, 1.0, dx2.ptr, 0.0, dy2.ptr, None, None)
    assert rc == 0
    for b in (dx, dy, other, dx2, dy2, other2):
        b.free()
    blk.free()
    c2.close()


def test_unaligned_views(ctx):
    """colind/values that are not 16-byte aligned take the element-wise path
    and stay exact."""
    rng = np.random.default_rng(11)
    rp, ci, va = random_csr(rng, 900, 900, 6)
    x = rng.uniform(-1, 1, 900)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    nnz = len(va)
    dci = ctx.upload(np.concatenate([[0], ci]).astype(np.int32))
    dva = ctx.upload(np.concatenate([[0.0], va]))
    drp = ctx.upload(rp)
    dx, dy = ctx.upload(x), ctx.zeros(900, np.float64)
    plan = C.c_void_p()
    hip.call("spmv_hip_csr_plan_create", ctx.h, 900, 900, nnz, drp.ptr,
             dci.at(1), 0, hip.ALGO_ROWBLOCK, C.byref(plan))
    hip.call("spmv_hip_csr_spmv_f64", ctx.h, plan, 900, 900, nnz, drp.ptr,
             dci.at(1), dva.at(1), None, 1.0, dx.ptr, 0.0, dy.ptr, None, None)
    assert np.array_equal(dy.numpy(), y_ref)
    hip.call("spmv_hip_csr_plan_destroy", plan)
    for b in (dci, dva, drp, dx, dy):
        b.free()


# ---------------------------------------------------------------------------
# symmetric kernel
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [4, 9, 16, 33])
def test_poisson_symmetric(ctx, n):
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    N = n ** 3
    x = oracle.gaussian_x_fast(N)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x)
    bound = 16 * U * abs_bound(rp, ci, va, x)
    for alpha, beta in [(1.0, 0.0), (0.5, 2.0), (1.0, 1.0)]:
        y0 = np.cos(np.arange(N))
        y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, alpha, beta, y0)
        b = 16 * U * abs_bound(rp, ci, va, x, alpha, beta, y0)
        # the default: atomic-free, the reference's order, the reference's bits
        y = run_spmv(ctx, lrp, lci, lva, x, N, N, alpha, beta,
                     None if beta == 0 else y0, diagonal=dg, symmetric=True)
        assert np.array_equal(y, y_ref), (alpha, beta)
        # atomic kernels: plain per-entry atomics (0) and LDS-window variants
        for window, srows in ((0, 1024), (256, 512), (1024, 1024),
                              (4096, 2048)):
            y = run_spmv(ctx, lrp, lci, lva, x, N, N, alpha, beta,
                         None if beta == 0 else y0, diagonal=dg, symmetric=True,
                         knobs=dict(sym_det=0, sym_window=window, sym_rows=srows))
            assert np.all(np.abs(y - y_ref) <= b), (window, srows)
    # and against the general kernel on the full matrix
    y_gen = oracle.csr_spmv(rp, ci, va, x)
    y = run_spmv(ctx, lrp, lci, lva, x, N, N, diagonal=dg, symmetric=True)
    assert np.all(np.abs(y - y_gen) <= bound)


def test_symmetric_random_and_diag_only(ctx):
    rng = np.random.default_rng(3)
    n = 1500
    rp, ci, va = random_csr(rng, n, n, 9, long_rows=2, long_len=1400)
    rows = np.repeat(np.arange(n), np.diff(rp))
    keep = ci < rows
    lrp = np.zeros(n + 1, np.int64)
    np.add.at(lrp, rows[keep] + 1, 1)
    lrp = np.cumsum(lrp).astype(np.int32)
    lci, lva = ci[keep], va[keep]
    dg = rng.uniform(1, 2, n)
    x = rng.uniform(-1, 1, n)
    y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x, 1.5, 0.0)
    full = np.zeros(n)
    np.add.at(full, rows[keep], np.abs(lva * x[lci]))
    np.add.at(full, lci, np.abs(lva * x[rows[keep]]))
    full += np.abs(dg * x)
    terms = np.diff(lrp) + np.bincount(lci, minlength=n) + 1
    y = run_spmv(ctx, lrp, lci, lva, x, n, n, 1.5, 0.0, diagonal=dg,
                 symmetric=True)
    assert np.array_equal(y, y_ref)  # ragged rows, repeated columns, long rows
    y0 = rng.uniform(-1, 1, n)
    y = run_spmv(ctx, lrp, lci, lva, x, n, n, -0.75, 0.5, y0, diagonal=dg,
                 symmetric=True)
    assert np.array_equal(y, oracle.csr_spmv_sym(lrp, lci, lva, dg, x, -0.75, 0.5, y0))
    for window in (0, 256, 1024):  # targets both inside and below the window
        y = run_spmv(ctx, lrp, lci, lva, x, n, n, 1.5, 0.0, diagonal=dg,
                     symmetric=True, knobs=dict(sym_det=0, sym_window=window))
        assert np.all(np.abs(y - y_ref) <= (16 + terms) * U * 1.5 * full), window
    # diagonal-only symmetric block (nnz == 0, diagonal != NULL)
    y = run_spmv(ctx, None, None, None, x, n, n, 2.0, 0.0, diagonal=dg,
                 symmetric=True)
    assert np.array_equal(y, 2.0 * (dg * x))


# ---------------------------------------------------------------------------
# fp32 instantiations (device_executor.h:88-99)
# ---------------------------------------------------------------------------
def test_fp32(ctx):
    rng = np.random.default_rng(5)
    rp, ci, va = random_csr(rng, 2000, 1800, 8, dtype=np.float32)
    x = rng.uniform(-1, 1, 1800).astype(np.float32)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    for algo in EXACT_ALGOS:
        y = run_spmv(ctx, rp, ci, va, x, 2000, 1800, algo=algo,
                     dtype=np.float32)
        assert np.array_equal(y, y_ref)
    y = run_spmv(ctx, rp, ci, va, x, 2000, 1800, algo=hip.ALGO_VECTOR,
                 dtype=np.float32)
    assert np.allclose(y, y_ref, rtol=0, atol=64 * 2.0 ** -24 * 8)
    n = 16
    rp, ci, va = poisson.poisson3d_csr(n, dtype=np.float32)
    ci = ci.astype(np.int32)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    x = oracle.gaussian_x_fast(n ** 3).astype(np.float32)
    y_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x)
    y = run_spmv(ctx, lrp, lci, lva, x, n ** 3, n ** 3, diagonal=dg,
                 symmetric=True, dtype=np.float32)
    assert np.array_equal(y, y_ref)
    y = run_spmv(ctx, lrp, lci, lva, x, n ** 3, n ** 3, diagonal=dg,
                 symmetric=True, dtype=np.float32, knobs=dict(sym_det=0))
    assert np.allclose(y, y_ref, rtol=0, atol=16 * 2.0 ** -24 * 12)


# ---------------------------------------------------------------------------
# gather, dot, fills
# ---------------------------------------------------------------------------
def test_gather(ctx):
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, 10000)
    for n in (0, 1, 255, 256, 257, 5000):
        idx = rng.integers(0, 10000, n).astype(np.int32)
        dx, di = ctx.upload(x), ctx.upload(idx)
        do = ctx.empty(max(n, 1), np.float64)
        ctx.gather(di, dx, do, n)
        if n:
            assert np.array_equal(do.numpy(n), oracle.gather_ghosts(idx, x))
        for b in (dx, di, do):
            b.free()


def test_dot_and_fill(ctx):
    rng = np.random.default_rng(13)
    for n in (1, 2, 1001, 1 << 20):
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        dx, dy = ctx.upload(x), ctx.upload(y)
        d = ctx.dot(n, dx.ptr, dy.ptr)
        ref = oracle.ddot(x, y)
        assert abs(d - ref) <= 8 * U * n ** 0.5 * np.sum(np.abs(x * y)) + 1e-300
        # deterministic: same bits on a second run
        assert d == ctx.dot(n, dx.ptr, dy.ptr)
        dx.free(), dy.free()
    N = 4096
    g = ctx.empty(N, np.float64)
    ctx.fill_gaussian(N, 0, N, g.ptr)
    ref = oracle.gaussian_x_fast(N)
    assert np.allclose(g.numpy(), ref, rtol=4 * 2.0 ** -52, atol=1e-300)
    ctx.fill_const(N, 1.25, g.ptr)
    assert np.array_equal(g.numpy(), np.full(N, 1.25))
    g.free()


# ---------------------------------------------------------------------------
# device Poisson generator vs the host generator
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,P", [(4, 1), (6, 1), (8, 2), (8, 4), (9, 3), (16, 8)])
def test_device_poisson_matches_host(ctx, n, P):
    """The device generator writes exactly the blocks create_matrix builds
    from the host CSR (oracle.create_matrix = spmv/Matrix.cpp:295-480)."""
    N = n ** 3
    ranges = poisson.owner_ranges(P, N)
    grp, gci, gva = poisson.poisson3d_csr(n)
    for r in range(P):
        r0, r1 = int(ranges[r]), int(ranges[r + 1])
        lrp, lci, lva, ghosts = oracle.localise_rows(grp, gci, gva, r0, r1)
        nloc = r1 - r0
        single = oracle.create_matrix(r, ranges, ranges, lrp, lci, lva, ghosts,
                                      False, oracle.P2P_BLOCKING)
        split = oracle.create_matrix(r, ranges, ranges, lrp, lci, lva, ghosts,
                                     False, oracle.P2P_NONBLOCKING)
        sym = oracle.create_matrix(r, ranges, ranges, lrp, lci, lva, ghosts,
                                   True, oracle.P2P_BLOCKING)
        expect = {hip.PART_ALL: single["local"], hip.PART_LOCAL: split["local"],
                  hip.PART_REMOTE: split["remote"],
                  hip.PART_LOCAL_LOWER: sym["local"]}
        for part, (erp, eci, eva) in expect.items():
            blk = hip.poisson3d_block(ctx, n, r0, r1, part,
                                      with_diagonal=(part == hip.PART_LOCAL_LOWER))
            assert blk.ghosts_below + blk.ghosts_above == len(ghosts)
            assert blk.nnz == len(eva)
            if blk.nnz:  # an empty block owns no arrays (csr_matrix.cpp:34)
                assert np.array_equal(blk.rowptr.numpy(), erp)
                assert np.array_equal(blk.colind.numpy(), eci)
                assert np.array_equal(blk.values.numpy(), eva)
            if blk.diagonal is not None:
                assert np.array_equal(blk.diagonal.numpy(), sym["diagonal"])
            blk.free()
        assert np.array_equal(sym["remote"][1], split["remote"][1])
    # the generator's non-symmetric variant (ctx option "poisson_skew_ppm"):
    # lower neighbours -1 - s, upper -1 + s; takes the FULL diagonal form
    ctx.set_option("poisson_skew_ppm", 250000)
    blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
    ctx.set_option("poisson_skew_ppm", 0)
    rows = np.repeat(np.arange(N), np.diff(grp))
    want = np.where(gci == rows, 6.0, np.where(gci < rows, -1.25, -0.75))
    assert np.array_equal(blk.colind.numpy(), gci)
    assert np.array_equal(blk.values.numpy(), want)
    blk.free()


@pytest.mark.parametrize("n,parts", [(6, (2, 2, 2)), (7, (3, 2, 1)),
                                     (9, (1, 2, 4)), (5, (1, 1, 1)),
                                     (8, (4, 1, 2))])
def test_device_box_generator_matches_host_rows(ctx, n, parts):
    """spmv_hip_poisson3d_box_* (the blocks of Matrix::create_poisson3d_boxes,
    generated on the device) against the host generator
    Matrix::poisson3d_box_rows for every rank of the partition: same ghost
    count, same rows (as sets of (column, value): the host lists a row by
    GLOBAL column, the device in create_matrix's local order), columns
    ascending within a row, and the LOCAL / REMOTE / LOCAL_LOWER parts the
    matching subsets."""
    from spmv_amd import host

    def axis_part(n, p, i):
        q, rem = divmod(n, p)
        return i * q + min(i, rem), q + (1 if i < rem else 0)

    px, py, pz = parts
    for rank in range(px * py * pz):
        rp_h, ci_h, va_h, ghosts, _, box = host.poisson3d_box_rows(n, parts, rank)
        idx = (rank % px, (rank // px) % py, rank // (px * py))
        fl = [axis_part(n, p, i) for p, i in zip(parts, idx)]
        first = np.array([f for f, _ in fl], np.int32)
        length = np.array([l for _, l in fl], np.int32)
        assert tuple(length) == tuple(box)
        nloc = int(np.prod(length))
        fp, lp = first.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p)
        ng = C.c_int64()
        hip.call("spmv_hip_poisson3d_box_count", ctx.h, n, fp, lp, hip.PART_ALL,
                 None, None, C.byref(ng), None)
        assert ng.value == len(ghosts)
        got = {}
        for part in (hip.PART_ALL, hip.PART_LOCAL, hip.PART_REMOTE,
                     hip.PART_LOCAL_LOWER):
            d_rp = ctx.empty(nloc + 1, np.int32)
            nnz = C.c_int64()
            hip.call("spmv_hip_poisson3d_box_count", ctx.h, n, fp, lp, part,
                     d_rp.ptr, C.byref(nnz), None, None)
            d_ci = ctx.empty(max(nnz.value, 1), np.int32)
            d_va = ctx.empty(max(nnz.value, 1), np.float64)
            d_dg = ctx.empty(nloc, np.float64)
            hip.call("spmv_hip_poisson3d_box_fill_f64", ctx.h, n, fp, lp, part,
                     d_rp.ptr, d_ci.ptr, d_va.ptr, d_dg.ptr, None)
            got[part] = (d_rp.numpy(), d_ci.numpy()[:nnz.value],
                         d_va.numpy()[:nnz.value])
            assert np.all(d_dg.numpy() == 6.0)
            for b in (d_rp, d_ci, d_va, d_dg):
                b.free()
        rp, ci, va = got[hip.PART_ALL]
        assert np.array_equal(rp, rp_h)
        for i in range(nloc):
            dev = list(zip(ci[rp[i]:rp[i + 1]], va[rp[i]:rp[i + 1]]))
            assert [c for c, _ in dev] == sorted(c for c, _ in dev), (rank, i)
            hst = sorted(zip(ci_h[rp_h[i]:rp_h[i + 1]], va_h[rp_h[i]:rp_h[i + 1]]))
            assert dev == hst, (rank, i)
            for part, pred in ((hip.PART_LOCAL, lambda c: c < nloc),
                               (hip.PART_REMOTE, lambda c: c >= nloc),
                               (hip.PART_LOCAL_LOWER, lambda c: c < i)):
                prp, pci, pva = got[part]
                sub = list(zip(pci[prp[i]:prp[i + 1]], pva[prp[i]:prp[i + 1]]))
                assert sub == [(c, v) for c, v in dev if pred(c)], (rank, i, part)


def _put_window(ctx, stage_bytes):
    put = C.c_void_p()
    handle = (C.c_ubyte * 64)()
    raw, pid = C.c_uint64(), C.c_int64()
    hip.call("spmv_hip_put_create", ctx.h, stage_bytes, C.byref(put), handle,
             C.byref(raw), C.byref(pid))
    return put, handle, raw.value, pid.value


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_put_exchange_with_itself(ctx, dtype):
    """The one-sided halo's kernel at unit level (include/spmv_hip.h,
    spmv_hip_put_*): a window connected to ITSELF in two slots -- slot 0 sends
    into slot 1's segment and the other way round -- goes through the whole
    protocol (free flags, stores, data flags, staging -> ghost tail) for several
    epochs; every epoch delivers that epoch's data."""
    n0, n1 = 1000, 37  # elements of the two segments
    esz = np.dtype(dtype).itemsize
    put, handle, raw, pid = _put_window(ctx, 8 * (n0 + n1))
    # slot k: send segment k of the send buffer to where the OTHER slot reads
    hip.call("spmv_hip_put_connect", put, 0, handle, raw, pid, 8 * (n0 + n1),
             n0, 1, 0, n1, 0, n0, 1)     # my n1 items -> segment [n0, n0 + n1)
    hip.call("spmv_hip_put_connect", put, 1, handle, raw, pid, 8 * (n0 + n1),
             0, 0, n1, n0, n0, n1, 1)    # my n0 items -> segment [0, n0)
    hip.call("spmv_hip_put_finish", put)
    rng = np.random.default_rng(5)
    ghost = ctx.upload(np.full(n0 + n1, np.nan, dtype), dtype)
    for epoch in range(4):
        send = rng.uniform(-1, 1, n0 + n1).astype(dtype)
        d_send = ctx.upload(send, dtype)
        hip.call("spmv_hip_put_exchange", ctx.h, put, esz, d_send.ptr, ghost.ptr,
                 None)
        ctx.stream_sync()
        failed = C.c_int()
        hip.call("spmv_hip_put_status", put, C.byref(failed))
        assert failed.value == 0
        got = ghost.numpy()
        # slot 0 sent send[0:n1] to [n0, n0+n1); slot 1 sent send[n1:] to [0, n0)
        assert np.array_equal(got[n0:], send[:n1]), epoch
        assert np.array_equal(got[:n0], send[n1:]), epoch
        d_send.free()
    ghost.free()
    hip.call("spmv_hip_put_destroy", put)


def test_put_exchange_times_out_instead_of_hanging(ctx):
    """A neighbour that never answers: the bounded waits of the put kernel end
    after about 4 s, the window reports the failure, and every later exchange
    returns SPMV_HIP_EPEER at once -- the GPU is never left with a kernel that
    polls for ever."""
    import time
    n = 64
    put, handle, raw, pid = _put_window(ctx, 8 * n)
    # connected to itself in the WRONG slot: it signals slot 3, waits on slot 0
    hip.call("spmv_hip_put_connect", put, 0, handle, raw, pid, 8 * n, 0, 3, 0, n,
             0, n, 1)
    hip.call("spmv_hip_put_finish", put)
    hip.call("spmv_hip_put_label", put, 7, 0, 9)  # I am rank 7, slot 0 is rank 9
    buf = C.create_string_buffer(512)
    hip.call("spmv_hip_peer_error_detail", ctx.h, buf, 512)
    assert buf.value == b""  # nothing has failed
    # a segment outside the staging buffers is refused at connect time
    with pytest.raises(Exception):
        hip.call("spmv_hip_put_connect", put, 1, handle, raw, pid, 8 * n, 1, 3, 0,
                 n, 0, n, 1)
    ctx.set_option("put_timeout_ms", 3000)  # (the default is a minute)
    send, ghost = ctx.upload(np.ones(n)), ctx.upload(np.zeros(n))
    t0 = time.perf_counter()
    hip.call("spmv_hip_put_exchange", ctx.h, put, 8, send.ptr, ghost.ptr, None)
    # the host learns of it where it waits for the device anyway
    with pytest.raises(Exception, match="did not answer"):
        ctx.stream_sync()
    waited = time.perf_counter() - t0
    assert 2.0 < waited < 20.0, waited
    failed = C.c_int()
    hip.call("spmv_hip_put_status", put, C.byref(failed))
    assert failed.value == 1
    # ... and the failure says WHICH wait gave up: the first exchange (epoch 1)
    # never saw slot 0's FREE flag move (it signalled slot 3 instead)
    hip.call("spmv_hip_peer_error_detail", ctx.h, buf, 512)
    msg = buf.value.decode()
    assert "rank 7: put kernel of epoch 1" in msg and "FREE flag" in msg, msg
    assert "neighbour slot 0 (rank 9)" in msg and "shows epoch 0" in msg, msg
    assert "never started" in msg
    # nothing was delivered, and the ghosts cannot be taken for valid ones: NaN
    out = np.empty(n)
    hip.call("spmv_hip_copy_d2h_async", ctx.h, out.ctypes.data_as(C.c_void_p),
             ghost.ptr, 8 * n, None)
    with pytest.raises(Exception, match="did not answer"):
        ctx.synchronize()
    assert np.all(np.isnan(out))
    with pytest.raises(Exception, match="did not answer"):
        hip.call("spmv_hip_put_exchange", ctx.h, put, 8, send.ptr, ghost.ptr, None)
    hip.call("spmv_hip_put_destroy", put)  # ... until the window is gone
    ctx.set_option("put_timeout_ms", 60000)
    ctx.stream_sync()
    send.free(), ghost.free()


def test_unstructured_generator_matches_numpy_twin(ctx):
    """spmv_hip_unstructured_fill_f64 (the benchmark's matrix without lattice
    structure) against spmv_amd.poisson.unstructured_csr: same arrays."""
    for N, per_row, band, far in ((1, 1, 0, 0), (777, 5, 16, 500),
                                  (50_000, 7, 2048, 100), (3000, 32, 100, 1000)):
        rp, ci, va = poisson.unstructured_csr(N, per_row, band, far, seed=99 + N)
        d_rp = ctx.empty(N + 1, np.int32)
        d_ci = ctx.empty(N * per_row, np.int32)
        d_va = ctx.empty(N * per_row, np.float64)
        hip.call("spmv_hip_unstructured_fill_f64", ctx.h, N, per_row, band, far,
                 99 + N, d_rp.ptr, d_ci.ptr, d_va.ptr, None)
        assert np.array_equal(d_rp.numpy(), rp)
        assert np.array_equal(d_ci.numpy(), ci)
        assert np.array_equal(d_va.numpy(), va)
        assert np.all(np.diff(ci.reshape(N, per_row), axis=1) >= 0)
        for b in (d_rp, d_ci, d_va):
            b.free()
    with pytest.raises(Exception):
        hip.call("spmv_hip_unstructured_fill_f64", ctx.h, 10, 33, 5, 0, 1, 1, 1,
                 1, None)



# ---------------------------------------------------------------------------
# CG building blocks: drive the kernels exactly as spmv::cg does and compare
# with the oracle's CG (cg.cpp:21-98)
# ---------------------------------------------------------------------------
def gpu_cg(ctx, blk, b, kmax, rtol, fused_dot=True, regrouped=False):
    n = blk.nrows
    ws = C.c_void_p()
    hip.call("spmv_hip_cg_ws_create", ctx.h, kmax, C.byref(ws))
    hip.call("spmv_hip_cg_ws_reset", ws, rtol, None)
    part = C.c_void_p()
    hip.call("spmv_hip_cg_ws_partials", ws, C.byref(part))
    r, p = ctx.upload(b), ctx.upload(b)
    x, Ap = ctx.zeros(n, np.float64), ctx.zeros(n, np.float64)
    hip.call("spmv_hip_cg_dot_rr_f64", ctx.h, ws, n, r.ptr, None)
    hip.call("spmv_hip_cg_reduce_rr", ctx.h, ws, 0, None)
    for k in range(1, kmax + 1):
        if fused_dot and not blk.symmetric:
            blk.mult(1.0, p.ptr, 0.0, Ap.ptr, dot_partials=part)
        else:
            blk.mult(1.0, p.ptr, 0.0, Ap.ptr)
            hip.call("spmv_hip_dot_partial_f64", ctx.h, n, p.ptr, Ap.ptr, part,
                     None)
        hip.call("spmv_hip_cg_reduce_pAp", ctx.h, ws, k, None)
        if regrouped:  # what spmv::cg issues: p is read once per iteration
            hip.call("spmv_hip_cg_update_r_f64", ctx.h, ws, k, n, Ap.ptr, r.ptr,
                     None)
            hip.call("spmv_hip_cg_reduce_rr", ctx.h, ws, k, None)
            hip.call("spmv_hip_cg_update_xp_f64", ctx.h, ws, k, n, r.ptr, x.ptr,
                     p.ptr, None)
        else:
            hip.call("spmv_hip_cg_update_xr_f64", ctx.h, ws, k, n, p.ptr,
                     Ap.ptr, x.ptr, r.ptr, None)
            hip.call("spmv_hip_cg_reduce_rr", ctx.h, ws, k, None)
            hip.call("spmv_hip_cg_update_p_f64", ctx.h, ws, k, n, r.ptr, p.ptr,
                     None)
    flags = np.zeros(2, np.int32)
    rr = np.zeros(kmax + 1)
    # a destination shorter than the device history is refused at the boundary
    # (ABI 2), and nothing is written
    short = np.full(kmax, -7.0)
    with pytest.raises(Exception):
        hip.call("spmv_hip_cg_ws_read_async", ws,
                 flags.ctypes.data_as(C.c_void_p),
                 short.ctypes.data_as(C.c_void_p), kmax, None)
    ctx.stream_sync()
    assert np.all(short == -7.0)
    cap = C.c_int()
    hip.call("spmv_hip_cg_ws_capacity", ws, C.byref(cap))
    assert cap.value == kmax
    hip.call("spmv_hip_cg_ws_read_async", ws, flags.ctypes.data_as(C.c_void_p),
             rr.ctypes.data_as(C.c_void_p), kmax + 1, None)
    ctx.stream_sync()
    xs = x.numpy()
    hip.call("spmv_hip_cg_ws_destroy", ws)
    for buf in (r, p, x, Ap):
        buf.free()
    return xs, flags, np.sqrt(rr)


@pytest.mark.parametrize("symmetric", [False, True])
def test_cg_kernels_match_oracle(ctx, symmetric):
    n = 12
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    b = oracle.csr_spmv(rp, ci, va, np.ones(N))  # exact solution = ones
    if symmetric:
        lrp, lci, lva, dg = lower_split(rp, ci, va)
        blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
        x_ref, k_ref, hist_ref = oracle.cg(lrp, lci, lva, b, 200, 1e-10, dg)
    else:
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va)
        x_ref, k_ref, hist_ref = oracle.cg(rp, ci, va, b, 200, 1e-10)
    assert k_ref < 200
    x, flags, hist = gpu_cg(ctx, blk, b, 200, 1e-10)
    # both groupings of the vector updates are the same arithmetic
    x2, flags2, hist2 = gpu_cg(ctx, blk, b, 200, 1e-10, regrouped=True)
    assert np.array_equal(flags, flags2)
    if not symmetric:  # deterministic kernels: bit-identical
        assert np.array_equal(x, x2) and np.array_equal(hist, hist2)
    else:
        assert np.linalg.norm(x - x2) <= 1e-9 * np.linalg.norm(x)
    # the host enqueued all 200 iterations; the device stopped itself
    assert flags[0] == 1 and abs(int(flags[1]) - k_ref) <= 1
    k = int(flags[1])
    m = min(k, k_ref, 50)
    assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6, atol=0)
    assert hist[k] / hist[0] < 1e-10
    assert np.linalg.norm(x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    assert np.linalg.norm(x - 1.0) <= 1e-8 * np.sqrt(N)
    blk.free()


def test_cg_kmax_reached_and_unfused_dot(ctx):
    n = 10
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    b = oracle.gaussian_x_fast(N)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va)
    x_ref, k_ref, hist_ref = oracle.cg(rp, ci, va, b, 7, 1e-30)
    assert k_ref == 7
    for fused in (True, False):
        x, flags, hist = gpu_cg(ctx, blk, b, 7, 1e-30, fused_dot=fused)
        assert flags[0] == 0  # never converged: host returns kmax
        assert np.allclose(hist, hist_ref, rtol=1e-10, atol=0)
        assert np.linalg.norm(x - x_ref) <= 1e-12 * np.linalg.norm(x_ref)
    blk.free()


# ---------------------------------------------------------------------------
# size-independent properties at benchmark scale (no oracle run needed)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [128])
def test_full_size_properties(ctx, n):
    """A * ones is an exact small integer per row (6 - #neighbours), so both
    the general and the atomic symmetric kernel must reproduce it bit for
    bit at full size; linearity A(2x) == 2 A x is exact as well."""
    N = n ** 3
    blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
    assert blk.nnz == poisson.poisson3d_nnz(n)
    x, y = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
    ctx.fill_const(N, 1.0, x.ptr)
    blk.mult(1.0, x.ptr, 0.0, y.ptr)
    i = np.arange(N)
    xx, yy, zz = i % n, (i // n) % n, i // (n * n)
    nb = ((xx > 0).astype(int) + (xx < n - 1) + (yy > 0) + (yy < n - 1)
          + (zz > 0) + (zz < n - 1))
    expect = (6 - nb).astype(np.float64)
    assert np.array_equal(y.numpy(), expect)
    sym = hip.poisson3d_block(ctx, n, 0, N, hip.PART_LOCAL_LOWER,
                              with_diagonal=True)
    assert sym.nnz == (poisson.poisson3d_nnz(n) - N) // 2
    sym.mult(1.0, x.ptr, 0.0, y.ptr)
    assert np.array_equal(y.numpy(), expect)
    # linearity + general == symmetric within the stated tolerance
    ctx.fill_gaussian(N, 0, N, x.ptr)
    blk.mult(1.0, x.ptr, 0.0, y.ptr)
    y1 = y.numpy()
    blk.mult(2.0, x.ptr, 0.0, y.ptr)
    assert np.array_equal(y.numpy(), 2.0 * y1)
    sym.mult(1.0, x.ptr, 0.0, y.ptr)
    xh = x.numpy()
    assert np.all(np.abs(y.numpy() - y1) <= 16 * U * 12 * np.abs(xh).max())
    # checksum: sum(A x) == sum over boundary-weighted x (A symmetric)
    assert abs(y1.sum() - float(expect @ xh)) <= 1e-9 * np.abs(xh).sum()
    for b in (x, y):
        b.free()
    blk.free(), sym.free()


def test_spmv_dot_partials_all_general_kernels(ctx):
    """The p.Ap share every general kernel leaves next to y: the partials add
    up to x.(A x) whatever kernel (and row grouping) produced them."""
    n = 20
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va)
    x = oracle.gaussian_x_fast(N)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    dx, dy = ctx.upload(x), ctx.empty(N, np.float64)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    expect = float(x @ y_ref)
    for knobs in (dict(), dict(algo=hip.ALGO_SCALAR), dict(algo=hip.ALGO_VECTOR)):
        for k, v in knobs.items():
            blk.set(k, v)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr, dot_partials=part.ptr)
        got = float(np.sum(part.numpy()))
        assert abs(got - expect) <= 1e-12 * abs(expect), knobs
        assert np.array_equal(dy.numpy(), y_ref) or knobs.get("algo") == hip.ALGO_VECTOR
        blk.set("algo", hip.ALGO_ROWBLOCK)
    for b in (dx, dy, part):
        b.free()
    blk.free()


@pytest.mark.parametrize("window", [-1, 0, 256])
def test_symmetric_fused_dot(ctx, window):
    """x.(A x) produced by the symmetric kernels themselves: the deterministic
    one (window -1) has the finished row, the atomic ones use the mirror
    identity sum_i x_i (2 (d_i x_i + (L x)_i) - d_i x_i)."""
    n = 18
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, lrp, lci, lva, dg, True)
    assert blk.get("sym_det") == 1
    if window >= 0:
        blk.set("sym_det", 0)
        blk.set("sym_window", window)
    dx, dy = ctx.upload(x), ctx.empty(N, np.float64)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    res = ctx.empty(1, np.float64)
    for alpha in (1.0, -0.5):
        blk.mult(alpha, dx.ptr, 0.0, dy.ptr, dot_partials=part.ptr)
        hip.call("spmv_hip_reduce_partials_f64", ctx.h, part.ptr, res.ptr, None)
        got = res.numpy()[0]
        expect = alpha * float(x @ y_ref)
        scale = abs(alpha) * float(np.abs(x) @ (np.abs(va[np.arange(len(va))]) @ np.ones(1) if False else np.abs(y_ref) + 12 * np.abs(x)))
        assert abs(got - expect) <= 1e-13 * scale
        assert np.all(np.abs(dy.numpy() - alpha * y_ref) <= 16 * U * 12 * abs(alpha))
    for b in (dx, dy, part, res):
        b.free()
    blk.free()


# ---------------------------------------------------------------------------
# SURVEY 8d "robustness" inputs
# ---------------------------------------------------------------------------
def test_tridiagonal_ten_million_rows(ctx):
    """1-D operator of demos/CreateA.cpp (gamma = 0.1) at N = 1e7."""
    N = 10_000_000
    rp, ci, va = oracle.tridiag_csr(N)
    x = oracle.gaussian_x_fast(N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    for algo in EXACT_ALGOS + [hip.ALGO_AUTO]:
        y = run_spmv(ctx, rp, ci, va, x, N, N, algo=algo)
        assert np.array_equal(y, y_ref), algo
    bound = 16 * U * abs_bound(rp, ci, va, x)
    y = run_spmv(ctx, rp, ci, va, x, N, N, algo=hip.ALGO_VECTOR)
    assert np.all(np.abs(y - y_ref) <= bound)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    ys_ref = oracle.csr_spmv_sym(lrp, lci, lva, dg, x)
    for knobs in (dict(), dict(sym_det=0), dict(sym_det=0, sym_window=0)):
        ys = run_spmv(ctx, lrp, lci, lva, x, N, N, diagonal=dg, symmetric=True,
                      knobs=knobs)
        if not knobs:
            assert np.array_equal(ys, ys_ref)
        assert np.all(np.abs(ys - y_ref) <= bound), knobs


@pytest.mark.parametrize("n", [64, 216])
def test_poisson_sparsity_splitmix_values(ctx, n):
    """Poisson sparsity with values and x drawn from splitmix64(0x5EED0001);
    n = 216 is the north-star size (10,077,696 rows)."""
    rp, ci, _ = oracle.poisson3d(n)
    N = n ** 3
    u = oracle.splitmix64_unit(len(ci) + N)
    va, x = u[:len(ci)].copy(), u[len(ci):].copy()
    y_ref = oracle.csr_spmv(rp, ci, va, x, -0.75, 0.0)
    y = run_spmv(ctx, rp, ci, va, x, N, N, alpha=-0.75)
    assert np.array_equal(y, y_ref)
    y0 = oracle.splitmix64_unit(N, seed=0x5EED0002)
    y_ref = oracle.csr_spmv(rp, ci, va, x, 1.0, 1.0, y0)
    y = run_spmv(ctx, rp, ci, va, x, N, N, beta=1.0, y0=y0)
    assert np.array_equal(y, y_ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_scatter_add_distinct_indices(ctx, dtype):
    """Owner-side accumulate of L2GMap::reverse_update (L2GMap.cpp:921-922)."""
    rng = np.random.default_rng(3)
    n, m = 100_000, 37_111
    idx = rng.permutation(n)[:m].astype(np.int32)
    src = rng.uniform(-1, 1, m).astype(dtype)
    dst = rng.uniform(-1, 1, n).astype(dtype)
    want = dst.copy()
    want[idx] += src
    d_i, d_s, d_d = ctx.upload(idx, np.int32), ctx.upload(src, dtype), ctx.upload(dst, dtype)
    ctx.scatter_add(d_i, d_s, d_d, m, dtype=dtype)
    ctx.scatter_add(d_i, d_s, d_d, 0, dtype=dtype)  # empty call is a no-op
    assert np.array_equal(d_d.numpy(), want)
    for b in (d_i, d_s, d_d):
        b.free()


def test_fuzz_shapes_all_kernels(ctx):
    """Sixty random shapes in one test: tiny and empty matrices, one column,
    nnz just below / at / above tile boundaries, block-boundary row counts,
    all-empty rows; exact kernels bit-exact, the others inside their bound;
    symmetric storage derived from a random symmetric pattern."""
    # SPMV_FUZZ_SEED / SPMV_FUZZ_TRIALS: other seeds, longer runs (by hand)
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", str(0xF022))))
    specials = [(1, 1, 0.0), (1, 1, 3.0), (255, 7, 2.0), (256, 256, 1.0),
                (257, 300, 2.0), (511, 1, 4.0), (512, 2000, 0.0), (513, 50, 9.0),
                (1024, 1024, 0.5), (2049, 4096, 2.0)]
    for case in range(int(os.environ.get("SPMV_FUZZ_TRIALS", "60"))):
        if case < len(specials):
            nrows, ncols, avg = specials[case]
        else:
            nrows = int(rng.integers(1, 6000))
            ncols = int(rng.integers(1, 6000))
            avg = float(rng.choice([0.2, 1, 2, 5, 11, 40]))
        rp, ci, va = random_csr(rng, nrows, ncols, avg,
                                empty_frac=float(rng.choice([0.0, 0.1, 0.6])),
                                long_rows=int(rng.integers(0, 3)),
                                long_len=int(rng.integers(300, 3000)))
        x = rng.uniform(-1, 1, ncols)
        alpha, beta = float(rng.choice([1.0, -2.0, 0.5])), float(rng.choice([0.0, 1.0, -0.25]))
        y0 = rng.uniform(-1, 1, nrows)
        y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
        bound = (16 + np.diff(rp)) * U * abs_bound(rp, ci, va, x, alpha, beta, y0)
        for algo in (hip.ALGO_AUTO, hip.ALGO_ROWBLOCK, hip.ALGO_SCALAR,
                     hip.ALGO_VECTOR, hip.ALGO_ROWLIST):
            if algo == hip.ALGO_ROWLIST and len(va) == 0:
                continue  # a row list of an empty block is refused (EINVAL)
            y = run_spmv(ctx, rp, ci, va, x, nrows, ncols, alpha, beta,
                         None if beta == 0 else y0, algo=algo)
            if algo in (hip.ALGO_ROWBLOCK, hip.ALGO_SCALAR, hip.ALGO_ROWLIST):
                assert np.array_equal(y, y_ref), (case, algo)
            else:
                assert np.all(np.abs(y - y_ref) <= bound + 1e-300), (case, algo)
        if case % 3 == 0:  # symmetric storage of a random symmetric matrix
            n = min(nrows, 1500)
            dense = rng.random((n, n)) < min(0.5, (avg + 1) / n)
            dense = dense | dense.T
            vals = rng.uniform(-1, 1, (n, n))
            vals = (vals + vals.T) / 2
            srp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
            sci = np.nonzero(dense)[1].astype(np.int32)
            sva = vals[dense]
            xs = rng.uniform(-1, 1, n)
            ys0 = rng.uniform(-1, 1, n)
            ref = oracle.csr_spmv(srp, sci, sva, xs, alpha, beta, ys0)
            lrp, lci, lva, dg = lower_split(srp, sci, sva)
            sb = (16 + 2 * np.diff(srp)) * U * abs_bound(srp, sci, sva, xs, alpha, beta, ys0)
            sref = oracle.csr_spmv_sym(lrp, lci, lva, dg, xs, alpha, beta, ys0)
            for knobs in (dict(), dict(sym_det=0), dict(sym_det=0, sym_window=0),
                          dict(sym_det=0, sym_window=512, sym_rows=512)):
                ys = run_spmv(ctx, lrp, lci, lva, xs, n, n, alpha, beta,
                              None if beta == 0 else ys0, diagonal=dg,
                              symmetric=True, knobs=knobs)
                if not knobs:
                    assert np.array_equal(ys, sref), case
                assert np.all(np.abs(ys - ref) <= sb + 1e-300), (case, knobs)
