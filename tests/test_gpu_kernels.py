"""GPU parity tests of the HIP kernels, called through the C ABI
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
from spmv_amd import hip, poisson
from util import U, abs_bound, lower_split, random_csr

pytestmark = pytest.mark.gpu

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


# ---------------------------------------------------------------------------
# LX form: LDS-staged x windows + 16-bit local column offsets
# ---------------------------------------------------------------------------
@pytest.fixture()
def lx_ctx():
    c = hip.Context(0)
    c.set_option("lx_min_nnz", 0)  # build the form for small test matrices too
    yield c
    c.close()


def _banded_mixed(rng, n):
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


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lx_form_bit_exact(lx_ctx, dtype):
    ctx = lx_ctx
    rng = np.random.default_rng(77)
    cases = []
    for n in (16, 20, 33):
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", rp, ci.astype(np.int32), va, n ** 3, n ** 3))
    rp, ci, va = _banded_mixed(rng, 5000)
    cases.append(("banded_mixed", rp, ci, va, 5000, 5000))
    rp, ci, va = oracle.tridiag_csr(70001)  # odd column count: rounded windows
    cases.append(("tridiag", rp, ci, va, 70001, 70001))
    for name, rp, ci, va, nrows, ncols in cases:
        va = va.astype(dtype)
        x = rng.uniform(-1, 1, ncols).astype(dtype)
        y0 = rng.uniform(-1, 1, nrows).astype(dtype)
        blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va, None, False,
                           hip.ALGO_ROWBLOCK, dtype)
        assert blk.get("lx") == 1, name
        nrb = (nrows + 255) // 256
        assert blk.get("lx_blocks") == nrb
        if name == "banded_mixed":
            assert 0 < blk.get("lx_staged") < nrb  # some blocks stay direct
        else:
            # (the DMA layout fetches x in aligned 16-byte chunks: the row
            # blocks that touch the last ncols % 4 columns stay direct)
            assert nrb - blk.get("lx_staged") <= (0 if ncols % 4 == 0 else 3)
        assert blk.get("lxw") == 1, name  # the LDS-DMA kernel is the default
        dx = ctx.upload(x, dtype)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)  # f32-aware
            for nt in (0, 1):
                blk.set("nontemporal", nt)
                # the DMA kernel, the register-staged kernel on the same
                # (padded) layout, and the plain gather kernel: same bits
                for lx, lxw in ((1, 1), (1, 0), (0, 0)):
                    blk.set("lx", lx)
                    blk.set("lxw", lxw)
                    dy = ctx.upload(np.full(nrows, np.nan, dtype) if beta == 0
                                    else y0, dtype)
                    blk.mult(alpha, dx.ptr, beta, dy.ptr)
                    y = dy.numpy()
                    dy.free()
                    assert np.array_equal(y, y_ref), (name, alpha, beta, nt, lx,
                                                      lxw)
        dx.free()
        blk.free()
    # the register-staged kernel's own layout (context option lx_dma = 0)
    ctx.set_option("lx_dma", 0)
    rp, ci, va = _banded_mixed(rng, 5000)
    va = va.astype(dtype)
    x = rng.uniform(-1, 1, 5000).astype(dtype)
    blk = hip.CsrBlock(ctx, 5000, 5000, rp, ci, va, None, False,
                       hip.ALGO_ROWBLOCK, dtype)
    ctx.set_option("lx_dma", 1)
    assert blk.get("lx") == 1 and blk.get("lxw") == 0
    with pytest.raises(Exception):
        blk.set("lxw", 1)  # its records were not built
    dx, dy = ctx.upload(x, dtype), ctx.upload(np.full(5000, np.nan, dtype), dtype)
    blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
    assert np.array_equal(dy.numpy(), oracle.csr_spmv(rp, ci, va, x))
    for b in (dx, dy):
        b.free()
    blk.free()


@pytest.mark.parametrize("n", [32, 33, 48])
def test_lxw_plane_walk_bit_exact(lx_ctx, n):
    """The LX form's DMA kernel on 3-D grids in the plane-walk order (forced
    tables with 1-3 runs; planes of whole row blocks for n = 32, 48, ragged
    ones for n = 33) -- random values and a third of the entries dropped, so no
    row block repeats its neighbour.  Same bits as the oracle with every
    combination, fused dot included."""
    ctx = lx_ctx
    rng = np.random.default_rng(500 + n)
    N = n ** 3
    offs = [-n * n, -n, -1, 0, 1, n, n * n]
    for drop in (0.0, 0.3):
        rp, ci, va = _stencil_csr(rng, N, offs, drop=drop)
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        assert blk.get("lx") == 1 and blk.get("lxw") == 1 and blk.get("lat") == 0
        assert blk.get("lattice_d2") == n * n
        dx = ctx.upload(x)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(), dict(zwalk_segments=1), dict(zwalk_segments=2),
                          dict(zwalk_segments=3), dict(zwalk=0),
                          dict(zwalk=1, lxw_blocks_per_cu=1),
                          dict(lxw_blocks_per_cu=0), dict(lxw=0), dict(lxw=1)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                dot = beta == 0
                blk.mult(alpha, dx.ptr, beta, dy.ptr,
                         dot_partials=part.ptr if dot else None)
                assert np.array_equal(dy.numpy(), y_ref), (n, drop, alpha, knobs)
                if dot:
                    want = float(np.dot(x, y_ref))
                    got = float(np.sum(part.numpy()))
                    assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
                dy.free()
        for b in (dx, part):
            b.free()
        blk.free()


def test_lx_fused_dot_and_row_block_orders(lx_ctx):
    ctx = lx_ctx
    n = 24
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    N = n ** 3
    x = oracle.gaussian_x_fast(N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("lx") == 1
    dx, dy = ctx.upload(x), ctx.upload(np.zeros(N))
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    want = float(np.dot(x, y_ref))
    for knobs in (dict(), dict(xcd_group=0), dict(xcd_group=3),
                  dict(blocks_per_cu=1), dict(blocks_per_cu=8)):
        for k, v in knobs.items():
            blk.set(k, v)
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr, dot_partials=part.ptr)
        assert np.array_equal(dy.numpy(), y_ref), knobs
        got = float(np.sum(part.numpy()))
        assert abs(got - want) <= 1e-12 * abs(want), knobs
    for b in (dx, dy, part):
        b.free()
    blk.free()


def test_options_and_plan_queries_reject_unknown_keys(ctx):
    with pytest.raises(Exception):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(Exception):
        ctx.set_option("blas1_nt_min_elems", -1)
    rp, ci, va = poisson.poisson3d_csr(6)
    blk = hip.CsrBlock(ctx, 216, 216, rp, ci.astype(np.int32), va, None, False,
                       hip.ALGO_ROWBLOCK)
    with pytest.raises(Exception):
        blk.get("no_such_key")
    with pytest.raises(Exception):
        blk.set("lx", 1)  # the form was not built for this small matrix
    assert blk.get("lx") == 0 and blk.get("algo") == hip.ALGO_ROWBLOCK
    blk.free()


def test_lx_fuzz_banded(lx_ctx):
    """Random banded matrices in LX form: empty rows, empty row blocks, ragged
    last block, duplicates, a row too long for the plan kernel (direct block),
    bands too wide to stage."""
    ctx = lx_ctx
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", str(0x1F))))
    for case in range(int(os.environ.get("SPMV_FUZZ_TRIALS", "24"))):
        nrows = int(rng.choice([1, 255, 256, 257, 700, 3001]))
        ncols = nrows + int(rng.integers(0, 50))
        half = int(rng.choice([3, 40, 200, 900, 4000]))
        lens = rng.poisson(float(rng.choice([1.0, 4.0, 9.0])), nrows)
        lens[rng.random(nrows) < float(rng.choice([0.0, 0.3]))] = 0
        if case % 5 == 0 and nrows > 600:
            lens[256:512] = 0          # a whole row block without entries
        if case % 7 == 0 and nrows > 300:
            lens[rng.integers(0, nrows)] = 5000  # > 4096 entries in one block
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(nrows), lens)
        ci = np.clip(rows + rng.integers(-half, half + 1, len(rows)), 0,
                     ncols - 1).astype(np.int32)
        va = rng.uniform(-1, 1, len(ci))
        x = rng.uniform(-1, 1, ncols)
        y0 = rng.uniform(-1, 1, nrows)
        if len(ci) == 0:
            continue
        try:
            blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va, None, False,
                               hip.ALGO_ROWBLOCK)
        except Exception as e:  # pragma: no cover
            raise AssertionError((case, nrows, half)) from e
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (0.5, -1.0)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            dy = ctx.upload(np.full(nrows, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr)
            assert np.array_equal(dy.numpy(), y_ref), (case, nrows, half,
                                                       blk.get("lx"),
                                                       blk.get("lx_staged"))
            dy.free()
        dx.free()
        blk.free()


# ---------------------------------------------------------------------------
# XW: the LDS-DMA kernel on the CALLER's CSR arrays (32-bit column indices
# streamed as they are, x windows staged; spmv_lxw.hip) -- what a plan without
# lattice / LX / sliced jagged form runs instead of the gather kernel
# ---------------------------------------------------------------------------
@pytest.fixture()
def xw_ctx():
    c = hip.Context(0)
    c.set_option("lx_min_nnz", 1 << 62)  # no LX form: the arrays stay the caller's
    c.set_option("lat_min_nnz", 1 << 62)
    c.set_option("sj_min_nnz", 1 << 62)
    c.set_option("xw_min_nnz", 0)
    c.set_option("xw_min_x_bytes", 0)
    yield c
    c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_xw_kernel_on_the_callers_arrays_bit_exact(xw_ctx, dtype):
    """Poisson grids (3 windows per row block), a banded matrix with far blocks
    and two row blocks of scattered columns (those gather), a tridiagonal matrix
    with an odd column count (the blocks at the end of x gather), a matrix whose
    row blocks need MORE than four windows (they gather): every alpha / beta,
    non-temporal loads on / off, the fused dot -- the oracle's bits, and the
    plain gather kernel on the same plan (xw = 0) too."""
    ctx = xw_ctx
    rng = np.random.default_rng(177)
    cases = []
    for n in (16, 20, 33):
        rp, ci, va = poisson.poisson3d_csr(n)
        cases.append((f"poisson{n}", rp, ci.astype(np.int32), va, n ** 3, n ** 3))
    rp, ci, va = _banded_mixed(rng, 5000)
    cases.append(("banded_mixed", rp, ci, va, 5000, 5000))
    rp, ci, va = oracle.tridiag_csr(70001)
    cases.append(("tridiag", rp, ci, va, 70001, 70001))
    # the diagonal and eight far bands, at most eight windows per row block
    # (staged) -- but for eight row blocks with a ninth band (those gather)
    N6 = 40000
    offs6 = [0] + [s * d for d in (3000, 6000, 9000, 20000) for s in (-1, 1)]
    rp, ci, va = _stencil_csr(rng, N6, offs6, drop=0.2)
    extra = np.arange(10240, 12288)
    rows = np.concatenate([np.repeat(np.arange(N6), np.diff(rp)), extra])
    cols = np.concatenate([ci, extra + 15000]).astype(np.int32)
    order = np.lexsort((cols, rows))
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N6))]
                        ).astype(np.int32)
    ci, va = cols[order], rng.uniform(-1, 1, len(cols))
    cases.append(("many_windows", rp, ci, va, N6, N6))
    # ... and bands every 1200 columns: more than eight windows everywhere
    offs9 = [0] + [s * 1200 * k for k in range(1, 12) for s in (-1, 1)]
    rp, ci, va = _stencil_csr(rng, N6, offs9, drop=0.2)
    cases.append(("too_many_windows", rp, ci, va, N6, N6))
    for name, rp, ci, va, nrows, ncols in cases:
        va = va.astype(dtype)
        x = rng.uniform(-1, 1, ncols).astype(dtype)
        y0 = rng.uniform(-1, 1, nrows).astype(dtype)
        blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va, None, False,
                           hip.ALGO_ROWBLOCK, dtype)
        nrb = (nrows + 255) // 256
        assert blk.get("lx") == 0 and blk.get("lat") == 0 and blk.get("sjds") == 0
        if name == "too_many_windows":
            # every row block would gather: the records are dropped
            assert blk.get("xw") == 0, name
            blk.free()
            continue
        assert blk.get("xw") == 1, name
        if name in ("banded_mixed", "many_windows"):
            assert 0 < blk.get("xw_staged") < nrb
        else:
            assert nrb - blk.get("xw_staged") <= (0 if ncols % 4 == 0 else 3)
        assert blk.get("plan_kib") <= (nrb * 144 + 4 * 2048 * 512) // 1024 + 2
        dx = ctx.upload(x, dtype)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.0), (2.0, 1.0), (1.0, -0.25)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)  # f32-aware
            for nt in (0, 1):
                blk.set("nontemporal", nt)
                for xw in (1, 0):
                    blk.set("xw", xw)
                    dy = ctx.upload(np.full(nrows, np.nan, dtype) if beta == 0
                                    else y0, dtype)
                    dot = dtype == np.float64 and beta == 0 and nrows == ncols
                    blk.mult(alpha, dx.ptr, beta, dy.ptr,
                             dot_partials=part.ptr if dot else None)
                    y = dy.numpy()
                    dy.free()
                    assert np.array_equal(y, y_ref), (name, alpha, beta, nt, xw)
                    if dot:
                        want = float(np.dot(x.astype(np.float64), y_ref))
                        got = float(np.sum(part.numpy()))
                        assert abs(got - want) <= 1e-12 * (np.abs(x) @ np.abs(y_ref))
            blk.set("xw", 1)
        for b in (dx, part):
            b.free()
        blk.free()


@pytest.mark.parametrize("n", [32, 33, 48])
def test_xw_plane_walk_bit_exact(xw_ctx, n):
    """The XW kernel on 3-D grids in the plane-walk order (forced tables with
    1-3 runs, table off, one workgroup per CU) -- random values, a third of the
    entries dropped.  Same bits as the oracle."""
    ctx = xw_ctx
    rng = np.random.default_rng(900 + n)
    N = n ** 3
    offs = [-n * n, -n, -1, 0, 1, n, n * n]
    for drop in (0.0, 0.3):
        rp, ci, va = _stencil_csr(rng, N, offs, drop=drop)
        x = rng.uniform(-1, 1, N)
        y0 = rng.uniform(-1, 1, N)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
        assert blk.get("xw") == 1 and blk.get("lx") == 0 and blk.get("lat") == 0
        assert blk.get("lattice_d2") == n * n
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (-0.5, 0.75)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            for knobs in (dict(), dict(zwalk_segments=1), dict(zwalk_segments=2),
                          dict(zwalk_segments=3), dict(zwalk=0),
                          dict(zwalk=1, lxw_blocks_per_cu=1),
                          dict(lxw_blocks_per_cu=0), dict(xw=0), dict(xw=1)):
                for k, v in knobs.items():
                    blk.set(k, v)
                dy = ctx.upload(np.full(N, np.nan) if beta == 0 else y0)
                blk.mult(alpha, dx.ptr, beta, dy.ptr)
                assert np.array_equal(dy.numpy(), y_ref), (n, drop, alpha, knobs)
                dy.free()
        dx.free()
        blk.free()


def test_xw_fuzz_banded(xw_ctx):
    """Random banded matrices through the XW kernel: empty rows, empty row
    blocks, ragged last block, repeated and unsorted columns, a row too long
    for the plan kernel, bands too wide to stage, rectangular blocks."""
    ctx = xw_ctx
    rng = np.random.default_rng(int(os.environ.get("SPMV_FUZZ_SEED", str(0x2F))))
    seen_xw = 0
    for case in range(int(os.environ.get("SPMV_FUZZ_TRIALS", "24"))):
        nrows = int(rng.choice([1, 255, 256, 257, 700, 3001]))
        ncols = nrows + int(rng.integers(0, 50))
        half = int(rng.choice([3, 40, 200, 900, 4000]))
        lens = rng.poisson(float(rng.choice([1.0, 4.0, 9.0])), nrows)
        lens[rng.random(nrows) < float(rng.choice([0.0, 0.3]))] = 0
        if case % 5 == 0 and nrows > 600:
            lens[256:512] = 0
        if case % 7 == 0 and nrows > 300:
            lens[rng.integers(0, nrows)] = 5000
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(nrows), lens)
        ci = np.clip(rows + rng.integers(-half, half + 1, len(rows)), 0,
                     ncols - 1).astype(np.int32)
        va = rng.uniform(-1, 1, len(ci))
        x = rng.uniform(-1, 1, ncols)
        y0 = rng.uniform(-1, 1, nrows)
        if len(ci) == 0:
            continue
        blk = hip.CsrBlock(ctx, nrows, ncols, rp, ci, va, None, False,
                           hip.ALGO_ROWBLOCK)
        seen_xw += blk.get("xw")
        dx = ctx.upload(x)
        for alpha, beta in ((1.0, 0.0), (0.5, -1.0)):
            y_ref = oracle.csr_spmv(rp, ci, va, x, alpha, beta, y0)
            dy = ctx.upload(np.full(nrows, np.nan) if beta == 0 else y0)
            blk.mult(alpha, dx.ptr, beta, dy.ptr)
            assert np.array_equal(dy.numpy(), y_ref), (case, nrows, half,
                                                       blk.get("xw"),
                                                       blk.get("xw_staged"))
            dy.free()
        dx.free()
        blk.free()
    assert seen_xw > 0


def test_xw_probe_lets_the_first_launches_choose(xw_ctx):
    """XW or the gather kernel: launches 0-3 of a plan with XW records alternate
    between the two under HIP events, a later launch reads the times and fixes
    the choice.  Every launch -- probing or not -- returns
    the oracle's bits; plan_set "xw_probe" restarts or ends the probe, "xw" = 1
    asks for the kernel by name; the context option switches the probe off."""
    ctx = xw_ctx
    n = 40
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    rng = np.random.default_rng(61)
    va = rng.uniform(-1, 1, len(va))
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("xw") == 1 and blk.get("xw_pick") == -1
    dx = ctx.upload(x)
    part = ctx.empty(ctx.dot_partials_len, np.float64)

    def launch(dot=False):
        dy = ctx.upload(np.full(N, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr, dot_partials=part.ptr if dot else None)
        y = dy.numpy()  # (synchronises)
        dy.free()
        assert np.array_equal(y, y_ref)
    for i in range(4):
        assert blk.get("xw_pick") == -1, i
        launch(dot=bool(i & 1))
    launch()  # the four are complete: this one reads them
    pick = blk.get("xw_pick")
    assert pick in (0, 1)
    assert blk.get("xw_probe_xw_us") > 0 and blk.get("xw_probe_gather_us") > 0
    assert pick == (blk.get("xw_probe_xw_us") <= blk.get("xw_probe_gather_us")) \
        or blk.get("xw_probe_xw_us") == blk.get("xw_probe_gather_us")
    launch(dot=True)
    assert blk.get("xw_pick") == pick
    blk.set("xw_probe", 1)  # again
    assert blk.get("xw_pick") == -1
    for _ in range(6):
        launch()
    assert blk.get("xw_pick") in (0, 1)
    blk.set("xw_probe", 0)
    assert blk.get("xw_pick") == 1
    blk.set("xw_probe", 1)
    blk.set("xw", 1)  # by name
    assert blk.get("xw_pick") == 1
    launch()
    blk.free()
    ctx.set_option("xw_probe", 0)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("xw") == 1 and blk.get("xw_pick") == 1
    launch()
    blk.free()
    for b in (dx, part):
        b.free()


def test_sliced_jagged_form_declined_leaves_the_xw_kernel():
    """a plan that wanted the sliced jagged form and could not have
    it (here: 8 staged chunks per block leave nearly every entry far) stages the
    x windows over the caller's arrays instead of gathering -- with the default
    sj_min_nnz / xw_min_nnz thresholds (both 2^20)."""
    ctx = hip.Context(0)
    ctx.set_option("lx_min_nnz", 1 << 62)
    ctx.set_option("lat_min_nnz", 1 << 62)
    ctx.set_option("sj_max_chunks", 8)
    ctx.set_option("xw_min_x_bytes", 0)
    n = 64
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    rng = np.random.default_rng(62)
    va = rng.uniform(-1, 1, len(va))
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_ROWBLOCK)
    assert blk.get("xw") == 0  # the sliced jagged form is to come with the values
    with pytest.raises(Exception):  # SPMV_HIP_ENOTSUP: no form holds the values
        blk.bake()
    assert blk.get("sjds") == 0 and blk.get("lx") == 0 and blk.get("lat") == 0
    assert blk.get("xw") == 1 and blk.get("xw_staged") == (N + 255) // 256
    with pytest.raises(Exception):  # ... and the analysis is not repeated
        blk.bake()
    dx = ctx.upload(x)
    for _ in range(6):
        dy = ctx.upload(np.full(N, np.nan))
        blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
        assert np.array_equal(dy.numpy(), y_ref)
        dy.free()
    assert blk.get("xw_pick") in (0, 1)
    dx.free()
    blk.free()
    ctx.close()


def test_csr_in_place_plans_take_xw_at_the_default_thresholds():
    """The context option csr_in_place: no copy of the index or value stream (no
    LX form, no sliced jagged form) -- a banded matrix whose x outgrows the
    caches (17 M columns = 134 MB >= xw_min_x_bytes) gets the XW kernel with
    every threshold at its default; without the option the LX form."""
    N = 17_000_000
    rng = np.random.default_rng(63)
    rows = np.arange(N, dtype=np.int64)
    cols = np.stack([rows + 60 * k - 120 + rng.integers(0, 50, N) for k in range(4)],
                    axis=1)
    ci = np.clip(cols, 0, N - 1).astype(np.int32).ravel()
    rp = (np.arange(N + 1, dtype=np.int64) * 4).astype(np.int32)
    va = rng.uniform(-1, 1, len(ci))
    x = rng.uniform(-1, 1, N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    for in_place in (1, 0):
        ctx = hip.Context(0)
        ctx.set_option("csr_in_place", in_place)
        blk = hip.CsrBlock(ctx, N, N, rp, ci, va, None, False, hip.ALGO_AUTO)
        assert blk.get("lat") == 0 and blk.get("sjds") == 0
        if in_place:
            assert blk.get("xw") == 1 and blk.get("lx") == 0
            # 144 B per row block, nothing per entry
            assert blk.get("plan_kib") <= ((N + 255) // 256 * 144) // 1024 + 8
        else:
            assert blk.get("xw") == 0 and blk.get("lx") == 1
        dx = ctx.upload(x)
        for _ in range(6):
            dy = ctx.upload(np.full(N, np.nan))
            blk.mult(1.0, dx.ptr, 0.0, dy.ptr)
            assert np.array_equal(dy.numpy(), y_ref), in_place
            dy.free()
        if in_place:
            assert blk.get("xw_pick") in (0, 1)
        dx.free()
        blk.free()
        ctx.close()


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


def _stencil_csr(rng, N, offsets, drop=0.0, dtype=np.float64):
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
