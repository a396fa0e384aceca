"""GPU parity tests of the LX form and of the XW kernel (spmv_lxw.hip,
spmv_csr_forms.hip): x windows staged in LDS, 16-bit offsets or the caller's
32-bit column indices -- bit-exact against the CPU oracle (split from
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
# LX form: LDS-staged x windows + 16-bit local column offsets
# ---------------------------------------------------------------------------
@pytest.fixture()
def lx_ctx():
    c = hip.Context(0)
    c.set_option("lx_min_nnz", 0)  # build the form for small test matrices too
    yield c
    c.close()


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
