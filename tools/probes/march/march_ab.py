"""PROBE (round 6): A/B of the marched exchange (march_probe.hip, plan built on the
host by march_build.py) against the product's kernel for symmetric storage of a
matrix without lattice structure (the merged matrix in the sliced jagged form,
spmv_sjds.hip) -- same matrix (the benchmark's FEM-like matrix in symmetric
storage), same x, ONE box, HIP events; the two results compared bit for bit.

    make -C tools/probes/march          # the kernels (hipcc, gfx950)
    python tools/probes/march/march_ab.py [--rows 10000000] [--lseg 20] [--no-sort]
                                        [--tile 512]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import march_build  # noqa: E402
from spmv_amd import hip, poisson  # noqa: E402
from spmv_amd.host import FemParams  # noqa: E402


def timed(ctx, fn, reps):
    e0, e1 = ctx.event_create(), ctx.event_create()
    fn()
    best = None
    for _ in range(3):
        ctx.event_record(e0)
        for _ in range(reps):
            fn()
        ctx.event_record(e1)
        ctx.event_sync(e1)
        ms = ctx.elapsed_ms(e0, e1) / reps
        best = ms if best is None else min(best, ms)
    ctx.event_destroy(e0), ctx.event_destroy(e1)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--lseg", type=int, default=0,
                    help="levels per unit (0: about two units per CU)")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--no-sort", action="store_true",
                    help="lanes in row order (no sort by length inside a tile)")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--tile", type=int, default=1024, choices=[1024, 512],
                    help="rows per tile (512: 8 waves, ~70 KB of LDS, two "
                         "workgroups per CU)")
    ap.add_argument("--fem", nargs="*", default=[], help="generator key=value ...")
    args = ap.parse_args()
    suf = "" if args.tile == 1024 else "_512"
    march_build.B = args.tile
    march_build.SLICES = args.tile // 64
    wgs_per_cu = 1 if args.tile == 1024 else 2
    lib = C.CDLL(os.path.join(HERE, f"libmarch_probe{suf}.so"))
    stash, xcap = C.c_int(), C.c_int()
    lib.march_limits(C.byref(stash), C.byref(xcap))
    ctx = hip.Context(0)
    N = args.rows
    kw = {k: int(v) for k, v in (kv.split("=") for kv in args.fem)}
    prm = FemParams(**poisson.fem_params(N, **kw))
    # the matrix on the device: general block -> symmetric storage
    d_rp = ctx.empty(N + 1, np.int32)
    nnz = C.c_int64()
    hip.call("spmv_hip_fem_count", ctx.h, C.byref(prm), d_rp.ptr, C.byref(nnz), None)
    d_ci, d_va = ctx.empty(nnz.value, np.int32), ctx.empty(nnz.value, np.float64)
    hip.call("spmv_hip_fem_fill_f64", ctx.h, C.byref(prm), nnz.value, d_rp.ptr,
             d_ci.ptr, d_va.ptr, None)
    l_rp = ctx.empty(N + 1, np.int32)
    lnnz = C.c_int64()
    hip.call("spmv_hip_csr_lower_split_count", ctx.h, N, d_rp.ptr, d_ci.ptr, l_rp.ptr,
             C.byref(lnnz), None)
    l_ci, l_va = ctx.empty(lnnz.value, np.int32), ctx.empty(lnnz.value, np.float64)
    l_dg = ctx.empty(N, np.float64)
    hip.call("spmv_hip_csr_lower_split_fill_f64", ctx.h, N, d_rp.ptr, d_ci.ptr,
             d_va.ptr, l_rp.ptr, l_ci.ptr, l_va.ptr, l_dg.ptr, None)
    ctx.synchronize()
    for b in (d_rp, d_ci, d_va):
        b.free()
    # ---- the product's kernel --------------------------------------------------
    blk = hip.CsrBlock(ctx, N, N, l_rp, l_ci, l_va, l_dg, True)
    blk.bake()
    assert blk.get("sym_sj") == 1
    x = ctx.empty(N, np.float64)
    ctx.fill_gaussian(N, 0, N, x.ptr)
    y_ref, y = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
    part = ctx.empty(ctx.dot_partials_len, np.float64)
    ms_ref = timed(ctx, lambda: blk.mult(1.0, x.ptr, 0.0, y_ref.ptr,
                                         dot_partials=part.ptr), args.reps)
    B_sym = lnnz.value * 12 + (N + 1) * 4 + 3 * N * 8
    # ---- the marched plan, on the host -----------------------------------------
    rp, ci, va, dg = l_rp.numpy(), l_ci.numpy(), l_va.numpy(), l_dg.numpy()
    t0 = time.perf_counter()
    S = march_build.far_offset(rp, ci)
    assert S > march_build.B, S
    nlev = (N + S - 1) // S
    nchain = (S + march_build.B - 1) // march_build.B
    lseg = args.lseg or max(2, int(np.ceil(nlev * nchain
                                           / (2.0 * wgs_per_cu * ctx.num_cus))))
    p = march_build.build(rp, ci, va, dg, S, lseg, stash.value, xcap.value,
                          sort_rows=not args.no_sort)
    t_build = time.perf_counter() - t0
    dev = {k: ctx.upload(v) for k, v in p.items()
           if isinstance(v, np.ndarray) and k != "diag"}
    grid = args.grid or min(p["nunits"], wgs_per_cu * ctx.num_cus)
    ptr = lambda k: C.c_void_p(dev[k].ptr)  # noqa: E731

    def run():
        rc = lib.march_spmv(
            C.c_int(grid), C.c_int(p["nunits"]), ptr("unit_step0"), ptr("step_tile"),
            ptr("step_chunk0"), ptr("step_own0"), ptr("chunks"), ptr("tile_row0"),
            ptr("meta"), ptr("sb0"), ptr("sb1"), ptr("sb2"), ptr("sb3"), ptr("sb4"),
            ptr("a_val"), ptr("a_code"), ptr("bc_code"), ptr("bs_val"),
            ptr("bs_code"), ptr("cc_code"), ptr("cs_val"), ptr("cs_code"),
            C.c_void_p(l_dg.ptr), C.c_int64(N), C.c_void_p(x.ptr), C.c_void_p(y.ptr),
            C.c_void_p(part.ptr), None)
        assert rc == 0, rc
    ctx.memset(y.ptr, 0xFF, 8 * N)
    ms = timed(ctx, run, args.reps)
    ctx.synchronize()
    yr, ym = y_ref.numpy(), y.numpy()
    same = bool(np.array_equal(yr, ym))
    bad = int(np.sum(yr != ym))
    # ---- the pipelined version --------------------------------------------------
    lib2 = C.CDLL(os.path.join(HERE, f"libmarch_probe2{suf}.so"))
    p2 = march_build.pack_v2(p)
    dev2 = {k: ctx.upload(v) for k, v in p2.items()}
    ptr2 = lambda k: C.c_void_p(dev2[k].ptr)  # noqa: E731

    def run2():
        rc = lib2.march2_spmv(
            C.c_int(grid), C.c_int(p["nunits"]), ptr("unit_step0"), ptr2("steps"),
            ptr2("chunks_padded"), ptr("meta"), ptr2("dval"), ptr2("sbp"),
            ptr("a_val"), ptr2("a_code32"), ptr("bc_code"), ptr("bs_val"),
            ptr("bs_code"), ptr("cc_code"), ptr("cs_val"), ptr("cs_code"),
            C.c_int64(N), C.c_void_p(x.ptr), C.c_void_p(y.ptr),
            C.c_void_p(part.ptr), None)
        assert rc == 0, rc
    ctx.memset(y.ptr, 0xFF, 8 * N)
    ms2 = timed(ctx, run2, args.reps)
    ctx.synchronize()
    ym2 = y.numpy()
    same2 = bool(np.array_equal(yr, ym2))
    bad2 = int(np.sum(yr != ym2))
    moved = march_build.bytes_moved(p)
    st = p["stats"]
    out = {"rows": N, "stored_entries": int(lnnz.value), "far_offset": S,
           "levels_per_unit": lseg, "grid": grid, "tile_rows": args.tile, "sorted_lanes": not args.no_sort,
           "product_kernel_ms": ms_ref, "product_frac_of_B_sym": B_sym / ms_ref / 8e9,
           "marched_ms": ms, "marched_frac_of_B_sym": B_sym / ms / 8e9,
           "bit_equal": same, "rows_that_differ": bad,
           "marched_pipelined_ms": ms2,
           "marched_pipelined_frac_of_B_sym": B_sym / ms2 / 8e9,
           "pipelined_bit_equal": same2, "pipelined_rows_that_differ": bad2,
           "marched_bytes_per_launch": moved,
           "marched_bytes_per_stored_entry":
               (st["stored"] * 10 + (st["captured_near"] + st["captured_far"]) * 2
                + (st["streamed_early"] + st["streamed_late"]) * 10) / st["stored"],
           "captured_share": (st["captured_near"] + st["captured_far"]) / st["stored"],
           "marched_gbs": moved / ms / 1e6, "host_build_s": t_build, "stats": st}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
