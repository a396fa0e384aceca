"""Kernel sweep on one GPU: times SpMV variants on the device-generated
Poisson matrix, all variants interleaved in ONE process (guide rule 24).

    python tools/kbench.py --n 216 512 --reps 20 --out gpurun_out/kbench.json

Algorithmic bytes follow SURVEY section 8d.  Prints one JSON line per variant.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import hip, poisson  # noqa: E402

HBM_PEAK = 8000.0  # GB/s, MI355X spec


def time_ms(ctx, fn, reps, rounds=3):
    e0, e1 = ctx.event_create(), ctx.event_create()
    fn()
    ctx.synchronize()
    best = []
    for _ in range(rounds):
        ctx.event_record(e0)
        for _ in range(reps):
            fn()
        ctx.event_record(e1)
        ctx.event_sync(e1)
        best.append(ctx.elapsed_ms(e0, e1) / reps)
    ctx.event_destroy(e0), ctx.event_destroy(e1)
    return min(best), float(np.median(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[128, 216])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--fp32", action="store_true",
                    help="also time the fp32 instantiation (host-built matrix)")
    ap.add_argument("--only", default=None, help="comma list of variants")
    args = ap.parse_args()
    ctx = hip.Context(0)
    results = []

    def emit(**kw):
        results.append(kw)
        print(json.dumps(kw), flush=True)

    for n in args.n:
        N = n ** 3
        blk = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
        x, y = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
        ctx.fill_gaussian(N, 0, N, x.ptr)
        part = ctx.empty(ctx.dot_partials_len, np.float64)
        bytes_csr = poisson.csr_bytes(N, N, blk.nnz)
        reps = max(3, args.reps if n < 400 else args.reps // 4)

        # known-good ceiling on this device: d2d copy of the same byte count
        nb = min(bytes_csr // 2, 4 << 30) // 8 * 8
        src, dst = ctx.empty(nb // 8, np.float64), ctx.empty(nb // 8, np.float64)
        tmin, tmed = time_ms(ctx, lambda: ctx.copy(dst.ptr, src.ptr, nb), reps)
        emit(n=n, variant="memcpy_d2d", ms=tmin, ms_med=tmed,
             gbs=2 * nb / tmin / 1e6, frac=2 * nb / tmin / 1e6 / HBM_PEAK)
        src.free(), dst.free()

        # the three forms of the general kernel, each on its own plan: lattice
        # (default for this matrix), LX (lattice switched off), plain gather
        forms = [("lattice", blk, dict())]
        ctx.set_option("lat_min_nnz", 1 << 62)
        lx = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
        forms.append(("lx", lx, dict()))
        forms.append(("gather", lx, dict(lx=0)))
        ctx.set_option("lat_min_nnz", 1 << 20)
        for name, b, kn in forms:
            for k, v in kn.items():
                b.set(k, v)
            for dot in (False, True):
                tmin, tmed = time_ms(
                    ctx, lambda: b.mult(1.0, x.ptr, 0.0, y.ptr,
                                        dot_partials=part.ptr if dot else None),
                    reps)
                emit(n=n, variant=name + ("+dot" if dot else ""),
                     form=dict(lat=b.get("lat"), lx=b.get("lx")), ms=tmin,
                     ms_med=tmed, gbs=bytes_csr / tmin / 1e6,
                     frac=bytes_csr / tmin / 1e6 / HBM_PEAK)
        lx.set("lx", 1)
        for name, kn in (("scalar", dict(algo=hip.ALGO_SCALAR)),
                         ("vector", dict(algo=hip.ALGO_VECTOR, lanes_per_row=8))):
            for k, v in kn.items():
                lx.set(k, v)
            tmin, tmed = time_ms(
                ctx, lambda: lx.mult(1.0, x.ptr, 0.0, y.ptr), reps)
            emit(n=n, variant=name, knobs=kn, ms=tmin, ms_med=tmed,
                 gbs=bytes_csr / tmin / 1e6,
                 frac=bytes_csr / tmin / 1e6 / HBM_PEAK)
        lx.free()
        blk.free()

        if args.fp32 and n <= 256:
            rp32, ci32, va32 = poisson.poisson3d_csr(n, dtype=np.float32)
            b32 = hip.CsrBlock(ctx, N, N, rp32, ci32.astype(np.int32), va32,
                               dtype=np.float32)
            del rp32, ci32, va32
            x32, y32 = ctx.zeros(N, np.float32), ctx.empty(N, np.float32)
            bytes32 = poisson.csr_bytes(N, N, b32.nnz, value_bytes=4)
            for ch, nt in ((1, 0), (1, 1), (2, 1), (4, 1)):
                b32.set("chunks", ch)
                b32.set("nontemporal", nt)
                tmin, tmed = time_ms(
                    ctx, lambda: b32.mult(1.0, x32.ptr, 0.0, y32.ptr), reps)
                emit(n=n, variant="rowblock_fp32",
                     knobs=dict(chunks=ch, nontemporal=nt), ms=tmin,
                     ms_med=tmed, gbs=bytes32 / tmin / 1e6,
                     frac=bytes32 / tmin / 1e6 / HBM_PEAK)
            b32.free(), x32.free(), y32.free()

        # symmetric
        sym = hip.poisson3d_block(ctx, n, 0, N, hip.PART_LOCAL_LOWER,
                                  with_diagonal=True)
        bytes_sym = poisson.sym_csr_bytes(N, sym.nnz)
        for window, srows, nt in ((0, 1024, 0), (256, 512, 0), (512, 512, 0),
                                  (256, 1024, 0), (512, 1024, 0),
                                  (512, 1024, 1), (512, 2048, 0),
                                  (768, 1024, 0)):
            sym.set("nontemporal", nt)
            sym.set("sym_window", window)
            sym.set("sym_rows", srows)
            tmin, tmed = time_ms(
                ctx, lambda: sym.mult(1.0, x.ptr, 0.0, y.ptr), reps)
            emit(n=n, variant="symmetric",
                 knobs=dict(sym_window=window, sym_rows=srows,
                            nontemporal=nt), ms=tmin,
                 ms_med=tmed, gbs=bytes_sym / tmin / 1e6,
                 frac=bytes_sym / tmin / 1e6 / HBM_PEAK,
                 gbs_equiv_general=bytes_csr / tmin / 1e6)
        sym.free()

        # CG vector kernels at this size
        ws = C.c_void_p()
        hip.call("spmv_hip_cg_ws_create", ctx.h, 4, C.byref(ws))
        hip.call("spmv_hip_cg_ws_reset", ws, 0.0, None)
        r, p = ctx.empty(N, np.float64), ctx.empty(N, np.float64)
        ctx.fill_const(N, 1.0, r.ptr), ctx.fill_const(N, 1.0, p.ptr)
        ctx.fill_const(N, 1.0, y.ptr)
        hip.call("spmv_hip_cg_dot_rr_f64", ctx.h, ws, N, r.ptr, None)
        hip.call("spmv_hip_cg_reduce_rr", ctx.h, ws, 0, None)
        hip.call("spmv_hip_dot_partial_f64", ctx.h, N, p.ptr, y.ptr, part.ptr,
                 None)
        pslot = C.c_void_p()
        hip.call("spmv_hip_cg_ws_pAp", ws, 1, C.byref(pslot))
        hip.call("spmv_hip_reduce_partials_f64", ctx.h, part.ptr, pslot, None)
        rslot = C.c_void_p()
        hip.call("spmv_hip_cg_ws_rr", ws, 1, C.byref(rslot))
        ctx.copy(rslot, pslot, 8)
        tmin, tmed = time_ms(ctx, lambda: hip.call(
            "spmv_hip_cg_update_xr_f64", ctx.h, ws, 1, N, p.ptr, y.ptr, x.ptr,
            r.ptr, None), reps)
        emit(n=n, variant="cg_update_xr", ms=tmin, ms_med=tmed,
             gbs=6 * N * 8 / tmin / 1e6, frac=6 * N * 8 / tmin / 1e6 / HBM_PEAK)
        ctx.copy(rslot, pslot, 8)
        tmin, tmed = time_ms(ctx, lambda: hip.call(
            "spmv_hip_cg_update_p_f64", ctx.h, ws, 1, N, r.ptr, p.ptr, None),
            reps)
        emit(n=n, variant="cg_update_p", ms=tmin, ms_med=tmed,
             gbs=3 * N * 8 / tmin / 1e6, frac=3 * N * 8 / tmin / 1e6 / HBM_PEAK)
        tmin, tmed = time_ms(ctx, lambda: hip.call(
            "spmv_hip_cg_reduce_rr", ctx.h, ws, 2, None), reps)
        emit(n=n, variant="cg_reduce", ms=tmin, ms_med=tmed)
        # the same three kernels in CG order, each bracketed by its own events
        gen = hip.poisson3d_block(ctx, n, 0, N, hip.PART_ALL)
        evs = [ctx.event_create() for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        rounds = max(5, reps)
        for it in range(rounds + 2):
            ctx.copy(rslot, pslot, 8)
            ctx.event_record(evs[0])
            gen.mult(1.0, p.ptr, 0.0, y.ptr, dot_partials=part.ptr)
            ctx.event_record(evs[1])
            hip.call("spmv_hip_cg_update_xr_f64", ctx.h, ws, 1, N, p.ptr, y.ptr,
                     x.ptr, r.ptr, None)
            ctx.event_record(evs[2])
            ctx.copy(rslot, pslot, 8)
            hip.call("spmv_hip_cg_update_p_f64", ctx.h, ws, 1, N, r.ptr, p.ptr,
                     None)
            ctx.event_record(evs[3])
            ctx.event_sync(evs[3])
            if it >= 2:
                for i in range(3):
                    acc[i] += ctx.elapsed_ms(evs[i], evs[i + 1])
        emit(n=n, variant="sequence[spmv+dot,xr,p]",
             ms_spmv=acc[0] / rounds, ms_xr=acc[1] / rounds,
             ms_p=acc[2] / rounds, ms=sum(acc) / rounds, ms_med=0.0)
        gen.free()
        hip.call("spmv_hip_cg_ws_destroy", ws)
        for b in (x, y, part, r, p):
            b.free()
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
