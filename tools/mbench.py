"""SpMV on the general (non-lattice) test matrices, one JSON line per variant:
the kernel the plan picks (auto) and each forced algorithm / knob, timed with HIP
events in ONE process, priced with SURVEY 8d's CSR bytes, and compared bit for
bit with the one-lane-per-row kernel (the reference loop verbatim).

    python tools/mbench.py --kind fem fem_tail fem81 unstructured
    python tools/mbench.py --kind fem_tail --variants auto rowblock vector16
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import _lib, host, poisson  # noqa: E402

KINDS = {
    "fem": dict(),
    "fem_tail": dict(tail_permille=10),
    "fem81": dict(min_len=81, max_len=81),
    "fem15": dict(min_len=15, max_len=15),  # the mean of "fem", every row alike
    "fem_long": dict(min_len=40, max_len=120),  # ragged AND long
    "fem_mid": dict(min_len=20, max_len=80),
    # symmetric storage (strictly lower part + diagonal; run with --no-check)
    "fem_sym": dict(symmetric=True),
    "fem_tail_sym": dict(symmetric=True, tail_permille=10),
}
ALGO = {"rowblock": 1, "vector": 2, "scalar": 3}
FORM_KEYS = ("algo", "sym_sj", "sj_long_rows", "sj_long_table", "lat", "lx", "lxw", "wdia", "wdia_half", "wdia_hbox", "sdia", "sjds", "sj_wpb", "sj_unit", "sj_sigma",
             "sj_max_chunks", "sj_far_permille", "sj_staged_bytes_per_entry_x100",
             "sj_wide", "sj_long_rows", "lx_staged", "lx_blocks", "blocks_per_cu", "nontemporal")


def make(kind, rows, comm, exec_):
    if kind.startswith("poisson"):  # poisson512: the 7-point matrix, general
        return host.Matrix.create_poisson3d(comm, exec_, int(kind[7:]), False,
                                            host.P2P_BLOCKING)
    if kind == "unstructured":
        return host.Matrix.create_unstructured(comm, exec_, rows)
    return host.Matrix.create_fem_like(comm, exec_, rows, **KINDS[kind])


def timed(exec_, A, d_x, d_y, reps):
    ctx = exec_.context
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("spmv_hip_event_create", ctx, 1, C.byref(e0))
    _lib.call("spmv_hip_event_create", ctx, 1, C.byref(e1))
    A.mult(d_x, d_y)
    best = None
    for _ in range(3):
        _lib.call("spmv_hip_event_record", ctx, e0, None)
        for _ in range(reps):
            A.mult(d_x, d_y)
        _lib.call("spmv_hip_event_record", ctx, e1, None)
        _lib.call("spmv_hip_event_synchronize", ctx, e1)
        ms = C.c_float()
        _lib.call("spmv_hip_event_elapsed_ms", ctx, e0, e1, C.byref(ms))
        best = ms.value / reps if best is None else min(best, ms.value / reps)
    _lib.call("spmv_hip_event_destroy", ctx, e0)
    _lib.call("spmv_hip_event_destroy", ctx, e1)
    return best


def form_of(A):
    out = {}
    for k in FORM_KEYS:
        try:
            out[k] = A.plan_get(k)
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", nargs="+", default=["fem", "fem_tail", "fem81",
                                                  "unstructured"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", nargs="+",
                    default=["auto", "rowblock", "vector", "scalar"],
                    help="auto | rowblock | vector[LPR] | scalar | key=value[,key=value] "
                         "(plan_set knobs on the auto plan)")
    ap.add_argument("--set", nargs="*", default=[], help="ctx option=value ...")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    exec_ = host.HipExecutor(0)
    comm = host.Comm.self_comm()
    ctx = exec_.context
    for kv in args.set:
        k, v = kv.split("=")
        _lib.call("spmv_hip_ctx_set_option", ctx, k.encode(), int(v))
    for kind in args.kind:
        A = make(kind, args.rows, comm, exec_)
        rows, cols, nnz = A.blocks()["local"]
        algo_bytes = poisson.csr_bytes(rows, cols, nnz)
        d_x, d_y = exec_.alloc(cols), exec_.alloc(rows)
        _lib.call("spmv_hip_fill_gaussian_f64", ctx, cols, 0, cols, d_x, None)
        auto_algo = A.plan_get("algo")
        y_ref = None
        if not args.no_check:
            A.plan_set("algo", ALGO["scalar"])
            exec_.memset(d_y, 0xFF, 8 * rows)
            A.mult(d_x, d_y)
            y_ref = exec_.copy_to_host(d_y, rows)
            A.plan_set("algo", auto_algo)
        base = {"kind": kind, "rows": rows, "nnz": nnz,
                "avg_row": nnz / rows, "B_csr": algo_bytes,
                "plan_ms": A.plan_get("plan_us") / 1e3,
                "plan_mem_ms": A.plan_get("plan_mem_us") / 1e3,
                "plan_kib": A.plan_get("plan_kib")}
        for var in args.variants:
            A.plan_set("algo", auto_algo)
            undo = []
            try:
                if var == "auto":
                    pass
                elif var.startswith("vector"):
                    A.plan_set("algo", ALGO["vector"])
                    if var[6:]:
                        A.plan_set("lanes_per_row", int(var[6:]))
                elif var in ALGO:
                    A.plan_set("algo", ALGO[var])
                else:
                    for kv in var.split(","):
                        k, v = kv.split("=")
                        try:
                            undo.append((k, A.plan_get(k)))
                        except Exception:
                            pass
                        A.plan_set(k, int(v))
            except Exception as e:  # a knob this plan does not have
                print(json.dumps(dict(base, variant=var, error=str(e))), flush=True)
                continue
            exec_.memset(d_y, 0xFF, 8 * rows)
            ms = timed(exec_, A, d_x, d_y, args.reps)
            rec = dict(base, variant=var, ms=round(ms, 5),
                       gbs_csr=round(algo_bytes / ms / 1e6, 1),
                       frac_csr=round(algo_bytes / ms / 1e6 / 8000.0, 4),
                       form=form_of(A))
            if y_ref is not None:
                y = exec_.copy_to_host(d_y, rows)
                rec["bit_equal_scalar"] = bool(np.array_equal(y, y_ref))
                if not rec["bit_equal_scalar"]:
                    d = np.abs(y - y_ref)
                    rec["max_abs_diff"] = float(np.nanmax(d))
                    rec["finite"] = bool(np.isfinite(y).all())
            print(json.dumps(rec), flush=True)
            for k, v in undo:
                A.plan_set(k, v)
        A.close()
        exec_.free(d_x), exec_.free(d_y)
    comm.close()
    exec_.close()


if __name__ == "__main__":
    main()
