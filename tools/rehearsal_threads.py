"""REHEARSAL (never a benchmark result): BASELINE configs[4] -- the 512^3 Poisson
matrix row-partitioned over 8 ranks -- with the ranks as 8 THREADS of one
process on ONE GPU (tests/thread_world.py: halo = device-to-device copies
between the ranks' buffers, reductions through the host).  A 1-GPU box admits
at most 6 processes on its card, so `bench.py --transport gloo` stops at 6
ranks; threads reach the real rank count and with it the per-rank shapes of the
8-GPU run: local blocks of 512 x 512 x 64, remote blocks of 2 x 262,144 rows,
2 MiB ghost planes.

What it proves: every rank's local block takes the form the 1-rank matrix takes
(constant-diagonal / diagonal form), the remote block the row-list kernel; the
halo self-check (every ghost = g(global index)) passes; 20 CG iterations land on
the 1-rank residual ||r_10|| / ||r_0|| = 6.635343844806121 to rounding.  Times
are NOT performance numbers (8 ranks share one GPU).

    python tools/rehearsal_threads.py [--cm onesided_put_active] [--grid 512]
"""
import argparse
import json
import os
import sys
import time

# threaded ranks share the process's hardware queues: ask for enough of them
# before HIP starts (tests/conftest.py does the same)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from spmv_amd import _lib, host, poisson  # noqa: E402
from thread_world import ThreadWorld  # noqa: E402

K10_512 = 6.635343844806121


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cm", default="p2p_nonblocking",
                    choices=["p2p_blocking", "p2p_nonblocking",
                             "onesided_put_active"])
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--peer-reduce", action="store_true",
                    help="the CG scalars by the deterministic peer reduction "
                         "(Comm::enable_peer_reduce) instead of the transport's "
                         "all-reduce")
    ap.add_argument("--poll-every", type=int, default=0,
                    help="CgOptions::poll_every: iterations the host may run "
                         "ahead of the device (0 = default 16)")
    ap.add_argument("--allow-pair", action="store_true",
                    help="PROBE: --peer-reduce together with the one-sided halo "
                         "(the library refuses the pair; SPMV_ALLOW_PUT_WITH_PEER_"
                         "REDUCE=1 lifts that for this probe)")
    ap.add_argument("--timeout-ms", type=int, default=20000,
                    help="put / peer-reduce wait bound (ctx put_timeout_ms)")
    args = ap.parse_args()
    if args.peer_reduce and args.cm.startswith("onesided"):
        if not args.allow_pair:
            raise SystemExit("--peer-reduce goes with the two-sided halo models "
                             "(--allow-pair: the probe of the pair)")
        os.environ["SPMV_ALLOW_PUT_WITH_PEER_REDUCE"] = "1"
    P, n = args.ranks, args.grid
    N = n ** 3
    cm = getattr(host, args.cm.upper())
    tw = ThreadWorld(P, timeout=120.0 if args.peer_reduce else 600.0)
    ranks = [None] * P
    t_all = time.perf_counter()

    def rank_body(rank, comm, exec_):
        import ctypes as C
        ctx = exec_.context
        stream = C.c_void_p()
        _lib.call("spmv_hip_stream_create", ctx, C.byref(stream))
        _lib.call("spmv_hip_set_stream", ctx, stream)
        A = host.Matrix.create_poisson3d(comm, exec_, n, args.symmetric, cm)
        l2g = A.col_map()
        M, ng = l2g.local_size(), l2g.num_ghosts()
        rec = {"rank": rank, "rows": M, "ghosts": ng, "neighbours": int(l2g._nn),
               "onesided": bool(l2g.onesided()),
               "local": {k: A.plan_get(k) for k in
                         ("lat", "slat", "sdia", "sdia_const", "sdia_tile", "zwalk")},
               "blocks": A.blocks()}
        if A.blocks()["remote"][2] > 0:
            rec["remote_algo"] = A.plan_get("algo", remote=True)  # 4 = ROWLIST
        # halo self-check on this run's own partition
        d_v = exec_.alloc(M + ng)
        exec_.memset(d_v, 0xFF, 8 * (M + ng))
        _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M, d_v,
                  None)
        tw.bar.wait()
        l2g.update(d_v)
        l2g.update_finalise(d_v)
        exec_.synchronize()
        got = exec_.copy_to_host(d_v, M + ng)
        gidx = np.asarray(l2g.ghosts(), dtype=np.float64)
        want = np.exp(-10 * (5 * (gidx / float(N) - 0.5)) ** 2)
        rec["halo_selfcheck"] = "ok" if np.allclose(got[M:], want, rtol=1e-12,
                                                     atol=1e-300) else "FAILED"
        exec_.free(d_v)
        d_b, d_x = exec_.alloc(M), exec_.alloc(M)
        _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M, d_b,
                  None)
        ws = host.CgWorkspace(exec_)
        host.cg_ex(comm, exec_, A, d_b, d_x, 0, 1e-30, ws)  # sizes the workspace
        if args.peer_reduce:
            # AFTER the workspace exists: hipMalloc / hipFree wait for the whole
            # device -- with the ranks as threads of one process, for a peer's
            # reduction kernel that in turn waits for this rank's (ranks in
            # processes of their own do not share that wait).  A stuck wait
            # fails in 20 s.
            _lib.call("spmv_hip_ctx_set_option", ctx, b"put_timeout_ms",
                      args.timeout_ms)
            rec["peer_reduce"] = bool(comm.enable_peer_reduce(exec_))
        tw.bar.wait()
        t0 = time.perf_counter()
        try:
            k, hist, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, args.steps, 0.0,
                                       ws, history=True,
                                       poll_every=args.poll_every)
            exec_.synchronize()
        except Exception as e:
            # every rank's own account of the wait that gave up (the first
            # failure alone would hide what the OTHER rank was waiting for);
            # give the peers' kernels time to run into their own bound first
            time.sleep(args.timeout_ms / 1e3 + 1.0)
            buf = C.create_string_buffer(512)
            _lib.call("spmv_hip_peer_error_detail", ctx, buf, 512)
            print(f"rank {rank} FAILED: {e}\n  rank {rank} record: "
                  f"{buf.value.decode() or '(none)'}", flush=True)
            raise
        rec["cg_wall_s"] = time.perf_counter() - t0
        rec["k"] = k
        rec["k10"] = float(hist[min(10, len(hist) - 1)] / hist[0])
        ranks[rank] = rec
        tw.bar.wait()
        ws.close()
        A.close()
        exec_.free(d_b), exec_.free(d_x)
        tw.bar.wait()
        exec_.synchronize()
        _lib.call("spmv_hip_set_stream", ctx, None)
        _lib.call("spmv_hip_stream_destroy", ctx, stream)

    tw.run(rank_body, gpu=True)
    k10 = ranks[0]["k10"]
    out = {"data": "synthetic (REHEARSAL: %d ranks as threads of one process on one "
                   "GPU; not a performance number)" % P,
           "config": {"workload": f"poisson3d_{n}^3_csr_fp64_cg", "rows": N,
                      "nnz": poisson.poisson3d_nnz(n), "ranks": P, "halo": args.cm,
                      "storage": "symmetric-csr" if args.symmetric else "csr"},
           "steps": args.steps,
           "halo_selfcheck": "ok" if all(r["halo_selfcheck"] == "ok" for r in ranks)
           else "FAILED",
           "cg_rel_residual": {"k10": k10, "k10_expected": K10_512 if n == 512 else None,
                               "k10_ok": (bool(abs(k10 / K10_512 - 1) < 1e-9)
                                          if n == 512 else None)},
           "every_rank_same_k10": bool(all(r["k10"] == k10 for r in ranks)),
           "local_block_form": ranks[0]["local"],
           "every_local_block_same_form": bool(all(r["local"] == ranks[0]["local"]
                                                   for r in ranks)),
           "remote_block_algo": sorted({r.get("remote_algo") for r in ranks}),
           "onesided_put_path": bool(all(r["onesided"] for r in ranks)),
           "peer_reduce": bool(all(r.get("peer_reduce") for r in ranks)),
           "ranks": [{k: r[k] for k in ("rank", "rows", "ghosts", "neighbours")}
                     for r in ranks],
           "wall_s": time.perf_counter() - t_all}
    print(json.dumps(out), flush=True)
    ok = (out["halo_selfcheck"] == "ok" and out["every_rank_same_k10"]
          and out["cg_rel_residual"]["k10_ok"] in (True, None))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
