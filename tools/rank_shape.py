"""A MODEL, not a measurement of scaling: what ONE rank of BASELINE configs[4]
(the 512^3 Poisson matrix row-partitioned over P GPUs) does per CG iteration,
timed alone on one GPU.

The P ranks exist as threads of this process only for the SET-UP (the plan
collectives of L2GMap and create_matrix need every rank: tests/thread_world.py);
then ONE interior rank -- local block 512 x 512 x (512 / P), its remote block
over two ghost planes (one at the ends of the chain) -- runs the P > 1 launch
sequence of spmv::cg (host/cg.cpp, reference cg.cpp:55-86) with a NO-OP
transport: the halo exchange and the two scalar all-reduces return at once and
move nothing.  What is left is exactly what the rank's own GPU executes between
two collectives: its kernels and the boundaries between them.  The other
ranks' threads idle at a barrier meanwhile, so nothing shares the GPU.

    python tools/rank_shape.py --ranks 2 4 8 [--grid 512] [--steps 100]
    rocprofv3 --kernel-trace --stats ... -- python3 tools/rank_shape.py --ranks 8
        (tools/rank_shape_report.py turns the trace into the boundary share)

Reported per P: ms per iteration (HIP events around the timed iterations on
the CG stream, and the wall clock), kernel launches per iteration, the SpMV's
live-timed launches; under rocprofv3 the report adds the sum of the kernels'
own durations -- the difference is kernel boundaries.  Adding a transport's
latency per collective to this gives the per-rank critical path of the N-GPU
run.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def model(P, n=512, steps=100, plain=True, cm="p2p_nonblocking", symmetric=False):
    """-> the record of one interior rank of P (see the module docstring)"""
    import ctypes as C

    from spmv_amd import _lib, host
    from thread_world import ThreadWorld

    class ModelWorld(ThreadWorld):
        """the thread world with a transport that moves nothing"""

        def device_transport(self, rank, ctx_handle, asynchronous=False):
            def exchange(user, elem, nn, nbrs, send_buf, scnt, soff, recv_base,
                         rcnt, roff, stream):
                return 0

            def allreduce(user, dev, count, stream):
                return 0
            return exchange, allreduce

    tw = ModelWorld(P, timeout=600.0)
    me = P // 2  # an interior rank (two neighbours) from P = 3 on
    out = {}
    N = n ** 3

    def rank_body(rank, comm, exec_):
        ctx = exec_.context
        opts = {}
        if plain and not symmetric:
            opts[b"lat_min_nnz"] = (1 << 62, 1 << 20)
        if plain and symmetric:
            opts[b"const_diagonals"] = (0, 1)
        for k, (v, _) in opts.items():
            _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
        A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                         getattr(host, cm.upper()))
        l2g = A.col_map()
        M, ng = l2g.local_size(), l2g.num_ghosts()
        tw.bar.wait()
        if rank == me:
            d_b, d_x = exec_.alloc(M), exec_.alloc(M)
            _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M,
                      d_b, None)
            ws = host.CgWorkspace(exec_)
            host.cg_ex(comm, exec_, A, d_b, d_x, 5, 0.0, ws)  # warm-up
            ws.reserve_timing(steps)
            exec_.synchronize()
            t0 = time.perf_counter()
            k, hist, spmv_ms, spmv_n = host.cg_ex(comm, exec_, A, d_b, d_x, steps,
                                                  0.0, ws, time_spmv=True,
                                                  history=True)
            exec_.synchronize()
            wall = time.perf_counter() - t0
            blocks = A.blocks()
            out.update({
                "ranks": P, "rank": rank, "rows": M, "ghosts": ng,
                "neighbours": int(l2g._nn),
                "local_block": list(blocks["local"]),
                "remote_block": list(blocks["remote"]),
                "local_form": {k_: A.plan_get(k_) for k_ in
                               ("lat", "lx", "lxw", "sdia", "sdia_const",
                                "sdia_tile")},
                "remote_algo": (A.plan_get("algo", remote=True)
                                if blocks["remote"][2] > 0 else None),
                "iterations": k,
                "ms_per_iteration": wall / steps * 1e3,
                "local_spmv_ms": spmv_ms / max(spmv_n, 1),
                # SpMV local + remote, p.Ap reducer, r update, r.r reducer,
                # x / p update (host/cg.cpp:271-298); one rank: 3
                "launches_per_iteration": 5 + (1 if blocks["remote"][2] > 0 else 0),
                "collectives_per_iteration": 3,  # halo exchange + 2 all-reduces
                "finite": bool(all(h == h for h in hist)),
            })
            ws.close()
            exec_.free(d_b), exec_.free(d_x)
        tw.bar.wait()
        A.close()
        for k, (_, d) in opts.items():
            _lib.call("spmv_hip_ctx_set_option", ctx, k, d)
        tw.bar.wait()

    tw.run(rank_body, gpu=True)
    out["plan"] = ("csr-order" if plain else "AUTO (specialised)")
    out["what"] = ("MODEL: one interior rank of %d at %d^3, no-op transport, alone on "
                   "one GPU -- kernels and kernel boundaries of the P > 1 launch "
                   "sequence, no collective latency" % (P, n))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--specialised", action="store_true",
                    help="the AUTO plan (constant diagonals) instead of CSR order")
    ap.add_argument("--both", action="store_true", help="both plans")
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    recs = {}
    plans = [True, False] if args.both else [not args.specialised]
    for plain in plans:
        for P in args.ranks:
            r = model(P, args.grid, args.steps, plain, symmetric=args.symmetric)
            recs[("csr_order" if plain else "specialised") + f"_P{P}"] = r
            print("rank_shape", json.dumps(r), file=sys.stderr, flush=True)
    doc = {"grid": args.grid, "steps": args.steps, "records": recs}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    print(json.dumps(doc), flush=True)


if __name__ == "__main__":
    main()
