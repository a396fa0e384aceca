"""Benchmark of the hot path: fp64 CG on the 3-D Poisson CSR matrix.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one CG iteration of spmv::cg (spmv/cg.cpp:55-86): halo update of
p, SpMV (local + remote block) with the fused p.Ap, x/r update with the fused
r.r, p update -- through the C++ host mirror and the HIP kernels.  The matrix
is the 512^3 7-point Poisson matrix (BASELINE.json configs[2]/[4]) split into
contiguous row slabs over the N ranks (strong scaling: total work is fixed),
generated on the device, inputs resident in HBM before the timed region.

Prints ONE JSON line on rank 0.  `value` = CG iterations per second of the
whole job; `roofline` describes the dominant kernel (the local-block CSR
SpMV), timed live with HIP events on its own stream inside the timed region;
`cpu_baseline` is the oracle's OpenMP CG (= the reference's CPU path,
restated) on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dmabuf IPC is the only mode the host driver supports; RCCL's intra-node
# transport fails with "hipIpcGetMemHandle: invalid argument" without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", dest="n", type=int, default=512,
                    help="grid points per side (not --n: torchrun's own parser "
                         "treats that as an abbreviation of its options)")
    ap.add_argument("--symmetric", action="store_true",
                    help="symmetric-CSR storage (BASELINE configs[3])")
    ap.add_argument("--cm", default="p2p_nonblocking",
                    choices=["p2p_blocking", "p2p_nonblocking"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"],
                    help="gloo: REHEARSAL ONLY -- halo and reductions staged "
                         "through the host so that several ranks can share one "
                         "GPU (implies --share-gpu); never a benchmark result")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks use GPU 0 (rehearsal on a 1-GPU box)")
    ap.add_argument("--cpu-n", type=int, default=256,
                    help="grid of the bounded CPU sample")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--reducer-kernels", action="store_true",
                    help="finish dot products with the single-workgroup reducer "
                         "launches even on one rank (experiments)")
    ap.add_argument("--no-lx", action="store_true",
                    help="do not build the LX form of the matrix (experiments)")
    ap.add_argument("--blas1-nt-min", type=int, default=None,
                    help="override the context option blas1_nt_min_elems "
                         "(experiments)")
    return ap.parse_args()


def cpu_baseline(args, n_gpu, rows_gpu):
    """Oracle OpenMP CG on a bounded sample, scaled by row count to the GPU
    workload (a CG iteration is O(rows) for this matrix)."""
    import numpy as np

    import oracle
    n = min(args.cpu_n, n_gpu)
    rp, ci, va = oracle.poisson3d(n)
    b = np.ones(n ** 3)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    oracle.time_cg(rp, ci, va, b, 1, threads)  # first touch / warm-up
    best = None
    for t in sorted({threads, 1}, reverse=True):
        secs, its = oracle.time_cg(rp, ci, va, b, args.cpu_iters, t)
        rate = its / secs * (n ** 3) / rows_gpu
        if best is None or rate > best[0]:
            best = (rate, t, secs)
    rate, t, secs = best
    out = {"value": rate, "unit": "iters/s", "cores": t, "kind": "port",
           "sample": (f"oracle OpenMP CG (restated spmv/openmp path), "
                      f"{n}^3 Poisson, {args.cpu_iters} iterations in "
                      f"{secs:.2f} s on {t} threads, scaled by rows "
                      f"{n ** 3}/{rows_gpu} to the {n_gpu}^3 workload")}
    # SURVEY 8d extras on the same sample: plain SpMV on the OpenMP path and
    # the 1-thread ReferenceExecutor-equivalent loop (BASELINE configs[0] at
    # 128^3), both in algorithmic GB/s (same formula as the GPU figure)
    try:
        from spmv_amd import poisson
        x = np.ones(n ** 3)
        nbytes = poisson.csr_bytes(n ** 3, n ** 3, len(va))
        s_omp = oracle.time_spmv(rp, ci, va, x, reps=10, num_threads=t)
        out["spmv_omp"] = {"grid": n, "threads": t, "ms_per_apply": s_omp * 1e3,
                           "GB/s": nbytes / s_omp / 1e9}
        if n > 128:
            rp, ci, va = oracle.poisson3d(128)
            x = np.ones(128 ** 3)
            nbytes = poisson.csr_bytes(128 ** 3, 128 ** 3, len(va))
        s_ref = oracle.time_spmv(rp, ci, va, x, reps=5, num_threads=1)
        out["spmv_reference_1thread"] = {"grid": min(n, 128),
                                         "ms_per_apply": s_ref * 1e3,
                                         "GB/s": nbytes / s_ref / 1e9}
        with open("/proc/cpuinfo") as f:
            models = [ln.split(":", 1)[1].strip() for ln in f
                      if ln.startswith("model name")]
        out["host"] = {"model": models[0] if models else "unknown",
                       "logical_cpus": os.cpu_count(), "usable": cores}
    except Exception as e:  # extras only; the baseline itself is above
        out["extras_error"] = repr(e)
    return out


def north_star_spmv(exec_, comm, host, _lib, poisson, n=216, reps=200):
    """BASELINE.json's target line: plain fp64 CSR SpMV (y = A x, the
    demos/spmv.cpp protocol: 1 warm-up + timed applies) on the ~10 M-row
    Poisson matrix, one GPU, HIP events around the applies."""
    import ctypes as C
    N = n ** 3
    A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_BLOCKING)
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    ctx = exec_.context
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_x, None)
    e0, e1 = C.c_void_p(), C.c_void_p()
    _lib.call("spmv_hip_event_create", ctx, 1, C.byref(e0))
    _lib.call("spmv_hip_event_create", ctx, 1, C.byref(e1))
    A.col_map().update(d_x)
    A.mult(d_x, d_y)  # warm-up
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
    nnz = A.non_zeros()
    nbytes = poisson.csr_bytes(N, N, nnz)
    _lib.call("spmv_hip_event_destroy", ctx, e0)
    _lib.call("spmv_hip_event_destroy", ctx, e1)
    A.close()
    exec_.free(d_x), exec_.free(d_y)
    gbs = nbytes / (best * 1e-3) / 1e9
    return {"workload": f"poisson3d_{n}^3_csr_fp64_spmv", "rows": N, "nnz": nnz,
            "ms_per_apply": best, "algorithmic_bytes": nbytes, "GB/s": gbs,
            "frac_of_8TBs": gbs / HBM_PEAK_GBS, "applies_timed": reps}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch "
                         "N>1 through torch.distributed.run")

    import numpy as np
    import torch
    import torch.distributed as dist

    from spmv_amd import _lib, host, poisson

    rehearsal = args.transport == "gloo"
    dev = 0 if (args.share_gpu or rehearsal) else local_rank
    torch.cuda.set_device(dev)
    exec_ = host.HipExecutor(dev)
    if world > 1 and rehearsal:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import dist_util  # gloo-backed CallbackComm transport (tests/)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ex, ar = dist_util.make_device_transport(exec_.context)
        comm = host.Comm.callback(rank, world, dist_util.make_allgather(world),
                                  ex, ar)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", dev))
        ident = [host.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        comm = host.Comm.rccl(exec_, world, rank, ident[0])
    else:
        comm = host.Comm.self_comm()

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.n
    N = n ** 3
    if args.no_lx:
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"lx_min_nnz",
                  1 << 62)
    cm = getattr(host, args.cm.upper())
    A = host.Matrix.create_poisson3d(comm, exec_, n, args.symmetric, cm)
    l2g = A.col_map()
    M = l2g.local_size()
    blocks = A.blocks()
    nnz_local = blocks["local"][2]

    # RHS b = Gaussian bump (demos/spmv.cpp:63-67) -- resident before timing
    ctx = exec_.context
    if args.blas1_nt_min is not None:
        _lib.call("spmv_hip_ctx_set_option", ctx, b"blas1_nt_min_elems",
                  args.blas1_nt_min)
    d_b, d_x = exec_.alloc(M), exec_.alloc(M)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M, d_b,
              None)
    ws = host.CgWorkspace(exec_)
    exec_.synchronize()

    # Several ranks: before anything is timed, prove the transport on this run's
    # own partition -- after one halo update of the vector x_i = g(i) every
    # ghost entry must hold g(its global index), and the locally owned part must
    # be untouched.  A wrong offset or a lost message fails here, loudly,
    # instead of producing a fast wrong number.
    if world > 1:
        ng = l2g.num_ghosts()
        d_v = exec_.alloc(M + ng)
        exec_.memset(d_v, 0xFF, 8 * (M + ng))  # NaN pattern in the ghost tail
        _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M,
                  d_v, None)
        l2g.update(d_v)
        l2g.update_finalise(d_v)  # non-blocking models complete here
        exec_.synchronize()
        got = exec_.copy_to_host(d_v, M + ng)
        gidx = np.asarray(l2g.ghosts(), dtype=np.float64)
        want = np.exp(-10 * (5 * (gidx / float(N) - 0.5)) ** 2)
        own = np.arange(l2g.global_offset(), l2g.global_offset() + M,
                        dtype=np.float64)
        own = np.exp(-10 * (5 * (own / float(N) - 0.5)) ** 2)
        bad = (not np.allclose(got[M:], want, rtol=1e-12, atol=1e-300)
               or not np.allclose(got[:M], own, rtol=1e-12, atol=1e-300))
        exec_.free(d_v)
        if bad:
            raise SystemExit(f"rank {rank}: halo self-check FAILED "
                             f"({ng} ghosts) -- refusing to benchmark")

    # warm-up: W untimed iterations (also sizes the workspace, RCCL rings)
    if args.warmup > 0:
        host.cg_ex(comm, exec_, A, d_b, d_x, args.warmup, 0.0, ws,
                   consumer_reductions=not args.reducer_kernels)
    ws.reserve_timing(args.steps)  # HIP events created outside the timed region
    torch.cuda.synchronize()
    barrier()

    # ---- timed region: exactly K iterations (rtol = 0 never converges) ----
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    k, hist, spmv_ms, spmv_launches = host.cg_ex(comm, exec_, A, d_b, d_x,
                                                 args.steps, 0.0, ws,
                                                 time_spmv=True, history=True,
                                                 consumer_reductions=not args.reducer_kernels)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert k == args.steps, (k, args.steps)

    # algorithmic bytes of the dominant kernel on THIS rank (DESIGN.md,
    # SURVEY 8d): entries*12 + (rows+1)*4 + x (cols*8) + y (rows*8)
    rows_b, cols_b, nnz_b = blocks["local"]
    if args.symmetric:
        kernel_bytes = poisson.sym_csr_bytes(rows_b, nnz_b)
        kernel = "csr_sym_window_kernel<double> (local lower block + diagonal)"
    else:
        kernel_bytes = poisson.csr_bytes(rows_b, cols_b, nnz_b)
        kernel = ("csr_rowblock_kernel<double> (local block, fused p.Ap)"
                  if args.no_lx else
                  "csr_rowblock_lx_kernel<double> (local block, LX form: x "
                  "windows staged in LDS, 16-bit column offsets; fused p.Ap)")
    iter_bytes = kernel_bytes + 9 * M * 8  # + fused BLAS-1 minimum, SURVEY 8d

    # max over ranks of the times, sum over ranks of the bytes
    if world > 1:
        dev_t = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed, spmv_ms / max(spmv_launches, 1)],
                         dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, spmv_ms_avg = float(t[0]), float(t[1])
        bsum = torch.tensor([kernel_bytes, iter_bytes], dtype=torch.float64,
                            device=dev_t)
        dist.all_reduce(bsum, op=dist.ReduceOp.SUM)
        kernel_bytes_all, iter_bytes_all = float(bsum[0]), float(bsum[1])
    else:
        spmv_ms_avg = spmv_ms / max(spmv_launches, 1)
        kernel_bytes_all, iter_bytes_all = kernel_bytes, iter_bytes

    if rank == 0:
        achieved = kernel_bytes / (spmv_ms_avg * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc) and world == 1 and n == 512:
            try:
                summary = json.load(open(pmc))
                if args.symmetric:
                    summary = summary.get("symmetric_kernel", {})
                elif args.no_lx:
                    summary = summary.get("gather_kernel", {})
                traffic = summary.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "fp64 CG iters/sec (SpMV effective GB/s vs HBM roofline)",
            "value": args.steps / elapsed,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (REHEARSAL: gloo transport, shared GPU)"
                                   if rehearsal else ""),
            "config": {"workload": f"poisson3d_{n}^3_csr_fp64_cg",
                       "rows": N, "nnz": poisson.poisson3d_nnz(n),
                       "storage": "symmetric-csr" if args.symmetric else "csr",
                       "partition": f"row-slab x{world}",
                       "halo": args.cm + " (RCCL send/recv on a side stream)"
                       if world > 1 else "none (1 rank)"},
            "roofline": {"bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel,
                         "algorithmic_bytes_per_launch": kernel_bytes,
                         "avg_launch_ms": spmv_ms_avg,
                         "launches_timed": spmv_launches},
            # whole-iteration effective bandwidth: SpMV + fused BLAS-1 minimum
            # (9 vectors of 8 B per row)
            # ||r_k|| / ||r_0|| from the device-side history: after 10
            # iterations the same number to ~1e-12 whatever the rank count
            # or storage (a cross-check of the distributed path); after all K,
            # where CG has amplified the different summation orders
            "cg_rel_residual": {"k10": float(hist[min(10, len(hist) - 1)] / hist[0]),
                                "kK": float(hist[-1] / hist[0])},
            "cg_gbs_per_gpu": iter_bytes / (elapsed / args.steps) / 1e9,
            # all ranks together: sum of bytes / time of the slowest rank
            "spmv_gbs_aggregate": kernel_bytes_all / (spmv_ms_avg * 1e-3) / 1e9,
            "cg_gbs_aggregate": iter_bytes_all / (elapsed / args.steps) / 1e9,
        }
        if world == 1 and not args.symmetric:
            out["north_star_spmv"] = north_star_spmv(exec_, comm, host, _lib,
                                                     poisson)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n, N)
        print(json.dumps(out), flush=True)

    ws.close()
    A.close()
    exec_.free(d_b), exec_.free(d_x)
    comm.close()
    exec_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
