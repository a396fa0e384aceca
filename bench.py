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
`cpu_baseline` is the oracle's OpenMP SpMV + CG (= the reference's CPU path,
restated) on the host cores this job may use, rank 0 at N = 1 only.

What the headline times (round 6).  `value`, `ms_per_step` and `roofline`
belong to the CSR-ORDER plan: the matrix as a general CSR matrix, the
plan-time lattice analysis (and with it every stencil form) switched off, so
that the SpMV streams every stored value and an index per entry -- the kernel
any banded CSR matrix of this shape gets (the LX form: values + 16-bit column
offsets by LDS-DMA, x windows staged).  `roofline.frac` = SURVEY 8d's
algorithmic bytes B_csr (12 B per entry, row pointer, x, y) / the kernel's
average launch time measured live / 8 TB/s; `bytes_per_launch` = B_csr; and
(B_csr + 9 N 8) / ms_per_step stays below the peak.  What the AUTO plan does
with THIS matrix -- it finds the lattice and that every diagonal is constant,
and streams no matrix at all -- is faster and is reported beside it as
`roofline.specialised` (`--specialised` makes it the main line, as in rounds
1-5): there `frac_physical` prices the bytes that kernel's own format moves and
`frac_csr_equivalent` prices B_csr over its time -- a speed-up statement that
exceeds 1, not a bandwidth.  Likewise `symmetric` (BASELINE configs[3]) is the
symmetric storage with its values STREAMED, priced with B_sym, the
constant-diagonal run beside it.
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
# the CPU baseline's threads: one per core, spread over the sockets (read by
# the OpenMP runtime when the oracle library is loaded)
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "spread")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# ||r_10|| / ||r_0|| of the 512^3 Gaussian right-hand side on ONE rank: what
# every N-rank line must reproduce (to the rounding of its dot products)
K10_512 = 6.635343844806121


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", dest="n", type=int, default=512,
                    help="grid points per side (not --n: torchrun's own parser "
                         "treats that as an abbreviation of its options)")
    ap.add_argument("--symmetric", action="store_true",
                    help="symmetric-CSR storage (BASELINE configs[3]) as the "
                         "main line (the default line carries it as a "
                         "sub-record)")
    ap.add_argument("--peer-reduce", action="store_true",
                    help="N > 1: the two scalar reductions of a CG iteration by the "
                         "deterministic peer reduction (Comm::enable_peer_reduce: "
                         "one single-wave kernel per rank, values added in rank "
                         "order) instead of RCCL's all-reduce.  Opt-in: validated "
                         "on one device only")
    ap.add_argument("--cm", default="p2p_nonblocking",
                    choices=["p2p_blocking", "p2p_nonblocking",
                             "onesided_put_active"],
                    help="halo model; onesided_put_active: peer stores into "
                         "IPC-mapped windows instead of RCCL send/recv "
                         "(DESIGN.md section 6; never yet run over xGMI)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="main line only: no symmetric / LX / 216^3 sub-records")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"],
                    help="gloo: REHEARSAL ONLY -- halo and reductions staged "
                         "through the host so that several ranks can share one "
                         "GPU (implies --share-gpu); never a benchmark result")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks use GPU 0 (rehearsal on a 1-GPU box)")
    ap.add_argument("--cpu-n", type=int, default=0,
                    help="grid of the CPU baseline (0 = the benchmark's own, "
                         "or 256 when host memory is short)")
    ap.add_argument("--cpu-iters", type=int, default=100)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="0 = physical cores this job may use")
    ap.add_argument("--reducer-kernels", action="store_true",
                    help="finish dot products with the single-workgroup reducer "
                         "launches even on one rank (experiments)")
    ap.add_argument("--specialised", action="store_true",
                    help="the main line on the AUTO plan of the Poisson matrix "
                         "(lattice + constant diagonals: no matrix stream), as in "
                         "rounds 1-5; default: the CSR-order plan, the "
                         "specialised run in roofline.specialised")
    ap.add_argument("--no-lattice", action="store_true",
                    help="(the default since round 6) no lattice form: the LX "
                         "form as any matrix without lattice structure gets it")
    ap.add_argument("--no-bake", action="store_true",
                    help="no plan-time symmetry check of the general matrix: the "
                         "lattice kernel on the caller's CSR values (experiments)")
    ap.add_argument("--no-const", action="store_true",
                    help="stream the matrix values even though every diagonal of "
                         "the Poisson matrix is constant: the diagonal form as a "
                         "lattice matrix with varying coefficients gets it")
    ap.add_argument("--no-lx", action="store_true",
                    help="with --no-lattice: the plain gather kernel")
    ap.add_argument("--mixed-grid", type=int, default=216,
                    help="grid of the mixed-precision CG sub-record (512 takes "
                         "~15 s more)")
    ap.add_argument("--stencil27-grid", type=int, default=256,
                    help="grid of the 27-point sub-record")
    ap.add_argument("--unstructured-rows", type=int, default=10_000_000,
                    help="rows of the unstructured sub-record")
    ap.add_argument("--fem-rows", type=int, default=10_000_000,
                    help="rows of the ragged-row (FEM-like) sub-records")
    ap.add_argument("--petsc-matrix", default=None,
                    help="PETSc binary matrix file (spmv/read_petsc.cpp:40-228, as "
                         "demos/cg.cpp:47 reads it): the main line's matrix instead "
                         "of the generated Poisson matrix")
    ap.add_argument("--petsc-rhs", default=None,
                    help="PETSc binary vector file for the right-hand side "
                         "(demos/cg.cpp:51); default: the Gaussian vector")
    ap.add_argument("--rank-shape", type=int, nargs="*", default=None, metavar="P",
                    help="MODEL, instead of the benchmark: one interior rank of "
                         "P (default 2 4 8) ranks at --grid, alone on this GPU, "
                         "the P > 1 launch sequence of cg() with a no-op "
                         "transport -- ms per iteration and launches per "
                         "iteration of the per-rank critical path "
                         "(tools/rank_shape.py; DESIGN.md section 6)")
    ap.add_argument("--put-timeout-ms", type=int, default=None,
                    help="bound of the waits of the one-sided halo and of the peer "
                         "reduction (ctx option put_timeout_ms; default 60 s)")
    ap.add_argument("--detail", default=None,
                    help="where the full record goes (every sub-record, plan "
                         "costs, cross-checks, notes); default "
                         "gpurun_out/bench_detail.json.  stdout carries ONE "
                         "compact line (<= 6 KB) made from it")
    ap.add_argument("--blas1-nt-min", type=int, default=None,
                    help="override the context option blas1_nt_min_elems "
                         "(experiments)")
    return ap.parse_args()


# ---------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------
def usable_cores():
    """Physical cores this process may keep busy: affinity mask, distinct
    (socket, core) pairs, and the cgroup CPU quota."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    phys, model = set(), "unknown"
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ":" in ln:
                    k, v = (s.strip() for s in ln.split(":", 1))
                    cur[k] = v
                elif not ln.strip():
                    if int(cur.get("processor", -1)) in cpus:
                        phys.add((cur.get("physical id", "0"),
                                  cur.get("core id", cur.get("processor"))))
                        model = cur.get("model name", model)
                    cur = {}
    except OSError:
        pass
    cores = len(phys) or len(cpus)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        pass
    n = cores if quota is None else max(1, min(cores, int(quota)))
    return dict(threads=n, physical_cores=cores, logical_cpus=len(cpus),
                cgroup_cpu_quota=quota, model=model)


def cpu_baseline(args, n_gpu, rows_gpu, host):
    """The oracle's OpenMP path (restated spmv/openmp/*) on the benchmark's own
    matrix: first touch by the owning thread, threads = usable physical cores
    spread over the sockets, only the apply / iteration loops timed.  `host` =
    usable_cores(), taken before any OpenMP runtime bound this thread."""
    import oracle
    from spmv_amd import poisson
    threads = args.cpu_threads or host["threads"]
    n = args.cpu_n or n_gpu
    try:
        with open("/proc/meminfo") as f:
            avail_kib = next(int(ln.split()[1]) for ln in f
                             if ln.startswith("MemAvailable"))
    except (OSError, StopIteration):
        avail_kib = 0
    need = 12 * 7 * n ** 3 + 44 * n ** 3  # CSR + rowptr + five vectors
    if not args.cpu_n and avail_kib * 1024 < 2 * need:
        n = min(n, 256)
    r = oracle.cpu_baseline(n, threads, 5, args.cpu_iters)
    N, nnz = n ** 3, 7 * n ** 3 - 6 * n ** 2
    nbytes = poisson.csr_bytes(N, N, nnz)
    its = r["cg_iters"] / r["cg_loop_s"]
    scaled = its * N / rows_gpu
    out = {"value": scaled, "unit": "iters/s", "cores": threads, "kind": "port",
           "sample": (f"oracle OpenMP CG (restated spmv/openmp path) on the "
                      f"{n}^3 Poisson matrix itself, {r['cg_iters']} iterations "
                      f"in {r['cg_loop_s']:.2f} s (loop only; set-up "
                      f"{r['setup_s']:.1f} s with owner first touch), "
                      f"{threads} threads = the physical cores this job may "
                      f"use, OMP_PLACES=cores OMP_PROC_BIND=spread"
                      + ("" if n == n_gpu else
                         f"; scaled by rows {N}/{rows_gpu} to {n_gpu}^3")),
           "sample_short": (f"oracle OpenMP CG on the {n}^3 Poisson matrix, "
                            f"{r['cg_iters']} iterations in {r['cg_loop_s']:.2f} s, "
                            f"{threads} threads"
                            + ("" if n == n_gpu else f", scaled by rows to {n_gpu}^3")),
           "spmv_omp": {"grid": n, "threads": threads,
                        "ms_per_apply": r["spmv_s_per_apply"] * 1e3,
                        "GB/s": nbytes / r["spmv_s_per_apply"] / 1e9,
                        "bytes": "algorithmic CSR bytes, as for the GPU"},
           "cg_rel_residual_after": r["rel_residual"],
           # the same system as the GPU line (Gaussian right-hand side): compare
           # with the line's cg_rel_residual.k10 when the grids are the same
           "cg_rel_residual_k10": r["rel_residual_k10"] or None,
           "rhs": "b_i = exp(-10 (5 (i/N - 1/2))^2), as the GPU line",
           "host": host}
    try:  # BASELINE configs[0]: the 1-thread ReferenceExecutor loop at 128^3
        import numpy as np
        rp, ci, va = oracle.poisson3d(128)
        s_ref = oracle.time_spmv(rp, ci, va, np.ones(128 ** 3), reps=5,
                                 num_threads=1)
        nb = poisson.csr_bytes(128 ** 3, 128 ** 3, len(va))
        out["spmv_reference_1thread"] = {"grid": 128, "ms_per_apply": s_ref * 1e3,
                                         "GB/s": nb / s_ref / 1e9}
    except Exception as e:  # extras only
        out["extras_error"] = repr(e)
    return out


def oracle_parity_checks(exec_, comm, host, _lib):
    """Part of the cpu_baseline leg (the only place bench.py may call the
    oracle): oracle-sized instances of the sub-records' matrices through the
    product path, compared bit for bit with the oracle's reference loop
    (csr_kernels.cpp:41-51) -- the same checks tests/test_gpu_matrix.py makes."""
    import numpy as np
    import oracle
    from spmv_amd import poisson
    ctx = exec_.context
    out = {}

    def run(A, x):
        N = len(x)
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        exec_.memset(d_y, 0xFF, 8 * N)
        A.mult(d_x, d_y)
        y = exec_.copy_to_host(d_y, N)
        exec_.free(d_x), exec_.free(d_y)
        return y
    try:
        n = 33
        rp, ci, va = poisson.stencil27_csr(n)
        x = oracle.gaussian_x_fast(n ** 3) + 0.25
        y_ref = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
        for k, v in ((b"poisson_stencil", 27), (b"lat_min_nnz", 0),
                     (b"lx_min_nnz", 0)):
            _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
        try:
            A = host.Matrix.create_poisson3d(comm, exec_, n, False,
                                             host.P2P_BLOCKING)
        finally:
            for k, v in ((b"poisson_stencil", 7), (b"lat_min_nnz", 1 << 20),
                         (b"lx_min_nnz", 1 << 20)):
                _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
        out["stencil27_33^3"] = {"kernel": kernel_of(A, False)[0].split(" ")[0],
                                 "bit_exact_vs_oracle":
                                     bool(np.array_equal(run(A, x), y_ref))}
        A.close()
        N = 300_000
        rp, ci, va = poisson.unstructured_csr(N)
        x = oracle.gaussian_x_fast(N) + 0.25
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        _lib.call("spmv_hip_ctx_set_option", ctx, b"lx_min_nnz", 0)
        try:
            A = host.Matrix.create_unstructured(comm, exec_, N)
        finally:
            _lib.call("spmv_hip_ctx_set_option", ctx, b"lx_min_nnz", 1 << 20)
        out["unstructured_300000"] = {
            "kernel": kernel_of(A, False)[0].split(" ")[0],
            "bit_exact_vs_oracle": bool(np.array_equal(run(A, x), y_ref))}
        A.close()
    except Exception as e:  # extras only
        out["error"] = repr(e)
    return out


# ---------------------------------------------------------------------------
# what the local block's plan turned into, and the bytes that form moves
# ---------------------------------------------------------------------------
def kernel_of(A, symmetric):
    rows, cols, nnz = A.blocks()["local"]
    nrb = (rows + 255) // 256
    y_x = rows * 8 + cols * 8
    if symmetric:
        algo = rows * 8 * 3 + (rows + 1) * 4 + nnz * 12  # SURVEY 8d B_sym
        if A.plan_get("sdia") and A.plan_get("sdia_const"):
            R = A.plan_get("sdia_tile")
            name = (f"csr_const_dia_tile_kernel<double, {R} lattice lines per lane>"
                    if R > 1 else "csr_const_dia_kernel<double>")
            return (name + " (symmetric storage whose "
                    "diagonals are constant bit for bit: the plan keeps one "
                    "number per diagonal and a mask byte per row, no values are "
                    "streamed; same products and sums in the reference's order, "
                    "atomic-free, bit-exact)",
                    algo, rows * 1 + y_x)
        if A.plan_get("sdia"):
            nd = A.plan_get("sdia_offsets")
            return ("csr_sym_dia_kernel<double> (symmetric diagonal form: the "
                    "plan's copy of the values by offset, own and column windows "
                    "by LDS-DMA, no index stream, atomic-free, bit-exact)",
                    algo, rows * (8 * nd + 8 + 1) + y_x)
        if A.plan_get("slat"):
            return ("csr_sym_lattice_kernel<double> (symmetric lattice form: own "
                    "and column value windows by LDS-DMA, no index stream, "
                    "atomic-free, bit-exact)",
                    algo, nnz * 8 + rows * (4 + 1 + 8) + y_x)
        if A.plan_get("sym_sj"):
            wpb = A.plan_get("sj_wpb")
            nlong = A.plan_get("sj_long_rows")
            return (f"csr_sjds_kernel<double, {wpb} slices per block, symmetric "
                    "storage>"
                    + (f" + csr_sjds_longt_kernel ({nlong} long rows of the stored "
                       "block: their lower part from the caller's arrays, before "
                       "the slices)" if nlong else "")
                    + " (no lattice structure: the merged matrix -- per row "
                    "its stored lower entries, then its column's entries in the "
                    "reference's order -- in the sliced jagged form: the plan's "
                    "copy of the values and 16-bit column codes, x staged in LDS; "
                    "one pass, the sum turning into y = alpha (d x + L x) + beta y "
                    "where the column's entries begin; atomic-free, bit-exact; "
                    "fused p.Ap)",
                    # every stored entry twice (value + code); long rows: their
                    # lower part once from the CSR arrays (12 B) + once merged
                    algo, nnz * 20 + rows * (4 + 8 + 8 + 8 + 8) + cols * 8)
        if A.plan_get("sym_det"):
            return ("csr_symt_kernel<double> (transposed map, atomic-free, "
                    "bit-exact)", algo, algo + (rows + 1) * 4 + nnz * 8)
        return ("csr_sym_window_kernel<double> (LDS window + global atomics)",
                algo, algo)
    algo = nnz * 12 + (rows + 1) * 4 + y_x  # SURVEY 8d B_csr
    if A.plan_get("wdia") and A.plan_get("wdia_const") and A.plan_get("wdia_box"):
        R = A.plan_get("wdia_box")
        return (f"csr_box27_const_kernel<double, {R} lattice lines per lane> (a "
                "27-point box stencil whose diagonals are constant bit for bit: "
                "the plan keeps 27 numbers and a 32-bit presence mask per row, no "
                "values are streamed; a lane keeps the lines around its rows in "
                "registers and hands them on from plane to plane; rows summed in "
                "the CSR kernel's order: bit-exact; fused p.Ap)",
                algo, rows * 4 + y_x)
    if A.plan_get("wdia") and A.plan_get("wdia_const"):
        K = A.plan_get("wdia_offsets")
        return (f"csr_wdia_kernel<double, constant> (wide diagonal form: the "
                f"matrix sits on {K} diagonals and every one of them is constant "
                "bit for bit; the plan keeps one number per diagonal and a 32-bit "
                "presence mask per row, no values are streamed; rows summed in "
                "the CSR kernel's order: bit-exact; fused p.Ap)",
                algo, rows * 4 + y_x)
    if A.plan_get("wdia") and A.plan_get("wdia_half") and A.plan_get("wdia_hbox"):
        return ("csr_box27_half_kernel<double> (27-point box with varying "
                "coefficients, found symmetric bit for bit: the half form -- 13 "
                "lower diagonals + the diagonal + a 32-bit mask per row -- walked "
                "down the planes in tiles of 1024 rows, the plane's values handed "
                "on through LDS (the upper entries are the plane above's lower "
                "ones), x from an LDS ring of three plane windows; rows summed in "
                "the CSR kernel's order: bit-exact; fused p.Ap)",
                algo, rows * (8 * 14 + 4) + y_x)
    if A.plan_get("wdia") and A.plan_get("wdia_half"):
        K = A.plan_get("wdia_offsets")
        return (f"csr_wdia_kernel<double, half> (wide diagonal form: the matrix "
                f"sits on {K} diagonals and was found symmetric bit for bit; the "
                f"plan keeps the values of the {(K + 1) // 2} diagonals <= 0 and a "
                "32-bit presence mask per row, an upper entry is read as the "
                "lower entry of its column's row; rows summed in the CSR "
                "kernel's order: bit-exact; fused p.Ap)",
                algo, rows * (8 * ((K + 1) // 2) + 4) + y_x)
    if A.plan_get("wdia"):
        K = A.plan_get("wdia_offsets")
        return (f"csr_wdia_kernel<double> (wide diagonal form: the matrix sits on "
                f"{K} diagonals; the plan keeps its values by offset and a 32-bit "
                "presence mask per row; every load coalesced, no index stream, "
                "rows summed in the CSR kernel's order: bit-exact; fused p.Ap)",
                algo, rows * (8 * K + 4) + y_x)
    if A.plan_get("sdia") and A.plan_get("sdia_const"):
        R = A.plan_get("sdia_tile")
        name = (f"csr_const_dia_tile_kernel<double, general order, {R} lattice "
                "lines per lane>" if R > 1
                else "csr_const_dia_kernel<double, general order>")
        return (name + " (every diagonal of "
                "the matrix is constant bit for bit: the plan keeps one number "
                "per diagonal and a mask byte per row, no values are streamed; "
                "the kernel does the CSR kernel's multiplications and additions "
                "in its order with the constant in a register: bit-exact; fused "
                "p.Ap)",
                algo, rows * 1 + y_x)
    if A.plan_get("sdia") and A.plan_get("sdia_general") == 2:
        nd = A.plan_get("sdia_offsets")
        return ("csr_sym_dia_kernel<double, general order, full> (lattice matrix "
                "that is not symmetric: the plan keeps ALL its values by offset, "
                "windows by LDS-DMA, no index stream, rows summed in the CSR "
                "kernel's order: bit-exact; fused p.Ap)",
                algo, rows * (8 * (2 * nd + 1) + 1) + y_x)
    if A.plan_get("sdia"):
        nd = A.plan_get("sdia_offsets")
        return ("csr_sym_dia_kernel<double, general order> (the general matrix "
                "was found symmetric bit for bit at plan time: the plan keeps "
                "its lower half + diagonal by offset, windows by LDS-DMA, no "
                "index stream, rows summed in the CSR kernel's order: "
                "bit-exact; fused p.Ap)",
                algo, rows * (8 * nd + 8 + 1) + y_x)
    if A.plan_get("lat"):
        return ("csr_lattice_kernel<double> (lattice form: constant column "
                "offsets per row block, values by LDS-DMA one block ahead, no "
                "index stream; fused p.Ap)",
                algo, nnz * 8 + rows * 1 + nrb * 48 + y_x)
    if A.plan_get("sjds"):
        E, wpb = A.plan_get("sj_unit"), A.plan_get("sj_wpb")
        stored = nnz * A.plan_get("sj_pad_permille") // 1000  # short rows, padded
        code = 4 if A.plan_get("sj_wide") else 2
        staged = nnz * A.plan_get("sj_staged_bytes_per_entry_x100") // 100
        nlong = A.plan_get("sj_long_rows")
        return (f"csr_sjds_kernel<double, {wpb} slices per block, {E} entries per "
                "lane and step> (sliced jagged form: lane = row in 64-row slices "
                "stored as jagged diagonals, the plan's copy of the values and "
                f"{8 * code}-bit column codes, x staged in LDS per block"
                + (f", {nlong} long rows by 8-lane groups from the CSR arrays"
                   if nlong else "")
                + "; rows summed in the CSR kernel's order: bit-exact; fused p.Ap)",
                # the padded short rows (long rows: their CSR entries, counted in
                # `stored` at 12 B below is close enough: their share of the
                # values copy is never read), lenperm + slice bases, x, y, the
                # staged chunks (served by the L2s)
                algo, stored * (8 + code) + rows * 4 + (rows // 64 + 1) * 4
                + y_x + staged)
    if A.plan_get("lx") and A.plan_get("lxw"):
        return ("csr_lxw_kernel<double> (LX form, LDS-DMA kernel: values, 16-bit "
                "column offsets and x windows arrive by LDS-DMA one row block "
                "ahead; fused p.Ap)",
                algo, nnz * 10 + (rows + 1) * 4 + nrb * 80 + y_x)
    if A.plan_get("lx"):
        return ("csr_rowblock_lx_kernel<double> (LX form: x windows staged in "
                "LDS, 16-bit column offsets; fused p.Ap)",
                algo, nnz * 10 + (rows + 1) * 4 + nrb * 144 + y_x)
    if A.plan_get("xw") and A.plan_get("xw_pick") == 0:
        return ("csr_rowblock_kernel<double> (gather; the plan's first launches "
                "timed it against the XW kernel -- x windows staged over the same "
                "arrays -- and found it faster on this device: "
                f"{A.plan_get('xw_probe_gather_us')} against "
                f"{A.plan_get('xw_probe_xw_us')} us; fused p.Ap)", algo, algo)
    if A.plan_get("xw"):
        return ("csr_lxw_kernel<double, XW> (the caller's CSR arrays as they "
                "are: values and 32-bit column indices by LDS-DMA one row block "
                "ahead, the row block's x windows staged in LDS, a column turned "
                "into its staged position by the block's window list; fused p.Ap)",
                algo, algo + nrb * 144)
    return ("csr_rowblock_kernel<double> (gather; fused p.Ap)", algo, algo)


def plan_record(A):
    rows, cols, nnz = A.blocks()["local"]
    return {"plan_ms": A.plan_get("plan_us") / 1e3,
            # ... of it inside hipMalloc / hipFree (the plan's own arrays, the
            # scratch of its analysis): where a stalled allocator would show
            "plan_mem_ms": A.plan_get("plan_mem_us") / 1e3,
            "plan_extra_bytes": A.plan_get("plan_kib") * 1024,
            # the caller's CSR arrays the plan's memory comes on top of
            "csr_bytes": nnz * 12 + (rows + 1) * 4,
            "form": {k: A.plan_get(k) for k in
                     ("lat", "lx", "lxw", "xw", "sjds", "sym_sj", "wdia", "wdia_const",
                      "wdia_hbox", "slat", "sdia", "sdia_const", "sym_det", "zwalk")}}


def pmc_traffic(record, kernel_name, n, world):
    """HBM-side bytes per launch of this record's kernel from the committed PMC
    passes (profiles/): a constant of an EARLIER run on another box, returned
    with its source, or (None, None) when no pass covers it.  Round 3's summary
    is keyed by bench record; round 2's by kernel name and grid."""
    if world != 1:
        return None, None
    try:
        for rnd in ("r06", "r05", "r04", "r03"):
            path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")
            if os.path.exists(path):
                rec = json.load(open(path)).get("records", {}).get(record)
                if rec and rec.get("grid") == n:
                    return rec["fabric_bytes_per_launch"], rec["source"]
        path = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")
        if record in ("main", "symmetric") and os.path.exists(path):
            for rec in json.load(open(path))["kernels"]:
                if rec["grid"] == n and kernel_name.startswith(rec["kernel_prefix"]):
                    return rec["fabric_bytes_per_launch"], rec["source"]
    except Exception:
        pass
    return None, None


def price(ms, algo_bytes, req_bytes, traffic=None):
    """the roofline fields of one launch (see the module docstring)"""
    gbs = req_bytes / ms / 1e6
    out = {"GB/s": gbs, "frac": gbs / HBM_PEAK_GBS,
           "requested_bytes": req_bytes, "frac_requested": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes": algo_bytes,
           "csr_equivalent_gbs": algo_bytes / ms / 1e6,
           "frac_csr_equivalent": algo_bytes / ms / 1e6 / HBM_PEAK_GBS}
    if traffic:
        out["traffic"] = traffic
        out["frac_traffic"] = traffic / ms / 1e6 / HBM_PEAK_GBS
    return out


def timed_spmv(exec_, A, N, _lib, reps, crosscheck=False):
    """plain y = A x applies (demos/spmv.cpp:73-96 protocol), best of 3 rounds
    of `reps`, HIP events on the executor's stream.  crosscheck: afterwards the
    same product with the plan switched to the one-lane-per-row kernel (the
    reference loop verbatim) -- returns (ms, bit_equal)."""
    import ctypes as C
    ctx = exec_.context
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
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
    _lib.call("spmv_hip_event_destroy", ctx, e0)
    _lib.call("spmv_hip_event_destroy", ctx, e1)
    same = None
    if crosscheck:
        import numpy as np
        y = exec_.copy_to_host(d_y, N)
        algo0, sjds0 = A.plan_get("algo"), A.plan_get("sjds")
        if crosscheck == "symt":  # symmetric storage: the transposed-map kernel
            A.plan_set("sjds", 0)
        else:
            A.plan_set("algo", 3)  # SPMV_HIP_ALGO_SCALAR
        exec_.memset(d_y, 0xFF, 8 * N)
        A.mult(d_x, d_y)
        same = bool(np.array_equal(y, exec_.copy_to_host(d_y, N))
                    and np.isfinite(y).all())
        if crosscheck == "symt":  # ... timed too: what the form replaces
            exec_.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                A.mult(d_x, d_y)
            exec_.synchronize()
            timed_spmv.other_ms = (time.perf_counter() - t0) * 1e3 / 5
            if sjds0:
                A.plan_set("sjds", sjds0)
        else:
            A.plan_set("algo", algo0)
    exec_.free(d_x), exec_.free(d_y)
    return (best, same) if crosscheck else best


def spmv_record(exec_, comm, host, _lib, n, symmetric, reps, lattice=True,
                bake=True, skew_ppm=0, lx=True, record=None, stencil=7,
                const=True, sj=True, xw=True):
    """one plain-SpMV sub-record on the n^3 matrix in the given storage/form
    (skew_ppm: the generator's non-symmetric variant of the matrix; stencil 27:
    the 27-point operator)"""
    ctx = exec_.context
    opts = {b"poisson_skew_ppm": (skew_ppm, 0), b"poisson_stencil": (stencil, 7)}
    if not lattice:
        opts[b"lat_min_nnz"] = (1 << 62, 1 << 20)
    if not lx:
        opts[b"lx_min_nnz"] = (1 << 62, 1 << 20)
    if not sj:  # nor the sliced jagged form: the plain row-block gather kernel
        opts[b"sj_min_nnz"] = (1 << 62, 1 << 20)
    if not xw:  # ... nor staged x windows: the gather kernel
        opts[b"xw_min_nnz"] = (1 << 62, 1 << 20)
    if not bake:
        opts[b"bake_general"] = (0, 1)
    if not const:  # stream the values even where the diagonals are constant
        opts[b"const_diagonals"] = (0, 1)
    for k, (v, _) in opts.items():
        _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
    try:
        A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                         host.P2P_BLOCKING)
    finally:
        for k, (_, d) in opts.items():
            _lib.call("spmv_hip_ctx_set_option", ctx, k, d)
    name = f"poisson3d_{n}^3" if stencil == 7 else f"stencil27_{n}^3"
    rec = matrix_spmv_record(
        exec_, A, _lib, symmetric, reps, record, n,
        f"{name}_{'symmetric-csr' if symmetric else 'csr'}_fp64_spmv"
        + (f"_skew{skew_ppm}ppm" if skew_ppm else "")
        + ("" if const else "_values_streamed"), crosscheck=stencil != 7)
    A.close()
    return rec


def matrix_spmv_record(exec_, A, _lib, symmetric, reps, record, grid, workload,
                       crosscheck=False):
    rows, cols, nnz = A.blocks()["local"]
    kernel, algo, req = kernel_of(A, symmetric)  # before any plan_set
    plan = plan_record(A)
    r = timed_spmv(exec_, A, rows, _lib, reps, crosscheck)
    ms, same = r if crosscheck else (r, None)
    if A.plan_get("xw"):  # XW or gather: the plan's first launches decided
        kernel, algo, req = kernel_of(A, symmetric)
    traffic, source = pmc_traffic(record, kernel, grid, 1)
    rec = {"workload": workload, "rows": rows, "nnz_stored": nnz,
           "kernel": kernel, "ms_per_apply": ms, "applies_timed": reps}
    rec.update(price(ms, algo, req, traffic))
    if traffic:
        rec["traffic_source"] = source
    if same is not None:
        rec["crosscheck"] = {"against": ("csr_symt_kernel (transposed map: the "
                                         "reference loop of csr_kernels.cpp:26-40 "
                                         "seen from the row) on the same matrix "
                                         "and x" if crosscheck == "symt" else
                                         "csr_scalar_kernel (one lane per row, the "
                                         "reference loop of csr_kernels.cpp:41-51 "
                                         "verbatim) on the same matrix and x"),
                             "bit_equal": same}
    rec.update(plan)
    return rec


def cg_record(exec_, comm, host, _lib, A, N, steps, symmetric, record, n, workload):
    """`steps` timed CG iterations (3 warm-up) on matrix A with the Gaussian
    right-hand side: iterations/s, the SpMV kernel's live-timed launches and
    their pricing (SURVEY 8d's bytes and the format's own)."""
    ctx = exec_.context
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_b, None)
    ws = host.CgWorkspace(exec_)
    host.cg_ex(comm, exec_, A, d_b, d_x, 3, 0.0, ws)
    ws.reserve_timing(steps)
    exec_.synchronize()
    t0 = time.perf_counter()
    _, h, ms, launches = host.cg_ex(comm, exec_, A, d_b, d_x, steps, 0.0, ws,
                                    time_spmv=True, history=True)
    exec_.synchronize()
    el = time.perf_counter() - t0
    kern, algo, req = kernel_of(A, symmetric)
    ms /= max(launches, 1)
    tr, src = pmc_traffic(record, kern, n, 1)
    rec = {"workload": workload, "iters/s": steps / el, "steps": steps,
           "ms_per_step": el / steps * 1e3, "kernel": kern, "avg_launch_ms": ms,
           "launches_timed": launches,
           "cg_rel_residual_k10": float(h[min(10, len(h) - 1)] / h[0])}
    rec.update(price(ms, algo, req, tr))
    rec["traffic_source"] = src
    rec.update(plan_record(A))
    ws.close()
    exec_.free(d_b), exec_.free(d_x)
    return rec



def mixed_precision_record(exec_, comm, host, _lib, n, rtol=1e-10, kmax=6000):
    """the same solve (Gaussian right-hand side, to rtol) with
    the fp64 values and with CgOptions::mixed -- iterations, wall time, the
    TRUE final residual and the distance between the two solutions."""
    import numpy as np
    N = n ** 3
    ctx = exec_.context
    # both legs on the diagonal form WITH its values streamed (constant-diagonal
    # detection off -- with it no matrix value is read in either precision and
    # there is nothing for fp32 to save): fp64 copy of the values against the
    # fp32 copy (33 against 17 B of matrix data per row)
    _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals", 0)
    try:
        A = host.Matrix.create_poisson3d(comm, exec_, n, False,
                                         host.P2P_NONBLOCKING)
    finally:
        _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals", 1)
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_b, None)
    ws = host.CgWorkspace(exec_)
    rec = {"workload": f"poisson3d_{n}^3_csr_cg_to_{rtol:g}_values_streamed",
           "rows": N}
    sols = {}
    for name in ("fp64", "mixed"):
        for rep in range(2):  # first pass: warm-up (fp32 copy, workspace)
            exec_.synchronize()
            t0 = time.perf_counter()
            if name == "fp64":
                k, hist, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, kmax, rtol,
                                           ws, history=True)
                st = {}
            else:
                k, hist, st = host.cg_mixed(comm, exec_, A, d_b, d_x, kmax, rtol,
                                            replace_every=50, workspace=ws)
            exec_.synchronize()
            secs = time.perf_counter() - t0
        sols[name] = exec_.copy_to_host(d_x, N)
        rec[name] = {"iterations": k, "seconds": secs, "iters/s": k / secs,
                     "recurrence_rel_residual": float(hist[-1] / hist[0])}
        rec[name].update({k_: st[k_] for k_ in
                          ("replacements", "true_rel_residual",
                           "continuation_iterations",
                           "final_true_rel_residual") if k_ in st})
    rec["speedup"] = rec["fp64"]["seconds"] / rec["mixed"]["seconds"]
    rec["kernel"] = ("csr_sym_dia_kernel, fp64 / fp32 copy of the values"
                     if A.plan_get("sdia_mixed") else
                     "fp64: " + kernel_of(A, False)[0].split(" ")[0]
                     + "; mixed: CSR-order fp32 values")
    rec["x_rel_diff"] = float(np.linalg.norm(sols["mixed"] - sols["fp64"])
                              / np.linalg.norm(sols["fp64"]))
    rec["note"] = ("fp32 copy of the matrix values in the SpMV, fp64 vectors and "
                   "arithmetic, residual replacement every 50 iterations; the "
                   "Poisson values are exact in fp32, so the iteration differs "
                   "from the fp64 one only at the replaced residuals")
    ws.close()
    A.close()
    exec_.free(d_b), exec_.free(d_x)
    return rec


# ---------------------------------------------------------------------------
# the line the driver reads: contract keys + roofline + cpu_baseline, <= 6 KB;
# everything else lives in the detail file
# ---------------------------------------------------------------------------
LINE_MAX = 6144


def sig(x, n=6):
    """a float with n significant digits (None stays None)"""
    return None if x is None else float(f"{float(x):.{n}g}")


def kname(kernel):
    """the kernel's name without its description"""
    return kernel.split(" (")[0]


def compact_line(out, detail_path):
    r = out["roofline"]
    line = {k: out[k] for k in
            ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
             "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = out["config"]
    roof = {"bound": r["bound"], "achieved": sig(r["achieved"]), "peak": r["peak"],
            "unit": r["unit"], "frac": sig(r["frac"]),
            "traffic": r["traffic"], "traffic_source": r["traffic_source"],
            "frac_traffic": sig(r["frac_traffic"]),
            "kernel": kname(r["kernel"]), "avg_launch_ms": sig(r["avg_launch_ms"]),
            "launches_timed": r["launches_timed"],
            "bytes_per_launch": r["bytes_per_launch"],
            "format_bytes_per_launch": r["format_bytes_per_launch"],
            "frac_format": sig(r["frac_format"]),
            "step_algorithmic_gbs": sig(r["step_algorithmic_gbs"]),
            "plan": r["plan"]}
    sp = r.get("specialised")
    if sp:
        roof["specialised"] = {k: (sig(v) if isinstance(v, float) else v)
                               for k, v in sp.items()
                               if k in ("kernel", "iters_per_s", "ms",
                                        "frac_physical", "frac_csr_equivalent")}
    if "north_star_rowblock_spmv" in out:
        # 216^3 (the north star's 10 M rows), priced with SURVEY 8d's B_csr
        ns = {"rows": out["north_star_rowblock_spmv"]["rows"]}
        for key, rec in (("rowblock", "north_star_rowblock_spmv"),
                         ("lx", "north_star_lx_spmv"),
                         ("default", "north_star_spmv")):
            ns[key + "_ms"] = sig(out[rec]["ms_per_apply"])
            ns[key + "_frac"] = sig(out[rec]["frac_csr_equivalent"])
        roof["north_star"] = ns
    if "csr_rowblock_spmv" in out:
        # the caller's CSR arrays streamed as they are: [ms, frac of B_csr,
        # fabric bytes] with the x windows staged (XW), and the gather kernel
        for key, rec in (("csr_rowblock", "csr_rowblock_spmv"),
                         ("csr_gather", "csr_gather_spmv")):
            if rec in out:
                roof[key] = [sig(out[rec]["ms_per_apply"]),
                             sig(out[rec]["frac_csr_equivalent"]),
                             out[rec].get("traffic")]
    if "ragged" in r:
        # name: [ms per apply, frac of SURVEY 8d's bytes (B_sym for *_sym_*)]
        roof["ragged"] = {k: [sig(v["ms_per_apply"]), sig(v["frac"])]
                          for k, v in r["ragged"].items()
                          if isinstance(v, dict) and "frac" in v}
        cg = r["ragged"].get("fem_sym_cg")
        if cg:
            roof["ragged"]["fem_sym_cg_iters_per_s"] = sig(cg["iters/s"])
        # plan memory on top of the caller's CSR arrays (x their bytes); with
        # CSRMatrix::release_csr the general sliced jagged plans without long
        # rows give colind and values back: resident = 1 + this - 12 nnz / csr
        roof["ragged_plan_over_csr"] = {
            k: sig(out[k]["plan_extra_bytes"] / out[k]["csr_bytes"], 3)
            for k in r["ragged"] if k in out and "csr_bytes" in out[k]}
    if "plan" in out and r["algorithmic_bytes_per_launch"]:
        roof["plan_extra_over_csr_bytes"] = sig(
            out["plan"]["plan_extra_bytes"] / out["plan"]["csr_bytes"], 3)
    line["roofline"] = roof
    line["cg_rel_residual"] = {k: v for k, v in out["cg_rel_residual"].items()
                               if k in ("k10", "k10_ok")}
    if "symmetric" in out:  # BASELINE configs[3]
        s_ = out["symmetric"]
        # frac = SURVEY 8d's B_sym / avg launch / peak (values streamed)
        line["symmetric"] = {"iters_per_s": sig(s_["iters/s"]),
                             "kernel": kname(s_["kernel"]),
                             "avg_launch_ms": sig(s_["avg_launch_ms"]),
                             "frac": sig(s_["frac_csr_equivalent"]),
                             "frac_format": sig(s_["frac"]),
                             "traffic": s_.get("traffic")}
        sp = s_.get("specialised")
        if sp:
            line["symmetric"]["specialised"] = {
                "iters_per_s": sig(sp["iters/s"]), "kernel": kname(sp["kernel"]),
                "ms": sig(sp["avg_launch_ms"]), "frac_physical": sig(sp["frac"]),
                "frac_of_B_sym": sig(sp["frac_csr_equivalent"])}
    for k in ("halo_selfcheck", "rccl", "cg_scalar_reductions", "launcher"):
        if k in out:
            line[k] = out[k]
    if "ranks" in out:  # per rank: [neighbours, ghosts, rows]
        line["ranks"] = [[r_["neighbours"], r_["ghosts"], r_["rows"]]
                         for r_ in out["ranks"]]
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {
            "value": sig(c["value"]), "unit": c["unit"], "cores": c["cores"],
            "kind": c["kind"], "sample": c["sample_short"],
            "spmv_omp_gbs": sig(c["spmv_omp"]["GB/s"]),
            "k10": c["cg_rel_residual_k10"]}
        pc = c.get("parity_checks")
        if pc:
            line["cpu_baseline"]["parity_checks_bit_exact"] = bool(
                "error" not in pc and all(v["bit_exact_vs_oracle"]
                                          for v in pc.values()))
    line["detail"] = detail_path
    return line


def emit(out, args):
    """detail -> file (+ stderr, prefixed so that no other line starts with a
    brace), then the compact line, last, on stdout"""
    path = args.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        shown = f"not written ({e.strerror})"
    print("bench_detail " + json.dumps(out), file=sys.stderr, flush=True)
    text = json.dumps(compact_line(out, shown), separators=(",", ":"))
    if len(text) > LINE_MAX:
        raise SystemExit(f"bench line is {len(text)} bytes (> {LINE_MAX}): the "
                         "driver would not parse it")
    print(text, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as
    CHILD processes through torch.distributed.run and wait.  This parent has
    made no HIP / torch.cuda call (torch is not even imported) and never
    does; nothing that touched a GPU is ever re-exec'd.  Rank 0's line goes to
    the inherited stdout; a failing child gives a non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, SPMV_BENCH_LAUNCHER="self")
    print(f"bench.py: no WORLD_SIZE in the environment, starting {args.gpus} "
          f"ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rank_shape is not None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import rank_shape
        recs = {}
        for plain in (True, False):
            for P in (args.rank_shape or [2, 4, 8]):
                recs[("csr_order" if plain else "specialised") + f"_P{P}"] = \
                    rank_shape.model(P, args.n, args.steps, plain,
                                     symmetric=args.symmetric)
        print(json.dumps({"rank_shape": recs, "grid": args.n, "steps": args.steps,
                          "data": "MODEL: one rank of P alone on one GPU, no-op "
                                  "transport; not a scaling measurement"}),
              flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # before torch is imported: with OMP_PROC_BIND set, the first OpenMP runtime
    # that starts pins this thread to one core and the mask would read "1 core"
    host_cores = usable_cores()

    import numpy as np
    import torch
    import torch.distributed as dist

    from spmv_amd import _lib, host, poisson

    rehearsal = args.transport == "gloo"
    dev = 0 if (args.share_gpu or rehearsal) else local_rank
    torch.cuda.set_device(dev)
    exec_ = host.HipExecutor(dev)
    rccl = None
    if world > 1 and rehearsal:
        from spmv_amd import gloo_transport as dist_util  # CallbackComm over gloo
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
        rccl = comm.rccl_info()
        if rccl["nranks"] != world or rccl["rank"] != rank:
            raise SystemExit(f"rank {rank}: RCCL reports rank {rccl['rank']} of "
                             f"{rccl['nranks']}, expected {rank} of {world}")
    else:
        comm = host.Comm.self_comm()
    # (--peer-reduce with a one-sided --cm: the LIBRARY refuses the pair -- the
    # L2GMap built below falls back to the two-sided exchange on every rank;
    # the line's config.halo says what ran)
    if args.put_timeout_ms is not None:
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"put_timeout_ms",
                  args.put_timeout_ms)
    peer_reduce = bool(world > 1 and args.peer_reduce
                       and comm.enable_peer_reduce(exec_))

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.n
    N = n ** 3
    ctx = exec_.context
    # The headline's plan (module docstring): CSR order -- no lattice analysis
    # for general storage, values streamed for symmetric storage -- unless
    # --specialised; a PETSc file always gets the AUTO plan.
    plain = not (args.specialised or args.petsc_matrix)
    main_opts = {}
    if args.no_lattice or args.no_lx or (plain and not args.symmetric):
        main_opts[b"lat_min_nnz"] = (1 << 62, 1 << 20)
    if args.no_bake:
        main_opts[b"bake_general"] = (0, 1)
    if args.no_const or (plain and args.symmetric):
        main_opts[b"const_diagonals"] = (0, 1)
    if args.no_lx:
        main_opts[b"lx_min_nnz"] = (1 << 62, 1 << 20)
    cm = getattr(host, args.cm.upper())
    exec_.synchronize()
    t_create = time.perf_counter()
    for k_, (v_, _) in main_opts.items():
        _lib.call("spmv_hip_ctx_set_option", ctx, k_, v_)
    try:
        if args.petsc_matrix:
            # the reference demos' input path (demos/cg.cpp:47-51,
            # demos/spmv.cpp:43): every rank reads its row slab of the file
            # (spmv/read_petsc.cpp:40-228)
            A = host.read_petsc_binary_matrix(args.petsc_matrix, comm, exec_,
                                              args.symmetric, cm)
        else:
            A = host.Matrix.create_poisson3d(comm, exec_, n, args.symmetric, cm)
    finally:  # the sub-records below set what they need themselves
        for k_, (_, d_) in main_opts.items():
            _lib.call("spmv_hip_ctx_set_option", ctx, k_, d_)
    exec_.synchronize()
    t_create = time.perf_counter() - t_create  # generator + upload-free plan
    l2g = A.col_map()
    M = l2g.local_size()
    blocks = A.blocks()
    nnz_global = poisson.poisson3d_nnz(n)
    if args.petsc_matrix:
        n = 0  # no grid: none of the Poisson-only records below applies
        args.no_extras = True
        N, nnz_global = M, A.non_zeros()
        if world > 1:
            tot = torch.tensor([M, A.non_zeros()], dtype=torch.float64,
                               device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            N, nnz_global = int(tot[0]), int(tot[1])

    # RHS b = Gaussian bump (demos/spmv.cpp:63-67) -- resident before timing
    if args.blas1_nt_min is not None:
        _lib.call("spmv_hip_ctx_set_option", ctx, b"blas1_nt_min_elems",
                  args.blas1_nt_min)
    d_x = exec_.alloc(M)
    if args.petsc_rhs:
        d_b, m_b = host.read_petsc_binary_vector(comm, exec_, args.petsc_rhs)
        if m_b != M:
            raise SystemExit(f"rank {rank}: {args.petsc_rhs} gives {m_b} local "
                             f"entries, the matrix has {M} local rows")
    else:
        d_b = exec_.alloc(M)
        _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, l2g.global_offset(), M,
                  d_b, None)
    ws = host.CgWorkspace(exec_)
    exec_.synchronize()

    # Several ranks: before anything is timed, prove the transport on this run's
    # own partition -- after one halo update of the vector x_i = g(i) every
    # ghost entry must hold g(its global index), and the locally owned part must
    # be untouched.  A wrong offset or a lost message fails here, loudly,
    # instead of producing a fast wrong number.
    halo_selfcheck = None
    ng = l2g.num_ghosts()
    if world > 1:
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
        halo_selfcheck = "ok"

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
    k, hist, spmv_ms, spmv_launches = host.cg_ex(
        comm, exec_, A, d_b, d_x, args.steps, 0.0, ws, time_spmv=True,
        history=True, consumer_reductions=not args.reducer_kernels)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert k == args.steps, (k, args.steps)

    # algorithmic bytes of the dominant kernel on THIS rank (DESIGN.md,
    # SURVEY 8d), and the bytes its form really loads and stores
    kernel, kernel_bytes, requested_bytes = kernel_of(A, args.symmetric)
    iter_bytes = kernel_bytes + 9 * M * 8  # + fused BLAS-1 minimum, SURVEY 8d

    # max over ranks of the times, sum over ranks of the bytes; per-rank halo
    # shape for the record
    mine = [float(l2g._nn), float(ng), float(M)]
    if world > 1:
        dev_t = "cpu" if rehearsal else "cuda"
        t = torch.tensor([elapsed, spmv_ms / max(spmv_launches, 1)],
                         dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, spmv_ms_avg = float(t[0]), float(t[1])
        bsum = torch.tensor([kernel_bytes, iter_bytes, requested_bytes],
                            dtype=torch.float64, device=dev_t)
        dist.all_reduce(bsum, op=dist.ReduceOp.SUM)
        kernel_bytes_all, iter_bytes_all = float(bsum[0]), float(bsum[1])
        requested_all = float(bsum[2])
        shape = torch.zeros(world, 3, dtype=torch.float64, device=dev_t)
        shape[rank] = torch.tensor(mine, dtype=torch.float64)
        dist.all_reduce(shape, op=dist.ReduceOp.SUM)
        per_rank = shape.cpu().tolist()
    else:
        spmv_ms_avg = spmv_ms / max(spmv_launches, 1)
        kernel_bytes_all, iter_bytes_all = kernel_bytes, iter_bytes
        requested_all = requested_bytes
        per_rank = [mine]

    if rank == 0:
        traffic, traffic_source = pmc_traffic(
            "csr_order" if plain and not args.symmetric else
            "symmetric_value_stream_spmv" if plain else
            "symmetric" if args.symmetric else "main", kernel, n, world)
        pr = price(spmv_ms_avg, kernel_bytes, requested_bytes, traffic)
        k10 = float(hist[min(10, len(hist) - 1)] / hist[0])
        resid = {"k10": k10, "kK": float(hist[-1] / hist[0])}
        if n == 512 and len(hist) > 10:
            # the 1-rank value: an N-rank run must land on it (its dot products
            # are summed in another order, nothing else differs)
            resid["k10_expected"] = K10_512
            resid["k10_rel_deviation"] = abs(k10 / K10_512 - 1.0)
            resid["k10_ok"] = bool(abs(k10 / K10_512 - 1.0) < 1e-9)
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
            "config": {"workload": (f"poisson3d_{n}^3_csr_fp64_cg" if n else
                                    "petsc_binary_matrix_csr_fp64_cg: "
                                    + os.path.basename(args.petsc_matrix)),
                       "rows": N, "nnz": nnz_global,
                       "storage": "symmetric-csr" if args.symmetric else "csr",
                       "partition": f"row-slab x{world}",
                       "halo": (args.cm + (" (peer stores into IPC windows, "
                                            "one put kernel per exchange)"
                                            if l2g.onesided() else
                                            " (RCCL send/recv on a side stream)"))
                       if world > 1 else "none (1 rank)"},
            # roofline of the dominant kernel (the local block's SpMV), timed
            # live with HIP events on its stream.  CSR-order plan (default):
            # achieved / frac / bytes_per_launch are SURVEY 8d's algorithmic
            # bytes (B_csr, or B_sym for symmetric storage); --specialised or a
            # PETSc file: the bytes the AUTO plan's own format moves (PHYSICAL),
            # with the CSR-equivalent figure beside it.
            "roofline": {"bound": "hbm",
                         "achieved": (pr["csr_equivalent_gbs"] if plain
                                      else pr["GB/s"]),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (pr["frac_csr_equivalent"] if plain
                                  else pr["frac"]),
                         "frac_prices": ("SURVEY 8d algorithmic bytes ("
                                         + ("B_sym" if args.symmetric else "B_csr")
                                         + ") / avg_launch_ms / peak" if plain else
                                         "the bytes the plan's own format moves "
                                         "(physical) / avg_launch_ms / peak"),
                         "bytes_per_launch": (kernel_bytes if plain
                                              else requested_bytes),
                         # what the plan's format of the matrix really loads
                         # and stores (LX: 10 B per entry instead of CSR's 12)
                         "format_bytes_per_launch": requested_bytes,
                         "frac_format": pr["frac"],
                         "traffic": traffic, "traffic_source": traffic_source,
                         # ... and with the bytes the PMC passes saw cross the
                         # fabric: what the HBM side is really asked to do
                         "frac_traffic": pr.get("frac_traffic"),
                         "algorithmic_bytes_per_launch": kernel_bytes,
                         "csr_equivalent_gbs": pr["csr_equivalent_gbs"],
                         "frac_csr_equivalent": pr["frac_csr_equivalent"],
                         # SURVEY 8d's bytes of a whole step (SpMV + 9 vector
                         # passes) over ms_per_step: must stay below the peak
                         "step_algorithmic_gbs":
                             iter_bytes / (elapsed / args.steps) / 1e9,
                         "note": ("default: the CSR-order plan, achieved / frac / "
                                  "bytes_per_launch = SURVEY 8d's algorithmic "
                                  "bytes; format_bytes_per_launch / frac_format "
                                  "= what the plan's own format moves; "
                                  "roofline.specialised = the AUTO plan of this "
                                  "matrix in the same loop; sub-records "
                                  "value_stream_spmv (lattice matrix whose "
                                  "coefficients vary), csr_lx_spmv / "
                                  "csr_rowblock_spmv / unstructured_spmv are the "
                                  "kernels other matrices get"),
                         "plan": ("csr-order (lattice analysis off; symmetric "
                                  "storage: values streamed)" if plain else
                                  "AUTO (specialised to the matrix)"),
                         "kernel": kernel,
                         "avg_launch_ms": spmv_ms_avg,
                         "launches_timed": spmv_launches},
            # ||r_k|| / ||r_0|| from the device-side history: after 10
            # iterations the same number to ~1e-12 whatever the rank count
            # or storage (a cross-check of the distributed path); after all K,
            # where CG has amplified the different summation orders.  CG parity
            # is unpinned by the reference itself.
            "cg_rel_residual": resid,
            # whole-iteration effective bandwidth: SpMV + fused BLAS-1 minimum
            # (9 vectors of 8 B per row)
            "cg_gbs_per_gpu": iter_bytes / (elapsed / args.steps) / 1e9,
            # all ranks together: sum of bytes / time of the slowest rank
            "spmv_gbs_aggregate": requested_all / (spmv_ms_avg * 1e-3) / 1e9,
            "spmv_csr_equivalent_gbs_aggregate":
                kernel_bytes_all / (spmv_ms_avg * 1e-3) / 1e9,
            "cg_gbs_aggregate": iter_bytes_all / (elapsed / args.steps) / 1e9,
            "plan": plan_record(A),
            "matrix_create_ms": t_create * 1e3,
        }
        if os.environ.get("SPMV_BENCH_LAUNCHER"):
            out["launcher"] = "bench.py started its own ranks (child processes)"
        if world > 1:
            out["halo_selfcheck"] = halo_selfcheck
            out["ranks"] = [{"rank": r, "neighbours": int(v[0]),
                             "ghosts": int(v[1]), "rows": int(v[2])}
                            for r, v in enumerate(per_rank)]
            if rccl:
                out["rccl"] = {k_: rccl[k_] for k_ in
                               ("nranks", "version", "lib_path",
                                "separate_reduction_comm")}
            out["cg_scalar_reductions"] = (
                "peer windows, added in rank order (spmv_hip_reduce_*)"
                if peer_reduce else "the transport's all-reduce")
    # the main matrix is no longer needed: make room for the sub-records
    ws.close()
    A.close()
    exec_.free(d_b), exec_.free(d_x)

    if rank == 0:
        if world == 1 and not args.no_extras:
            self_comm = comm
            # Time to solution by the reference demo's protocol
            # (demos/cg.cpp:64-72: max_its = 100, rtol = 1e-10, wall clock
            # around cg(), which allocates its work vectors inside): plan
            # creation INCLUDED -- what one solve costs a caller who builds the
            # matrix, solves once and leaves.
            if True:
                exec_.synchronize()
                t0 = time.perf_counter()
                At = host.Matrix.create_poisson3d(self_comm, exec_, n,
                                                  args.symmetric, cm)
                exec_.synchronize()
                t_mat = time.perf_counter() - t0
                d_b, d_x = exec_.alloc(N), exec_.alloc(N)
                _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_b, None)
                exec_.synchronize()
                t0 = time.perf_counter()
                k100, _ = host.cg(self_comm, exec_, At, d_b, d_x, 100, 1e-10,
                                  history=False)
                exec_.synchronize()
                t_cg = time.perf_counter() - t0
                plan_ms = At.plan_get("plan_us") / 1e3
                out["time_to_solution"] = {
                    "protocol": "demos/cg.cpp:64-72 (max_its 100, rtol 1e-10, "
                                "wall clock around cg(); work vectors allocated "
                                "inside) + plan creation",
                    "iterations": k100, "cg_ms": t_cg * 1e3, "plan_ms": plan_ms,
                    "total_ms": t_cg * 1e3 + plan_ms,
                    "iters_per_s_incl_plan": k100 / (t_cg + plan_ms * 1e-3),
                    "matrix_generate_and_plan_ms": t_mat * 1e3}
                At.close()
                exec_.free(d_b), exec_.free(d_x)
            # BASELINE configs[3]: symmetric storage at the same size, the CG
            # loop on it.  First with the values STREAMED (constant-diagonal
            # detection off: the half diagonal form any symmetric lattice
            # matrix with varying coefficients gets), priced with SURVEY 8d's
            # B_sym; the AUTO plan's constant-diagonal run beside it.
            if not args.symmetric:
                steps = min(args.steps, 20)

                def sym_cg(const, record):
                    _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals",
                              1 if const else 0)
                    try:
                        As = host.Matrix.create_poisson3d(self_comm, exec_, n, True,
                                                          cm)
                    finally:
                        _lib.call("spmv_hip_ctx_set_option", ctx,
                                  b"const_diagonals", 1)
                    r_ = cg_record(exec_, self_comm, host, _lib, As, N, steps, True,
                                   record, n,
                                   f"poisson3d_{n}^3_symmetric-csr_fp64_cg"
                                   + ("" if const else "_values_streamed"))
                    r_["parity"] = ("bit-exact vs the oracle (atomic-free)"
                                    if "atomic-free" in r_["kernel"]
                                    else "tolerance (atomics)")
                    As.close()
                    return r_
                out["symmetric"] = sym_cg(False, "symmetric_value_stream_spmv")
                out["symmetric"]["specialised"] = sym_cg(True, "symmetric")
                # The same CG with the GENERAL matrix's values streamed (lattice
                # analysis on, constant-diagonal detection off): what the loop
                # costs on a lattice matrix whose coefficients vary -- the half
                # diagonal form of rounds 2-3
                if not (args.no_const or args.no_bake or args.no_lattice
                        or args.no_lx):
                    _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals", 0)
                    try:
                        Av = host.Matrix.create_poisson3d(self_comm, exec_, n,
                                                          False, cm)
                    finally:
                        _lib.call("spmv_hip_ctx_set_option", ctx,
                                  b"const_diagonals", 1)
                    out["value_stream_cg"] = cg_record(
                        exec_, self_comm, host, _lib, Av, N, steps, False,
                        "value_stream_spmv", n,
                        f"poisson3d_{n}^3_csr_fp64_cg_values_streamed")
                    Av.close()
                rec = lambda name, *a, **kw: spmv_record(  # noqa: E731
                    exec_, self_comm, host, _lib, *a, record=name, **kw)
                # a lattice matrix that is NOT symmetric (the generator's skewed
                # variant: lower neighbours -1.001, upper -0.999): the full
                # diagonal form; and the CSR-order lattice kernel (no baked
                # copy of the values) on the Poisson matrix itself
                if not (args.no_lattice or args.no_lx or args.no_bake):
                    # the kernels that STREAM the values (what a lattice matrix
                    # with varying coefficients gets), on the same matrix with
                    # the constant-diagonal detection switched off: the half
                    # diagonal form (symmetric matrix), symmetric storage, and
                    # the full diagonal form on the skewed variant
                    if not args.no_const:
                        out["value_stream_spmv"] = rec(
                            "value_stream_spmv", n, False, 20, const=False)
                        out["symmetric_value_stream_spmv"] = rec(
                            "symmetric_value_stream_spmv", n, True, 20,
                            const=False)
                    out["csr_nonsymmetric_spmv"] = rec(
                        "csr_nonsymmetric_spmv", n, False, 20, skew_ppm=1000,
                        const=False)
                    out["csr_lattice_spmv"] = rec("csr_lattice_spmv", n, False,
                                                  20, bake=False)
                # What a CSR matrix WITHOUT lattice structure gets, on the same
                # matrix with the lattice analysis switched off: the LX form
                # (banded matrices: x windows staged in LDS) and the plain
                # row-block gather kernel (everything else)
                if not (args.no_lattice or args.no_lx):
                    out["csr_lx_spmv"] = rec("csr_lx_spmv", n, False, 20,
                                             lattice=False)
                    # ... the caller's arrays as they are: x windows staged
                    # (XW), and the gather kernel
                    out["csr_rowblock_spmv"] = rec("csr_rowblock_spmv", n, False,
                                                   20, lattice=False, lx=False,
                                                   sj=False)
                    out["csr_gather_spmv"] = rec("csr_gather_spmv", n, False,
                                                 20, lattice=False, lx=False,
                                                 sj=False, xw=False)
                    # ... and the sliced jagged form on this matrix (7 entries
                    # per row are too few for it: the LX form is the AUTO choice)
                    out["csr_sjds_spmv"] = rec("csr_sjds_spmv", n, False, 20,
                                               lattice=False, lx=False)
            if not args.symmetric and plain:
                # WHAT THE AUTO PLAN DOES WITH THIS MATRIX, kept inside `roofline`
                # (the block the driver stores): the same CG loop with the
                # plan-time analysis on -- it finds the lattice and that every
                # diagonal is constant and streams no matrix at all (rounds 1-5's
                # main line).  frac_physical = the bytes that kernel's own format
                # moves / time / peak; frac_csr_equivalent = SURVEY 8d's CSR
                # bytes over the same time: a speed-up statement, exceeds 1.
                Ag = host.Matrix.create_poisson3d(self_comm, exec_, n, False, cm)
                r_ = cg_record(exec_, self_comm, host, _lib, Ag, N,
                               min(args.steps, 30), False, "main", n,
                               f"poisson3d_{n}^3_csr_fp64_cg_auto_plan")
                Ag.close()
                out["specialised_cg"] = r_
                out["roofline"]["specialised"] = {
                    "what": "the AUTO plan of this matrix (lattice analysis + "
                            "constant diagonals: no matrix stream) in the same "
                            "CG loop",
                    "kernel": r_["kernel"].split(" (")[0],
                    "iters_per_s": r_["iters/s"], "ms": r_["avg_launch_ms"],
                    "ms_per_step": r_["ms_per_step"],
                    "frac_physical": r_["frac"],
                    "frac_csr_equivalent": r_["frac_csr_equivalent"],
                    "bytes_per_launch": r_["requested_bytes"],
                    "traffic": r_.get("traffic"),
                    "cg_rel_residual_k10": r_["cg_rel_residual_k10"]}
            if not args.symmetric:
                # RAGGED ROWS (what the PETSc reader typically delivers): seeded
                # FEM-like matrices, 10 M rows -- row lengths 5-40 in three
                # clusters of columns (a bandwidth-reducing order); the same with
                # a 1 % tail of 200-2000-entry rows; 81 entries in every row.
                # The AUTO plan takes the sliced jagged form; cross-checked here
                # against the one-lane-per-row kernel, oracle-sized instances in
                # tests/test_gpu_matrix.py.  Summary inside `roofline`.
                ragged = {}
                for name, kw in (("fem_spmv", dict()),
                                 ("fem_tail_spmv", dict(tail_permille=10)),
                                 ("fem81_spmv", dict(min_len=81, max_len=81))):
                    Af = host.Matrix.create_fem_like(self_comm, exec_,
                                                     args.fem_rows, **kw)
                    r = matrix_spmv_record(
                        exec_, Af, _lib, False, 30, name, args.fem_rows,
                        f"fem_like_{args.fem_rows}rows_"
                        + ("len81" if kw.get("min_len") else "len5-40")
                        + ("_tail1pct_200-2000" if kw.get("tail_permille") else "")
                        + "_csr_fp64_spmv", crosscheck=True)
                    r["avg_row"] = r["nnz_stored"] / r["rows"]
                    out[name] = r
                    ragged[name] = {"ms_per_apply": r["ms_per_apply"],
                                    "frac": r["frac_csr_equivalent"],
                                    "kernel": r["kernel"].split(" (")[0],
                                    "bit_equal_one_lane_per_row":
                                        r["crosscheck"]["bit_equal"],
                                    "plan_ms": r["plan_ms"],
                                    "traffic": r.get("traffic")}
                    if name == "fem_spmv":
                        # ... the same product with the fp32 copy of the values
                        # (CgOptions::mixed: fp64 vectors and arithmetic)
                        Af.enable_mixed()
                        Af.use_mixed(True)
                        ms32 = timed_spmv(exec_, Af, args.fem_rows, _lib, 30)

                        def product():
                            Nf_ = args.fem_rows
                            d_x, d_y = exec_.alloc(Nf_), exec_.alloc(Nf_)
                            _lib.call("spmv_hip_fill_gaussian_f64", ctx, Nf_, 0, Nf_,
                                      d_x, None)
                            exec_.memset(d_y, 0xFF, 8 * Nf_)
                            Af.mult(d_x, d_y)
                            y = exec_.copy_to_host(d_y, Nf_)
                            exec_.free(d_x), exec_.free(d_y)
                            return y
                        y_sj = product()
                        mixed_form = Af.plan_get("sj_mixed")
                        Af.plan_set("sjds", 0)  # the CSR-order kernel, fp32 values
                        same32 = bool(np.array_equal(y_sj, product())
                                      and np.isfinite(y_sj).all())
                        Af.plan_set("sjds", 1)
                        Af.use_mixed(False)
                        ragged["fem_mixed_spmv"] = {
                            "ms_per_apply": ms32,
                            "speedup_over_fp64": r["ms_per_apply"] / ms32,
                            "kernel": ("csr_sjds_kernel<double, float values>"
                                       if mixed_form else "CSR-order fp32 values"),
                            "bit_equal_csr_order_kernel": same32,
                            "requested_bytes": r["nnz_stored"] * 6
                                               + r["rows"] * 12 + r["rows"] * 8,
                            "frac_requested": (r["nnz_stored"] * 6 + r["rows"] * 20)
                                              / ms32 / 1e6 / HBM_PEAK_GBS}
                    Af.close()
                # ... and the first of them in SYMMETRIC storage (its strictly
                # lower part + diagonal, as create_matrix(symmetric = true) keeps
                # it): both blocks sliced jagged; priced with SURVEY 8d's B_sym
                Af = host.Matrix.create_fem_like(self_comm, exec_, args.fem_rows,
                                                 symmetric=True)
                r = matrix_spmv_record(
                    exec_, Af, _lib, True, 30, "fem_sym_spmv", args.fem_rows,
                    f"fem_like_{args.fem_rows}rows_len5-40_lower+diag_symmetric_"
                    "storage_fp64_spmv",
                    # (nothing to compare where the merged form was refused and
                    # the transposed-map kernel is the plan's own)
                    crosscheck="symt" if Af.plan_get("sym_sj") else False)
                out["fem_sym_spmv"] = r
                ragged["fem_sym_spmv"] = {
                    "ms_per_apply": r["ms_per_apply"],
                    "frac": r["frac_csr_equivalent"],
                    "frac_note": "SURVEY 8d B_sym (12 B per stored lower entry, "
                                 "row pointer, diagonal, x, y) / time / 8 TB/s; the "
                                 "kernel streams 20 B per stored entry",
                    "frac_requested": r["frac_requested"],
                    "kernel": r["kernel"].split(" (")[0],
                    "bit_equal_transposed_map_kernel":
                        r.get("crosscheck", {}).get("bit_equal"),
                    "transposed_map_kernel_ms":
                        getattr(timed_spmv, "other_ms", None)
                        if "crosscheck" in r else r["ms_per_apply"],
                    "plan_ms": r["plan_ms"], "traffic": r.get("traffic")}
                # ... and CG on it (strictly diagonally dominant: SPD), the
                # reference's loop (cg.cpp:21-98) end to end on a matrix without
                # stencil structure: 3 launches per iteration + the second pass
                Nf = args.fem_rows
                d_b, d_x = exec_.alloc(Nf), exec_.alloc(Nf)
                _lib.call("spmv_hip_fill_gaussian_f64", ctx, Nf, 0, Nf, d_b, None)
                wsf = host.CgWorkspace(exec_)
                stepsf = min(args.steps, 30)
                host.cg_ex(self_comm, exec_, Af, d_b, d_x, 3, 0.0, wsf)
                exec_.synchronize()
                t0 = time.perf_counter()
                _, hf, _, _ = host.cg_ex(self_comm, exec_, Af, d_b, d_x, stepsf, 0.0,
                                         wsf, history=True)
                exec_.synchronize()
                elf = time.perf_counter() - t0
                ragged["fem_sym_cg"] = {
                    "iters/s": stepsf / elf, "iterations": stepsf,
                    "rel_residual_after": float(hf[-1] / hf[0]),
                    "what": "cg() on the 10 M-row FEM-like matrix in symmetric "
                            "storage, Gaussian right-hand side"}
                wsf.close()
                exec_.free(d_b), exec_.free(d_x)
                Af.close()
                # ... and the matrix with the 1 % tail of 200-2000-entry rows in
                # symmetric storage: long rows AND long columns of the stored
                # lower part (Matrix.cpp:337-349 on a FEM matrix with a dense-ish
                # tail); beside its general-storage time above
                Af = host.Matrix.create_fem_like(self_comm, exec_, args.fem_rows,
                                                 symmetric=True, tail_permille=10)
                r = matrix_spmv_record(
                    exec_, Af, _lib, True, 30, "fem_tail_sym_spmv", args.fem_rows,
                    f"fem_like_{args.fem_rows}rows_len5-40_tail1pct_200-2000_"
                    "lower+diag_symmetric_storage_fp64_spmv",
                    crosscheck="symt" if Af.plan_get("sym_sj") else False)
                out["fem_tail_sym_spmv"] = r
                ragged["fem_tail_sym_spmv"] = {
                    "ms_per_apply": r["ms_per_apply"],
                    "frac": r["frac_csr_equivalent"],
                    "over_general_storage": r["ms_per_apply"]
                                            / ragged["fem_tail_spmv"]["ms_per_apply"],
                    "kernel": r["kernel"].split(" (")[0],
                    "bit_equal_transposed_map_kernel":
                        r.get("crosscheck", {}).get("bit_equal"),
                    "transposed_map_kernel_ms":
                        getattr(timed_spmv, "other_ms", None)
                        if "crosscheck" in r else r["ms_per_apply"],
                    "plan_ms": r["plan_ms"], "traffic": r.get("traffic")}
                Af.close()
                out["roofline"]["ragged"] = dict(
                    ragged, note="frac = SURVEY 8d CSR bytes (12 B per entry, row "
                                 "pointer, x, y) / launch time / 8 TB/s")
            if not args.symmetric:
                rec = lambda name, *a, **kw: spmv_record(  # noqa: E731
                    exec_, self_comm, host, _lib, *a, record=name, **kw)
                # BASELINE north_star: plain SpMV on the ~10 M-row matrix: the
                # default plan, and the CSR-order kernels next to it
                out["north_star_spmv"] = rec("north_star_spmv", 216, False, 200)
                out["north_star_value_stream_spmv"] = rec(
                    "north_star_value_stream_spmv", 216, False, 200, const=False)
                out["north_star_lattice_spmv"] = rec(
                    "north_star_lattice_spmv", 216, False, 200, bake=False)
                out["north_star_lx_spmv"] = rec(
                    "north_star_lx_spmv", 216, False, 200, lattice=False)
                out["north_star_rowblock_spmv"] = rec(
                    "north_star_rowblock_spmv", 216, False, 200, lattice=False,
                    lx=False, sj=False)
                # Matrices that are NOT the 7-point stencil.  (a) the 27-point
                # operator on a 256^3 grid (HPCG's matrix; 16.8 M rows, 449 M
                # entries); (b) a seeded unstructured matrix (10 M rows, 7
                # entries per row: 90 % within 2,048 columns of the diagonal,
                # 10 % anywhere) -- no lattice, column windows too wide to
                # stage.  Each is cross-checked in this run against the
                # one-lane-per-row kernel; the oracle-sized instances are in
                # tests/test_gpu_matrix.py and cpu_baseline.parity_checks.
                out["stencil27_spmv"] = rec("stencil27_spmv", args.stencil27_grid,
                                            False, 20, stencil=27)
                # ... with its values streamed (26 / -1 are constants too): the
                # half form, the matrix being symmetric
                out["stencil27_value_stream_spmv"] = rec(
                    "stencil27_value_stream_spmv", args.stencil27_grid, False, 20,
                    stencil=27, const=False)
                Au = host.Matrix.create_unstructured(self_comm, exec_,
                                                     args.unstructured_rows)
                out["unstructured_spmv"] = matrix_spmv_record(
                    exec_, Au, _lib, False, 50, "unstructured_spmv",
                    args.unstructured_rows,
                    f"unstructured_{args.unstructured_rows}rows_7per_row_band2048_"
                    "far10pct_csr_fp64_spmv", crosscheck=True)
                Au.close()
                if "ragged" in out["roofline"]:
                    r = out["unstructured_spmv"]
                    out["roofline"]["ragged"]["unstructured_spmv"] = {
                        "ms_per_apply": r["ms_per_apply"],
                        "frac": r["frac_csr_equivalent"],
                        "kernel": r["kernel"].split(" (")[0],
                        "bit_equal_one_lane_per_row": r["crosscheck"]["bit_equal"],
                        "plan_ms": r["plan_ms"], "traffic": r.get("traffic")}
                # mixed-precision CG against pure fp64, to 1e-10
                out["mixed_precision_cg"] = mixed_precision_record(
                    exec_, self_comm, host, _lib, args.mixed_grid)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, n, N, host_cores)
            if not args.no_extras:
                out["cpu_baseline"]["parity_checks"] = oracle_parity_checks(
                    exec_, comm, host, _lib)
        emit(out, args)

    comm.close()
    exec_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
