"""bench.py contract checks on a small grid: the single-GPU line and a
2-rank rehearsal (gloo transport, both ranks on GPU 0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup",
            "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check(d, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "iters/s" and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - steps / (d["ms_per_step"] * steps / 1e3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_timed"] == steps and r["avg_launch_ms"] > 0
    # physical: the bytes the kernel's format moves over the measured time
    assert abs(r["achieved"] - r["bytes_per_launch"] / r["avg_launch_ms"] / 1e6) \
        < 1e-9 * r["achieved"]
    assert r["frac_requested"] == r["frac"] and r["frac"] > 0
    # ... and the CSR-equivalent figure under its own name (may exceed 1)
    assert r["algorithmic_bytes_per_launch"] >= r["bytes_per_launch"]
    assert abs(r["csr_equivalent_gbs"] - r["algorithmic_bytes_per_launch"]
               / r["avg_launch_ms"] / 1e6) < 1e-9 * r["csr_equivalent_gbs"]
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert (r["traffic"] is None) == (r["frac_traffic"] is None) and "note" in r
    assert d["cg_rel_residual"]["k10"] > 0 and d["cg_rel_residual"]["kK"] > 0
    assert d["plan"]["plan_ms"] >= 0 and d["plan"]["plan_extra_bytes"] >= 0


def test_bench_single_gpu_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "64", "--steps", "20", "--warmup", "3",
                          "--cpu-n", "32", "--cpu-iters", "3", "--mixed-grid",
                          "48", "--stencil27-grid", "40", "--unstructured-rows",
                          "200000", "--fem-rows", "200000"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 1, 20, 3)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["cores"] <= c["host"]["physical_cores"] and "32^3" in c["sample"]
    assert c["spmv_omp"]["GB/s"] > 0
    assert "sample" in c and d["north_star_spmv"]["rows"] == 216 ** 3
    # BASELINE configs[3] in the default line: symmetric storage, atomic-free
    sym = d["symmetric"]
    assert sym["frac"] > 0 and sym["iters/s"] > 0 and "atomic-free" in sym["kernel"]
    assert d["csr_lx_spmv"]["form"]["lx"] == 1 and d["csr_lx_spmv"]["form"]["lat"] == 0
    # every diagonal of the Poisson matrix is constant: no values are streamed
    assert d["plan"]["form"]["sdia"] == 1 and d["plan"]["form"]["sdia_const"] == 1
    assert "csr_const_dia_tile_kernel<double, general order, 4 lattice lines" \
        in d["roofline"]["kernel"]
    assert d["roofline"]["bytes_per_launch"] == 17 * 64 ** 3
    # (symmetric storage of 64^3 is below the lattice analysis' size: the
    # transposed map; tests/test_gpu_matrix.py covers the larger grids)
    # ... the same matrix with its values streamed (what a lattice matrix with
    # varying coefficients gets): symmetric, so the plan keeps its lower half
    vs, svs = d["value_stream_spmv"], d["symmetric_value_stream_spmv"]
    assert vs["form"]["sdia"] == 1 and vs["form"]["sdia_const"] == 0
    assert "csr_sym_dia_kernel<double, general order>" in vs["kernel"]
    assert svs["form"]["sdia_const"] == 0 and "atomic-free" in svs["kernel"]
    vcg = d["value_stream_cg"]
    assert vcg["form"]["sdia_const"] == 0 and vcg["iters/s"] > 0
    assert "csr_sym_dia_kernel<double, general order>" in vcg["kernel"]
    assert abs(vcg["cg_rel_residual_k10"] / d["cg_rel_residual"]["k10"] - 1) < 1e-9
    assert d["north_star_spmv"]["form"]["sdia_const"] == 1
    assert d["north_star_value_stream_spmv"]["form"]["sdia_const"] == 0
    lat = d["csr_lattice_spmv"]["form"]
    assert lat["lat"] == 1 and lat["sdia"] == 0
    # ... and a matrix that is not symmetric keeps all its values by offset
    ns = d["csr_nonsymmetric_spmv"]
    assert ns["form"]["sdia"] == 1 and "full" in ns["kernel"]
    assert d["north_star_spmv"]["form"]["lat"] == 1
    # the kernels of matrices WITHOUT lattice structure, measured and checked
    assert d["csr_rowblock_spmv"]["form"] == dict(lat=0, lx=0, lxw=0, sjds=0,
                                                  sym_sj=0, wdia=0, wdia_const=0,
                                                  wdia_hbox=0, slat=0,
                                                  sdia=0, sdia_const=0, sym_det=0,
                                                  zwalk=0)
    assert d["csr_sjds_spmv"]["form"]["sjds"] == 1
    # the general-CSR line inside `roofline` (the block the driver keeps): the
    # AUTO plan without the lattice analysis, same CG loop, SURVEY 8d's bytes
    co = r_ = d["roofline"]["csr_order"]
    assert "csr_lxw_kernel" in co["kernel"] and 0 < co["frac"] <= 1.5
    assert abs(co["frac"] - co["algorithmic_bytes_per_launch"] / co["ms_per_apply"]
               / 1e6 / 8000.0) < 1e-9
    assert co["algorithmic_bytes_per_launch"] == d["roofline"][
        "algorithmic_bytes_per_launch"]
    assert abs(co["cg_rel_residual_k10"] / d["cg_rel_residual"]["k10"] - 1) < 1e-9
    assert d["roofline"]["general_cg_iters_per_s"] > 0
    # ragged rows: the sliced jagged form, bit-equal to the reference loop
    rg = d["roofline"]["ragged"]
    # ... and in symmetric storage both blocks of it, bit-equal to the
    # transposed-map kernel
    fs = rg["fem_sym_spmv"]
    assert fs["bit_equal_transposed_map_kernel"] is True and fs["frac"] > 0
    assert d["fem_sym_spmv"]["form"]["sym_sj"] == 1 and "symmetric" in fs["kernel"]
    assert 5 < d["fem_sym_spmv"]["nnz_stored"] / d["fem_sym_spmv"]["rows"] < 9
    assert rg["fem_sym_cg"]["iters/s"] > 0 and rg["fem_sym_cg"]["rel_residual_after"] < 1e-3
    assert rg["fem_mixed_spmv"]["ms_per_apply"] > 0
    assert "float values" in rg["fem_mixed_spmv"]["kernel"]
    assert rg["fem_mixed_spmv"]["bit_equal_csr_order_kernel"] is True
    for k in ("fem_spmv", "fem_tail_spmv", "fem81_spmv", "unstructured_spmv"):
        assert rg[k]["bit_equal_one_lane_per_row"] is True and rg[k]["frac"] > 0
        assert d[k]["rows"] == 200000 and d[k]["crosscheck"]["bit_equal"] is True
    for k in ("fem_spmv", "fem_tail_spmv", "fem81_spmv"):
        assert "csr_sjds_kernel" in rg[k]["kernel"] and d[k]["form"]["sjds"] == 1
    assert 13 < d["fem_spmv"]["avg_row"] < 17 and d["fem81_spmv"]["avg_row"] > 79
    assert d["fem_tail_spmv"]["avg_row"] > 20
    assert c["cg_rel_residual_k10"] is None or c["cg_rel_residual_k10"] > 0
    for k in ("north_star_lattice_spmv", "north_star_lx_spmv",
              "north_star_rowblock_spmv"):
        assert d[k]["rows"] == 216 ** 3 and d[k]["form"]["sdia"] == 0
        assert 0 < d[k]["frac"] and d[k]["plan_ms"] >= 0
    assert d["north_star_lx_spmv"]["form"]["lx"] == 1
    s27, un = d["stencil27_spmv"], d["unstructured_spmv"]
    assert s27["rows"] == 40 ** 3 and s27["nnz_stored"] == (3 * 40 - 2) ** 3
    assert un["rows"] == 200000 and un["nnz_stored"] == 7 * 200000
    s27v = d["stencil27_value_stream_spmv"]
    assert s27["form"]["wdia_const"] == 1 and "csr_box27_const_kernel" in s27["kernel"]
    assert s27v["form"]["wdia"] == 1 and s27v["form"]["wdia_const"] == 0
    assert "half" in s27v["kernel"]
    for r in (s27, s27v, un):
        assert r["crosscheck"]["bit_equal"] is True and r["frac"] > 0
    pc = c["parity_checks"]
    assert "error" not in pc, pc
    assert pc["stencil27_33^3"]["bit_exact_vs_oracle"] is True
    assert pc["unstructured_300000"]["bit_exact_vs_oracle"] is True
    tts = d["time_to_solution"]
    assert tts["iterations"] == 100 and tts["total_ms"] > tts["plan_ms"] >= 0
    mp = d["mixed_precision_cg"]
    assert mp["mixed"]["final_true_rel_residual"] < 1.001e-10
    assert abs(mp["mixed"]["iterations"] - mp["fp64"]["iterations"]) <= 5
    assert mp["x_rel_diff"] < 1e-7


def test_bench_without_the_symmetry_check():
    """--no-bake: the line of a lattice matrix whose values stay in CSR order
    (the CSR-order lattice kernel; DESIGN.md section 7, 'reading the headline')."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "128", "--steps", "10", "--warmup", "2",
                          "--no-bake", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 1, 10, 2)
    assert d["plan"]["form"]["sdia"] == 0 and d["plan"]["form"]["lat"] == 1
    assert "csr_lattice_kernel" in d["roofline"]["kernel"]
    assert d["roofline"]["frac"] <= d["roofline"]["frac_csr_equivalent"]


def test_bench_with_the_values_streamed():
    """--no-const: the headline of a lattice matrix whose coefficients vary (the
    half diagonal form streams the lower values)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "128", "--steps", "10", "--warmup", "2",
                          "--no-const", "--no-extras", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 1, 10, 2)
    assert d["plan"]["form"]["sdia"] == 1 and d["plan"]["form"]["sdia_const"] == 0
    assert "csr_sym_dia_kernel<double, general order>" in d["roofline"]["kernel"]


def test_bench_two_rank_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus",
         "2", "--steps", "10", "--warmup", "2", "--grid", "64", "--transport",
         "gloo"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 2, 10, 2)
    assert "REHEARSAL" in d["data"] and "cpu_baseline" not in d
    assert d["halo_selfcheck"] == "ok" and len(d["ranks"]) == 2
    assert [r["neighbours"] for r in d["ranks"]] == [1, 1]
    assert [r["ghosts"] for r in d["ranks"]] == [64 * 64, 64 * 64]
    assert sum(r["rows"] for r in d["ranks"]) == 64 ** 3
    # the distributed run reproduces the one-rank residual after 10 iterations
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "64", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    k10 = _line(one.stdout)["cg_rel_residual"]["k10"]
    assert abs(d["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10


def test_bench_two_rank_rehearsal_onesided_halo():
    """The same rehearsal with --cm onesided_put_active: the two ranks are
    processes sharing GPU 0, the halo moves by peer stores into IPC-mapped
    windows (one put kernel per exchange), and the run lands on the same
    residual."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus",
         "2", "--steps", "10", "--warmup", "2", "--grid", "64", "--transport",
         "gloo", "--cm", "onesided_put_active"],
        capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 2, 10, 2)
    assert "peer stores" in d["config"]["halo"] and d["halo_selfcheck"] == "ok"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "64", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    k10 = _line(one.stdout)["cg_rel_residual"]["k10"]
    assert abs(d["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10


def _gpu_count():
    import torch
    return torch.cuda.device_count()  # does not initialise the GPU


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs (RCCL refuses two "
                    "ranks on one device); the 1-GPU boxes run the gloo rehearsal")
def test_bench_two_gpus_over_rccl():
    """The real transport: two ranks, one GPU each, RCCL send/recv for the halo
    on the side stream and RCCL all-reduce (its own communicator) for the
    scalars.  The line must prove itself: RCCL reports 2 ranks, the halo
    self-check passed, and the run lands on the one-rank residual."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus",
         "2", "--steps", "12", "--warmup", "2", "--grid", "128"],
        capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d, 2, 12, 2)
    assert d["rccl"]["nranks"] == 2 and d["rccl"]["separate_reduction_comm"]
    assert d["halo_selfcheck"] == "ok"
    assert [r["ghosts"] for r in d["ranks"]] == [128 * 128] * 2
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--grid", "128", "--steps", "12", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extras"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    k10 = _line(one.stdout)["cg_rel_residual"]["k10"]
    assert abs(d["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10
