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
LINE_MAX = 6144  # what the driver parses; BENCH_r04's 29,947-byte line it could not


def _line(out):
    """the ONE line on stdout that starts with a brace: the compact record"""
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    assert out.rstrip().endswith(lines[0]), "the line must come last"
    assert len(lines[0]) <= LINE_MAX, len(lines[0])
    return json.loads(lines[0])


def _run(args, tmp_path, launcher=(), timeout=600):
    """bench.py with its detail file under tmp_path -> (line, detail)"""
    detail = str(tmp_path / "detail.json")
    res = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"),
                          *args, "--detail", detail],
                         capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    # stderr carries the detail too, never as a line starting with a brace
    assert not [ln for ln in res.stderr.splitlines() if ln.startswith("{")]
    line = _line(res.stdout)
    assert line["detail"] == detail
    return line, json.load(open(detail))


def _rel(a, b):
    return abs(a - b) <= 2e-5 * abs(b)


def _check(line, d, n_gpus, steps, warmup, plain=True):
    for k in REQUIRED:
        assert k in line and k in d, k
        if k != "roofline":
            assert line[k] == d[k], k
    assert line["n_gpus"] == n_gpus and line["steps"] == steps
    assert line["warmup"] == warmup
    assert line["unit"] == "iters/s" and line["higher_is_better"] is True
    assert line["dtype"] == "f64" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    c = line["roofline"]
    assert c["bound"] == "hbm" and c["peak"] == 8000.0 and c["unit"] == "GB/s"
    assert " " not in c["kernel"].split("<")[0]  # a name, not a paragraph
    assert _rel(c["frac"], c["achieved"] / c["peak"])
    assert c["launches_timed"] == steps and c["avg_launch_ms"] > 0
    assert _rel(c["achieved"], c["bytes_per_launch"] / c["avg_launch_ms"] / 1e6)
    r = d["roofline"]
    if plain:
        # the headline prices SURVEY 8d's algorithmic bytes of the CSR-order
        # plan: every stored value and an index per entry are streamed
        assert c["plan"].startswith("csr-order")
        assert c["bytes_per_launch"] == r["algorithmic_bytes_per_launch"]
        assert c["format_bytes_per_launch"] <= c["bytes_per_launch"]
        rows = d["plan"]["csr_bytes"]  # nnz * 12 + (rows + 1) * 4
        assert c["format_bytes_per_launch"] >= 0.8 * rows
    else:
        # physical: the bytes the kernel's format moves over the measured time
        assert c["plan"].startswith("AUTO")
        assert c["bytes_per_launch"] == c["format_bytes_per_launch"]
        assert r["algorithmic_bytes_per_launch"] >= c["bytes_per_launch"]
    assert _rel(c["frac_format"], c["format_bytes_per_launch"]
                / c["avg_launch_ms"] / 1e6 / 8000.0)
    assert _rel(r["frac_csr_equivalent"], r["algorithmic_bytes_per_launch"]
                / r["avg_launch_ms"] / 1e6 / 8000.0)
    assert (c["traffic"] is None) == (c["frac_traffic"] is None)
    assert (c["traffic"] is None) == (c["traffic_source"] is None)
    # ... the detail file: the same numbers at full precision, with the notes
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "note" in r
    assert _rel(c["frac"], r["frac"]) and _rel(c["avg_launch_ms"], r["avg_launch_ms"])
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert line["cg_rel_residual"]["k10"] == d["cg_rel_residual"]["k10"] > 0
    assert d["cg_rel_residual"]["kK"] > 0
    assert d["plan"]["plan_ms"] >= 0 and d["plan"]["plan_extra_bytes"] >= 0
    assert d["plan"]["csr_bytes"] > 0


def test_bench_single_gpu_line(tmp_path):
    line, d = _run(["--grid", "64", "--steps", "20", "--warmup", "3",
                    "--cpu-n", "32", "--cpu-iters", "3", "--mixed-grid",
                    "48", "--stencil27-grid", "40", "--unstructured-rows",
                    "200000", "--fem-rows", "200000"], tmp_path)
    _check(line, d, 1, 20, 3)
    # the compact line: what the driver keeps
    lc, lr = line["cpu_baseline"], line["roofline"]
    assert set(lc) >= {"value", "unit", "cores", "kind", "sample", "spmv_omp_gbs",
                       "k10"}
    assert lc["kind"] == "port" and lc["cores"] >= 1 and lc["value"] > 0
    assert "32^3" in lc["sample"] and len(lc["sample"]) < 160
    assert lc["parity_checks_bit_exact"] is True
    # the headline is the CSR-order plan's kernel ...
    assert lr["kernel"] == "csr_lxw_kernel<double>"
    # ... and what the AUTO plan does with this matrix stands beside it
    sp = lr["specialised"]
    assert sp["kernel"].startswith("csr_const_dia_tile_kernel<double, general order")
    assert sp["iters_per_s"] > 0 and sp["ms"] > 0
    assert sp["frac_csr_equivalent"] > sp["frac_physical"] > 0
    ns = lr["north_star"]
    assert ns["rows"] == 216 ** 3 and 0 < ns["rowblock_frac"] < 1
    assert 0 < ns["lx_frac"] < 1 and ns["default_frac"] > 0
    assert set(lr["ragged"]) >= {"fem_spmv", "fem_tail_spmv", "fem81_spmv",
                                 "fem_sym_spmv", "fem_tail_sym_spmv",
                                 "unstructured_spmv"}
    for k, v in lr["ragged"].items():
        if k != "fem_sym_cg_iters_per_s":
            assert len(v) == 2 and v[0] > 0 and v[1] > 0, k
    assert lr["plan_extra_over_csr_bytes"] >= 0
    assert line["symmetric"]["iters_per_s"] > 0
    assert line["symmetric"]["specialised"]["iters_per_s"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["cores"] <= c["host"]["physical_cores"] and "32^3" in c["sample"]
    assert c["spmv_omp"]["GB/s"] > 0
    assert "sample" in c and d["north_star_spmv"]["rows"] == 216 ** 3
    # BASELINE configs[3] in the default line: symmetric storage, atomic-free
    sym = d["symmetric"]
    assert sym["frac"] > 0 and sym["iters/s"] > 0 and "atomic-free" in sym["kernel"]
    assert sym["form"]["sdia_const"] == 0 and sym["specialised"]["iters/s"] > 0
    assert abs(sym["specialised"]["cg_rel_residual_k10"]
               / sym["cg_rel_residual_k10"] - 1) < 1e-9
    assert d["csr_lx_spmv"]["form"]["lx"] == 1 and d["csr_lx_spmv"]["form"]["lat"] == 0
    # the main plan: the LX form, the lattice analysis off
    assert d["plan"]["form"]["lx"] == 1 and d["plan"]["form"]["lat"] == 0
    assert d["plan"]["form"]["sdia"] == 0
    # the AUTO plan: every diagonal of the Poisson matrix is constant, no values
    # are streamed
    spc = d["specialised_cg"]
    assert spc["form"]["sdia"] == 1 and spc["form"]["sdia_const"] == 1
    assert "csr_const_dia_tile_kernel<double, general order, 4 lattice lines" \
        in spc["kernel"]
    assert spc["requested_bytes"] == 17 * 64 ** 3
    assert abs(spc["cg_rel_residual_k10"] / d["cg_rel_residual"]["k10"] - 1) < 1e-9
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
    # the caller's CSR arrays as they are: the gather kernel, and (from 2^20
    # entries on -- 64^3 has more) the XW kernel with the x windows staged
    assert d["csr_gather_spmv"]["form"] == dict(lat=0, lx=0, lxw=0, xw=0, sjds=0,
                                                sym_sj=0, wdia=0, wdia_const=0,
                                                wdia_hbox=0, slat=0,
                                                sdia=0, sdia_const=0, sym_det=0,
                                                zwalk=0)
    assert "csr_rowblock_kernel" in d["csr_gather_spmv"]["kernel"]
    # (XW only where x outgrows the caches: not at this test's 64^3)
    rbf = d["csr_rowblock_spmv"]["form"]
    assert rbf["xw"] == 0 and rbf["lx"] == 0 and rbf["sjds"] == 0 and rbf["lat"] == 0
    assert len(lr["csr_rowblock"]) == 3 and len(lr["csr_gather"]) == 3
    assert d["csr_sjds_spmv"]["form"]["sjds"] == 1
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


def test_bench_without_the_symmetry_check(tmp_path):
    """--no-bake: the line of a lattice matrix whose values stay in CSR order
    (the CSR-order lattice kernel; DESIGN.md section 7, 'reading the headline')."""
    line, d = _run(["--grid", "128", "--steps", "10", "--warmup", "2",
                    "--specialised", "--no-bake", "--no-extras",
                    "--no-cpu-baseline"], tmp_path)
    _check(line, d, 1, 10, 2, plain=False)
    assert line["roofline"]["kernel"] == "csr_lattice_kernel<double>"
    assert d["plan"]["form"]["sdia"] == 0 and d["plan"]["form"]["lat"] == 1
    assert "csr_lattice_kernel" in d["roofline"]["kernel"]
    assert d["roofline"]["frac"] <= d["roofline"]["frac_csr_equivalent"]


def test_bench_with_the_values_streamed(tmp_path):
    """--no-const: the headline of a lattice matrix whose coefficients vary (the
    half diagonal form streams the lower values)."""
    line, d = _run(["--grid", "128", "--steps", "10", "--warmup", "2",
                    "--specialised", "--no-const", "--no-extras",
                    "--no-cpu-baseline"], tmp_path)
    _check(line, d, 1, 10, 2, plain=False)
    assert d["plan"]["form"]["sdia"] == 1 and d["plan"]["form"]["sdia_const"] == 0
    assert "csr_sym_dia_kernel<double, general order>" in d["roofline"]["kernel"]


def _torchrun(nproc):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
            str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]


def _k10_one_rank(grid, steps, tmp_path):
    (tmp_path / "one").mkdir(exist_ok=True)
    line, _ = _run(["--grid", str(grid), "--steps", str(steps), "--warmup", "2",
                    "--no-cpu-baseline", "--no-extras"], tmp_path / "one")
    return line["cg_rel_residual"]["k10"]


def test_bench_two_rank_rehearsal_self_launched(tmp_path, monkeypatch):
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the
    environment (the shape of the driver's N = 1 command): bench.py starts the
    two ranks itself as child processes -- the parent never touches the GPU --
    and forwards rank 0's line."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    line, d = _run(["--gpus", "2", "--steps", "10", "--warmup", "2", "--grid", "64",
                    "--transport", "gloo"], tmp_path)
    _check(line, d, 2, 10, 2)
    assert "child processes" in line["launcher"]
    assert "REHEARSAL" in line["data"] and "cpu_baseline" not in line
    assert line["halo_selfcheck"] == "ok" and len(line["ranks"]) == 2
    # per rank: [neighbours, ghosts, rows]
    assert [r[0] for r in line["ranks"]] == [1, 1]
    assert [r[1] for r in line["ranks"]] == [64 * 64, 64 * 64]
    assert sum(r[2] for r in line["ranks"]) == 64 ** 3
    assert [r["ghosts"] for r in d["ranks"]] == [64 * 64, 64 * 64]
    # the distributed run reproduces the one-rank residual after 10 iterations
    k10 = _k10_one_rank(64, 10, tmp_path)
    assert abs(line["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10


def test_bench_self_launch_reports_a_failing_rank(tmp_path, monkeypatch):
    """a child that dies gives the parent a non-zero exit code and no line"""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus",
                          "2", "--steps", "4", "--warmup", "1", "--grid", "32",
                          "--transport", "gloo", "--petsc-matrix",
                          str(tmp_path / "no_such_file.dat")],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_two_rank_rehearsal_onesided_halo(tmp_path):
    """The rehearsal under torch.distributed.run (the driver's N > 1 command)
    with --cm onesided_put_active: the two ranks are processes sharing GPU 0,
    the halo moves by peer stores into IPC-mapped windows (one put kernel per
    exchange), and the run lands on the same residual."""
    line, d = _run(["--gpus", "2", "--steps", "10", "--warmup", "2", "--grid", "64",
                    "--transport", "gloo", "--cm", "onesided_put_active"],
                   tmp_path, launcher=_torchrun(2))
    _check(line, d, 2, 10, 2)
    assert "launcher" not in line
    assert "peer stores" in line["config"]["halo"]
    assert line["halo_selfcheck"] == "ok"
    k10 = _k10_one_rank(64, 10, tmp_path)
    assert abs(line["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10


def test_bench_rehearsal_onesided_halo_with_peer_reduction(tmp_path):
    """The pair the library refuses for thread ranks -- one-sided halo + the
    deterministic peer reduction of the CG scalars -- in the production
    topology, one PROCESS per rank (4 ranks sharing GPU 0 over IPC handles):
    neither exchange touches the host, 40 iterations, the one-rank residual."""
    line, d = _run(["--gpus", "4", "--steps", "40", "--warmup", "3", "--grid", "64",
                    "--transport", "gloo", "--cm", "onesided_put_active",
                    "--peer-reduce", "--put-timeout-ms", "10000"],
                   tmp_path, launcher=_torchrun(4))
    _check(line, d, 4, 40, 3)
    assert "peer stores" in line["config"]["halo"]
    assert line["cg_scalar_reductions"].startswith("peer windows")
    assert line["halo_selfcheck"] == "ok"
    k10 = _k10_one_rank(64, 40, tmp_path)
    assert abs(line["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10


def _gpu_count():
    import torch
    return torch.cuda.device_count()  # does not initialise the GPU


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs (RCCL refuses two "
                    "ranks on one device); the 1-GPU boxes run the gloo rehearsal")
def test_bench_two_gpus_over_rccl(tmp_path, monkeypatch):
    """The real transport: two ranks, one GPU each, RCCL send/recv for the halo
    on the side stream and RCCL all-reduce (its own communicator) for the
    scalars -- started the way the driver may start it, `python bench.py --gpus
    2` without a launcher.  The line must prove itself: RCCL reports 2 ranks,
    the halo self-check passed, and the run lands on the one-rank residual."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    line, d = _run(["--gpus", "2", "--steps", "12", "--warmup", "2", "--grid",
                    "128"], tmp_path, timeout=900)
    _check(line, d, 2, 12, 2)
    assert line["rccl"]["nranks"] == 2 and line["rccl"]["separate_reduction_comm"]
    assert line["halo_selfcheck"] == "ok"
    assert [r[1] for r in line["ranks"]] == [128 * 128] * 2
    k10 = _k10_one_rank(128, 12, tmp_path)
    assert abs(line["cg_rel_residual"]["k10"] - k10) <= 1e-10 * k10
