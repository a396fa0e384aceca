set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( time timeout -k 10 1500 python bench.py > gpurun_out/r04/bench_default.log 2> gpurun_out/r04/bench_default.err ) 2> gpurun_out/r04/bench_default.time || { tail -20 gpurun_out/r04/bench_default.err; exit 1; }
cat gpurun_out/r04/bench_default.time
python tools/show_bench.py gpurun_out/r04/bench_default.log 2>/dev/null | head -80 || true
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/bench_default.log").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
r=d["roofline"]; print("main frac", r["frac"], r["kernel"][:60], r["avg_launch_ms"])
print("csr_order", json.dumps(r.get("csr_order"), indent=None)[:600])
print("general_cg", r.get("general_cg_iters_per_s"))
print("ragged", json.dumps(r.get("ragged"))[:1500])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cg_rel_residual_k10"), d["cg_rel_residual"])
for k in ("csr_lx_spmv","csr_rowblock_spmv","csr_sjds_spmv","north_star_lx_spmv","north_star_rowblock_spmv","stencil27_value_stream_spmv","unstructured_spmv","value_stream_spmv"):
    if k in d: print(k, d[k]["ms_per_apply"], d[k]["frac"], d[k]["frac_csr_equivalent"], d[k].get("plan_ms"))
PY
