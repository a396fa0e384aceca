set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04/gputests_full.log 2>&1 || { tail -40 gpurun_out/r04/gputests_full.log; exit 1; }
tail -2 gpurun_out/r04/gputests_full.log
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default.log 2>&1 || { tail -20 gpurun_out/r04/bench_default.log; exit 1; }
python tools/show_bench.py gpurun_out/r04/bench_default.log | head -4
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04/bench_default.log") if l.startswith("{")][-1])
for k,v in d["roofline"]["ragged"].items():
    if isinstance(v, dict): print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_apply","frac","speedup_over_fp64","iters/s","bit_equal_csr_order_kernel","bit_equal_transposed_map_kernel","bit_equal_one_lane_per_row")})
PY
