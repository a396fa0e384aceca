set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py tests/test_gpu_bench.py -q -x -m gpu -k "symmetric or sliced_jagged or values_changed or long_rows or mixed or bench_single or kat" > gpurun_out/r04/t43.log 2>&1 || { tail -60 gpurun_out/r04/t43.log; exit 1; }
tail -2 gpurun_out/r04/t43.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04/bench43.log 2>&1 || { tail -20 gpurun_out/r04/bench43.log; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04/bench43.log") if l.startswith("{")][-1])
print(d["value"])
for k,v in d["roofline"]["ragged"].items():
    if isinstance(v, dict): print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_apply","frac","frac_requested","iters/s","bit_equal_transposed_map_kernel","plan_ms","transposed_map_kernel_ms")})
PY
