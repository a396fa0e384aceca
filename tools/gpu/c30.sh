set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench.py -q -x -m gpu -k "lower_split or symmetric_storage or bench_single" > gpurun_out/r04/t30.log 2>&1 || { tail -60 gpurun_out/r04/t30.log; exit 1; }
tail -2 gpurun_out/r04/t30.log
timeout -k 10 400 python bench.py > gpurun_out/r04/bench30.log 2>&1 || { tail -20 gpurun_out/r04/bench30.log; exit 1; }
python tools/show_bench.py gpurun_out/r04/bench30.log | grep -i "value\|fem\|unstr"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04/bench30.log") if l.startswith("{")][-1])
print(json.dumps(d["roofline"]["ragged"]["fem_sym_spmv"], indent=1))
PY
