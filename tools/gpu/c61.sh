set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py -q -x -m gpu -k "sliced_jagged or fem_like or unstructured or mixed_precision_sliced" > gpurun_out/r04/t61.log 2>&1 || { tail -40 gpurun_out/r04/t61.log; exit 1; }
tail -2 gpurun_out/r04/t61.log
timeout -k 10 300 python tools/mbench.py --kind unstructured fem --variants auto > gpurun_out/r04/c61.log 2>&1
grep '^{' gpurun_out/r04/c61.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'), d['form'].get('sj_unit'), d['form'].get('sj_sigma'), d.get('plan_ms'))"
