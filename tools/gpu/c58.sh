set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "sliced_jagged or fem_like or long_rows or symmetric_storage or mixed_precision_sliced" > gpurun_out/r04/t58.log 2>&1 || { tail -40 gpurun_out/r04/t58.log; exit 1; }
tail -2 gpurun_out/r04/t58.log
timeout -k 10 300 python tools/mbench.py --kind fem81 fem_long fem --variants auto > gpurun_out/r04/c58.log 2>&1
grep '^{' gpurun_out/r04/c58.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], round(d['avg_row'],1), d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'), d.get('plan_ms'))"
