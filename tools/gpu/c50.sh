set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "sliced_jagged or long_rows or values_changed or symmetric_storage or fem_like" > gpurun_out/r04/t50.log 2>&1 || { tail -50 gpurun_out/r04/t50.log; exit 1; }
tail -2 gpurun_out/r04/t50.log
rm -f gpurun_out/r04/c50.log
for sg in 1 0 1 0; do
  timeout -k 10 300 python tools/mbench.py --kind fem fem_tail fem81 unstructured --variants auto --set sj_sigma=$sg >> gpurun_out/r04/c50.log 2>&1
done
grep '^{' gpurun_out/r04/c50.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'), d['form'].get('sj_wpb'), d.get('plan_ms'))"
