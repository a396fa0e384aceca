set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "fem_like" > gpurun_out/r04/t1.log 2>&1 || { tail -30 gpurun_out/r04/t1.log; exit 1; }
tail -3 gpurun_out/r04/t1.log
python tools/mbench.py --kind fem fem_tail fem81 unstructured --variants auto rowblock vector vector8 vector16 vector32 vector64 scalar > gpurun_out/r04/mbench1.jsonl 2> gpurun_out/r04/mbench1.err || { tail -20 gpurun_out/r04/mbench1.err; exit 1; }
cat gpurun_out/r04/mbench1.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'),d.get('avg_row'),d.get('plan_ms'), d.get('error',''))
"
