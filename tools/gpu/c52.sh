cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c52.log
for u in 2 4 1 2 4; do
  timeout -k 10 300 python tools/mbench.py --kind fem unstructured --variants auto --set sj_unit=$u >> gpurun_out/r04/c52.log 2>&1
done
grep '^{' gpurun_out/r04/c52.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], d['form'].get('sj_unit'), d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'))"
