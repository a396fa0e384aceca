cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c57.log
for sg in 0 2 0 2; do
  timeout -k 10 300 python tools/mbench.py --kind fem_long fem_mid fem81 --rows 5000000 --variants auto --set sj_sigma=$sg >> gpurun_out/r04/c57.log 2>&1
done
grep '^{' gpurun_out/r04/c57.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], round(d['avg_row'],1), d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'), d['form'].get('sj_wpb'))"
