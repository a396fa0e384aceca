cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c40.log
for w in 0 4 8 16; do
  for k in fem unstructured; do
  timeout -k 10 200 python tools/mbench.py --kind $k --variants auto --set sj_wpb=$w >> gpurun_out/r04/c40.log 2>&1
  done
done
grep '^{' gpurun_out/r04/c40.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kind'], d['form'].get('sj_wpb'), d.get('ms'), d.get('frac_csr'), d['form'].get('sj_max_chunks'), d['form'].get('sj_staged_bytes_per_entry_x100'), d.get('bit_equal_scalar'))"
