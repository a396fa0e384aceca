cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_SP timeout -k 10 200 python tools/mbench.py --kind fem --reps 1 --no-check --variants auto > gpurun_out/r04/c39.log 2>&1
grep SJPROBE gpurun_out/r04/c39.log | tail -6
grep '^{' gpurun_out/r04/c39.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], d.get('ms'), d['form'].get('sj_wpb'), d['form'].get('sj_max_chunks'))"
