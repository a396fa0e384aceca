set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_P timeout -k 10 200 python tools/mbench.py --kind fem_tail --reps 1 --no-check --variants sj_phases=1 > gpurun_out/r04/c25.log 2>&1
grep LTPROBE gpurun_out/r04/c25.log | tail -8
grep '^{' gpurun_out/r04/c25.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['variant'], d.get('ms'))"
