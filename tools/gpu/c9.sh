set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/fold_probe.log
for v in A B A B; do
  echo "variant $v" >> gpurun_out/r04/fold_probe.log
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 600 python bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'spmv_ms':d['roofline']['avg_launch_ms'],'k10':d['cg_rel_residual']['k10']}))" >> gpurun_out/r04/fold_probe.log
done
cat gpurun_out/r04/fold_probe.log
