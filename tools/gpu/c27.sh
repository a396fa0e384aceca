set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c27.log
for v in P0 P1 P2 P3; do
  echo "variant $v" >> gpurun_out/r04/c27.log
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 200 python tools/mbench.py --kind fem_tail --reps 1 --no-check --variants sj_phases=1 > gpurun_out/r04/c27_$v.log 2>&1
  grep "LTPROBE wg 0 " gpurun_out/r04/c27_$v.log | tail -2 >> gpurun_out/r04/c27.log
  grep "LTPROBE wg 300 " gpurun_out/r04/c27_$v.log | tail -2 >> gpurun_out/r04/c27.log
done
cat gpurun_out/r04/c27.log
