set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
./tools/probes/lds_probe > gpurun_out/r04/lds_probe.log 2>&1
cat gpurun_out/r04/lds_probe.log
R=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$R
export PMC_GROUPS="ea write tcc"
PMC_SCRIPT=tools/prof_matrix.py tools/pmc_passes.sh $OUT/fem_tail_spmv fem_tail_spmv --kind fem_tail --rows 10000000 > gpurun_out/r04/pmc_femtail.log 2>&1
python3 tools/pmc_to_profiles.py $OUT/fem_tail_spmv $R --grid 10000000 --record fem_tail_spmv --out-dir gpurun_out/profiles_$R >> gpurun_out/r04/pmc_femtail.log 2>&1
tail -5 gpurun_out/r04/pmc_femtail.log
