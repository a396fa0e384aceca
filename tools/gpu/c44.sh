set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/profiles_r04b
export PMC_GROUPS="ea write tcc"
for rec in fem_sym fem_tail; do
  PMC_SCRIPT=tools/prof_matrix.py bash tools/pmc_passes.sh gpurun_out/pmc_r04b/${rec}_spmv ${rec}_spmv --kind $rec --rows 10000000 > gpurun_out/r04/pmc44_$rec.log 2>&1
done
ls gpurun_out/pmc_r04b/*
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default.log 2>&1 || { tail -20 gpurun_out/r04/bench_default.log; exit 1; }
python tools/show_bench.py gpurun_out/r04/bench_default.log | sed -n 1,3p
