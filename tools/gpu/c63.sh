set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export PMC_GROUPS="ea write tcc"
PMC_SCRIPT=tools/prof_matrix.py bash tools/pmc_passes.sh gpurun_out/pmc_r04f/unstructured_spmv unstructured_spmv --kind unstructured --rows 10000000 > gpurun_out/r04/pmc63.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r04 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_under_rocprof.log 2>&1
cp "$(find /tmp/rp_r04 -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r04_rocprof_bench_n512_kernel_stats.csv
echo done
