set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/profiles_r04d
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04/gputests_full.log 2>&1 || { tail -40 gpurun_out/r04/gputests_full.log; exit 1; }
tail -2 gpurun_out/r04/gputests_full.log
export PMC_GROUPS="ea write tcc"
for rec in fem fem_tail fem81 fem_sym unstructured; do
  PMC_SCRIPT=tools/prof_matrix.py bash tools/pmc_passes.sh gpurun_out/pmc_r04d/${rec}_spmv ${rec}_spmv --kind $rec --rows 10000000 > gpurun_out/r04/pmc51_$rec.log 2>&1
done
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default.log 2>&1 || { tail -20 gpurun_out/r04/bench_default.log; exit 1; }
python tools/show_bench.py gpurun_out/r04/bench_default.log | sed -n 1,2p
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r04 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_under_rocprof.log 2>&1
cp "$(find /tmp/rp_r04 -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r04_rocprof_bench_n512_kernel_stats.csv
echo done
