set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
rm -rf /tmp/rp_r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r04 -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r04_bench_under_rocprof.log 2>&1
cp "$(find /tmp/rp_r04 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r04_rocprof_bench_n512_kernel_stats.csv
head -8 $R/gpurun_out/r04_rocprof_bench_n512_kernel_stats.csv | cut -c1-200
