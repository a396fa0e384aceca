set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/profiles_r04
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r04/gputests_full.log 2>&1 || { tail -40 gpurun_out/r04/gputests_full.log; exit 1; }
tail -2 gpurun_out/r04/gputests_full.log
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_default.log 2>&1 || { tail -20 gpurun_out/r04/bench_default.log; exit 1; }
python tools/show_bench.py gpurun_out/r04/bench_default.log | head -30
timeout -k 10 1500 bash tools/pmc_round.sh r04 > gpurun_out/r04/pmc_round.log 2>&1 || { tail -30 gpurun_out/r04/pmc_round.log; exit 1; }
tail -3 gpurun_out/r04/pmc_round.log
