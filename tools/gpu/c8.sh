set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python tools/rehearsal_threads.py --grid 64 --ranks 4 > gpurun_out/r04/rehearsal_small.log 2> gpurun_out/r04/rehearsal_small.err || { tail -20 gpurun_out/r04/rehearsal_small.err; exit 1; }
cut -c1-600 gpurun_out/r04/rehearsal_small.log
timeout -k 10 900 python tools/rehearsal_threads.py > gpurun_out/r04/rehearsal_8rank_default.log 2> gpurun_out/r04/rehearsal_8rank_default.err || { tail -20 gpurun_out/r04/rehearsal_8rank_default.err; exit 1; }
cut -c1-1500 gpurun_out/r04/rehearsal_8rank_default.log
timeout -k 10 900 python tools/rehearsal_threads.py --cm onesided_put_active > gpurun_out/r04/rehearsal_8rank_put.log 2> gpurun_out/r04/rehearsal_8rank_put.err || { tail -20 gpurun_out/r04/rehearsal_8rank_put.err; exit 1; }
cut -c1-1500 gpurun_out/r04/rehearsal_8rank_put.log
