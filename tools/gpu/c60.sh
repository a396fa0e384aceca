cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 200 python tools/rehearsal_threads.py > gpurun_out/r04/rehearsal_default.log 2>&1; echo "default rc=$?"
tail -c 400 gpurun_out/r04/rehearsal_default.log | head -c 300; echo
timeout -k 10 200 python tools/rehearsal_threads.py --cm onesided_put_active > gpurun_out/r04/rehearsal_put.log 2>&1; echo "put rc=$?"
tail -c 400 gpurun_out/r04/rehearsal_put.log | head -c 300; echo
