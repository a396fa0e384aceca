cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 240 python tools/rehearsal_threads.py --peer-reduce --cm onesided_put_active > gpurun_out/r04/rehearsal_peer_reduce_put.log 2>&1 || { grep -v "^  File\|^    \|threading.py" gpurun_out/r04/rehearsal_peer_reduce_put.log | head -12; exit 1; }
tail -c 700 gpurun_out/r04/rehearsal_peer_reduce_put.log | head -c 500
