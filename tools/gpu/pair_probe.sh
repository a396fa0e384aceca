#!/bin/bash
# PROBE of the peer-reduce + one-sided-put pair with the ranks as threads
# (VERDICT r04 #3c): does bounding the host's run-ahead make it complete?
# Every wait is bounded (3 s) and every run has an outer timeout.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for pe in 1 2 16; do
  echo "=== pair, 8 thread ranks, 128^3, poll_every=$pe" 
  ( time timeout -k 10 150 python tools/rehearsal_threads.py --grid 128 --steps 24 \
      --cm onesided_put_active --peer-reduce --allow-pair --timeout-ms 3000 \
      --poll-every $pe ) > gpurun_out/r05/pair_threads_poll$pe.log 2>&1
  echo "rc=$?"; tail -c 600 gpurun_out/r05/pair_threads_poll$pe.log
done
echo "=== pair, 4 PROCESS ranks sharing GPU 0 over IPC, 64^3 (bench.py --transport gloo)"
( time SPMV_ALLOW_PUT_WITH_PEER_REDUCE=1 timeout -k 10 200 python bench.py --gpus 4 --grid 64 \
    --steps 40 --warmup 5 --transport gloo --cm onesided_put_active --peer-reduce --put-timeout-ms 3000 \
    --detail gpurun_out/r05/pair_procs_detail.json ) > gpurun_out/r05/pair_procs.log 2>&1
echo "rc=$?"; grep '^{' gpurun_out/r05/pair_procs.log | tail -c 1500; tail -5 gpurun_out/r05/pair_procs.log | cut -c1-300
exit 0
