#!/bin/bash
# PROBE 2 of the peer-reduce + one-sided-put pair with the ranks as threads:
# which rank count (= how many HSA queues one process holds: 3 per rank + 1)
# still completes?  Bounded waits (3 s), outer timeouts.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for r in 2 4 6 7; do
  echo "=== pair, $r thread ranks, 128^3, poll_every=1"
  ( time timeout -k 10 120 python tools/rehearsal_threads.py --ranks $r --grid 128 --steps 24 \
      --cm onesided_put_active --peer-reduce --allow-pair --timeout-ms 3000 \
      --poll-every 1 ) > gpurun_out/r05/pair_threads_ranks$r.log 2>&1
  echo "rc=$?"; tail -c 400 gpurun_out/r05/pair_threads_ranks$r.log
done
echo "=== pair, 6 PROCESS ranks sharing GPU 0 over IPC, 128^3"
( time SPMV_ALLOW_PUT_WITH_PEER_REDUCE=1 timeout -k 10 200 python bench.py --gpus 6 --grid 128 \
    --steps 60 --warmup 5 --transport gloo --cm onesided_put_active --peer-reduce --put-timeout-ms 3000 \
    --detail gpurun_out/r05/pair_procs6_detail.json ) > gpurun_out/r05/pair_procs6.log 2>&1
echo "rc=$?"; grep '^{' gpurun_out/r05/pair_procs6.log | tail -c 700
exit 0
