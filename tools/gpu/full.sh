set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( timeout -k 10 1700 python -m pytest tests -q -x -m gpu > gpurun_out/r04/gputest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gputest_full.log ) &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 60; tail -c 200 gpurun_out/r04/gputest_full.log | tail -1; done
tail -5 gpurun_out/r04/gputest_full.log
