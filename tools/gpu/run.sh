#!/bin/bash
# one gpurun call: tools/gpu/run.sh <tag> -- the default bench line (stdout and
# stderr kept apart, as the driver sees them) and the bench contract tests
set -e
cd "$GRAFT_REPO_ROOT"
TAG="${1:-r05}"
mkdir -p gpurun_out/$TAG
( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 \
    --detail gpurun_out/$TAG/bench_detail.json \
    > gpurun_out/$TAG/bench_default.log 2> gpurun_out/$TAG/bench_default.err ) \
    2> gpurun_out/$TAG/bench_default.time || { tail -20 gpurun_out/$TAG/bench_default.err; exit 1; }
cat gpurun_out/$TAG/bench_default.time
wc -c gpurun_out/$TAG/bench_default.log
cat gpurun_out/$TAG/bench_default.log
