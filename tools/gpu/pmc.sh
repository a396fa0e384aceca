set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( bash tools/pmc_round.sh r04 > gpurun_out/r04/pmc_round.log 2>&1; echo "rc=$?" >> gpurun_out/r04/pmc_round.log ) &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 45; tail -1 gpurun_out/r04/pmc_round.log; done
tail -5 gpurun_out/r04/pmc_round.log
ls gpurun_out/profiles_r04 | wc -l
