set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_matrix.py -q -x -m gpu -k "peer_reduce or multirank_on_one_gpu or onesided" > gpurun_out/r04/t35.log 2>&1 || { tail -80 gpurun_out/r04/t35.log; exit 1; }
tail -3 gpurun_out/r04/t35.log
