set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wide_diagonal" > gpurun_out/r04/t10.log 2>&1 || { tail -30 gpurun_out/r04/t10.log; exit 1; }
tail -2 gpurun_out/r04/t10.log
timeout -k 10 900 python -m pytest tests/test_gpu_matrix.py -q -x -m gpu -k "27_point" >> gpurun_out/r04/t10.log 2>&1 || { tail -30 gpurun_out/r04/t10.log; exit 1; }
tail -2 gpurun_out/r04/t10.log
python tools/plan_cost.py 2>&1 | tail -12
