set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py -q -x -m gpu -k "mixed or sliced_jagged or long_rows or values_changed" > gpurun_out/r04/t33.log 2>&1 || { tail -60 gpurun_out/r04/t33.log; exit 1; }
tail -2 gpurun_out/r04/t33.log
