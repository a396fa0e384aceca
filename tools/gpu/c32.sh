set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py -q -x -m gpu -k "symmetric or lower_split or kat" > gpurun_out/r04/t32.log 2>&1 || { tail -60 gpurun_out/r04/t32.log; exit 1; }
tail -2 gpurun_out/r04/t32.log
