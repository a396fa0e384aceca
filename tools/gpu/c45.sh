set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_matrix.py -q -x -m gpu -k "petsc_file" > gpurun_out/r04/t45.log 2>&1 || { tail -40 gpurun_out/r04/t45.log; exit 1; }
tail -2 gpurun_out/r04/t45.log
