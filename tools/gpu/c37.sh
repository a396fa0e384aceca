cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 400 python -m pytest tests/test_gpu_matrix.py -q -m gpu -k "peer_reduce" > gpurun_out/r04/t37.log 2>&1
grep -v "^  File\|^    \|threading.py\|^$" gpurun_out/r04/t37.log | tail -25
