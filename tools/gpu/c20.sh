set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py -q -x -k "wide_diagonal or box27 or wdia or stencil27 or values_changed" > gpurun_out/r04/c20_test.log 2>&1 || { tail -40 gpurun_out/r04/c20_test.log; exit 1; }
tail -3 gpurun_out/r04/c20_test.log
SPMV_WDIA_TRACE=1 timeout -k 10 300 python tools/prof_matrix.py --kind stencil27 --n 256 --set const_diagonals=0 --reps 2 > gpurun_out/r04/c20_trace.log 2>&1
cat gpurun_out/r04/c20_trace.log
timeout -k 10 300 python tools/mbench.py --kind poisson256 --set poisson_stencil=27 const_diagonals=0 --variants auto wdia_hbox=0 auto 2>&1 | cut -c1-400
