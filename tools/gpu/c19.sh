set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SPMV_WDIA_TRACE=1 timeout -k 10 300 python tools/prof_matrix.py --kind stencil27 --n 256 --set const_diagonals=0 --reps 2 > gpurun_out/r04/c19_trace.log 2>&1
cat gpurun_out/r04/c19_trace.log
timeout -k 10 300 python tools/mbench.py --kind poisson256 --set poisson_stencil=27 const_diagonals=0 --variants auto --no-check 2>&1 | cut -c1-300
