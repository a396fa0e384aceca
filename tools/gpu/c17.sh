set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
R=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$R
export PMC_GROUPS="ea write tcc"
PMC_SCRIPT=tools/prof_matrix.py tools/pmc_passes.sh $OUT/stencil27_value_stream_spmv stencil27_value_stream_spmv --kind stencil27 --n 256 --set const_diagonals=0 > gpurun_out/r04/pmc_s27v.log 2>&1
python3 tools/pmc_to_profiles.py $OUT/stencil27_value_stream_spmv $R --grid 256 --record stencil27_value_stream_spmv --out-dir gpurun_out/profiles_$R >> gpurun_out/r04/pmc_s27v.log 2>&1
tail -3 gpurun_out/r04/pmc_s27v.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp27
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp27 -o s27 -- python3 $GRAFT_REPO_ROOT/tools/prof_matrix.py --kind stencil27 --n 256 --set const_diagonals=0 > $GRAFT_REPO_ROOT/gpurun_out/r04/rp27.log 2>&1
cp $(find /tmp/rp27 -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04/rp27_kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/r04/rp27_kernel_stats.csv | cut -c1-200
