#!/bin/bash
# All PMC passes behind bench.py's `traffic` fields, one record after the other
# (run through gpurun; ~5 minutes), and the rocprofv3 kernel statistics of the
# bench command itself.
#   tools/pmc_round.sh r03
set -euo pipefail
R="$1"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_$R"
export PMC_GROUPS="${PMC_GROUPS:-ea write tcc}"
ONLY="${PMC_ONLY:-}" # PMC_ONLY="rec1 rec2": just those records, no bench profile
run() { # record grid script args...
  if [ -n "$ONLY" ] && ! [[ " $ONLY " == *" $1 "* ]]; then return 0; fi
  local rec="$1" grid="$2" script="$3"; shift 3
  PMC_SCRIPT="$script" "$ROOT/tools/pmc_passes.sh" "$OUT/$rec" "$rec" "$@"
  python3 "$ROOT/tools/pmc_to_profiles.py" "$OUT/$rec" "$R" --grid "$grid" \
      --record "$rec" --out-dir "$ROOT/gpurun_out/profiles_$R"
}
run main 512 tools/prof_spmv.py --n 512 --reps 6 --dot
run symmetric 512 tools/prof_spmv.py --n 512 --reps 6 --dot --symmetric
run value_stream_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --ctx const_diagonals=0
run symmetric_value_stream_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --symmetric --ctx const_diagonals=0
run csr_nonsymmetric_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --asym
run csr_lattice_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --no-bake
run csr_lx_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --no-lat
OFF=4611686018427387904
# (the probe between XW and the gather kernel off: this record is the XW kernel)
run csr_rowblock_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --no-lx --ctx sj_min_nnz=$OFF xw_probe=0
run csr_gather_spmv 512 tools/prof_spmv.py --n 512 --reps 6 --no-lx --ctx sj_min_nnz=$OFF xw_min_nnz=$OFF
run stencil27_spmv 256 tools/prof_matrix.py --kind stencil27 --n 256
run stencil27_value_stream_spmv 256 tools/prof_matrix.py --kind stencil27 --n 256 --set const_diagonals=0
run unstructured_spmv 10000000 tools/prof_matrix.py --kind unstructured --rows 10000000
run fem_spmv 10000000 tools/prof_matrix.py --kind fem --rows 10000000
run fem_tail_spmv 10000000 tools/prof_matrix.py --kind fem_tail --rows 10000000
run fem81_spmv 10000000 tools/prof_matrix.py --kind fem81 --rows 10000000
run fem_sym_spmv 10000000 tools/prof_matrix.py --kind fem_sym --rows 10000000
run fem_tail_sym_spmv 10000000 tools/prof_matrix.py --kind fem_tail_sym --rows 10000000
run csr_order 512 tools/prof_spmv.py --n 512 --reps 6 --no-lat --dot
if [ -n "$ONLY" ]; then echo "pmc round $R done (only: $ONLY)"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$R -o bench -- \
    python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/gpurun_out/${R}_bench_under_rocprof.log" 2>&1
cp "$(find /tmp/rp_$R -name '*kernel_stats.csv' | head -1)" \
   "$ROOT/gpurun_out/${R}_rocprof_bench_n512_kernel_stats.csv"
echo "pmc round $R done"
