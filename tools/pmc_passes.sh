#!/bin/bash
# rocprofv3 PMC passes over tools/prof_spmv.py: one counter group per pass
# (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2), kernel trace only.
#   tools/pmc_passes.sh OUTDIR TAG [prof_spmv.py arguments...]
# writes OUTDIR/TAG_<group>_counter_collection.csv; summarise with
#   python tools/pmc_summary.py OUTDIR
set -euo pipefail
OUT="$1"; TAG="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
declare -A CGROUPS=(
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
  [ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"
  [ea32]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"
  [tcc]="TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum"
  [sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
  [tcp]="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
)
# PMC_GROUPS="ea write" runs a subset of the passes; PMC_SCRIPT=tools/prof_matrix.py
# profiles the non-Poisson test matrices instead
for g in ${PMC_GROUPS:-fetch write ea tcc sq tcp}; do
  rm -rf "/tmp/pmc_${TAG}_$g"
  rocprofv3 --pmc ${CGROUPS[$g]} --kernel-trace --output-format csv \
      -d "/tmp/pmc_${TAG}_$g" -o p -- python3 "$ROOT/${PMC_SCRIPT:-tools/prof_spmv.py}" "$@" \
      > "/tmp/pmc_${TAG}_$g.log" 2>&1
  f=$(find "/tmp/pmc_${TAG}_$g" -name "*counter_collection.csv" | head -1)
  cp "$f" "$OUT/${TAG}_${g}_counter_collection.csv"
  echo "$TAG $g done"
done
