#!/bin/bash
# one PMC pass (EA read requests + TCC) per probe library
set -euo pipefail
ROOT="$1"; OUT="$ROOT/gpurun_out/probe"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in base P1 P2 P3; do
  if [ "$v" = base ]; then unset SPMV_AMD_LIBDIR; else export SPMV_AMD_LIBDIR="$ROOT/spmv_amd/lib_$v"; fi
  rm -rf /tmp/probe_$v
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv \
    -d /tmp/probe_$v -o p -- python3 "$ROOT/tools/prof_spmv.py" --n 512 --symmetric --dot > /tmp/probe_$v.log 2>&1
  f=$(find /tmp/probe_$v -name "*counter_collection.csv" | head -1)
  cp "$f" "$OUT/${v}_ea_counter_collection.csv"
  echo "$v done"
done
