#!/bin/bash
# PMC passes of the two kernels that stream the caller's CSR arrays, 512^3
set -e
cd "$GRAFT_REPO_ROOT"
R=r05; OUT=gpurun_out/pmc_$R; OFF=4611686018427387904
export PMC_GROUPS="ea write tcc"
PMC_SCRIPT=tools/prof_spmv.py tools/pmc_passes.sh $OUT/csr_rowblock_spmv csr_rowblock_spmv --n 512 --reps 6 --no-lx --ctx sj_min_nnz=$OFF
python3 tools/pmc_to_profiles.py $OUT/csr_rowblock_spmv $R --grid 512 --record csr_rowblock_spmv --out-dir gpurun_out/profiles_$R
PMC_SCRIPT=tools/prof_spmv.py tools/pmc_passes.sh $OUT/csr_gather_spmv csr_gather_spmv --n 512 --reps 6 --no-lx --ctx sj_min_nnz=$OFF xw_min_nnz=$OFF
python3 tools/pmc_to_profiles.py $OUT/csr_gather_spmv $R --grid 512 --record csr_gather_spmv --out-dir gpurun_out/profiles_$R
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/profiles_r05/r05_pmc_summary.json"))
for k,v in d["records"].items():
    print(k, v["kernel_prefix"], "read", v["fabric_read_bytes"]/1e9, "write", v["write_bytes"]/1e9, "ms", v["ms_profiled"])
PY
