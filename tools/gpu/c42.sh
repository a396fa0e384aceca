cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/ea32
export PMC_GROUPS="ea32"
PMC_SCRIPT=tools/prof_matrix.py bash tools/pmc_passes.sh gpurun_out/r04/ea32 unstructured_spmv --kind unstructured --rows 10000000 > gpurun_out/r04/ea32.log 2>&1
PMC_SCRIPT=tools/prof_matrix.py bash tools/pmc_passes.sh gpurun_out/r04/ea32 fem_tail_spmv --kind fem_tail --rows 10000000 >> gpurun_out/r04/ea32.log 2>&1
python tools/pmc_summary.py gpurun_out/r04/ea32 > gpurun_out/r04/ea32_summary.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/ea32_summary.json"))
for tag, ks in d.items():
    for k, c in ks.items():
        if k.startswith("csr_"): print(tag, k, {a: b for a, b in c.items()})
PY
