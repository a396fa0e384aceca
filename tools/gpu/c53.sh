set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py -q -x -m gpu -k "symmetric or kat or petsc_file or values_changed" > gpurun_out/r04/t53.log 2>&1 || { tail -50 gpurun_out/r04/t53.log; exit 1; }
tail -2 gpurun_out/r04/t53.log
timeout -k 10 300 python tools/probes/sym_general.py --rows 4000000 > gpurun_out/r04/sym_general.log 2>&1 || { tail -20 gpurun_out/r04/sym_general.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/sym_general.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["symmetric"], d["ms"], d["frac_csr"], d["frac_sym"], d["max_rel_diff"])
PY
