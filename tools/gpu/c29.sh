set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "symmetric or sliced_jagged or values_changed or long_rows" > gpurun_out/r04/t29.log 2>&1 || { tail -60 gpurun_out/r04/t29.log; exit 1; }
tail -2 gpurun_out/r04/t29.log
timeout -k 10 600 python tools/probes/sym_general.py --rows 4000000 > gpurun_out/r04/sym_general.log 2>&1 || { tail -20 gpurun_out/r04/sym_general.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/sym_general.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["symmetric"], d["ms"], d["frac_csr"], d["frac_sym"], d["max_rel_diff"], d["plan_ms"], {k:v for k,v in d["form"].items() if k in ("sjds","sj_wpb","sj_unit")})
PY
