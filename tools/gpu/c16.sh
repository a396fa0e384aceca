set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -q -x -k "box27_half_marched or wide_diagonal_half" > gpurun_out/r04/c16_test.log 2>&1 || { tail -40 gpurun_out/r04/c16_test.log; exit 1; }
tail -3 gpurun_out/r04/c16_test.log
timeout -k 10 300 python tools/mbench.py --kind poisson256 --set poisson_stencil=27 const_diagonals=0 --variants auto wdia_hbox=0 wdia_hbox=1 wdia_hbox_segs=2 wdia_hbox_segs=8 wdia_hbox_segs=4 auto > gpurun_out/r04/c16_mbench.log 2>&1 || { tail -20 gpurun_out/r04/c16_mbench.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/c16_mbench.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["variant"], d.get("ms"), d.get("bit_equal_scalar"), d.get("plan_ms"), {k:d["form"].get(k) for k in ("wdia","wdia_half","wdia_hbox")} if "form" in d else d.get("error"))
PY
