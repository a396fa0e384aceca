set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_matrix.py -q -x -k "wide_diagonal or box27 or wdia or stencil27 or values_changed or diagonal" > gpurun_out/r04/c18_test.log 2>&1 || { tail -40 gpurun_out/r04/c18_test.log; exit 1; }
tail -3 gpurun_out/r04/c18_test.log
timeout -k 10 300 python tools/mbench.py --kind poisson256 --set poisson_stencil=27 const_diagonals=0 --variants auto wdia_hbox=0 auto > gpurun_out/r04/c18_mbench.log 2>&1 || { tail -20 gpurun_out/r04/c18_mbench.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/c18_mbench.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["variant"], d.get("ms"), d.get("bit_equal_scalar"), d.get("plan_ms"), {k:d["form"].get(k) for k in ("wdia","wdia_half","wdia_hbox")} if "form" in d else d.get("error"))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp27
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp27 -o s27 -- python3 $GRAFT_REPO_ROOT/tools/prof_matrix.py --kind stencil27 --n 256 --set const_diagonals=0 > $GRAFT_REPO_ROOT/gpurun_out/r04/rp27.log 2>&1
cp $(find /tmp/rp27 -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04/rp27_kernel_stats.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/r04/rp27_kernel_stats.csv | cut -c1-120,200-400
