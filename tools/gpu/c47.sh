set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "long_rows_table or sliced_jagged or mixed_precision_sliced" > gpurun_out/r04/t47.log 2>&1 || { tail -40 gpurun_out/r04/t47.log; exit 1; }
tail -2 gpurun_out/r04/t47.log
rm -f gpurun_out/r04/c47.log
for v in C0 C1 C0 C1; do
  echo "variant $v" >> gpurun_out/r04/c47.log
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 200 python tools/mbench.py --kind fem_tail --variants auto sj_phases=1 >> gpurun_out/r04/c47.log 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/c47.log"):
    if l.startswith("variant"): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("  ", d["variant"], d.get("ms"), d.get("frac_csr"), d.get("bit_equal_scalar"), d.get("plan_ms"))
PY
