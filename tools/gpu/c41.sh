cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c41.log
for v in I0 I1 I0 I1; do
  echo "variant $v" >> gpurun_out/r04/c41.log
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 200 python tools/mbench.py --kind fem_tail --variants auto sj_phases=1 >> gpurun_out/r04/c41.log 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/c41.log"):
    if l.startswith("variant"): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("  ", d["variant"], d.get("ms"), d.get("frac_csr"), d.get("bit_equal_scalar"))
PY
