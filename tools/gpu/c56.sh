cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/c56.log
for v in A0 A1 A0 A1; do
  echo "variant $v" >> gpurun_out/r04/c56.log
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 200 python tools/mbench.py --kind fem unstructured --variants auto >> gpurun_out/r04/c56.log 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/c56.log"):
    if l.startswith("variant"): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("  ", d["kind"], d.get("ms"), d.get("frac_csr"), d.get("bit_equal_scalar"))
PY
