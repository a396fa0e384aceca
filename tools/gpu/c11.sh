set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/mbench11.jsonl
for v in A B C D A B C D; do
  echo "{\"variant_build\": \"$v\"}" >> gpurun_out/r04/mbench11.jsonl
  SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v timeout -k 10 600 python tools/mbench.py --kind fem_tail --no-check --variants auto sj_phases=1 >> gpurun_out/r04/mbench11.jsonl 2>> gpurun_out/r04/mbench11.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench11.jsonl"):
    d=json.loads(l)
    if "variant_build" in d: print("build", d["variant_build"]); continue
    print(" ", d['variant'], d.get('ms'), d.get('frac_csr'))
PY
