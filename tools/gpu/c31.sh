cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for i in 1 2; do
SPMV_PLAN_TRACE=1 timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04/bench31_$i.log 2> gpurun_out/r04/bench31_$i.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04/bench31_$i.log") if l.startswith("{")][-1])
print($i, d["value"], {k: d[k]["plan_ms"] for k in d if isinstance(d[k], dict) and "plan_ms" in d[k]})
PY
grep -B3 -A6 "sdia_bake fill  *[0-9]\{3,\}\.\|plan_create .* [0-9]\{3,\}\." gpurun_out/r04/bench31_$i.err | head -40
done
