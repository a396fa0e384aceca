set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "sliced_jagged" > gpurun_out/r04/t2.log 2>&1 || { tail -40 gpurun_out/r04/t2.log; exit 1; }
tail -3 gpurun_out/r04/t2.log
rm -f gpurun_out/r04/mbench2.jsonl
for u in 1 2 4; do for w in 8 16; do
timeout -k 10 900 python tools/mbench.py --kind fem fem_tail fem81 unstructured --variants auto sj_phases=2 --set sj_wpb=$w sj_unit=$u >> gpurun_out/r04/mbench2.jsonl 2>> gpurun_out/r04/mbench2.err || { tail -20 gpurun_out/r04/mbench2.err; exit 1; }
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench2.jsonl"):
    d=json.loads(l); f=d.get('form',{})
    print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'),d.get('plan_ms'),'wpb',f.get('sj_wpb'),'E',f.get('sj_unit'),f.get('sj_max_chunks'),f.get('sj_far_permille'),f.get('sj_long_rows'), d.get('error',''))
PY
