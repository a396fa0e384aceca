set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "sliced_jagged or values_changed" > gpurun_out/r04/t5.log 2>&1 || { tail -40 gpurun_out/r04/t5.log; exit 1; }
tail -3 gpurun_out/r04/t5.log
rm -f gpurun_out/r04/mbench5.jsonl
timeout -k 10 900 python tools/mbench.py --kind fem_tail --variants auto sj_long_panels=0 sj_phases=1 sj_phases=2 > gpurun_out/r04/mbench5.jsonl 2>> gpurun_out/r04/mbench5.err || { tail -20 gpurun_out/r04/mbench5.err; exit 1; }
timeout -k 10 900 python tools/mbench.py --kind fem_tail --variants auto sj_phases=1 --set sj_wpb=8 >> gpurun_out/r04/mbench5.jsonl 2>> gpurun_out/r04/mbench5.err || { tail -20 gpurun_out/r04/mbench5.err; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench5.jsonl"):
    d=json.loads(l); f=d.get('form',{})
    print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'),d.get('plan_ms'),'wpb',f.get('sj_wpb'),'E',f.get('sj_unit'),f.get('sj_max_chunks'),f.get('sj_long_rows'), d.get('error',''))
PY
