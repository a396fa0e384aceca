set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "long_rows_table or sliced_jagged or values_changed" > gpurun_out/r04/t23.log 2>&1 || { tail -40 gpurun_out/r04/t23.log; exit 1; }
tail -2 gpurun_out/r04/t23.log
timeout -k 10 600 python tools/mbench.py --kind fem_tail --variants auto sj_long_table=0 auto sj_long_table=0 sj_phases=1 sj_phases=1,sj_long_table=0 sj_phases=3 > gpurun_out/r04/mbench23.jsonl 2> gpurun_out/r04/mbench23.err || { tail -20 gpurun_out/r04/mbench23.err; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench23.jsonl"):
    d=json.loads(l); print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'), d.get('plan_ms'), d.get('error',''))
PY
