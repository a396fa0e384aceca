set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/mbench7.jsonl
for w in 4 8 16; do
timeout -k 10 900 python tools/mbench.py --kind fem_tail --no-check --variants sj_phases=1 sj_blocks_per_cu=1 sj_blocks_per_cu=2 sj_blocks_per_cu=3 sj_blocks_per_cu=4 sj_long_panels=0 --set sj_wpb=$w >> gpurun_out/r04/mbench7.jsonl 2>> gpurun_out/r04/mbench7.err || { tail -20 gpurun_out/r04/mbench7.err; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench7.jsonl"):
    d=json.loads(l); f=d.get('form',{})
    print(d['kind'],d['variant'],d.get('ms'),'wpb',f.get('sj_wpb'),'K',f.get('sj_max_chunks'),'bpc',f.get('sj_blocks_per_cu'), d.get('error',''))
PY
