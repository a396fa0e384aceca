set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/mbench4.jsonl
# Poisson without the lattice analysis: LX (DMA kernel), then SJDS, then the plain row-block kernel
timeout -k 10 900 python tools/mbench.py --kind poisson216 poisson512 --variants auto --reps 10 --set lat_min_nnz=4611686018427387904 >> gpurun_out/r04/mbench4.jsonl 2>> gpurun_out/r04/mbench4.err
for u in 1 2; do
timeout -k 10 900 python tools/mbench.py --kind poisson216 poisson512 --variants auto sjds=0 --reps 10 --set lat_min_nnz=4611686018427387904 lx_min_nnz=4611686018427387904 sj_unit=$u >> gpurun_out/r04/mbench4.jsonl 2>> gpurun_out/r04/mbench4.err
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench4.jsonl"):
    d=json.loads(l); f=d.get('form',{})
    print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'),d.get('plan_ms'),d.get('plan_kib'),'lx',f.get('lx'),f.get('lxw'),'sj',f.get('sjds'),'wpb',f.get('sj_wpb'),'E',f.get('sj_unit'),f.get('sj_max_chunks'), d.get('error',''))
PY
timeout -k 10 1100 python -m pytest tests/test_gpu_matrix.py -q -x -m gpu -k "fem_like or unstructured_matrix" > gpurun_out/r04/t4.log 2>&1 || { tail -40 gpurun_out/r04/t4.log; exit 1; }
tail -3 gpurun_out/r04/t4.log
