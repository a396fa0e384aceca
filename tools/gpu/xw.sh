#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "xw or lx" 2>&1 | tail -15 || exit 1
OFF=4611686018427387904
for k in poisson512 poisson216; do
timeout -k 10 400 python tools/mbench.py --kind $k --reps 10 \
  --set lat_min_nnz=$OFF lx_min_nnz=$OFF sj_min_nnz=$OFF \
  --variants auto xw=0 zwalk=0 nontemporal=1 nontemporal=0 lxw_blocks_per_cu=1 zwalk_segments=1 zwalk_segments=4 \
  2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['kind'], d['variant'], d.get('ms'), d.get('frac_csr'), d.get('bit_equal_scalar'), d.get('error',''), d.get('plan_ms'), d.get('plan_kib'))
    else: print(ln.rstrip()[:200])
" | tee -a gpurun_out/r05/xw.log
done
