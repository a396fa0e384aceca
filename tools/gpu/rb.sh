#!/bin/bash
# the plain row-block kernel on the 7-point matrix, plane-walk order on / off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
OFF=4611686018427387904
for k in poisson512 poisson216; do
timeout -k 10 400 python tools/mbench.py --kind $k --reps 10 --no-check \
  --set lat_min_nnz=$OFF lx_min_nnz=$OFF sj_min_nnz=$OFF \
  --variants auto zwalk=0 nontemporal=1 nontemporal=1,zwalk=0 nontemporal=0 zwalk_segments=1 zwalk_segments=4 zwalk_segments=8 \
     chunks=1 chunks=4 blocks_per_cu=4 blocks_per_cu=6 nontemporal=1,chunks=4 nontemporal=1,blocks_per_cu=4 \
  2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['kind'], d['variant'], d.get('ms'), d.get('frac_csr'), d.get('error',''), {k:d['form'].get(k) for k in ('nontemporal','blocks_per_cu')} if 'form' in d else '')
    else: print(ln.rstrip()[:200])
" | tee -a gpurun_out/r05/rb_walk.log
done
