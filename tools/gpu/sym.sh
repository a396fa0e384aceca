#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "symmetric_storage_sliced or too_long_for_the_sigma or sliced_jagged_form or long_rows_table" 2>&1 | tail -15 || exit 1
timeout -k 10 600 python tools/mbench.py --kind fem_sym fem_tail_sym fem_tail --reps 20 --no-check \
  --variants auto sjds=0 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['kind'], d['variant'], d.get('ms'), d.get('frac_csr'), d.get('error',''), d.get('plan_ms'), d.get('plan_kib'), {k:d['form'].get(k) for k in ('sym_sj','sj_long_rows','sj_long_table','sj_sigma','sj_wpb')} if 'form' in d else '')
    else: print(ln.rstrip()[:200])
" | tee -a gpurun_out/r05/sym.log
