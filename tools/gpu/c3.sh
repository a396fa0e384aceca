set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python tools/mbench.py --kind fem_tail fem --variants auto sj_phases=1 sj_phases=2 sj_blocks_per_cu=1 sj_blocks_per_cu=2 sj_blocks_per_cu=3 sj_blocks_per_cu=4 sj_xcd_group=0 sj_xcd_group=2 sj_xcd_group=32 --set sj_wpb=8 > gpurun_out/r04/mbench3.jsonl 2> gpurun_out/r04/mbench3.err || { tail -20 gpurun_out/r04/mbench3.err; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r04/mbench3.jsonl"):
    d=json.loads(l); f=d.get('form',{})
    print(d['kind'],d['variant'],d.get('ms'),d.get('frac_csr'),d.get('bit_equal_scalar'),f.get('sj_wpb'),f.get('sj_long_rows'), d.get('error',''))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fem -o p -- python3 $GRAFT_REPO_ROOT/tools/mbench.py --kind fem --variants auto --no-check --set sj_wpb=8 > /dev/null 2>&1
cp /tmp/prof_fem/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r04/fem_kernel_stats.csv 2>/dev/null || cp $(find /tmp/prof_fem -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04/fem_kernel_stats.csv
head -5 $GRAFT_REPO_ROOT/gpurun_out/r04/fem_kernel_stats.csv | cut -c1-300
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  n=$(echo $g | cut -d' ' -f1)
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_$n -o p -- python3 $GRAFT_REPO_ROOT/tools/mbench.py --kind fem --variants auto --no-check --reps 2 --set sj_wpb=8 > /tmp/pmc_$n.log 2>&1 || { tail -5 /tmp/pmc_$n.log; continue; }
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r04/fem_pmc_$n.csv
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/r04/
