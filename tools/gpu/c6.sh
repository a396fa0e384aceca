set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  n=$(echo $g | cut -d' ' -f1)
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pmc_$n -o p -- python3 $GRAFT_REPO_ROOT/tools/mbench.py --kind fem_tail --variants sj_phases=1 --no-check --reps 2 > /tmp/pmc_$n.log 2>&1 || { tail -5 /tmp/pmc_$n.log; continue; }
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r04/tail_pmc_$n.csv
done
cd $GRAFT_REPO_ROOT/gpurun_out/r04 && python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob("tail_pmc_*.csv")):
    acc2=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "csr_sjds" in r["Kernel_Name"]:
            acc2[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
    for k,v in acc2.items():
        vals=list(v.values())
        # first dispatch is the full-phase warm-up? all are phases=1 except none
        print(f[9:-4][:20], k, "per-launch %.4g"%(sorted(vals)[len(vals)//2]), "n",len(vals))
PY
