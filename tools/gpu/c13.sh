set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
cat > /tmp/p27.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from spmv_amd import _lib, host
exec_ = host.HipExecutor(0); comm = host.Comm.self_comm(); ctx = exec_.context
for k, v in ((b"poisson_stencil", 27), (b"const_diagonals", 0)):
    _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
for rep in range(2):
    A = host.Matrix.create_poisson3d(comm, exec_, 256, False, host.P2P_BLOCKING)
    print("plan_ms", A.plan_get("plan_us") / 1e3, "half", A.plan_get("wdia_half"))
    A.close()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp27 -o p -- python3 /tmp/p27.py 2>&1 | grep plan_ms
cp $(find /tmp/rp27 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04/plan27_kernel_stats.csv
cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/r04/plan27_kernel_stats.csv | cut -c1-150
