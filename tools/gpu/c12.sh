set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "x_ring or wide_diagonal" > gpurun_out/r04/t12.log 2>&1 || { tail -30 gpurun_out/r04/t12.log; exit 1; }
tail -2 gpurun_out/r04/t12.log
python - <<'PY'
import ctypes as C, json, sys, os
sys.path.insert(0, os.getcwd())
from spmv_amd import _lib, host
import bench
exec_ = host.HipExecutor(0); comm = host.Comm.self_comm(); ctx = exec_.context
for k, v in ((b"poisson_stencil", 27), (b"const_diagonals", 0)):
    _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
A = host.Matrix.create_poisson3d(comm, exec_, 256, False, host.P2P_BLOCKING)
N = 256 ** 3
print("plan_ms", A.plan_get("plan_us") / 1e3, "half", A.plan_get("wdia_half"), "zwalk", A.plan_get("wdia_zwalk"))
for ring in (1, 0, 1, 0):
    A.plan_set("wdia_ring", ring)
    ms = bench.timed_spmv(exec_, A, N, _lib, 20)
    print("ring", ring, "ms", round(ms, 4))
PY
