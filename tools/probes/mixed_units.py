"""Probe: the mixed-precision SpMV (fp32 twin of the jagged copy) against the
fp64 one on the FEM-like matrix, for 1 / 2 / 4 entries per lane and step."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spmv_amd import _lib, host  # noqa: E402
from tools.mbench import timed  # noqa: E402


def main():
    exec_ = host.HipExecutor(0)
    comm = host.Comm.self_comm()
    ctx = exec_.context
    N = 10_000_000
    for kind, kw in (("fem", dict()), ("fem81", dict(min_len=81, max_len=81))):
        for unit in (2, 4):
            _lib.call("spmv_hip_ctx_set_option", ctx, b"sj_unit", unit)
            A = host.Matrix.create_fem_like(comm, exec_, N, **kw)
            d_x, d_y = exec_.alloc(N), exec_.alloc(N)
            _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_x, None)
            ms64 = timed(exec_, A, d_x, d_y, 20)
            A.enable_mixed()
            A.use_mixed(True)
            ms32 = timed(exec_, A, d_x, d_y, 20)
            print(json.dumps(dict(kind=kind, unit=unit, ms_fp64=round(ms64, 4),
                                  ms_mixed=round(ms32, 4),
                                  sj_mixed=A.plan_get("sj_mixed"),
                                  wpb=A.plan_get("sj_wpb"))), flush=True)
            A.close()
            exec_.free(d_x), exec_.free(d_y)
    comm.close()
    exec_.close()


if __name__ == "__main__":
    main()
