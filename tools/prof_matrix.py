"""A fixed number of SpMV launches on one of the non-Poisson test matrices, for
rocprofv3 passes (tools/pmc_passes.sh takes the script through PMC_SCRIPT).

    PMC_SCRIPT=tools/prof_matrix.py tools/pmc_passes.sh OUT TAG --kind unstructured
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import _lib, host  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="unstructured",
                    choices=["unstructured", "stencil27", "fem", "fem_tail",
                             "fem81", "fem_sym", "fem_tail_sym"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--set", nargs="*", default=[], help="ctx option=value ...")
    args = ap.parse_args()
    exec_ = host.HipExecutor(0)
    comm = host.Comm.self_comm()
    ctx = exec_.context
    for kv in args.set:
        k, v = kv.split("=")
        _lib.call("spmv_hip_ctx_set_option", ctx, k.encode(), int(v))
    if args.kind == "unstructured":
        A = host.Matrix.create_unstructured(comm, exec_, args.rows)
        N = args.rows
    elif args.kind.startswith("fem"):
        kw = {"fem": dict(), "fem_tail": dict(tail_permille=10),
              "fem81": dict(min_len=81, max_len=81),
              "fem_sym": dict(symmetric=True),
              "fem_tail_sym": dict(symmetric=True, tail_permille=10)}[args.kind]
        A = host.Matrix.create_fem_like(comm, exec_, args.rows, **kw)
        N = args.rows
    else:
        _lib.call("spmv_hip_ctx_set_option", ctx, b"poisson_stencil", 27)
        A = host.Matrix.create_poisson3d(comm, exec_, args.n, False,
                                         host.P2P_BLOCKING)
        N = args.n ** 3
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_x, None)
    for _ in range(args.reps):
        A.mult(d_x, d_y)
    exec_.synchronize()
    print("done", A.blocks())
    A.close()


if __name__ == "__main__":
    main()
