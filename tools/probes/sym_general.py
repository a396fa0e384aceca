"""Probe: symmetric STORAGE of a general (non-lattice) matrix -- what does the
symmetric SpMV (csr_kernels.cpp:26-40) cost on a FEM-like matrix, next to the
same matrix in general storage?  S = L + D + L^T from the strictly lower part of
the seeded FEM-like matrix.  One JSON line per storage."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spmv_amd import _lib, host, poisson  # noqa: E402
from tools.mbench import timed, form_of  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    N = args.rows
    t0 = time.time()
    rp, ci, va = poisson.fem_like_csr(N)
    A = sp.csr_matrix((va, ci, rp), shape=(N, N))
    L = sp.tril(A, -1, format="csr")
    S = (L + L.T + sp.diags(A.diagonal())).tocsr()
    S.sort_indices()
    print("built S: nnz", S.nnz, "in %.1f s" % (time.time() - t0), file=sys.stderr)
    exec_ = host.HipExecutor(0)
    comm = host.Comm.self_comm()
    ctx = exec_.context
    x = np.random.default_rng(1).uniform(-1, 1, N)
    y_ref = None
    for symmetric in (False, True):
        M = host.Matrix.create_matrix(comm, exec_, S.indptr.astype(np.int32),
                                      S.indices.astype(np.int32), S.data, N, N,
                                      np.zeros(0, np.int64), np.zeros(0, np.int64),
                                      symmetric=symmetric)
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        ms = timed(exec_, M, d_x, d_y, args.reps)
        y = exec_.copy_to_host(d_y, N)
        if y_ref is None:
            y_ref = y
        nnz_l = L.nnz
        b_csr = poisson.csr_bytes(N, N, S.nnz)
        b_sym = poisson.sym_csr_bytes(N, nnz_l)
        print(json.dumps(dict(symmetric=symmetric, rows=N, nnz=int(S.nnz), ms=round(ms, 5),
                              frac_csr=round(b_csr / ms / 1e6 / 8000, 4),
                              frac_sym=round(b_sym / ms / 1e6 / 8000, 4),
                              max_rel_diff=float(np.max(np.abs(y - y_ref)) / np.max(np.abs(y_ref))),
                              form=form_of(M), plan_ms=M.plan_get("plan_us") / 1e3)), flush=True)
        M.close()
        exec_.free(d_x), exec_.free(d_y)
    comm.close()
    exec_.close()


if __name__ == "__main__":
    main()
