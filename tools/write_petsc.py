"""Write one of the seeded test matrices as PETSc binary files -- the input
format of the reference's demos (spmv/read_petsc.cpp:40-228 matrices, magic
1211216 at :75; :230-303 vectors, magic 1211214 at :259; everything big-endian)
-- so that `bench.py --petsc-matrix A.dat [--petsc-rhs b.dat]` and
host.read_petsc_binary_matrix drive the backend the way demos/cg.cpp:47-51 and
demos/spmv.cpp:43 drive the reference.

    python tools/write_petsc.py --kind fem --rows 1000000 --out /tmp/A.dat --rhs /tmp/b.dat
    python tools/write_petsc.py --kind poisson --grid 128 --out /tmp/P.dat

Kinds: fem / fem_tail / fem81 (spmv_amd.poisson.fem_like_csr), unstructured,
poisson (7-point, --grid).  The right-hand side is b = A * 1 (all-ones
solution) unless --rhs-gaussian.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmv_amd import poisson  # noqa: E402

MAT_ID, VEC_ID = 1211216, 1211214  # read_petsc.cpp:75, :259


def write_matrix(filename, rowptr, colind, values, ncols=None):
    """int32 {id, nrows, ncols, nnz}, nrows row lengths, nnz column ids (int32),
    nnz values (fp64) -- all big-endian (read_petsc.cpp:60-121)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    nrows = len(rowptr) - 1
    with open(filename, "wb") as f:
        np.array([MAT_ID, nrows, nrows if ncols is None else ncols, rowptr[-1]],
                 ">i4").tofile(f)
        np.diff(rowptr).astype(">i4").tofile(f)
        np.asarray(colind).astype(">i4").tofile(f)
        np.asarray(values).astype(">f8").tofile(f)


def write_vector(filename, x):
    with open(filename, "wb") as f:
        np.array([VEC_ID, len(x)], ">i4").tofile(f)
        np.asarray(x).astype(">f8").tofile(f)


def matrix(kind, rows, grid):
    if kind == "poisson":
        rp, ci, va = poisson.poisson3d_csr(grid)
        return rp, ci.astype(np.int32), va
    if kind == "unstructured":
        return poisson.unstructured_csr(rows)
    kw = {"fem": dict(), "fem_tail": dict(tail_permille=10),
          "fem81": dict(min_len=81, max_len=81)}[kind]
    return poisson.fem_like_csr(rows, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="fem",
                    choices=["fem", "fem_tail", "fem81", "unstructured", "poisson"])
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--out", required=True)
    ap.add_argument("--rhs", default=None)
    ap.add_argument("--rhs-gaussian", action="store_true")
    args = ap.parse_args()
    rp, ci, va = matrix(args.kind, args.rows, args.grid)
    write_matrix(args.out, rp, ci, va)
    n = len(rp) - 1
    if args.rhs:
        if args.rhs_gaussian:
            i = np.arange(n, dtype=np.float64)
            b = np.exp(-10 * (5 * (i / n - 0.5)) ** 2)
        else:  # A * 1: row sums, accumulated in the row's order
            b = np.add.reduceat(va, rp[:-1].astype(np.int64))
            b[np.diff(rp) == 0] = 0.0
        write_vector(args.rhs, b)
    print(f"{args.out}: {n} rows, {len(va)} entries"
          + (f"; {args.rhs}" if args.rhs else ""))


if __name__ == "__main__":
    main()
