"""Shared helpers for the parity tests (seeded inputs, tolerances)."""
import numpy as np

U = 2.0 ** -53  # fp64 unit roundoff


def random_csr(rng, nrows, ncols, avg, empty_frac=0.1, long_rows=0,
               long_len=0, dtype=np.float64):
    """Ragged CSR with empty rows and optional very long rows; columns are
    unsorted and may repeat (the kernels must not assume otherwise)."""
    lens = rng.poisson(avg, nrows).astype(np.int64)
    lens[rng.random(nrows) < empty_frac] = 0
    for r in rng.choice(nrows, size=min(long_rows, nrows), replace=False):
        lens[r] = long_len
    rowptr = np.zeros(nrows + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colind = rng.integers(0, ncols, nnz).astype(np.int32)
    values = rng.uniform(-1, 1, nnz).astype(dtype)
    return rowptr.astype(np.int32), colind, values


def abs_bound(rowptr, colind, values, x, alpha=1.0, beta=0.0, y0=None):
    """(|alpha| |A| |x| + |beta| |y0|)_i : scale of the rounding-error bound."""
    n = len(rowptr) - 1
    prod = np.abs(values.astype(np.float64)) * np.abs(x.astype(np.float64)[colind])
    out = np.zeros(n)
    np.add.at(out, np.repeat(np.arange(n), np.diff(rowptr)), prod)
    out *= abs(alpha)
    if y0 is not None:
        out += abs(beta) * np.abs(y0)
    return out


def lower_split(rowptr, colind, values):
    """Full symmetric CSR -> (strictly-lower CSR, diagonal), the storage of
    spmv/Matrix.cpp:337-349 for one rank."""
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    lo = colind < rows
    dg = colind == rows
    diag = np.zeros(n, values.dtype)
    np.add.at(diag, rows[dg], values[dg])
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows[lo] + 1, 1)
    return (np.cumsum(rp).astype(np.int32), colind[lo].astype(np.int32),
            values[lo].copy(), diag)
