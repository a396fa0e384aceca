"""Host-side 3-D 7-point Poisson generator (numpy).

Not in the reference (demos/CreateA.cpp is a 1-D
tridiagonal generator); this is the build's own synthetic input, used for the
sizes a host CSR is wanted (tests, CPU baseline).  Grid n^3, natural ordering
i = x + n (y + n z), diagonal 6, off-diagonal -1, neighbours outside the grid
dropped, columns ascending within a row.  The device generator
(spmv_hip_poisson3d_*) writes the same matrix without a host copy.
"""
import numpy as np


def poisson3d_csr(n, row_begin=0, row_end=None, dtype=np.float64):
    """Rows [row_begin,row_end) of the global matrix with GLOBAL column
    indices.  Returns (rowptr int32, colind int64, values dtype)."""
    N = n ** 3
    row_end = N if row_end is None else row_end
    i = np.arange(row_begin, row_end, dtype=np.int64)
    x, y, z = i % n, (i // n) % n, i // (n * n)
    offs = np.array([-n * n, -n, -1, 0, 1, n, n * n], dtype=np.int64)
    valid = np.stack([z > 0, y > 0, x > 0, np.ones_like(x, bool), x < n - 1,
                      y < n - 1, z < n - 1], axis=1)
    cols = (i[:, None] + offs[None, :])[valid]
    vals = np.broadcast_to(np.where(offs == 0, 6.0, -1.0).astype(dtype),
                           valid.shape)[valid]
    rowptr = np.zeros(len(i) + 1, np.int64)
    np.cumsum(valid.sum(axis=1), out=rowptr[1:])
    if rowptr[-1] > np.iinfo(np.int32).max:
        raise OverflowError("nnz exceeds the int32 row pointer of the format")
    return rowptr.astype(np.int32), cols, np.ascontiguousarray(vals)


def poisson3d_nnz(n):
    return 7 * n ** 3 - 6 * n ** 2


def owner_ranges(size, N):
    """Contiguous row ranges, first N % size ranks get one extra row
    (same rule as the reference's readers, spmv/read_petsc.cpp:20-37)."""
    q, r = divmod(int(N), int(size))
    return np.array([k * (q + 1) if k < r else k * q + r
                     for k in range(size + 1)], dtype=np.int64)


def csr_bytes(nrows, ncols, nnz, value_bytes=8):
    """Algorithmic (compulsory) bytes of one general CSR SpMV with beta = 0:
    every array touched once."""
    return nnz * (value_bytes + 4) + (nrows + 1) * 4 + ncols * value_bytes \
        + nrows * value_bytes


def sym_csr_bytes(nrows, nnz_lower, value_bytes=8):
    """Symmetric SpMV: lower entries + rowptr + diag + x + y (section 8d)."""
    return nnz_lower * (value_bytes + 4) + (nrows + 1) * 4 \
        + 3 * nrows * value_bytes
