"""Host-side generators of the synthetic inputs (numpy): the 3-D 7-point
Poisson matrix, the 27-point operator, a seeded unstructured matrix.

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


def stencil27_csr(n, dtype=np.float64):
    """The 27-point operator on the same grid (all neighbours with |dx|, |dy|,
    |dz| <= 1; diagonal 26, off-diagonal -1: HPCG's matrix), columns ascending
    within a row -- the host twin of the device generator with the context
    option "poisson_stencil" = 27.  Returns (rowptr int32, colind int64,
    values)."""
    assert n >= 3
    N = n ** 3
    i = np.arange(N, dtype=np.int64)
    x, y, z = i % n, (i // n) % n, i // (n * n)
    offs, valid = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                offs.append(dz * n * n + dy * n + dx)
                valid.append((x + dx >= 0) & (x + dx < n) & (y + dy >= 0)
                             & (y + dy < n) & (z + dz >= 0) & (z + dz < n))
    offs = np.array(offs, dtype=np.int64)
    valid = np.stack(valid, axis=1)
    cols = (i[:, None] + offs[None, :])[valid]
    vals = np.broadcast_to(np.where(offs == 0, 26.0, -1.0).astype(dtype),
                           valid.shape)[valid]
    rowptr = np.zeros(N + 1, np.int64)
    np.cumsum(valid.sum(axis=1), out=rowptr[1:])
    if rowptr[-1] > np.iinfo(np.int32).max:
        raise OverflowError("nnz exceeds the int32 row pointer of the format")
    return rowptr.astype(np.int32), cols, np.ascontiguousarray(vals)


def stencil27_nnz(n):
    return (3 * n - 2) ** 3


def _mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)"""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def unstructured_csr(nrows, per_row=7, band=2048, far_permille=100,
                     seed=0x5EED0003):
    """The seeded unstructured test matrix: numpy twin of the device generator
    spmv_hip_unstructured_fill_f64 (same hash, same columns, same values).
    Square, `per_row` entries per row; far_permille / 1000 of them anywhere, the
    rest within `band` columns of the diagonal; columns ascending within a row
    (repeats are kept: legal CSR, the reference loop just adds them), values
    uniform in [-1, 1).  No row block has constant column offsets and its
    column windows are mostly too wide to stage in LDS -- the shape of a badly
    ordered FEM matrix.  Returns (rowptr int32, colind int32, values)."""
    N = int(nrows)
    if N * per_row > np.iinfo(np.int32).max:
        raise OverflowError("nnz exceeds the int32 row pointer of the format")
    golden = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        e1 = (np.arange(N * per_row, dtype=np.uint64) + np.uint64(1)) * golden
        h = _mix64(e1 + np.uint64(seed))
        far = (h % np.uint64(1000)).astype(np.int64) < far_permille
        r = h >> np.uint64(10)
        i = np.repeat(np.arange(N, dtype=np.int64), per_row)
        near = i - band + (r % np.uint64(2 * band + 1)).astype(np.int64)
        cols = np.where(far, (r % np.uint64(N)).astype(np.int64), near)
        del h, r, near, far, i
        np.clip(cols, 0, N - 1, out=cols)
        cols = cols.astype(np.int32).reshape(N, per_row)
        cols.sort(axis=1)
        h2 = _mix64(e1 + np.uint64(seed + 1))
        vals = (h2 >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
    rowptr = (np.arange(N + 1, dtype=np.int64) * per_row).astype(np.int32)
    return rowptr, cols.reshape(-1), vals


def fem_params(nrows, min_len=5, max_len=40, layer=None, jitter=512,
               tail_permille=0, tail_min=200, tail_max=2000, tail_stride=16,
               seed=0x5EED0004):
    """Parameters of the FEM-like test matrix (spmv_hip_fem_params).  `layer`
    defaults to the level-set width of a 3-D mesh of nrows points in a
    bandwidth-reducing order, nrows^(2/3), at least 2 * jitter."""
    N = int(nrows)
    if layer is None:
        layer = max(2 * jitter, int(round(N ** (2.0 / 3.0))))
    return dict(num_rows=N, min_len=int(min_len), max_len=int(max_len),
                layer=int(layer), jitter=int(jitter),
                tail_permille=int(tail_permille), tail_min=int(tail_min),
                tail_max=int(tail_max), tail_stride=int(tail_stride),
                seed=int(seed))


def fem_like_csr(nrows, **params):
    """The seeded FEM-like test matrix: numpy twin of the device generator
    (spmv_hip_fem_count / spmv_hip_fem_fill_f64, spmv_amd/csrc/hip/poisson.hip;
    the construction is described there and followed here line by line).
    Ragged rows of min_len..max_len entries in three clusters (row - layer, row,
    row + layer, each within +-jitter), tail_permille / 1000 of the rows LONG
    (tail_min..tail_max entries, one per tail_stride columns); columns strictly
    ascending, diagonal always present.  Returns (rowptr int32, colind int32,
    values)."""
    p = fem_params(nrows, **params)
    N = p["num_rows"]
    golden = np.uint64(0x9E3779B97F4A7C15)
    seed = np.uint64(p["seed"])
    J, L = p["jitter"], p["layer"]
    W = 2 * J
    assert p["max_len"] <= W and L >= W and W <= N
    with np.errstate(over="ignore"):
        i = np.arange(N, dtype=np.int64)
        h = _mix64((i.astype(np.uint64) + np.uint64(1)) * golden + seed)
        tail = np.zeros(N, dtype=bool)
        if p["tail_permille"] > 0:
            assert p["tail_max"] * p["tail_stride"] <= N
            tail = (h % np.uint64(1000)).astype(np.int64) < p["tail_permille"]
        u = (h >> np.uint64(16)) & np.uint64(0xFFFF)
        f = (u * u * u + ((u * u) << np.uint64(16))) >> np.uint64(1)
        span = np.uint64(p["max_len"] - p["min_len"] + 1)
        n = p["min_len"] + ((span * f) >> np.uint64(48)).astype(np.int64)
        n0 = (3 * n) // 10
        n2 = (3 * n) // 10
        n1 = n - n0 - n2
        n0[(i - L - J < 0) | (i + J > N)] = 0
        n2[(i + L + J > N) | (i - J < 0)] = 0
        lo1 = np.clip(i - J, 0, N - W)
        lo0 = i - L - J
        lo2 = i + L - J
        s0 = np.where(n0 > 0, W // np.maximum(n0, 1), 1)
        s1 = W // n1
        s2 = np.where(n2 > 0, W // np.maximum(n2, 1), 1)
        if tail.any():
            nt = p["tail_min"] + ((h >> np.uint64(32)) % np.uint64(
                p["tail_max"] - p["tail_min"] + 1)).astype(np.int64)
            Wt = nt * p["tail_stride"]
            lot = np.clip(i - Wt // 2, 0, N - Wt)
            n0[tail] = 0
            n2[tail] = 0
            n1[tail] = nt[tail]
            lo1[tail] = lot[tail]
            s1[tail] = p["tail_stride"]
        kd = np.minimum((i - lo1) // s1, n1 - 1)
        lens = n0 + n1 + n2
        rowptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(lens, out=rowptr[1:])
        nnz = int(rowptr[-1])
        if nnz > np.iinfo(np.int32).max:
            raise OverflowError("nnz exceeds the int32 row pointer of the format")
        e = np.arange(nnz, dtype=np.int64)
        row = np.repeat(i, lens)
        k = e - rowptr[row]
        c0 = k < n0[row]
        c2 = k >= (n0 + n1)[row]
        c1 = ~(c0 | c2)
        kk = np.where(c0, k, np.where(c1, k - n0[row], k - (n0 + n1)[row]))
        lo = np.where(c0, lo0[row], np.where(c1, lo1[row], lo2[row]))
        s = np.where(c0, s0[row], np.where(c1, s1[row], s2[row]))
        e1 = (e.astype(np.uint64) + np.uint64(1)) * golden
        h1 = _mix64(e1 + seed + np.uint64(1))
        h2 = _mix64(e1 + seed + np.uint64(2))
        col = lo + kk * s + (h1 % s.astype(np.uint64)).astype(np.int64)
        val = (h2 >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0
        diag = c1 & (kk == kd[row])
        col[diag] = row[diag]
        val[diag] = lens[row][diag].astype(np.float64) + 1.0
    return rowptr.astype(np.int32), col.astype(np.int32), val


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
