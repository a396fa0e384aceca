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


def assembled_inputs(rng, P, N, density=0.2, symmetric=False):
    """A global matrix A and, per rank, the pieces a finite-element style
    assembly would hold: contributions to its own rows AND to rows owned by
    other ranks ("row ghosts", spmv/Matrix.cpp:188-292), some entries split
    over two ranks.  Values are dyadic so every partial sum is exact.
    Returns (A dense, ranges, inputs) with inputs[r] = (rowptr, colind local,
    values, row_ghosts, col_ghosts)."""
    dense = (rng.random((N, N)) < density) | np.eye(N, dtype=bool)
    vals = np.round(rng.uniform(-4, 4, (N, N)) * 8) / 8
    if symmetric:
        dense = dense | dense.T
        vals = np.round((vals + vals.T) * 4) / 8
    A = np.where(dense, vals, 0.0)
    q, rem = divmod(N, P)
    ranges = np.array([k * (q + 1) if k < rem else k * q + rem
                       for k in range(P + 1)], dtype=np.int64)
    owner = lambda i: int(np.searchsorted(ranges, i, side="right") - 1)  # noqa: E731
    contrib = [[] for _ in range(P)]
    for i, j in zip(*np.nonzero(dense)):
        v, o = A[i, j], owner(i)
        u = rng.random()
        if u < 0.25 and P > 1:      # split over the owner and another rank
            other = int(rng.integers(P))
            contrib[o].append((i, j, v / 2))
            contrib[other].append((i, j, v / 2))
        elif u < 0.5 and P > 1:     # assembled entirely elsewhere
            contrib[int(rng.integers(P))].append((i, j, v))
        else:
            contrib[o].append((i, j, v))
    inputs = []
    for r in range(P):
        r0, r1 = int(ranges[r]), int(ranges[r + 1])
        nloc = r1 - r0
        ent = contrib[r]
        rg = sorted({i for i, _, _ in ent if not r0 <= i < r1})
        cg = sorted({j for _, j, _ in ent if not r0 <= j < r1})
        rowmap = {**{i: i - r0 for i in range(r0, r1)},
                  **{g: nloc + k for k, g in enumerate(rg)}}
        colmap = {**{j: j - r0 for j in range(r0, r1)},
                  **{g: nloc + k for k, g in enumerate(cg)}}
        rows = [[] for _ in range(nloc + len(rg))]
        for i, j, v in ent:
            rows[rowmap[i]].append((colmap[j], v))
        rp, ci, va = [0], [], []
        for rr in rows:
            for c, v in rr:
                ci.append(c)
                va.append(v)
            rp.append(len(ci))
        inputs.append((np.array(rp, np.int32), np.array(ci, np.int64),
                       np.array(va, np.float64), np.array(rg, np.int64),
                       np.array(cg, np.int64)))
    return A, ranges, inputs


def box_partition(n, parts):
    """3-D block partition of the n^3 grid (Matrix::create_poisson3d_boxes):
    parts = (px, py, pz) boxes with the even rule per axis, rank = ix + px (iy
    + py iz), rank-major numbering with x fastest inside a box.  An
    independent numpy statement of it.  Returns (perm, ranges): perm[natural
    id x + n (y + n z)] = box-partition global id, ranges = row range of each
    rank."""
    def cuts(p):
        q, rem = divmod(n, p)
        return np.array([k * (q + 1) if k < rem else k * q + rem
                         for k in range(p + 1)], dtype=np.int64)

    cx, cy, cz = (cuts(p) for p in parts)
    px, py, pz = parts
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ix = np.searchsorted(cx, x, side="right") - 1
    iy = np.searchsorted(cy, y, side="right") - 1
    iz = np.searchsorted(cz, z, side="right") - 1
    rank = ix + px * (iy + py * iz)
    lx, ly, lz = np.diff(cx)[ix], np.diff(cy)[iy], np.diff(cz)[iz]
    sizes = np.zeros(px * py * pz, np.int64)
    np.add.at(sizes, rank.ravel(), 1)
    ranges = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    local = (x - cx[ix]) + lx * ((y - cy[iy]) + ly * (z - cz[iz]))
    del lz
    return (ranges[rank] + local).ravel(), ranges


def permute_csr(rowptr, colind, values, perm):
    """P A P^T for new id = perm[old id]; rows' entries ascending by column."""
    n = len(rowptr) - 1
    rows = perm[np.repeat(np.arange(n), np.diff(rowptr))]
    cols = perm[np.asarray(colind, dtype=np.int64)]
    order = np.lexsort((cols, rows))
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return (np.cumsum(rp).astype(np.int32), cols[order].astype(np.int64),
            np.asarray(values)[order].copy())
