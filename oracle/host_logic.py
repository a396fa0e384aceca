"""numpy restatement of the reference's HOST logic on the hot path
(TEST INFRASTRUCTURE ONLY): row partition, ghost-column localisation, the
L2GMap halo plan and exchange, Matrix::create_matrix's block split, the four
Matrix::mult variants and a P-rank in-process simulation of cg().

MPI collectives are simulated in-process: every "rank" is an entry of a
Python list.  Citations are relative to /root/reference.
"""
import math

import numpy as np

from . import _c

# CommunicationModel, spmv/mpi_utils.h:43-52 (same order => same ints)
P2P_BLOCKING = 0
P2P_NONBLOCKING = 1
COLLECTIVE_BLOCKING = 2
COLLECTIVE_NONBLOCKING = 3


def overlapping(cm):
    """L2GMap::overlapping, spmv/L2GMap.cpp:975-981."""
    return cm in (P2P_NONBLOCKING, COLLECTIVE_NONBLOCKING)


def owner_ranges(size, N):
    """spmv/read_petsc.cpp:20-37 (same rule as tests/test_spmv.cpp:25-41)."""
    n, r = divmod(int(N), int(size))
    return np.array([rank * (n + 1) if rank < r else rank * n + r
                     for rank in range(size + 1)], dtype=np.int64)


def gaussian_x(N, lo=0, hi=None):
    """x_i = exp(-10 (5 (i/N - 1/2))^2), tests/test_spmv.cpp:66-70,
    demos/spmv.cpp:63-67.  libm exp/pow evaluated per element like the
    reference (math.exp / math.pow call the same libm)."""
    hi = N if hi is None else hi
    out = np.empty(hi - lo)
    for k, i in enumerate(range(lo, hi)):
        z = float(i) / float(N)
        out[k] = math.exp(-10 * math.pow(5 * (z - 0.5), 2.0))
    return out


def gaussian_x_fast(N, lo=0, hi=None):
    """Vectorised variant for large N (numpy exp may differ from libm in the
    last ulp; used only where both sides consume the same array)."""
    hi = N if hi is None else hi
    z = np.arange(lo, hi, dtype=np.float64) / float(N)
    return np.exp(-10 * (5 * (z - 0.5)) ** 2)


def poisson3d_csr(n):
    """Independent 3-D 7-point Poisson generator (scipy kron), natural
    ordering i = x + n (y + n z), diag 6, off-diag -1, Dirichlet truncation
    (SURVEY section 8, row a13).  Used to cross-check spmv_amd.poisson."""
    import scipy.sparse as sp
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n), format="csr")
    I = sp.identity(n, format="csr")
    A = (sp.kron(sp.kron(I, I), T, format="csr")
         + sp.kron(sp.kron(I, T), I, format="csr")
         + sp.kron(sp.kron(T, I), I, format="csr")).tocsr()
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    return (A.indptr.astype(np.int32), A.indices.astype(np.int32),
            A.data.astype(np.float64))


def tridiag_csr(N, gamma=0.1):
    """Global CSR of the 1-D operator of demos/CreateA.cpp:31-64: rows 0 and
    N-1 hold (1-gamma, gamma) / (gamma, 1-gamma), interior rows
    (gamma, 1-2 gamma, gamma).  Rank slices come from localise_rows()."""
    N = int(N)
    cnt = np.full(N, 3, np.int64)
    cnt[0] = cnt[-1] = 2
    rowptr = np.zeros(N + 1, np.int64)
    np.cumsum(cnt, out=rowptr[1:])
    colind = np.empty(int(rowptr[-1]), np.int32)
    values = np.empty(int(rowptr[-1]), np.float64)
    i = np.arange(1, N - 1, dtype=np.int64)
    base = rowptr[1:N - 1]
    for k, v in ((0, gamma), (1, 1.0 - 2.0 * gamma), (2, gamma)):
        colind[base + k] = i - 1 + k
        values[base + k] = v
    colind[0:2] = (0, 1)
    values[0:2] = (1.0 - gamma, gamma)
    colind[-2:] = (N - 2, N - 1)
    values[-2:] = (gamma, 1.0 - gamma)
    return rowptr.astype(np.int32), colind, values


def splitmix64_unit(count, seed=0x5EED0001):
    """`count` doubles in [-1, 1) from the splitmix64 stream of `seed`
    (SURVEY section 8d "robustness" inputs): top 53 bits -> [0,1) -> 2u-1."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed)
             + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, count + 1, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0


def localise_rows(rowptr, colind, values, r0, r1):
    """tests/test_spmv.cpp:83-124: slice rows [r0,r1) of a global CSR, shift
    owned columns by -r0, append ghost columns in ascending global order.
    Returns (rowptr_local, colind_local, values_local, col_ghosts)."""
    rowptr = np.asarray(rowptr)
    a, b = int(rowptr[r0]), int(rowptr[r1])
    gcol = np.asarray(colind[a:b], dtype=np.int64)
    vals = np.asarray(values[a:b]).copy()
    rp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int32)
    ncols_local = r1 - r0
    is_ghost = (gcol < r0) | (gcol >= r1)
    col_ghosts = np.unique(gcol[is_ghost])  # std::set order, :100-108
    lcol = np.where(is_ghost,
                    ncols_local + np.searchsorted(col_ghosts, gcol),  # :117
                    gcol - r0)
    return rp, lcol.astype(np.int32), vals, col_ghosts.astype(np.int64)


# --------------------------------------------------------------------------
# L2GMap plan, spmv/L2GMap.cpp:346-479 (default, non-shmem branch)
# --------------------------------------------------------------------------
def l2g_plans(local_sizes, ghosts):
    """Build the halo plan of every rank.  `ghosts[r]` = sorted global ghost
    indices of rank r.  Returns a list of dicts with the reference's member
    names (without leading underscore)."""
    P = len(local_sizes)
    ranges = np.concatenate([[0], np.cumsum(local_sizes)]).astype(np.int64)
    ghost_count = np.zeros((P, P), np.int32)
    ghost_local = []
    for r in range(P):
        g = np.asarray(ghosts[r], dtype=np.int64)
        if np.any(np.diff(g) < 0):
            raise RuntimeError("Ghosts must be sorted")            # :362-363
        if np.any((g >= ranges[r]) & (g < ranges[r + 1])):
            raise RuntimeError("Ghost index in local range")       # :371-372
        owner = np.searchsorted(ranges, g, side="right") - 1       # :375-377
        for p in owner:
            ghost_count[r, p] += 1
        ghost_local.append((g - ranges[owner]).astype(np.int32))   # :380
    plans = []
    for r in range(P):
        remote_count = ghost_count[:, r]                           # :386-388
        nbrs, send_count, recv_count = [], [], []
        for i in range(P):                                         # :390-412
            c, rc = int(ghost_count[r, i]), int(remote_count[i])
            if c > 0 or rc > 0:
                nbrs.append(i)
                send_count.append(c)
                recv_count.append(rc)
        if not nbrs:                                               # :421-425
            send_count, recv_count = [0], [0]
        send_offset = np.concatenate([[0], np.cumsum(send_count)])
        recv_offset = np.concatenate([[0], np.cumsum(recv_count)])
        plans.append(dict(
            rank=r, ranges=ranges, local_size=int(local_sizes[r]),
            ghosts=np.asarray(ghosts[r], dtype=np.int64),
            neighbours=np.array(nbrs, np.int32),
            send_count=np.array(send_count, np.int32),
            recv_count=np.array(recv_count, np.int32),
            send_offset_raw=send_offset.astype(np.int32),
            send_offset=(send_offset + local_sizes[r]).astype(np.int32),  # :460
            recv_offset=recv_offset.astype(np.int32),
            num_indices=int(recv_offset[-1])))
    # indexbuf, :444-447 (Neighbor_alltoallv of ghost_local)
    for r in range(P):
        pl = plans[r]
        idx = np.zeros(pl["num_indices"], np.int32)
        for i, nb in enumerate(pl["neighbours"]):
            q = plans[nb]
            j = list(q["neighbours"]).index(r)
            seg = ghost_local[nb][q["send_offset_raw"][j]:
                                  q["send_offset_raw"][j] + q["send_count"][j]]
            assert len(seg) == pl["recv_count"][i]
            idx[pl["recv_offset"][i]:pl["recv_offset"][i] + len(seg)] = seg
        pl["indexbuf"] = idx
    return plans


def l2g_update(plans, vecs):
    """Forward halo, spmv/L2GMap.cpp:564-642: after this, for every rank
    vec[local_size + k] = owner's value of ghosts()[k].  In place."""
    send = [_c.gather_ghosts(pl["indexbuf"], v) if pl["num_indices"] else
            np.zeros(0) for pl, v in zip(plans, vecs)]             # :581,618
    for pl, v in zip(plans, vecs):
        for i, nb in enumerate(pl["neighbours"]):
            q = plans[nb]
            j = list(q["neighbours"]).index(pl["rank"])
            src = send[nb][q["recv_offset"][j]:
                           q["recv_offset"][j] + q["recv_count"][j]]
            off = pl["send_offset"][i]                             # :586,625
            v[off:off + pl["send_count"][i]] = src
    return vecs


def l2g_reverse_update(plans, vecs):
    """Reverse halo, spmv/L2GMap.cpp:907-950: every rank sends the slices of
    its ghost tail to their owners (:937-941); the owner receives them in
    index-buffer order and accumulates `vec[indexbuf[i]] += databuf[i]` for
    ascending i (:921-922,:947-948).  Ghost tails are left as they are.
    In place."""
    tails = [np.array(v, copy=True) for v in vecs]  # all sends read pre-update data
    for pl, v in zip(plans, vecs):
        databuf = np.zeros(pl["num_indices"], v.dtype)
        for i, nb in enumerate(pl["neighbours"]):
            q = plans[nb]
            j = list(q["neighbours"]).index(pl["rank"])
            src = tails[nb][q["send_offset"][j]:
                            q["send_offset"][j] + q["send_count"][j]]
            databuf[pl["recv_offset"][i]:pl["recv_offset"][i] + len(src)] = src
        for i in range(pl["num_indices"]):  # sequential: duplicates keep order
            v[pl["indexbuf"][i]] += databuf[i]
    return vecs


# --------------------------------------------------------------------------
# Matrix::create_matrix split, spmv/Matrix.cpp:295-480 (row_ghosts empty)
# --------------------------------------------------------------------------
def _csr_from_triplets(nrows, rows, cols, vals):
    """Eigen setFromTriplets: per-row ascending columns, duplicates summed."""
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    if len(rows):
        new = np.ones(len(rows), bool)
        new[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
        grp = np.cumsum(new) - 1
        v2 = np.zeros(int(grp[-1]) + 1, vals.dtype)
        np.add.at(v2, grp, vals)
        rows, cols, vals = rows[new], cols[new], v2
    rp = np.zeros(nrows + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return (np.cumsum(rp).astype(np.int32), cols.astype(np.int32), vals)


def create_matrix(rank, row_ranges, col_ranges, rowptr, colind, values,
                  col_ghosts, symmetric=False, cm=COLLECTIVE_BLOCKING):
    """Returns dict(local=(rp,ci,va)|None, remote=..., diagonal=..., ghosts,
    ncols_local, nnz, symmetric, overlapping)."""
    nrows = int(row_ranges[rank + 1] - row_ranges[rank])
    ncols_local = int(col_ranges[rank + 1] - col_ranges[rank])
    rowptr = np.asarray(rowptr)
    colind = np.asarray(colind, dtype=np.int64)
    values = np.asarray(values)
    col_ghosts = np.asarray(col_ghosts, dtype=np.int64)
    new_ghosts = np.unique(col_ghosts)                              # :295-318
    col = colind.copy()
    g = col >= ncols_local
    col[g] = ncols_local + np.searchsorted(new_ghosts,
                                           col_ghosts[col[g] - ncols_local])
    row = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(rowptr[:nrows + 1]))
    ncols = ncols_local + len(new_ghosts)
    out = dict(ghosts=new_ghosts, ncols_local=ncols_local, ncols=ncols,
               nrows=nrows, symmetric=bool(symmetric),
               overlapping=overlapping(cm), local=None, remote=None,
               diagonal=None)
    if symmetric:                                                   # :337-349
        in_local = col < ncols_local
        grow = row + row_ranges[rank]
        gcol = col + col_ranges[rank]
        lower = in_local & (grow > gcol)
        diag = in_local & (grow == gcol)
        rem = ~in_local
        out["local"] = _csr_from_triplets(nrows, row[lower], col[lower],
                                          values[lower])
        out["remote"] = _csr_from_triplets(nrows, row[rem], col[rem],
                                           values[rem])
        d = np.zeros(nrows, values.dtype)                           # :429-435
        np.add.at(d, row[diag], values[diag])
        out["diagonal"] = d
        out["nnz"] = (2 * len(out["local"][1]) + len(out["remote"][1])
                      + len(np.unique(row[diag])))                  # :443-444
    elif overlapping(cm):                                           # :350-355
        in_local = col < ncols_local
        out["local"] = _csr_from_triplets(nrows, row[in_local], col[in_local],
                                          values[in_local])
        out["remote"] = _csr_from_triplets(nrows, row[~in_local],
                                           col[~in_local], values[~in_local])
        out["nnz"] = len(out["local"][1]) + len(out["remote"][1])
    else:                                                           # :357
        out["local"] = _csr_from_triplets(nrows, row, col, values)
        out["nnz"] = len(out["local"][1])
    return out


def create_matrices_with_row_ghosts(ranges, inputs, symmetric=False,
                                    cm=COLLECTIVE_BLOCKING):
    """Matrix::create_matrix INCLUDING row-ghost elimination
    (spmv/Matrix.cpp:188-292, 295-318, 363-408) for all ranks at once.

    inputs[r] = (rowptr, colind, values, row_ghosts, col_ghosts): rowptr has
    nrows_local + len(row_ghosts) rows; the extra rows hold contributions to
    the global rows `row_ghosts` (owned elsewhere); colind is local (ghost
    columns >= ncols_local index into col_ghosts).  Returns one block dict per
    rank (as create_matrix)."""
    P = len(inputs)
    sent = [[] for _ in range(P)]  # sent[owner] = list of (src, grow, gcols, vals)
    for r, (rp, ci, va, rg, cg) in enumerate(inputs):
        nloc = int(ranges[r + 1] - ranges[r])
        rp, ci, va = np.asarray(rp), np.asarray(ci, dtype=np.int64), np.asarray(va)
        cg = np.asarray(cg, dtype=np.int64)
        for i, grow in enumerate(rg):                               # :229-251
            owner = int(np.searchsorted(ranges, grow, side="right") - 1)
            assert owner != r
            a, b = int(rp[nloc + i]), int(rp[nloc + i + 1])
            lc = ci[a:b]
            gc = np.where(lc < nloc, lc + ranges[r], cg[np.maximum(lc - nloc, 0)])
            sent[owner].append((r, int(grow), gc, va[a:b]))
    out = []
    for r, (rp, ci, va, rg, cg) in enumerate(inputs):
        nloc = int(ranges[r + 1] - ranges[r])
        rp = np.asarray(rp)
        ci = np.asarray(ci, dtype=np.int64)[:rp[nloc]]
        va = np.asarray(va)[:rp[nloc]]
        cg = np.asarray(cg, dtype=np.int64)
        recv = sorted(sent[r], key=lambda t: t[0])  # by source rank (alltoallv)
        extra_cols = [gc[(gc < ranges[r]) | (gc >= ranges[r + 1])]
                      for _, _, gc, _ in recv]
        new_ghosts = np.unique(np.concatenate([cg] + extra_cols)).astype(np.int64)
        # local entries (global columns), then received ones, in that order
        row = np.repeat(np.arange(nloc, dtype=np.int64), np.diff(rp[:nloc + 1]))
        gcol = np.where(ci < nloc, ci + ranges[r], cg[np.maximum(ci - nloc, 0)])
        rows = [row] + [np.full(len(gc), grow - ranges[r], np.int64)
                        for _, grow, gc, _ in recv]
        gcols = [gcol] + [gc for _, _, gc, _ in recv]
        vals = [va] + [v for _, _, _, v in recv]
        row, gcol, val = (np.concatenate(rows), np.concatenate(gcols),
                          np.concatenate(vals))
        owned = (gcol >= ranges[r]) & (gcol < ranges[r + 1])
        lcol = np.where(owned, gcol - ranges[r],
                        nloc + np.searchsorted(new_ghosts, gcol))
        # hand the merged triplets to the no-row-ghost splitter: one "row
        # block" whose columns are already final
        order = np.argsort(row, kind="stable")
        rp2 = np.zeros(nloc + 1, np.int64)
        np.add.at(rp2, row + 1, 1)
        rp2 = np.cumsum(rp2)
        # ghost list is already sorted/unique, so create_matrix keeps numbering
        A = create_matrix(r, ranges, ranges, rp2, lcol[order], val[order],
                          new_ghosts, symmetric, cm)
        out.append(A)
    return out


def _block_mult(block, alpha, x, beta, y, diagonal=None, symmetric=False):
    """CSRMatrix::mult, spmv/csr_matrix.cpp:81-87 (+ guard :85)."""
    rp, ci, va = block
    if len(va) == 0 and diagonal is None:
        return y
    if symmetric:
        return _c.csr_spmv_sym(rp, ci, va, diagonal, x, alpha, beta, y)
    return _c.csr_spmv(rp, ci, va, x, alpha, beta, y)


def matrix_mult(A, x, y=None):
    """Matrix::mult + the four variants, spmv/Matrix.cpp:131-141,483-552.
    `x` must already hold its ghost tail (update() happened before)."""
    y = np.zeros(A["nrows"], x.dtype) if y is None else y
    if A["symmetric"]:                                # :523-530 / :542-552
        y = _block_mult(A["local"], 1, x, 0, y, A["diagonal"], True)
        return _block_mult(A["remote"], 1, x, 1, y)
    if A["overlapping"]:                              # :498-511
        y = _block_mult(A["local"], 1, x, 0, y)
        beta = 1 if len(A["local"][2]) > 0 else 0
        return _block_mult(A["remote"], 1, x, beta, y)
    return _block_mult(A["local"], 1, x, 0, y)       # :483-486


def partition(P, rowptr, colind, values, symmetric=False,
              cm=COLLECTIVE_BLOCKING, ranges=None):
    """Row-block partition of a global square CSR over P ranks, exactly as
    tests/test_spmv.cpp:83-129 does on each rank.  Returns (ranges, mats,
    plans).  `ranges` replaces the even split by given contiguous row ranges
    (the 3-D block partition's box sizes)."""
    N = len(rowptr) - 1
    ranges = owner_ranges(P, N) if ranges is None else np.asarray(ranges, np.int64)
    assert len(ranges) == P + 1 and ranges[0] == 0 and ranges[-1] == N
    mats, ghosts = [], []
    for r in range(P):
        rp, ci, va, cg = localise_rows(rowptr, colind, values,
                                       int(ranges[r]), int(ranges[r + 1]))
        A = create_matrix(r, ranges, ranges, rp, ci, va, cg, symmetric, cm)
        mats.append(A)
        ghosts.append(A["ghosts"])
    plans = l2g_plans(np.diff(ranges), ghosts)
    return ranges, mats, plans


def dist_spmv(P, rowptr, colind, values, x, symmetric=False,
              cm=COLLECTIVE_BLOCKING, ranges=None):
    """`l2g->update(x); A->mult(x, y)` on P simulated ranks
    (tests/test_spmv.cpp:131-144).  Returns the global y."""
    ranges, mats, plans = partition(P, rowptr, colind, values, symmetric, cm,
                                    ranges)
    xs = []
    for r in range(P):
        v = np.zeros(mats[r]["ncols"], np.asarray(x).dtype)
        v[:mats[r]["ncols_local"]] = x[ranges[r]:ranges[r + 1]]
        xs.append(v)
    l2g_update(plans, xs)
    return np.concatenate([matrix_mult(A, v) for A, v in zip(mats, xs)])


def dist_cg(P, rowptr, colind, values, b, kmax, rtol, symmetric=False,
            cm=COLLECTIVE_BLOCKING, ranges=None):
    """spmv/cg.cpp:21-98 on P simulated ranks.  ddot = left-to-right per
    rank, MPI_Allreduce(SUM) = sum over ranks in rank order (both pinned by
    the oracle, unpinned in the reference).  Returns (x, k, rnorm_history)."""
    ranges, mats, plans = partition(P, rowptr, colind, values, symmetric, cm,
                                    ranges)
    M = [int(ranges[r + 1] - ranges[r]) for r in range(P)]

    def allreduce(parts):
        s = 0.0
        for v in parts:
            s += v
        return s

    r_ = [np.array(b[ranges[i]:ranges[i + 1]], dtype=np.float64)
          for i in range(P)]
    p_ = [np.zeros(mats[i]["ncols"]) for i in range(P)]
    x_ = [np.zeros(mats[i]["ncols"]) for i in range(P)]
    Ap = [np.zeros(M[i]) for i in range(P)]
    for i in range(P):
        p_[i][:M[i]] = r_[i]
    rnorm0 = math.sqrt(allreduce([_c.ddot(r_[i], r_[i]) for i in range(P)]))
    hist = [rnorm0]
    rnorm_old = rnorm0
    k = 0
    while k < kmax:
        k += 1
        l2g_update(plans, p_)                                       # :59
        for i in range(P):
            Ap[i] = matrix_mult(mats[i], p_[i], Ap[i])              # :60
        pdotAp = allreduce([_c.ddot(p_[i][:M[i]], Ap[i]) for i in range(P)])
        alpha = (rnorm_old * rnorm_old) / pdotAp                    # :66
        for i in range(P):
            x_[i][:M[i]] += alpha * p_[i][:M[i]]                    # :69
            r_[i] += (-alpha) * Ap[i]                               # :70
        rnorm_new = math.sqrt(allreduce([_c.ddot(r_[i], r_[i])
                                         for i in range(P)]))       # :73-76
        beta = (rnorm_new * rnorm_new) / (rnorm_old * rnorm_old)    # :77
        rnorm_old = rnorm_new
        hist.append(rnorm_new)
        if rnorm_new / rnorm0 < rtol:                               # :80-81
            break
        for i in range(P):
            p_[i][:M[i]] *= beta                                    # :84
            p_[i][:M[i]] += r_[i]                                   # :85
    x = np.concatenate([x_[i][:M[i]] for i in range(P)])
    return x, k, np.array(hist)
