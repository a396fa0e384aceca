"""PETSc binary files (TEST INFRASTRUCTURE ONLY): a writer for fixtures and a
numpy restatement of the reference reader, spmv/read_petsc.cpp:40-303.

Format, all big-endian: matrix = int32 {1211216, nrows, ncols, nnz}, nrows
int32 row lengths, nnz int32 column ids, nnz fp64 values; vector = int32
{1211214, n}, n fp64.
"""
import numpy as np

from .host_logic import owner_ranges

MAT_ID, VEC_ID = 1211216, 1211214


def write_matrix(filename, rowptr, colind, values, ncols=None):
    rowptr = np.asarray(rowptr, dtype=np.int64)
    nrows = len(rowptr) - 1
    ncols = nrows if ncols is None else ncols
    with open(filename, "wb") as f:
        np.array([MAT_ID, nrows, ncols, rowptr[-1]], ">i4").tofile(f)
        np.diff(rowptr).astype(">i4").tofile(f)
        np.asarray(colind).astype(">i4").tofile(f)
        np.asarray(values).astype(">f8").tofile(f)


def write_vector(filename, x):
    with open(filename, "wb") as f:
        np.array([VEC_ID, len(x)], ">i4").tofile(f)
        np.asarray(x).astype(">f8").tofile(f)


def read_matrix_rows(filename, rank, size):
    """read_petsc.cpp:56-151: this rank's rows with owned columns shifted to
    [0, ncols_local) and ghost columns numbered after them in ascending
    global order."""
    raw = np.fromfile(filename, dtype=np.uint8)
    head = raw[:16].view(">i4")
    if head[0] != MAT_ID:
        raise RuntimeError("Bad signature in PETSc Matrix file")   # :75-76
    nrows, ncols, nnz = int(head[1]), int(head[2]), int(head[3])
    lens = raw[16:16 + 4 * nrows].view(">i4").astype(np.int64)
    rr, cr = owner_ranges(size, nrows), owner_ranges(size, ncols)
    r0, r1, c0, c1 = int(rr[rank]), int(rr[rank + 1]), int(cr[rank]), int(cr[rank + 1])
    off = int(lens[:r0].sum())
    cnt = int(lens[r0:r1].sum())
    cbase = 16 + 4 * nrows
    vbase = cbase + 4 * nnz
    gcol = raw[cbase + 4 * off:cbase + 4 * (off + cnt)].view(">i4").astype(np.int64)
    vals = raw[vbase + 8 * off:vbase + 8 * (off + cnt)].view(">f8").astype(np.float64)
    ghost = (gcol < c0) | (gcol >= c1)
    ghosts = np.unique(gcol[ghost])                                 # :141-151
    lcol = np.where(ghost, (c1 - c0) + np.searchsorted(ghosts, gcol), gcol - c0)
    rowptr = np.concatenate([[0], np.cumsum(lens[r0:r1])]).astype(np.int32)
    return dict(nrows=nrows, ncols=ncols, nnz=nnz, row_begin=r0, row_end=r1,
                rowptr=rowptr, colind=lcol.astype(np.int32), values=vals,
                col_ghosts=ghosts.astype(np.int64))


def read_vector(filename, rank, size):
    """read_petsc.cpp:230-303"""
    raw = np.fromfile(filename, dtype=np.uint8)
    head = raw[:8].view(">i4")
    if head[0] != VEC_ID:
        raise RuntimeError("Bad signature in PETSc Vector file")    # :259-260
    ranges = owner_ranges(size, int(head[1]))
    r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
    return raw[8 + 8 * r0:8 + 8 * r1].view(">f8").astype(np.float64)
