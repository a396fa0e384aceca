"""Worker for tests/test_host_logic.py::test_l2g_plan_gloo_*: one process per
rank, gloo on CPU.  Builds the C++ L2GMap (HostExecutor: plan only) over a
CallbackComm whose allgather runs on torch.distributed, and compares every
plan array with the oracle's restatement of spmv/L2GMap.cpp:346-479.  Also
checks Matrix<double>::split_rows against oracle.create_matrix for this rank.
Exit code 0 = all ranks agree.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from spmv_amd import gloo_transport as dist_util  # noqa: E402
import oracle  # noqa: E402
from spmv_amd import host, poisson  # noqa: E402
from util import assembled_inputs, box_partition, permute_csr  # noqa: E402


def main():
    rank, world = dist_util.init_gloo()
    comm = host.Comm.callback(rank, world, dist_util.make_allgather(world))
    # one PROCESS per rank: no two ranks share one (per-process token = pid +
    # nonce, exchanged over the communicator; comm.cpp)
    assert comm.ranks_share_a_process() is False
    cases = []
    # KAT (tests/test_spmv.cpp:56-64) and small Poisson grids
    kat = (np.array([0, 3, 6, 9, 13, 15], np.int32),
           np.array([0, 1, 3, 0, 1, 3, 2, 3, 4, 0, 1, 2, 3, 2, 4], np.int64),
           np.array([1, -2, -3, -2, 5, 4, 6, 4, -4, -3, 4, 4, 8, -4, 8], float))
    cases.append(kat)
    for n in (3, 4, 6):
        cases.append(poisson.poisson3d_csr(n))
    # an unstructured matrix: every rank talks to every other rank
    rng = np.random.default_rng(42)
    N = 37
    dense = (rng.random((N, N)) < 0.2) | np.eye(N, dtype=bool)
    rp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    ci = np.nonzero(dense)[1].astype(np.int64)
    cases.append((rp, ci, rng.uniform(-1, 1, len(ci))))

    for rp, ci, va in cases:
        N = len(rp) - 1
        if N < world:
            continue
        ranges = oracle.owner_ranges(world, N)
        locs = [oracle.localise_rows(rp, ci, va, int(ranges[r]), int(ranges[r + 1]))
                for r in range(world)]
        plans = oracle.l2g_plans(np.diff(ranges), [l[3] for l in locs])
        lrp, lci, lva, ghosts = locs[rank]
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING,
                   host.COLLECTIVE_BLOCKING, host.COLLECTIVE_NONBLOCKING):
            m = host.L2GMap(comm, int(ranges[rank + 1] - ranges[rank]), ghosts,
                            None, cm)
            got, exp = m.plan(), plans[rank]
            assert np.array_equal(got.neighbours, exp["neighbours"]), (got.neighbours, exp["neighbours"])
            nn = len(exp["neighbours"])
            assert np.array_equal(got.send_count, exp["send_count"][:nn])
            assert np.array_equal(got.recv_count, exp["recv_count"][:nn])
            assert np.array_equal(got.send_offset, exp["send_offset"][:nn + 1])
            assert np.array_equal(got.recv_offset, exp["recv_offset"][:nn + 1])
            assert np.array_equal(got.indexbuf, exp["indexbuf"])
            m.close()
            for sym in (False, True):
                A = oracle.create_matrix(rank, ranges, ranges, lrp, lci, lva,
                                         ghosts, sym, cm)
                s = host.split_rows(lrp, lci, lva, ranges[rank + 1] - ranges[rank],
                                    ranges[rank + 1] - ranges[rank],
                                    ranges[rank], ranges[rank], ghosts, sym, cm)
                assert s["nnz"] == A["nnz"]
                assert np.array_equal(s["ghosts"], A["ghosts"])
                for name in ("local", "remote"):
                    if A[name] is None:
                        assert len(s[name][2]) == 0
                        continue
                    for a, b in zip(s[name], A[name]):
                        assert np.array_equal(a, b), name
                if sym:
                    assert np.array_equal(s["diagonal"], A["diagonal"])
    # 3-D block partition: this rank's generated box rows give
    # the plan and the split the oracle derives from the permuted matrix
    n, parts = 6, {2: (2, 1, 1), 3: (1, 3, 1)}.get(world, (1, 1, world))
    perm, ranges = box_partition(n, parts)
    brp, bci, bva = permute_csr(*poisson.poisson3d_csr(n), perm)
    locs = [oracle.localise_rows(brp, bci, bva, int(ranges[r]), int(ranges[r + 1]))
            for r in range(world)]
    plans = oracle.l2g_plans(np.diff(ranges), [l[3] for l in locs])
    lrp, lci, lva, ghosts, off, _ = host.poisson3d_box_rows(n, parts, rank)
    nloc = int(ranges[rank + 1] - ranges[rank])
    assert off == ranges[rank] and np.array_equal(ghosts, locs[rank][3])
    for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
        m = host.L2GMap(comm, nloc, ghosts, None, cm)
        got, exp = m.plan(), plans[rank]
        nn = len(exp["neighbours"])
        assert np.array_equal(got.neighbours, exp["neighbours"])
        assert np.array_equal(got.send_count, exp["send_count"][:nn])
        assert np.array_equal(got.recv_count, exp["recv_count"][:nn])
        assert np.array_equal(got.indexbuf, exp["indexbuf"])
        m.close()
        for sym in (False, True):
            A = oracle.create_matrix(rank, ranges, ranges, *locs[rank], sym, cm)
            s = host.split_rows_distributed(comm, lrp, lci, lva, nloc, nloc, [],
                                            ghosts, sym, cm)
            assert s["nnz"] == A["nnz"]
            for name in ("local", "remote"):
                if A[name] is None:
                    assert len(s[name][2]) == 0
                    continue
                for a, b in zip(s[name], A[name]):
                    assert np.array_equal(a, b), (name, sym, cm)
    # ghost-row elimination (Matrix.cpp:188-292): every rank holds pieces of
    # rows it does not own; the C++ exchange + split must equal the oracle's
    for seed, sym in ((1, False), (2, True)):
        rng = np.random.default_rng(seed)  # same stream on every rank
        A, ranges, inputs = assembled_inputs(rng, world, 29, symmetric=sym)
        rp, ci, va, rg, cg = inputs[rank]
        nloc = int(ranges[rank + 1] - ranges[rank])
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            exp = oracle.create_matrices_with_row_ghosts(ranges, inputs, sym,
                                                         cm)[rank]
            got = host.split_rows_distributed(comm, rp, ci, va, nloc, nloc, rg,
                                              cg, sym, cm)
            assert np.array_equal(got["ghosts"], exp["ghosts"])
            assert got["nnz"] == exp["nnz"]
            for name in ("local", "remote"):
                if exp[name] is None:
                    assert len(got[name][2]) == 0
                    continue
                for a, b in zip(got[name], exp[name]):
                    assert np.array_equal(a, b), (name, sym, cm)
            if sym:
                assert np.array_equal(got["diagonal"], exp["diagonal"])
    comm.close()
    print(f"rank {rank}/{world}: plan + split OK", flush=True)


if __name__ == "__main__":
    main()
