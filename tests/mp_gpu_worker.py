"""Worker for tests/test_gpu_matrix.py::test_multirank_on_one_gpu.

Every rank drives the C++ Matrix / L2GMap / cg on GPU 0 exactly as
tests/test_spmv_cuda.cpp does on rank r of an MPI job; the transport is a
CallbackComm over torch.distributed gloo (spmv_amd/gloo_transport.py).  Results are
compared with the oracle's P-rank simulation.
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
from util import U, abs_bound, assembled_inputs  # noqa: E402

EPS = np.finfo(float).eps


def main():
    rank, world = dist_util.init_gloo()
    exec_ = host.HipExecutor(0)
    ex, ar = dist_util.make_device_transport(exec_.context)
    comm = host.Comm.callback(rank, world, dist_util.make_allgather(world), ex, ar)

    kat = (np.array([0, 3, 6, 9, 13, 15], np.int32),
           np.array([0, 1, 3, 0, 1, 3, 2, 3, 4, 0, 1, 2, 3, 2, 4], np.int64),
           np.array([1, -2, -3, -2, 5, 4, 6, 4, -4, -3, 4, 4, 8, -4, 8], float))
    rng = np.random.default_rng(7)
    N = 53
    dense = (rng.random((N, N)) < 0.12) | np.eye(N, dtype=bool)
    dense = dense | dense.T
    rp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    ci = np.nonzero(dense)[1].astype(np.int64)
    vals = rng.uniform(-1, 1, (N, N))
    vals = (vals + vals.T) / 2
    unstructured = (rp, ci, vals[dense])
    trp, tci, tva = oracle.tridiag_csr(1001)  # demos/CreateA.cpp:31-64
    cases = [("kat", kat), ("poisson6", poisson.poisson3d_csr(6)),
             ("unstructured", unstructured),
             ("tridiag", (trp, tci.astype(np.int64), tva))]

    for name, (rp, ci, va) in cases:
        N = len(rp) - 1
        x = oracle.gaussian_x_fast(N) if name != "kat" else oracle.gaussian_x(N)
        y_seq = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
        norm_ref = float(np.sqrt(np.sum(y_seq * y_seq)))
        ranges = oracle.owner_ranges(world, N)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        lrp, lci, lva, ghosts = oracle.localise_rows(rp, ci, va, r0, r1)
        for symmetric in (False, True):
            for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING,
                       host.COLLECTIVE_NONBLOCKING, host.ONESIDED_PUT_ACTIVE,
                       host.SHMEM):
                # --- tests/test_spmv_cuda.cpp:127-160 ---
                A = host.Matrix.create_matrix(comm, exec_, lrp, lci, lva,
                                              r1 - r0, r1 - r0, [], ghosts,
                                              symmetric, cm)
                l2g = A.col_map()
                # the one-sided model moves the halo by peer stores into IPC
                # windows (the ranks are processes sharing GPU 0); every other
                # model through the two-sided exchange
                assert l2g.onesided() == (cm == host.ONESIDED_PUT_ACTIVE
                                          and world > 1 and l2g._nn > 0), (cm, name)
                assert l2g.local_size() == r1 - r0
                assert l2g.num_ghosts() == len(ghosts)
                assert l2g.global_size() == N and l2g.global_offset() == r0
                d_y = exec_.alloc(r1 - r0)
                exec_.memset(d_y, 0, 8 * (r1 - r0))
                d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
                exec_.copy_from_host(d_x, x[r0:r1])
                l2g.update(d_x)
                A.mult(d_x, d_y)
                exec_.synchronize()
                # the halo put the owners' values into the ghost tail
                xs = exec_.copy_to_host(d_x, l2g.local_size() + l2g.num_ghosts())
                assert np.array_equal(xs[r1 - r0:], x[ghosts])
                y = dist_util.gather_concat(exec_.copy_to_host(d_y, r1 - r0))
                y_ref = oracle.dist_spmv(world, rp, ci, va, x, symmetric,
                                         cm if cm < 4 else host.P2P_BLOCKING)
                norm = float(np.sqrt(np.sum(y * y)))
                if symmetric:
                    assert np.all(np.abs(y - y_seq) <= 16 * U * abs_bound(rp, ci, va, x)), name
                    assert abs(norm - norm_ref) <= 1e-14 * norm_ref
                else:  # bit-exact against the oracle's P-rank simulation
                    assert np.array_equal(y, y_ref), (name, cm)
                    assert abs(norm - norm_ref) <= 4 * EPS * norm_ref
                A.close()
                exec_.free(d_x), exec_.free(d_y)

    # device-generated Poisson blocks, slab partition (>= n^2 rows per rank)
    n = 8
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    x = oracle.gaussian_x_fast(N)
    y_seq = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    ranges = oracle.owner_ranges(world, N)
    r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
    for symmetric in (False, True):
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING,
                   host.ONESIDED_PUT_ACTIVE):
            A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric, cm)
            l2g = A.col_map()
            # one-sided: peer stores into the neighbours' IPC windows, 100+
            # exchanges (epochs) in the CG below
            assert l2g.onesided() == (cm == host.ONESIDED_PUT_ACTIVE and world > 1)
            ocm = cm if cm < 4 else host.P2P_BLOCKING  # the oracle's equivalent
            pl = l2g.plan()
            assert not l2g.packs, "slab halo must take the direct-send path"
            assert len(pl.neighbours) == (1 if rank in (0, world - 1) else 2)
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            d_y = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_x, x[r0:r1])
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = dist_util.gather_concat(exec_.copy_to_host(d_y, r1 - r0))
            if symmetric:
                assert np.all(np.abs(y - y_seq) <= 16 * U * abs_bound(rp, ci, va, x))
            else:
                assert np.array_equal(
                    y, oracle.dist_spmv(world, rp, ci, va, x, False, ocm))
            # CG on the distributed matrix vs the oracle's P-rank CG
            b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
            x_ref, k_ref, hist_ref = oracle.dist_cg(world, rp, ci, va, b, 100,
                                                    1e-10, symmetric, ocm)
            d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_b, b[r0:r1])
            k, hist = host.cg(comm, exec_, A, d_b, d_s, 100, 1e-10)
            xs = dist_util.gather_concat(exec_.copy_to_host(d_s, r1 - r0))
            assert abs(k - k_ref) <= 1 and k < 100, (k, k_ref)
            m = min(k, k_ref, 50)
            assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6)
            assert np.linalg.norm(xs - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
            A.close()
            for p in (d_x, d_y, d_b, d_s):
                exec_.free(p)
    # L2GMap::reverse_update (L2GMap.cpp:907-959) on stand-alone maps with
    # overlapping requests: ghost tails are added into the owners' entries in
    # the reference's order, so the result is bit-exact
    sizes = [17 + 5 * r for r in range(world)]
    rngs = np.concatenate([[0], np.cumsum(sizes)])
    rng = np.random.default_rng(11)
    ghosts_all, vec_all = [], []
    for r in range(world):
        others = np.setdiff1d(np.arange(rngs[-1]), np.arange(rngs[r], rngs[r + 1]))
        ghosts_all.append(np.sort(rng.choice(others, size=min(len(others), 12),
                                             replace=False)).astype(np.int64))
        vec_all.append(rng.uniform(-1, 1, sizes[r] + len(ghosts_all[r])))
    plans = oracle.l2g_plans(sizes, ghosts_all)
    for dtype, f32 in ((np.float64, False), (np.float32, True)):
        vecs = [v.astype(dtype) for v in vec_all]
        ref = oracle.l2g_reverse_update(plans, [v.copy() for v in vecs])
        for cm in (host.P2P_BLOCKING, host.COLLECTIVE_BLOCKING,
                   host.P2P_NONBLOCKING):
            m = host.L2GMap(comm, sizes[rank], ghosts_all[rank], exec_, cm)
            d_v = exec_.alloc(len(vecs[rank]), dtype)
            exec_.copy_from_host(d_v, vecs[rank])
            m.reverse_update(d_v, f32)
            m.reverse_update(d_v, f32)  # twice: staging buffer reuse
            exec_.synchronize()
            got = exec_.copy_to_host(d_v, len(vecs[rank]), dtype)
            twice = oracle.l2g_reverse_update(plans, [v.copy() for v in ref])
            assert np.array_equal(got, twice[rank]), (dtype, cm)
            exec_.free(d_v)
            m.close()

    # create_matrix WITH row ghosts (FEM-style assembly, Matrix.cpp:188-292):
    # dyadic values => the assembled product is exact, y must equal A x
    for seed, sym in ((3, False), (4, True)):
        rng = np.random.default_rng(seed)
        Ad, ranges, inputs = assembled_inputs(rng, world, 61, symmetric=sym)
        N = Ad.shape[0]
        x = np.round(rng.uniform(-2, 2, N) * 16) / 16
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        rp, ci, va, rg, cg = inputs[rank]
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            A = host.Matrix.create_matrix(comm, exec_, rp, ci, va, r1 - r0,
                                          r1 - r0, rg, cg, sym, cm)
            l2g = A.col_map()
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            d_y = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_x, x[r0:r1])
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = dist_util.gather_concat(exec_.copy_to_host(d_y, r1 - r0))
            assert np.array_equal(y, Ad @ x), (sym, cm)
            A.close()
            exec_.free(d_x), exec_.free(d_y)

    # PETSc binary ingest on every rank (demos/cg.cpp flow), unstructured
    import tempfile
    import torch.distributed as dist
    rp, ci, va = unstructured
    N = len(rp) - 1
    tmp = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(tmp, src=0)
    fa, fb = os.path.join(tmp[0], "A.dat"), os.path.join(tmp[0], "x.dat")
    x = oracle.gaussian_x_fast(N)
    if rank == 0:
        oracle.petsc_io.write_matrix(fa, rp, ci, va)
        oracle.petsc_io.write_vector(fb, x)
    dist.barrier()
    ranges = oracle.owner_ranges(world, N)
    r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
    for symmetric in (False, True):
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            A = host.read_petsc_binary_matrix(fa, comm, exec_, symmetric, cm)
            l2g = A.col_map()
            d_v, nloc = host.read_petsc_binary_vector(comm, exec_, fb)
            assert nloc == r1 - r0
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            exec_.copy(d_x, d_v, 8 * nloc)
            d_y = exec_.alloc(nloc)
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = dist_util.gather_concat(exec_.copy_to_host(d_y, nloc))
            y_ref = oracle.dist_spmv(world, rp, ci, va, x, symmetric, cm)
            if symmetric:
                assert np.all(np.abs(y - y_ref) <= 16 * U * abs_bound(rp, ci, va, x) + 1e-300)
            else:
                assert np.array_equal(y, y_ref)
            A.close()
            for ptr in (d_v, d_x, d_y):
                exec_.free(ptr)
    # the deterministic peer reduction of the CG scalars, the ranks being
    # PROCESSES: windows reached through IPC handles.  Sums in rank order: the
    # same bits on every rank; cg() through it follows the same history as
    # through the transport's all-reduce (which sums in rank order here too)
    n = 8
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    ranges = oracle.owner_ranges(world, N)
    r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
    A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_BLOCKING)
    d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
    exec_.copy_from_host(d_b, b[r0:r1])
    k0, hist0, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 40, 1e-30, history=True)
    assert comm.enable_peer_reduce(exec_) is (world > 1)
    if world > 1:
        d_v = exec_.alloc(3)
        vals = np.random.default_rng(99).uniform(-1, 1, (30, world, 3))
        vals[5, :, 0] = [1e16 * (-1) ** p for p in range(world)]
        for rnd in range(30):
            exec_.copy_from_host(d_v, vals[rnd, rank])
            comm.reduce_sum(d_v, 3 if rnd % 2 else 1)
            got = exec_.copy_to_host(d_v, 3)
            want = np.zeros(3)
            for p in range(world):
                want += vals[rnd, p]
            ncmp = 3 if rnd % 2 else 1
            assert np.array_equal(got[:ncmp], want[:ncmp]), (rnd, got, want)
        exec_.free(d_v)
        k1, hist1, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 40, 1e-30,
                                     history=True)
        assert k1 == k0 and np.array_equal(hist1, hist0)
    A.close()
    exec_.free(d_b), exec_.free(d_s)
    comm.close()
    exec_.close()
    print(f"rank {rank}/{world}: multirank OK", flush=True)


if __name__ == "__main__":
    main()
