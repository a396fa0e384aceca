"""GPU parity tests of the C++ host mirror (HipExecutor / Matrix / L2GMap /
cg) through its C facade.  The single-rank tests follow the reference's
tests/test_spmv_cuda.cpp step by step; the multi-rank ones launch one process
per rank (gloo transport, several ranks on the one GPU of the test box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from spmv_amd import hip, host, poisson
from util import U, abs_bound, box_partition, permute_csr

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(float).eps
CMS = [host.P2P_BLOCKING, host.P2P_NONBLOCKING, host.COLLECTIVE_BLOCKING,
       host.COLLECTIVE_NONBLOCKING]
# the seeded FEM-like matrices of the ragged-row records (spmv_amd.poisson)
FEM_KINDS = {"fem": dict(), "fem_tail": dict(tail_permille=10),
             "fem81": dict(min_len=81, max_len=81)}


@pytest.fixture(scope="module")
def exec_():
    e = host.HipExecutor(0)
    yield e
    e.synchronize()
    e.close()


@pytest.fixture(scope="module")
def comm():
    c = host.Comm.self_comm()
    yield c
    c.close()


def essentially_equal(a, b, eps):  # tests/test_spmv.cpp:20-23
    return abs(a - b) <= min(abs(a), abs(b)) * eps


def spmv_like_reference_test(exec_, comm, rowptr, colind, values, x, symmetric,
                             cm):
    """tests/test_spmv_cuda.cpp:118-160 on one rank."""
    N = len(rowptr) - 1
    A = host.Matrix.create_matrix(comm, exec_, rowptr, colind, values, N, N,
                                  [], [], symmetric, cm)
    l2g = A.col_map()
    d_y = exec_.alloc(N)
    exec_.memset(d_y, 0, 8 * N)
    d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
    exec_.copy_from_host(d_x, x)
    l2g.update(d_x)
    exec_.synchronize()
    A.mult(d_x, d_y)
    exec_.synchronize()
    y = exec_.copy_to_host(d_y, N)
    meta = dict(rows=A.rows(), cols=A.cols(), nnz=A.non_zeros(),
                symmetric=A.symmetric(), fmt=A.format_size(), blocks=A.blocks())
    A.close()
    exec_.free(d_y), exec_.free(d_x)
    return y, meta


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("cm", CMS)
def test_kat_like_reference(exec_, comm, symmetric, cm):
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    y, meta = spmv_like_reference_test(exec_, comm, k["rowptr"], k["colind"],
                                       k["values"], np.array(k["x"]),
                                       symmetric, cm)
    norm = float(np.sqrt(np.sum(y * y)))
    # the reference's own criterion, and bit-exact y -- both storages (the
    # reference's general and symmetric branches agree bit for bit on the KAT)
    assert essentially_equal(norm, k["norm_y"], EPS)
    assert list(y) == k["y"]
    assert meta["rows"] == 5 and meta["nnz"] == 15
    assert meta["symmetric"] == symmetric
    assert exec_.device_type == 2  # DeviceType::gpu


def test_all_eight_models_are_accepted(exec_, comm):
    """tests/test_spmv.cpp:180-262 loops the models; the one-sided and shmem
    ones behave like the blocking p2p model here."""
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
    for cm in (host.ONESIDED_PUT_ACTIVE, host.ONESIDED_PUT_PASSIVE, host.SHMEM,
               host.SHMEM_NODUP):
        y, meta = spmv_like_reference_test(exec_, comm, k["rowptr"], k["colind"],
                                           k["values"], np.array(k["x"]), False,
                                           cm)
        assert list(y) == k["y"] and meta["blocks"]["remote"] == (0, 0, 0)


@pytest.mark.parametrize("n", [5, 12])
@pytest.mark.parametrize("symmetric", [False, True])
def test_poisson_host_vs_device_generator(exec_, comm, n, symmetric):
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    x = oracle.gaussian_x_fast(N)
    y_ref = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    if symmetric:  # the symmetric branch's own order (csr_kernels.cpp:26-40)
        y_ref = oracle.csr_spmv_sym(*oracle.poisson3d_lower(n), x)
    for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
        y, meta = spmv_like_reference_test(exec_, comm, rp, ci, va, x,
                                           symmetric, cm)
        assert np.array_equal(y, y_ref)
        A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric, cm)
        assert (A.rows(), A.non_zeros(), A.format_size()) == (
            meta["rows"], meta["nnz"], meta["fmt"])
        assert A.blocks() == meta["blocks"]
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        A.col_map().update(d_x)
        A.mult(d_x, d_y)
        y2 = exec_.copy_to_host(d_y, N)
        assert np.array_equal(y2, y_ref)
        A.close()
        exec_.free(d_x), exec_.free(d_y)


@pytest.mark.parametrize("symmetric", [False, True])
def test_cg_matches_oracle(exec_, comm, symmetric):
    """spmv::cg vs oracle.cg (spmv/cg.cpp:21-98): same k (+-1), residual
    history to 1e-6 over the first 50 iterations, x to 1e-8."""
    n = 14
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci32 = ci.astype(np.int32)
    for rhs in ("A*ones", "gaussian"):
        b = (oracle.csr_spmv(rp, ci32, va, np.ones(N)) if rhs == "A*ones"
             else oracle.gaussian_x_fast(N))
        x_ref, k_ref, hist_ref = oracle.cg(rp, ci32, va, b, 100, 1e-10)
        A = host.Matrix.create_matrix(comm, exec_, rp, ci, va, N, N, [], [],
                                      symmetric, host.P2P_NONBLOCKING)
        d_b, d_x = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_b, b)
        k, hist = host.cg(comm, exec_, A, d_b, d_x, 100, 1e-10)
        x = exec_.copy_to_host(d_x, N)
        assert k_ref < 100 and abs(k - k_ref) <= 1
        assert len(hist) == k + 1 and hist[-1] / hist[0] < 1e-10
        m = min(k, k_ref, 50)
        assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6, atol=0)
        assert np.linalg.norm(x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
        # kmax smaller than needed: returns kmax (cg.cpp:55), x = iterate kmax
        k2, hist2 = host.cg(comm, exec_, A, d_b, d_x, 5, 1e-10)
        x5_ref, k5, h5 = oracle.cg(rp, ci32, va, b, 5, 1e-10)
        assert k2 == 5 == k5
        assert np.allclose(hist2, h5, rtol=1e-9)
        assert np.linalg.norm(exec_.copy_to_host(d_x, N) - x5_ref) <= 1e-11 * np.linalg.norm(x5_ref)
        A.close()
        exec_.free(d_b), exec_.free(d_x)


@pytest.mark.parametrize("kind", ["unstructured", "banded", "skewed_lattice"])
def test_cg_on_other_spd_matrices_vs_oracle(exec_, comm, kind):
    """spmv::cg off the Poisson matrix: a random unstructured SPD matrix (gather
    kernels / transposed map), a banded one (LX form), and a 3-D lattice with
    random symmetric positive-definite values (lattice + diagonal forms with
    values that are not -1 / 6) -- both storages, against oracle.cg."""
    import scipy.sparse as sp
    from spmv_amd import _lib
    rng = np.random.default_rng({"unstructured": 5, "banded": 6,
                                 "skewed_lattice": 7}[kind])
    for opt_ in (b"lx_min_nnz", b"lat_min_nnz"):
        _lib.call("spmv_hip_ctx_set_option", exec_.context, opt_, 0)
    if kind == "unstructured":
        N = 6000
        L = sp.random(N, N, density=4.0 / N, random_state=5, format="csr")
        L = sp.tril(L, -1)
    elif kind == "banded":
        N = 9000
        diags = [rng.uniform(-1, 1, N) for _ in range(5)]
        L = sp.diags(diags, [-1, -2, -3, -40, -41], shape=(N, N), format="csr")
    else:
        n = 16
        N = n ** 3
        rp0, ci0, _ = poisson.poisson3d_csr(n)
        A0 = sp.csr_matrix((rng.uniform(0.2, 1.0, len(ci0)), ci0, rp0), shape=(N, N))
        L = sp.tril(A0, -1)
    S = (L + L.T).tocsr()
    # strictly diagonally dominant with a positive diagonal: SPD
    d = np.asarray(abs(S).sum(1)).ravel() + rng.uniform(0.5, 1.5, N)
    A = (S + sp.diags(d)).tocsr()
    A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    b = oracle.csr_spmv(rp, ci, va, rng.uniform(-1, 1, N))
    x_ref, k_ref, hist_ref = oracle.cg(rp, ci, va, b, 200, 1e-10)
    assert 3 < k_ref < 200
    for symmetric in (False, True):
        M = host.Matrix.create_matrix(comm, exec_, rp, ci, va, N, N, [], [],
                                      symmetric, host.P2P_NONBLOCKING)
        d_b, d_x = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_b, b)
        k, hist = host.cg(comm, exec_, M, d_b, d_x, 200, 1e-10)
        x = exec_.copy_to_host(d_x, N)
        assert abs(k - k_ref) <= 1, (kind, symmetric, k, k_ref)
        m = min(k, k_ref, 50)
        assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6, atol=0)
        assert np.linalg.norm(x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
        if kind == "skewed_lattice":
            assert M.plan_get("sdia") == 1
        M.close()
        exec_.free(d_b), exec_.free(d_x)
    for opt_ in (b"lx_min_nnz", b"lat_min_nnz"):
        _lib.call("spmv_hip_ctx_set_option", exec_.context, opt_, 1 << 20)


def test_cg_early_convergence_with_long_queue(exec_, comm):
    """kmax far beyond convergence: the device stops itself, the host stops
    enqueuing at the next poll; k and x are those of the converged iterate."""
    n = 6
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    x_ref, k_ref, _ = oracle.cg(rp, ci.astype(np.int32), va, b, 2000, 1e-12)
    A = host.Matrix.create_matrix(comm, exec_, rp, ci, va, N, N, [], [], False,
                                  host.P2P_BLOCKING)
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_b, b)
    k, hist = host.cg(comm, exec_, A, d_b, d_x, 2000, 1e-12)
    assert abs(k - k_ref) <= 1 and k < 100
    assert np.linalg.norm(exec_.copy_to_host(d_x, N) - 1.0) < 1e-9 * np.sqrt(N)
    A.close()
    exec_.free(d_b), exec_.free(d_x)


@pytest.mark.parametrize("world,n", [(2, 10), (3, 12), (8, 16)])
def test_onesided_halo_ranks_threaded(world, n):
    """The one-sided models (L2GMap.cpp:645-682) with the ranks as THREADS of
    this process: every rank's window is reached by address, the halo is one
    put kernel per rank and exchange (signal free / store / raise the data
    flag / copy the staging buffer into the ghost tail).  SpMV bit-exact
    against the oracle's P-rank simulation, CG over 60+ exchanges follows its
    history; the ranks as PROCESSES (IPC handles) -- the deployment's shape --
    are test_multirank_on_one_gpu's business.

    The streams of ONE process share its hardware queues (four by default):
    a put kernel that polls can then sit in a queue in front of the very
    kernel it waits for -- the bounded waits expire and the exchange reports
    SPMV_HIP_EPEER, which is what they are for.  tests/conftest.py therefore
    asks the runtime for 24 queues before HIP starts and this test checks that
    the request is in effect (an environment that pins fewer queues skips: a
    property of ranks as threads, not of the protocol -- ranks in processes of
    their own have their own queues).  With the queues in place a timed-out
    exchange is a FAILURE."""
    from thread_world import ThreadWorld
    queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    if queues < 2 * world + 2:
        pytest.skip(f"GPU_MAX_HW_QUEUES={queues} pinned by the environment: "
                    f"{world} threaded ranks need {2 * world + 2} queues")
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    x = oracle.gaussian_x_fast(N)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    ranges = oracle.owner_ranges(world, N)
    refs = {sym: (oracle.dist_spmv(world, rp, ci, va, x, sym, host.P2P_BLOCKING),
                  oracle.dist_cg(world, rp, ci, va, b, 60, 1e-30, sym,
                                 host.P2P_BLOCKING))
            for sym in (False, True)}
    tw = ThreadWorld(world, timeout=60.0)

    def rank_body(rank, comm, exec_):
        import ctypes as C
        from spmv_amd import _lib
        # a compute stream of its own per rank, as ranks in processes of their
        # own have: threads would otherwise share the process's default stream,
        # and a rank whose stream waits for its halo would hold back the
        # neighbour's work queued behind it -- the very stores it waits for
        stream = C.c_void_p()
        _lib.call("spmv_hip_stream_create", exec_.context, C.byref(stream))
        _lib.call("spmv_hip_set_stream", exec_.context, stream)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        ws = host.CgWorkspace(exec_)
        for cm in (host.ONESIDED_PUT_ACTIVE, host.ONESIDED_PUT_PASSIVE):
            for sym, (y_ref, (x_ref, k_ref, hist_ref)) in refs.items():
                A = host.Matrix.create_poisson3d(comm, exec_, n, sym, cm)
                l2g = A.col_map()
                assert l2g.onesided() and not l2g.overlapping()
                ng = l2g.num_ghosts()
                d_x, d_y = exec_.alloc(r1 - r0 + ng), exec_.alloc(r1 - r0)
                d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
                exec_.copy_from_host(d_b, b[r0:r1])
                host.cg_ex(comm, exec_, A, d_b, d_s, 0, 1e-30, ws)  # sizes ws,
                #                                        exchanges nothing
                # Everything is allocated: from here to the barrier below no
                # rank calls hipMalloc / hipFree.  (Those wait for the whole
                # DEVICE -- with the ranks as threads of one process, for a
                # neighbour's put kernel that in turn waits for this rank's
                # next exchange.  Ranks in processes of their own, the real
                # deployment, do not share that wait.)
                tw.bar.wait()
                for rep in range(3):  # back to back: three more epochs
                    exec_.memset(d_x, 0xFF, 8 * (r1 - r0 + ng))  # NaN ghosts
                    exec_.copy_from_host(d_x, x[r0:r1])
                    l2g.update(d_x)
                    A.mult(d_x, d_y)
                got = exec_.copy_to_host(d_x, r1 - r0 + ng)
                assert np.array_equal(got[r1 - r0:], x[l2g.ghosts()])
                y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
                assert np.array_equal(y, y_ref), (cm, sym)
                k, hist, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 60, 1e-30,
                                           ws, history=True)
                xs = tw.gather(rank, exec_.copy_to_host(d_s, r1 - r0))
                assert k == 60 == k_ref
                assert np.allclose(hist, hist_ref, rtol=1e-7), (cm, sym)
                assert np.linalg.norm(xs - x_ref) <= 1e-9 * np.linalg.norm(x_ref)
                tw.bar.wait()  # nobody closes while a neighbour still exchanges
                A.close()
                for p in (d_x, d_y, d_b, d_s):
                    exec_.free(p)
                tw.bar.wait()
        ws.close()
        exec_.synchronize()
        _lib.call("spmv_hip_set_stream", exec_.context, None)
        _lib.call("spmv_hip_stream_destroy", exec_.context, stream)

    # a timed-out exchange (SPMV_HIP_EPEER) FAILS the test: with the queues
    # asked for above its cause is the protocol, not the harness
    tw.run(rank_body, gpu=True)


@pytest.mark.parametrize("world,model", [(2, "P2P_BLOCKING"), (3, "P2P_NONBLOCKING"),
                                         (8, "P2P_BLOCKING"), (8, "P2P_NONBLOCKING")])
def test_peer_reduce_ranks_threaded(world, model):
    """The deterministic peer reduction of the CG scalars (spmv_hip_reduce_*,
    Comm::enable_peer_reduce / reduce_sum) with the ranks as THREADS of this
    process: one single-wave kernel per rank and reduction, every rank's value
    stored into every rank's window, added in rank order.  The same bits on
    every rank, equal to the left-to-right sum in rank order (a case that
    cancels at 1e16 included), 1 and 3 values per reduction, 60 reductions back
    to back; cg() through it follows, bit for bit, the history it has through
    the transport's all-reduce (which adds in rank order here too)."""
    from thread_world import ThreadWorld
    queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    if queues < 2 * world + 2:
        pytest.skip(f"GPU_MAX_HW_QUEUES={queues} pinned by the environment: "
                    f"{world} threaded ranks need {2 * world + 2} queues")
    n = 10 if world < 8 else 16
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    ranges = oracle.owner_ranges(world, N)
    rounds = 60
    vals = np.random.default_rng(5).uniform(-1, 1, (rounds, world, 3))
    vals[7, :, 0] = [1e16 * (-1) ** p for p in range(world)]
    vals[7, world - 1, 0] += 1.0
    want = np.zeros((rounds, 3))
    for rnd in range(rounds):
        for p in range(world):  # left to right in rank order
            want[rnd] += vals[rnd, p]
    tw = ThreadWorld(world, timeout=60.0)

    def rank_body(rank, comm, exec_):
        import ctypes as C
        from spmv_amd import _lib
        stream = C.c_void_p()  # a compute stream of its own per rank
        _lib.call("spmv_hip_stream_create", exec_.context, C.byref(stream))
        _lib.call("spmv_hip_set_stream", exec_.context, stream)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        # (a stuck wait -- a reduction kernel parked in front of the kernel it
        # waits for -- fails in 20 s instead of 60)
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"put_timeout_ms", 20000)
        A = host.Matrix.create_poisson3d(comm, exec_, n, False, getattr(host, model))
        ws = host.CgWorkspace(exec_)
        d_b, d_s, d_v = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0), exec_.alloc(3)
        exec_.copy_from_host(d_b, b[r0:r1])
        k0, hist0, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 40, 1e-30, ws,
                                     history=True)
        assert comm.enable_peer_reduce(exec_)
        tw.bar.wait()  # (no hipMalloc / hipFree between here and the last wait)
        for rnd in range(rounds):
            exec_.copy_from_host(d_v, vals[rnd, rank])
            cnt = 3 if rnd % 2 else 1
            comm.reduce_sum(d_v, cnt)
            got = exec_.copy_to_host(d_v, 3)
            assert np.array_equal(got[:cnt], want[rnd, :cnt]), (rank, rnd, got)
        k1, hist1, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 40, 1e-30, ws,
                                     history=True)
        assert k1 == k0 and np.array_equal(hist1, hist0), rank
        exec_.synchronize()
        tw.bar.wait()
        ws.close()
        A.close()
        for p in (d_b, d_s, d_v):
            exec_.free(p)
        _lib.call("spmv_hip_set_stream", exec_.context, None)
        _lib.call("spmv_hip_stream_destroy", exec_.context, stream)

    tw.run(rank_body, gpu=True)


def test_library_refuses_peer_reduce_beside_the_onesided_halo():
    """The pair `deterministic peer reduction + one-sided halo` is refused by
    the LIBRARY, collectively (VERDICT r04 #3c): on a
    communicator that carries a one-sided L2GMap enable_peer_reduce() returns
    false on every rank and cg() keeps the transport's all-reduce; on a
    communicator with the peer reduction an L2GMap asked for
    onesided_put_active falls back to the two-sided exchange on every rank.
    Both orders end in a working solve with the history of the plain models."""
    from thread_world import ThreadWorld
    world, n = 3, 12
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    ranges = oracle.owner_ranges(world, N)
    tw = ThreadWorld(world, timeout=60.0)
    seen = {}

    def rank_body(rank, comm, exec_):
        import ctypes as C
        from spmv_amd import _lib
        stream = C.c_void_p()
        _lib.call("spmv_hip_stream_create", exec_.context, C.byref(stream))
        _lib.call("spmv_hip_set_stream", exec_.context, stream)
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"put_timeout_ms", 20000)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
        exec_.copy_from_host(d_b, b[r0:r1])
        ws = host.CgWorkspace(exec_)
        # the reference history: two-sided halo, transport's all-reduce
        A0 = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_BLOCKING)
        k0, hist0, _, _ = host.cg_ex(comm, exec_, A0, d_b, d_s, 30, 1e-30, ws,
                                     history=True)
        A0.close()
        # (1) the one-sided map first: the reduction is refused
        A = host.Matrix.create_poisson3d(comm, exec_, n, False,
                                         host.ONESIDED_PUT_ACTIVE)
        assert A.col_map().onesided()
        tw.bar.wait()
        assert comm.enable_peer_reduce(exec_) is False
        k1, hist1, _, _ = host.cg_ex(comm, exec_, A, d_b, d_s, 30, 1e-30, ws,
                                     history=True)
        assert k1 == k0 and np.array_equal(hist1, hist0), rank
        exec_.synchronize()
        tw.bar.wait()
        A.close()
        # (2) the reduction first: the map falls back to the two-sided exchange
        assert comm.enable_peer_reduce(exec_) is True
        B = host.Matrix.create_poisson3d(comm, exec_, n, False,
                                         host.ONESIDED_PUT_ACTIVE)
        assert not B.col_map().onesided()
        tw.bar.wait()
        k2, hist2, _, _ = host.cg_ex(comm, exec_, B, d_b, d_s, 30, 1e-30, ws,
                                     history=True)
        assert k2 == k0 and np.array_equal(hist2, hist0), rank
        exec_.synchronize()
        tw.bar.wait()
        seen[rank] = True
        ws.close()
        B.close()
        exec_.free(d_b), exec_.free(d_s)
        _lib.call("spmv_hip_set_stream", exec_.context, None)
        _lib.call("spmv_hip_stream_destroy", exec_.context, stream)

    tw.run(rank_body, gpu=True)
    assert len(seen) == world


@pytest.mark.parametrize("cm", ["p2p_nonblocking", "onesided_put_active"])
def test_rehearsal_of_config4_eight_ranks_at_512_cubed(cm):
    """BASELINE configs[4] at its real rank count and per-rank shapes on the one
    GPU of the test box: the 512^3 matrix on 8 ranks as threads of one process
    (tools/rehearsal_threads.py; a box admits 6 GPU processes).  Every local
    block 512 x 512 x 64 in the constant-diagonal tile form and the remote
    blocks ROWLIST (overlapping model); the halo self-check passes on every
    rank; 20 CG iterations land on the 1-rank residual after 10 iterations.
    Both halo models."""
    avail = _mem_available_gb()
    if avail < 8:
        pytest.skip(f"MemAvailable is {avail:.0f} GB")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="24")
    res = subprocess.run([sys.executable,
                          os.path.join(ROOT, "tools", "rehearsal_threads.py"),
                          "--cm", cm], capture_output=True, text=True, timeout=600,
                         env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["ranks"] == 8 and d["config"]["rows"] == 512 ** 3
    assert d["halo_selfcheck"] == "ok" and d["every_rank_same_k10"]
    assert d["cg_rel_residual"]["k10_ok"] is True
    assert [r["rows"] for r in d["ranks"]] == [512 * 512 * 64] * 8
    assert [r["ghosts"] for r in d["ranks"]] == [512 * 512] + [2 * 512 * 512] * 6 \
        + [512 * 512]
    assert d["every_local_block_same_form"]
    if cm == "p2p_nonblocking":
        f = d["local_block_form"]
        assert f["sdia"] == 1 and f["sdia_const"] == 1 and f["sdia_tile"] == 4
        assert d["remote_block_algo"] == [4]  # SPMV_HIP_ALGO_ROWLIST
        assert not d["onesided_put_path"]
    else:
        assert d["onesided_put_path"]


@pytest.mark.parametrize("world", [2, 3])
def test_multirank_on_one_gpu(world):
    """N>1 path: one process per rank, all on GPU 0, halo + reductions over a
    gloo-backed CallbackComm (tests/mp_gpu_worker.py)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mp_gpu_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-4000:]
    assert res.stdout.count("multirank OK") == world


def test_rccl_comm_single_rank(exec_):
    """RcclComm over a 1-rank communicator: exercises ncclGetUniqueId /
    ncclCommInitRank / in-stream all-reduce / staged all-gather through the
    library the process actually loaded (torch's bundled RCCL)."""
    ident = host.rccl_unique_id()
    assert len(ident) == 128 and any(ident)
    comm = host.Comm.rccl(exec_, 1, 0, ident)
    info = comm.rccl_info()  # as RCCL itself reports it
    assert info["nranks"] == 1 and info["rank"] == 0
    assert info["version_code"] >= 20000 and "rccl" in info["lib_path"].lower()
    assert info["separate_reduction_comm"] is False  # nothing to split off
    n = 9
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    A = host.Matrix.create_matrix(comm, exec_, rp, ci, va, N, N, [], [], False,
                                  host.P2P_NONBLOCKING)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_b, b)
    k, hist = host.cg(comm, exec_, A, d_b, d_x, 100, 1e-10)
    x_ref, k_ref, _ = oracle.cg(rp, ci.astype(np.int32), va, b, 100, 1e-10)
    assert abs(k - k_ref) <= 1
    assert np.linalg.norm(exec_.copy_to_host(d_x, N) - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    A.close()
    exec_.free(d_b), exec_.free(d_x)
    comm.close()


def test_rccl_grouped_send_recv_loopback(exec_):
    """The grouped ncclSend / ncclRecv of the halo exchange (hip/comm.hip,
    L2GMap.cpp:564-642) EXECUTED on a 1-GPU box: a one-rank RCCL communicator
    whose rank exchanges with itself -- two "neighbours" (both rank 0), distinct
    segments, fp64 and fp32, on a side stream behind a kernel that produces the
    send buffer (stream order, no host wait between them).  Not a transfer over
    xGMI -- but the calls, their grouping and their stream semantics run through
    the RCCL library the process loaded."""
    import ctypes as C
    from spmv_amd import _lib
    ctx = exec_.context
    ident = host.rccl_unique_id()
    comm = C.c_void_p()
    _lib.call("spmv_hip_comm_create", ctx, 1, 0, bytes(ident), C.byref(comm))
    stream = C.c_void_p()
    _lib.call("spmv_hip_stream_create", ctx, C.byref(stream))
    n0, n1 = 70_001, 262_144
    I32 = C.c_int32 * 2
    nb = I32(0, 0)
    scnt, soff = I32(n0, n1), I32(5, 5 + n0)
    rcnt, roff = I32(n0, n1), I32(n1 + 3, 3)  # the two segments swap places
    for elem, fn, dt in ((8, "spmv_hip_comm_neighbor_exchange_f64", np.float64),
                         (4, "spmv_hip_comm_neighbor_exchange_f32", np.float32)):
        total = 5 + n0 + n1
        d_send = exec_.alloc(total * elem // 8 + 1)
        d_recv = exec_.alloc((n0 + n1 + 3) * elem // 8 + 1)
        exec_.memset(d_recv, 0xFF, (n0 + n1 + 3) * elem)
        exec_.synchronize()
        # the send buffer is produced ON the stream, right before the exchange
        if elem == 8:
            _lib.call("spmv_hip_fill_gaussian_f64", ctx, total, 0, total, d_send,
                      stream)
            want = np.exp(-10 * (5 * (np.arange(total) / float(total) - 0.5)) ** 2)
        else:
            want = np.linspace(-1, 1, total).astype(np.float32)
            _lib.call("spmv_hip_copy_h2d_async", ctx, d_send,
                      want.ctypes.data_as(C.c_void_p), total * 4, stream)
        _lib.call(fn, comm, 2, nb, d_send, scnt, soff, d_recv, rcnt, roff, stream)
        _lib.call("spmv_hip_stream_synchronize", ctx, stream)
        got = np.empty(n0 + n1 + 3, dt)
        _lib.call("spmv_hip_copy_d2h_async", ctx, got.ctypes.data_as(C.c_void_p),
                  d_recv, (n0 + n1 + 3) * elem, None)
        exec_.synchronize()
        if elem == 8:
            assert np.allclose(got[n1 + 3:], want[5:5 + n0], rtol=1e-15, atol=0)
            assert np.allclose(got[3:3 + n1], want[5 + n0:], rtol=1e-15, atol=0)
        else:
            assert np.array_equal(got[n1 + 3:], want[5:5 + n0])
            assert np.array_equal(got[3:3 + n1], want[5 + n0:])
        assert np.all(np.isnan(got[:3]))  # untouched
        exec_.free(d_send), exec_.free(d_recv)
    # ... and the scalar all-reduce of cg() (cg.cpp:49,65,75) through ncclAllReduce
    # on the same stream: one rank, so the sum is the value itself
    vals = np.array([3.25, -1.5, 1e-300])
    d_v = exec_.alloc(3)
    _lib.call("spmv_hip_copy_h2d_async", ctx, d_v, vals.ctypes.data_as(C.c_void_p),
              24, stream)
    _lib.call("spmv_hip_comm_allreduce_sum_f64", comm, d_v, 3, stream)
    _lib.call("spmv_hip_stream_synchronize", ctx, stream)
    assert np.array_equal(exec_.copy_to_host(d_v, 3), vals)
    exec_.free(d_v)
    _lib.call("spmv_hip_stream_destroy", ctx, stream)
    _lib.call("spmv_hip_comm_destroy", comm)


@pytest.mark.parametrize("symmetric", [False, True])
def test_read_petsc_binary(exec_, comm, tmp_path, symmetric):
    """demos/cg.cpp flow: read A and b from PETSc binary files, solve."""
    n = 7
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    fa, fb = tmp_path / "A.dat", tmp_path / "b.dat"
    oracle.petsc_io.write_matrix(fa, rp, ci, va)
    oracle.petsc_io.write_vector(fb, b)
    A = host.read_petsc_binary_matrix(fa, comm, exec_, symmetric,
                                      host.P2P_NONBLOCKING)
    assert A.rows() == N and A.non_zeros() == len(va)
    d_b, nloc = host.read_petsc_binary_vector(comm, exec_, fb)
    assert nloc == N and np.array_equal(exec_.copy_to_host(d_b, N), b)
    d_x = exec_.alloc(N)
    k, hist = host.cg(comm, exec_, A, d_b, d_x, 100, 1e-10)
    x_ref, k_ref, _ = oracle.cg(rp, ci.astype(np.int32), va, b, 100, 1e-10)
    assert abs(k - k_ref) <= 1
    assert np.linalg.norm(exec_.copy_to_host(d_x, N) - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    A.close()
    exec_.free(d_b), exec_.free(d_x)


@pytest.mark.parametrize("kind", ["fem", "fem_tail"])
def test_petsc_file_of_a_ragged_matrix_one_million_rows(exec_, comm, tmp_path, kind):
    """demos/spmv.cpp:43 / demos/cg.cpp:47-51 on a file of the size the reader
    is for: tools/write_petsc.py writes the 1 M-row FEM-like matrix (ragged rows;
    with a tail of 200-2000-entry rows) as a PETSc binary file, the product
    reads it (spmv/read_petsc.cpp:40-228), builds the distributed matrix and
    multiplies -- every element identical to the oracle's loop on the same
    arrays; the plan is the sliced jagged form."""
    N = 1_000_000
    fa = tmp_path / "A.dat"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "write_petsc.py"),
                          "--kind", kind, "--rows", str(N), "--out", str(fa)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    rp, ci, va = poisson.fem_like_csr(N, **FEM_KINDS[kind])
    x = oracle.gaussian_x_fast(N) + 0.25
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    A = host.read_petsc_binary_matrix(fa, comm, exec_, False, host.P2P_NONBLOCKING)
    assert A.rows() == N and A.non_zeros() == len(va)
    assert A.plan_get("sjds") == 1
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_x, x)
    exec_.memset(d_y, 0xFF, 8 * N)
    A.col_map().update(d_x)
    A.mult(d_x, d_y)
    assert np.array_equal(exec_.copy_to_host(d_y, N), y_ref)
    A.close()
    # ... and read with symmetric = true (demos/cg.cpp:47: the reader keeps the
    # strictly lower part and the diagonal, Matrix.cpp:337-349): the merged
    # matrix in the sliced jagged form, the long rows of the stored block (the
    # tail) by the long-row kernels -- the reference's symmetric loop bit for bit
    from util import lower_split
    lrp, lci, lva, ldg = lower_split(rp, ci, va)
    A = host.read_petsc_binary_matrix(fa, comm, exec_, True, host.P2P_NONBLOCKING)
    assert A.symmetric() and A.rows() == N
    assert A.plan_get("sym_sj") == 1
    assert (A.plan_get("sj_long_rows") > 0) == (kind == "fem_tail")
    exec_.memset(d_y, 0xFF, 8 * N)
    A.mult(d_x, d_y)
    assert np.array_equal(exec_.copy_to_host(d_y, N),
                          oracle.csr_spmv_sym(lrp, lci, lva, ldg, x))
    A.close()
    exec_.free(d_x), exec_.free(d_y)


@pytest.mark.parametrize("symmetric", [False, True])
def test_matrix_fp32(exec_, comm, symmetric):
    """Matrix<float>: the fp32 visitors of the executor interface
    (device_executor.h:88-99)."""
    n = 9
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n, dtype=np.float32)
    x = oracle.gaussian_x_fast(N).astype(np.float32)
    y_ref = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
        A = host.MatrixF32(comm, exec_, rp, ci, va, N, N, [], [], symmetric, cm)
        assert A.info() == dict(rows=N, nnz=len(va), local_size=N, num_ghosts=0)
        d_x, d_y = exec_.alloc(N, np.float32), exec_.alloc(N, np.float32)
        exec_.copy_from_host(d_x, x)
        A.update(d_x)
        A.mult(d_x, d_y)
        y = exec_.copy_to_host(d_y, N, np.float32)
        if symmetric:
            assert np.allclose(y, y_ref, rtol=0, atol=16 * 2.0 ** -24 * 12)
        else:
            assert np.array_equal(y, y_ref)
        A.close()
        exec_.free(d_x), exec_.free(d_y)


def test_benchmark_scale_512_cubed(exec_, comm):
    """BASELINE.json's full size (134 M rows, 938 M entries, byte offsets far
    beyond 2^32) through the product path, checked by size-independent
    properties: A*1 is an exact small integer per row for BOTH storages
    (general, and symmetric with its atomic scatter), and CG on b = A*1 must
    drive the residual down 6 orders and return the all-ones solution."""
    n = 512
    N = n ** 3
    i = np.arange(N)
    xx, yy, zz = i % n, (i // n) % n, i // (n * n)
    expect = (6 - ((xx > 0).astype(np.int8) + (xx < n - 1) + (yy > 0)
                   + (yy < n - 1) + (zz > 0) + (zz < n - 1))).astype(np.float64)
    del i, xx, yy, zz
    ctx = exec_.context
    from spmv_amd import _lib
    d_one, d_b = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_const_f64", ctx, N, 1.0, d_one, None)
    for symmetric in (False, True):
        A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                         host.P2P_NONBLOCKING)
        assert A.rows() == N and A.non_zeros() == poisson.poisson3d_nnz(n)
        A.col_map().update(d_one)
        A.mult(d_one, d_b)
        assert np.array_equal(exec_.copy_to_host(d_b, N), expect), symmetric
        if not symmetric:
            d_x = exec_.alloc(N)
            k, hist = host.cg(comm, exec_, A, d_b, d_x, 3000, 1e-6)
            assert k < 3000 and hist[-1] / hist[0] < 1e-6
            x = exec_.copy_to_host(d_x, N)
            assert np.abs(x - 1.0).max() < 1e-3
            exec_.free(d_x)
        A.close()
    exec_.free(d_one), exec_.free(d_b)


@pytest.mark.parametrize("n", [448, 512])
def test_full_size_kernels_agree_bit_for_bit(exec_, comm, n):
    """Size-independent cross-check at the benchmark's scale (and at 448^3,
    where the plane-walk table has 32 runs and empty slots): every kernel is
    bit-exact against the oracle at the sizes the oracle can do, so at full
    size the DIAGONAL form (the default) must agree bit for bit with the
    CSR-order lattice kernels on the same plan, for the reference's Gaussian x
    and both storages; the mixed-precision copy likewise (Poisson values are
    exact in fp32)."""
    N = n ** 3
    ctx = exec_.context
    from spmv_amd import _lib
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_x, None)
    for symmetric in (False, True):
        A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                         host.P2P_NONBLOCKING)
        assert A.plan_get("sdia") == 1 and A.plan_get("zwalk") == 1
        A.mult(d_x, d_y)
        y_dia = exec_.copy_to_host(d_y, N)
        assert np.isfinite(y_dia).all() and np.abs(y_dia).max() > 0
        A.plan_set("sdia", 0)                     # lattice / symmetric lattice
        assert A.plan_get("lat" if not symmetric else "slat") == 1
        _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
        A.mult(d_x, d_y)
        assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia), symmetric
        A.plan_set("zwalk", 0)                    # ... in the plain order
        _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
        A.mult(d_x, d_y)
        assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia), symmetric
        A.plan_set("sdia", 1)
        if not symmetric:
            assert A.enable_mixed() and A.plan_get("sdia_mixed") == 1
            A.use_mixed(True)
            _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
            A.mult(d_x, d_y)
            A.use_mixed(False)
            assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia)
        A.close()
        # The VALUE-STREAMING diagonal form (what a lattice matrix with varying
        # coefficients gets, and what bench.py's value_stream_* records time):
        # the same matrix with the constant-diagonal detection off keeps its
        # values by offset; plane chain / ring on and off.  Same bits.
        _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals", 0)
        try:
            V = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                             host.P2P_NONBLOCKING)
        finally:
            _lib.call("spmv_hip_ctx_set_option", ctx, b"const_diagonals", 1)
        assert V.plan_get("sdia") == 1 and V.plan_get("sdia_const") == 0
        assert V.plan_get("sdia_general") == (0 if symmetric else 1)
        for chain in (1, 0):
            V.plan_set("sdia_chain", chain)
            _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
            V.mult(d_x, d_y)
            assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia), (
                "value-streaming diagonal form", symmetric, chain)
        V.close()
        if symmetric:
            continue
        # The CSR-ORDER kernels every matrix WITHOUT lattice structure gets --
        # the LX form, the plain row-block gather kernel -- and the one-lane-
        # per-row kernel (the reference loop verbatim) on the same matrix with
        # the lattice analysis switched off: 64-bit offsets into 7.5 GB of
        # values, row blocks beyond 2^19, the persistent grids.  Same bits.
        off = 1 << 62
        for name, opts, form in (
                ("lx", {b"lat_min_nnz": off}, dict(lat=0, lx=1, sjds=0)),
                ("sjds", {b"lat_min_nnz": off, b"lx_min_nnz": off},
                 dict(lat=0, lx=0, sjds=1)),
                ("rowblock", {b"lat_min_nnz": off, b"lx_min_nnz": off,
                              b"sj_min_nnz": off}, dict(lat=0, lx=0, sjds=0)),
                ("scalar", {b"lat_min_nnz": off, b"lx_min_nnz": off,
                            b"sj_min_nnz": off}, dict(lat=0, lx=0, sjds=0))):
            for k, v in opts.items():
                _lib.call("spmv_hip_ctx_set_option", ctx, k, v)
            try:
                B = host.Matrix.create_poisson3d(comm, exec_, n, False,
                                                 host.P2P_NONBLOCKING)
            finally:
                for k in (b"lat_min_nnz", b"lx_min_nnz", b"sj_min_nnz"):
                    _lib.call("spmv_hip_ctx_set_option", ctx, k, 1 << 20)
            for key, want in form.items():
                assert B.plan_get(key) == want, (name, key)
            assert B.plan_get("sdia") == 0
            if name == "scalar":
                B.plan_set("algo", 3)  # SPMV_HIP_ALGO_SCALAR
            _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
            B.mult(d_x, d_y)
            assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia), name
            if name == "rowblock":
                # that was the XW kernel (the caller's arrays, x windows
                # staged, plane-walk order); the gather kernel on the same plan
                assert B.plan_get("xw") == 1 and B.plan_get("zwalk") == 1
                B.plan_set("xw", 0)
                _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y,
                          None)
                B.mult(d_x, d_y)
                assert np.array_equal(exec_.copy_to_host(d_y, N), y_dia), "gather"
            B.close()
    exec_.free(d_x), exec_.free(d_y)


def test_release_csr_keeps_a_fem_matrix_near_one_copy(exec_, comm):
    """PLAN MEMORY at the mirror: a 10 M-row x 15 FEM-like
    matrix in the sliced jagged form holds the plan's copy (values + 16-bit
    codes) beside the caller's CSR arrays -- 1.9 times the CSR bytes resident.
    CSRMatrix::release_csr gives colind and values back: at most 1.2 times the
    CSR bytes stay, mult() returns the same bits, the fused dot still works, and
    the context option "release_csr" does it at creation."""
    from spmv_amd import _lib
    ctx = exec_.context
    N = 10_000_000
    A = host.Matrix.create_fem_like(comm, exec_, N)
    rows, cols, nnz = A.blocks()["local"]
    csr_bytes = nnz * 12 + (rows + 1) * 4
    plan_bytes = A.plan_get("plan_kib") * 1024
    assert A.plan_get("sjds") == 1 and A.plan_get("sj_long_rows") == 0
    assert 0.8 < plan_bytes / csr_bytes < 1.1
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_x, None)
    A.mult(d_x, d_y)
    y_before = exec_.copy_to_host(d_y, N)
    freed = A.release_csr()
    assert freed == nnz * 12
    assert A.release_csr() == 0  # once
    resident = csr_bytes - freed + plan_bytes
    assert resident <= 1.2 * csr_bytes, resident / csr_bytes
    exec_.memset(d_y, 0xFF, 8 * N)
    A.mult(d_x, d_y)
    assert np.array_equal(exec_.copy_to_host(d_y, N), y_before)
    with pytest.raises(Exception):  # the CSR-order kernels would read freed memory
        A.plan_set("sjds", 0)
    assert not A.enable_mixed()  # the fp64 values it would convert are gone
    # cg() on the released matrix (fused dot, 3 launches per iteration)
    d_b, d_s = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", ctx, N, 0, N, d_b, None)
    k, hist = host.cg(comm, exec_, A, d_b, d_s, 5, 0.0)
    assert k == 5 and np.isfinite(hist).all()
    A.close()
    # ... and by itself under the context option
    _lib.call("spmv_hip_ctx_set_option", ctx, b"release_csr", 1)
    try:
        B = host.Matrix.create_fem_like(comm, exec_, N)
    finally:
        _lib.call("spmv_hip_ctx_set_option", ctx, b"release_csr", 0)
    assert B.release_csr() == 0  # already given back
    exec_.memset(d_y, 0xFF, 8 * N)
    B.mult(d_x, d_y)
    assert np.array_equal(exec_.copy_to_host(d_y, N), y_before)
    k2, hist2 = host.cg(comm, exec_, B, d_b, d_s, 5, 0.0)
    assert np.array_equal(hist2, hist)
    B.close()
    # symmetric storage of the same matrix (lower part + diagonal): the merged
    # form duplicates the values, the plan stands at three times the stored
    # block; released, 1.9 times of it stay (merged copy + row pointer +
    # diagonal) -- less than the general matrix's CSR arrays alone
    S = host.Matrix.create_fem_like(comm, exec_, N, symmetric=True)
    srows, scols, snnz = S.blocks()["local"]
    sym_csr = snnz * 12 + (srows + 1) * 4 + srows * 8
    assert S.plan_get("sym_sj") == 1 and S.plan_get("sj_long_rows") == 0
    before = S.plan_get("plan_kib") * 1024
    assert before > 2.5 * sym_csr
    S.mult(d_x, d_y)
    ys = exec_.copy_to_host(d_y, N)
    assert S.release_csr() == snnz * 12
    after = S.plan_get("plan_kib") * 1024
    assert sym_csr - snnz * 12 + after <= 1.95 * sym_csr
    assert sym_csr - snnz * 12 + after < csr_bytes
    exec_.memset(d_y, 0xFF, 8 * N)
    S.mult(d_x, d_y)
    assert np.array_equal(exec_.copy_to_host(d_y, N), ys)
    k3, hist3 = host.cg(comm, exec_, S, d_b, d_s, 5, 0.0)
    assert k3 == 5 and np.isfinite(hist3).all()
    S.close()
    for p in (d_x, d_y, d_b, d_s):
        exec_.free(p)


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            return next(int(ln.split()[1]) for ln in f
                        if ln.startswith("MemAvailable")) / 2 ** 20
    except (OSError, StopIteration):
        return 0.0


def test_spmv_512_cubed_against_the_oracle_itself(exec_, comm):
    """BASELINE configs[2] and [3] compared with the ORACLE at full size (not
    kernel against kernel): the 512^3 Poisson matrix built on the host by the
    oracle's generator, the reference's Gaussian x (demos/spmv.cpp:63-67);
    general storage against oracle.omp_spmv (csr_kernels.cpp:41-51: every row
    is summed left to right whatever the thread count), symmetric storage
    against the sequential oracle.csr_spmv_sym (csr_kernels.cpp:26-40) -- the
    default plans (constant diagonals) and, for general storage, the plan a
    matrix without lattice structure gets (LX) and the two kernels that stream
    the caller's CSR arrays as they are (XW: x windows staged; the gather
    kernel).  Bit for bit."""
    avail = _mem_available_gb()
    if avail < 40:
        pytest.skip(f"MemAvailable is {avail:.0f} GB: the host copy of the 512^3 "
                    "matrix (11.8 GB) + vectors + the D2H staging need 40 GB")
    from spmv_amd import _lib
    n = 512
    N = n ** 3
    ctx = exec_.context
    x = oracle.gaussian_x_fast(N)
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_x, x)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))

    def product(A):
        _lib.call("spmv_hip_fill_const_f64", ctx, N, float("nan"), d_y, None)
        A.mult(d_x, d_y)
        return exec_.copy_to_host(d_y, N)

    # ---- general storage
    rp, ci, va = oracle.poisson3d(n)
    assert len(va) == 7 * N - 6 * n * n
    y_ref = oracle.omp_spmv(rp, ci, va, x, num_threads=threads)
    del rp, ci, va
    assert np.isfinite(y_ref).all() and np.abs(y_ref).max() > 0
    A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_NONBLOCKING)
    assert A.plan_get("sdia") == 1 and A.plan_get("sdia_const") == 1
    assert np.array_equal(product(A), y_ref), "default plan (constant diagonals)"
    A.close()
    off = 1 << 62
    for name, opts, form in (
            ("lx", {b"lat_min_nnz": off}, dict(lat=0, lx=1, sdia=0)),
            ("rowblock", {b"lat_min_nnz": off, b"lx_min_nnz": off,
                          b"sj_min_nnz": off}, dict(lat=0, lx=0, sjds=0, sdia=0))):
        B = _with_ctx_options(exec_, opts, lambda: host.Matrix.create_poisson3d(
            comm, exec_, n, False, host.P2P_NONBLOCKING))
        for key, want in form.items():
            assert B.plan_get(key) == want, (name, key)
        assert np.array_equal(product(B), y_ref), name
        if name == "rowblock":  # that was the XW kernel; now the gather kernel
            assert B.plan_get("xw") == 1
            B.plan_set("xw", 0)
            assert np.array_equal(product(B), y_ref), "gather"
        B.close()
    del y_ref

    # ---- symmetric storage: strictly lower part + diagonal
    rp, ci, va, dg = oracle.poisson3d_lower(n)
    assert len(va) == 3 * N - 3 * n * n
    y_ref = oracle.csr_spmv_sym(rp, ci, va, dg, x)
    del rp, ci, va, dg
    A = host.Matrix.create_poisson3d(comm, exec_, n, True, host.P2P_NONBLOCKING)
    assert A.plan_get("sdia") == 1
    assert np.array_equal(product(A), y_ref), "symmetric storage, default plan"
    A.plan_set("sdia", 0)  # the symmetric lattice kernel on the same plan
    assert A.plan_get("slat") == 1
    assert np.array_equal(product(A), y_ref), "symmetric storage, lattice kernel"
    A.close()
    exec_.free(d_x), exec_.free(d_y)


def _with_ctx_options(exec_, opts, fn):
    """run fn() with context options set, restore the defaults afterwards"""
    from spmv_amd import _lib
    defaults = {b"lat_min_nnz": 1 << 20, b"lx_min_nnz": 1 << 20,
                b"sj_min_nnz": 1 << 20, b"xw_min_nnz": 1 << 20,
                b"xw_min_x_bytes": 128 << 20,
                b"poisson_stencil": 7, b"bake_general": 1}
    for k, v in opts.items():
        _lib.call("spmv_hip_ctx_set_option", exec_.context, k, v)
    try:
        return fn()
    finally:
        for k in opts:
            _lib.call("spmv_hip_ctx_set_option", exec_.context, k, defaults[k])


@pytest.mark.parametrize("form", ["lx", "sjds", "rowblock"])
def test_spmv_production_size_csr_order_kernels(exec_, comm, form):
    """BASELINE configs[1] SpMV leg on the kernels a matrix WITHOUT lattice
    structure gets (the default plans of the Poisson matrix are the lattice /
    diagonal forms): 128^3 and the north-star 216^3 with the reference's
    Gaussian x, every element identical to csr_kernels.cpp:41-51."""
    opts = {b"lat_min_nnz": 1 << 62}
    if form in ("rowblock", "sjds"):
        opts[b"lx_min_nnz"] = 1 << 62
    if form == "rowblock":
        opts[b"sj_min_nnz"] = 1 << 62
        opts[b"xw_min_x_bytes"] = 0  # (by default only where x outgrows the caches)
    for n in (128, 216):
        N = n ** 3
        rp, ci, va = oracle.poisson3d(n)
        x = oracle.gaussian_x_fast(N)
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        del rp, ci, va
        A = _with_ctx_options(exec_, opts, lambda: host.Matrix.create_poisson3d(
            comm, exec_, n, False, host.P2P_NONBLOCKING))
        assert A.plan_get("lat") == 0 and A.plan_get("sdia") == 0
        assert A.plan_get("lx") == (1 if form == "lx" else 0)
        assert A.plan_get("sjds") == (1 if form == "sjds" else 0)
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        exec_.memset(d_y, 0xFF, 8 * N)
        A.col_map().update(d_x)
        A.mult(d_x, d_y)
        assert np.array_equal(exec_.copy_to_host(d_y, N), y_ref), (form, n)
        # the caller's arrays as they are: the XW kernel (x windows staged) is
        # the plan's choice, the gather kernel the same plan's fallback
        assert A.plan_get("xw") == (1 if form == "rowblock" else 0)
        if form == "rowblock":
            A.plan_set("xw", 0)
            exec_.memset(d_y, 0xFF, 8 * N)
            A.mult(d_x, d_y)
            assert np.array_equal(exec_.copy_to_host(d_y, N), y_ref), ("gather", n)
        A.close()
        exec_.free(d_x), exec_.free(d_y)


def test_27_point_stencil_device_generator_and_kernels(exec_, comm):
    """The 27-point operator (context option poisson_stencil = 27; HPCG's
    matrix): the device generator equals the host twin, and every kernel form
    the plan offers for it returns the oracle's bits (n = 12, 33); at 160^3
    (4 M rows, 108 M entries) the forms agree with the one-lane-per-row
    kernel, the reference loop verbatim."""
    from spmv_amd import _lib
    for n in (12, 33):
        N = n ** 3
        rp, ci, va = poisson.stencil27_csr(n)
        ci = ci.astype(np.int32)
        x = oracle.gaussian_x_fast(N) + 0.25
        y_ref = oracle.csr_spmv(rp, ci, va, x)
        for opts in ({b"lat_min_nnz": 0, b"lx_min_nnz": 0},
                     {b"lat_min_nnz": 1 << 62, b"lx_min_nnz": 0},
                     {b"lat_min_nnz": 1 << 62, b"lx_min_nnz": 1 << 62}):
            o = dict(opts)
            o[b"poisson_stencil"] = 27
            A = _with_ctx_options(exec_, o, lambda: host.Matrix.create_poisson3d(
                comm, exec_, n, False, host.P2P_BLOCKING))
            rows, cols, nnz = A.blocks()["local"]
            assert (rows, cols, nnz) == (N, N, poisson.stencil27_nnz(n))
            d_x, d_y = exec_.alloc(N), exec_.alloc(N)
            exec_.copy_from_host(d_x, x)
            exec_.memset(d_y, 0xFF, 8 * N)
            A.mult(d_x, d_y)
            assert np.array_equal(exec_.copy_to_host(d_y, N), y_ref), (n, opts)
            if A.plan_get("wdia"):
                # mixed precision on the wide diagonal form: the values (26,
                # -1) are exact in fp32, so the fp32 copy gives the same bits
                assert A.enable_mixed() and A.plan_get("wdia_mixed") == 1
                A.use_mixed(True)
                exec_.memset(d_y, 0xFF, 8 * N)
                A.mult(d_x, d_y)
                A.use_mixed(False)
                assert np.array_equal(exec_.copy_to_host(d_y, N), y_ref), n
            A.close()
            exec_.free(d_x), exec_.free(d_y)
    n = 160
    N = n ** 3
    d_x, d_y = exec_.alloc(N), exec_.alloc(N)
    _lib.call("spmv_hip_fill_gaussian_f64", exec_.context, N, 0, N, d_x, None)
    y_scalar = None
    for name, opts in (("scalar", {b"lat_min_nnz": 1 << 62, b"lx_min_nnz": 1 << 62}),
                       ("rowblock", {b"lat_min_nnz": 1 << 62, b"lx_min_nnz": 1 << 62}),
                       ("lx", {b"lat_min_nnz": 1 << 62}),
                       ("default", {})):
        o = dict(opts)
        o[b"poisson_stencil"] = 27
        A = _with_ctx_options(exec_, o, lambda: host.Matrix.create_poisson3d(
            comm, exec_, n, False, host.P2P_BLOCKING))
        if name == "scalar":
            A.plan_set("algo", 3)
        exec_.memset(d_y, 0xFF, 8 * N)
        A.mult(d_x, d_y)
        y = exec_.copy_to_host(d_y, N)
        if y_scalar is None:
            y_scalar = y
            assert np.isfinite(y).all() and np.abs(y).max() > 0
        assert np.array_equal(y, y_scalar), name
        A.close()
    exec_.free(d_x), exec_.free(d_y)


def test_unstructured_matrix_all_general_kernels(exec_, comm):
    """The seeded unstructured matrix of the benchmark's sub-record (banded
    with 10 % far entries: no lattice, few stageable windows), generated on the
    device: the oracle-sized instance is bit-exact against the oracle run on
    the numpy twin, on every general form; at the benchmark's 10 M rows the
    forms agree with the one-lane-per-row kernel (reference loop verbatim)."""
    for N, check_oracle in ((300_000, True), (10_000_000, False)):
        x = oracle.gaussian_x_fast(N) + 0.25
        y_ref = None
        if check_oracle:
            rp, ci, va = poisson.unstructured_csr(N)
            y_ref = oracle.csr_spmv(rp, ci, va, x)
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        for name, opts in (("scalar", {b"lx_min_nnz": 1 << 62}),
                           ("rowblock", {b"lx_min_nnz": 1 << 62}),
                           ("default", {b"lx_min_nnz": 0})):
            A = _with_ctx_options(exec_, opts,
                                  lambda: host.Matrix.create_unstructured(
                                      comm, exec_, N))
            assert A.plan_get("lat") == 0 and A.plan_get("sdia") == 0
            if name == "scalar":
                A.plan_set("algo", 3)
            exec_.memset(d_y, 0xFF, 8 * N)
            A.mult(d_x, d_y)
            y = exec_.copy_to_host(d_y, N)
            if y_ref is None:
                y_ref = y  # the scalar kernel's
                assert np.isfinite(y).all() and np.abs(y).max() > 0
            assert np.array_equal(y, y_ref), (N, name)
            A.close()
        exec_.free(d_x), exec_.free(d_y)


@pytest.mark.parametrize("kind", list(FEM_KINDS))
def test_fem_like_matrix_all_general_kernels(exec_, comm, kind):
    """The seeded FEM-like matrices of the benchmark's ragged-row records
    (row lengths 5-40; the same with a 1 % tail of 200-2000-entry rows; 81
    entries in every row), generated on the device: the 300 k-row instance is
    bit-exact against the oracle run on the numpy twin (csr_kernels.cpp:41-51)
    on every general form -- the VECTOR kernel, whose lanes sum in another
    order, by SURVEY 8d's bound --; at the benchmark's 10 M rows (fem81: 3 M)
    the forms agree bit for bit with the one-lane-per-row kernel."""
    big = 3_000_000 if kind == "fem81" else 10_000_000
    for N, check_oracle in ((300_000, True), (big, False)):
        x = oracle.gaussian_x_fast(N) + 0.25
        y_ref = bound = None
        if check_oracle:
            rp, ci, va = poisson.fem_like_csr(N, **FEM_KINDS[kind])
            y_ref = oracle.csr_spmv(rp, ci, va, x)
            bound = (16 + np.diff(rp)) * U * abs_bound(rp, ci, va, x)
        d_x, d_y = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_x, x)
        for name in ("scalar", "rowblock", "vector", "default"):
            A = host.Matrix.create_fem_like(comm, exec_, N, **FEM_KINDS[kind])
            assert A.plan_get("lat") == 0 and A.plan_get("sdia") == 0
            # the default plan of these matrices is the sliced jagged form
            assert A.plan_get("sjds") == 1 and A.plan_get("lx") == 0
            if kind == "fem_tail":
                assert A.plan_get("sj_long_rows") > N // 200
            if name in ("scalar", "rowblock", "vector"):
                A.plan_set("sjds", 0)
                A.plan_set("algo", {"rowblock": 1, "vector": 2, "scalar": 3}[name])
            exec_.memset(d_y, 0xFF, 8 * N)
            A.mult(d_x, d_y)
            y = exec_.copy_to_host(d_y, N)
            if y_ref is None:
                y_ref = y  # the scalar kernel's
                assert np.isfinite(y).all() and np.abs(y).max() > 0
            vector = name == "vector"
            if vector and bound is not None:
                assert np.all(np.abs(y - y_ref) <= bound), (N, name)
            elif not vector:
                assert np.array_equal(y, y_ref), (N, kind, name)
            A.close()
        exec_.free(d_x), exec_.free(d_y)


# ---------------------------------------------------------------------------
# Many ranks as threads of this process, all on GPU 0 (tests/thread_world.py):
# the 8-way slab layout of BASELINE configs[4] and unstructured halos with up
# to 7 neighbours per rank, end to end through Matrix / L2GMap / cg
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("world,n", [(8, 16), (5, 12), (8, 8)])
def test_slab_ranks_threaded_spmv_and_cg(world, n):
    from thread_world import ThreadWorld
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    x = oracle.gaussian_x_fast(N)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    y_seq = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    ranges = oracle.owner_ranges(world, N)
    refs = {}
    for sym in (False, True):
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            refs[(sym, cm)] = (
                oracle.dist_spmv(world, rp, ci, va, x, sym, cm),
                oracle.dist_cg(world, rp, ci, va, b, 200, 1e-10, sym, cm))
    tw = ThreadWorld(world, timeout=45.0)

    def rank_body(rank, comm, exec_):
        from spmv_amd import _lib
        # small as they are, the local blocks take the forms of the benchmark:
        # lattice analysis, and on it the diagonal form wherever the block is
        # square (every block but the single ghost-column block of the
        # blocking general model)
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"lx_min_nnz", 0)
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"lat_min_nnz", 0)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        for (sym, cm), (y_ref, (x_ref, k_ref, hist_ref)) in refs.items():
            A = host.Matrix.create_poisson3d(comm, exec_, n, sym, cm)
            if r1 - r0 >= 2 * n * n:  # at least two planes: all three offsets
                assert A.plan_get("sdia") == int(sym or cm == host.P2P_NONBLOCKING)
            l2g = A.col_map()
            assert l2g.local_size() == r1 - r0
            assert len(l2g.plan().neighbours) == (1 if rank in (0, world - 1) else 2)
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            d_y = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_x, x[r0:r1])
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
            # both storages: the oracle's P-rank simulation, bit for bit
            assert np.array_equal(y, y_ref), (sym, cm)
            assert np.all(np.abs(y - y_seq) <= 16 * U * abs_bound(rp, ci, va, x))
            d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_b, b[r0:r1])
            k, hist = host.cg(comm, exec_, A, d_b, d_s, 200, 1e-10)
            xs = tw.gather(rank, exec_.copy_to_host(d_s, r1 - r0))
            assert abs(k - k_ref) <= 1 and k < 200, (k, k_ref)
            m = min(k, k_ref, 50)
            assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6)
            assert np.linalg.norm(xs - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
            A.close()
            for p in (d_x, d_y, d_b, d_s):
                exec_.free(p)

    tw.run(rank_body, gpu=True)


@pytest.mark.parametrize("n,parts", [(12, (2, 2, 2)), (10, (3, 2, 1)),
                                     (9, (1, 2, 4))])
def test_box_partition_ranks_threaded_spmv_and_cg(n, parts):
    """the Poisson matrix on a 3-D block partition
    (Matrix::create_poisson3d_boxes), up to 6 face neighbours and packed
    sends per rank; y equals the oracle's simulation of the same partition of
    the permuted global matrix bit for bit, CG follows its history."""
    from thread_world import ThreadWorld
    world = parts[0] * parts[1] * parts[2]
    N = n ** 3
    perm, ranges = box_partition(n, parts)
    rp, ci, va = permute_csr(*poisson.poisson3d_csr(n), perm)
    x = oracle.gaussian_x_fast(N)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    y_seq = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    refs = {}
    for sym in (False, True):
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            refs[(sym, cm)] = (
                oracle.dist_spmv(world, rp, ci, va, x, sym, cm, ranges),
                oracle.dist_cg(world, rp, ci, va, b, 200, 1e-10, sym, cm, ranges))
    tw = ThreadWorld(world, timeout=45.0)

    def rank_body(rank, comm, exec_):
        from spmv_amd import _lib
        # lattice analysis (and the diagonal form on it) for these small boxes
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"lat_min_nnz", 0)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        for (sym, cm), (y_ref, (x_ref, k_ref, hist_ref)) in refs.items():
            A = host.Matrix.create_poisson3d_boxes(comm, exec_, n, parts, sym, cm)
            assert A.plan_get("sdia") == int(sym or cm == host.P2P_NONBLOCKING)
            l2g = A.col_map()
            assert l2g.local_size() == r1 - r0 == A.rows()
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            d_y = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_x, x[r0:r1])
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
            assert np.array_equal(y, y_ref), (sym, cm)
            assert np.all(np.abs(y - y_seq) <= 16 * U * abs_bound(rp, ci, va, x))
            d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_b, b[r0:r1])
            k, hist = host.cg(comm, exec_, A, d_b, d_s, 200, 1e-10)
            xs = tw.gather(rank, exec_.copy_to_host(d_s, r1 - r0))
            assert abs(k - k_ref) <= 1 and k < 200, (k, k_ref)
            m = min(k, k_ref, 50)
            assert np.allclose(hist[:m + 1], hist_ref[:m + 1], rtol=1e-6)
            assert np.linalg.norm(xs - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
            assert np.abs(xs - 1.0).max() < 1e-6
            A.close()
            for p in (d_x, d_y, d_b, d_s):
                exec_.free(p)

    tw.run(rank_body, gpu=True)


@pytest.mark.parametrize("world,N,seed", [(8, 120, 21), (6, 75, 22)])
def test_unstructured_ranks_threaded(world, N, seed):
    """Every rank talks to (almost) every other rank: packed sends, up to 7
    neighbours, forward halo + mult + reverse_update."""
    from thread_world import ThreadWorld
    rng = np.random.default_rng(seed)
    dense = rng.random((N, N)) < 0.12
    dense = dense | dense.T | np.eye(N, dtype=bool)
    rp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    ci = np.nonzero(dense)[1].astype(np.int64)
    vals = rng.uniform(-1, 1, (N, N))
    va = ((vals + vals.T) / 2)[dense]
    x = rng.uniform(-1, 1, N)
    y_seq = oracle.csr_spmv(rp, ci.astype(np.int32), va, x)
    ranges = oracle.owner_ranges(world, N)
    sizes = np.diff(ranges)
    locs = [oracle.localise_rows(rp, ci, va, int(ranges[r]), int(ranges[r + 1]))
            for r in range(world)]
    plans = oracle.l2g_plans(sizes, [l[3] for l in locs])
    assert max(len(p["neighbours"]) for p in plans) == world - 1
    # some ranks must pack (scattered requests), the dense ones send directly
    tails = [rng.uniform(-1, 1, int(sizes[r]) + len(locs[r][3])) for r in range(world)]
    rev_ref = oracle.l2g_reverse_update(plans, [t.copy() for t in tails])
    tw = ThreadWorld(world, timeout=45.0)

    def rank_body(rank, comm, exec_):
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        lrp, lci, lva, ghosts = locs[rank]
        for sym in (False, True):
            for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
                A = host.Matrix.create_matrix(comm, exec_, lrp, lci, lva, r1 - r0,
                                              r1 - r0, [], ghosts, sym, cm)
                l2g = A.col_map()
                d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
                d_y = exec_.alloc(r1 - r0)
                exec_.copy_from_host(d_x, x[r0:r1])
                l2g.update(d_x)
                A.mult(d_x, d_y)
                exec_.synchronize()
                xs = exec_.copy_to_host(d_x, l2g.local_size() + l2g.num_ghosts())
                assert np.array_equal(xs[r1 - r0:], x[ghosts])
                y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
                if sym:
                    bound = (16 + np.diff(rp)) * U * abs_bound(rp, ci, va, x)
                    assert np.all(np.abs(y - y_seq) <= bound)
                else:
                    assert np.array_equal(
                        y, oracle.dist_spmv(world, rp, ci, va, x, False, cm))
                A.close()
                exec_.free(d_x), exec_.free(d_y)
        m = host.L2GMap(comm, int(sizes[rank]), ghosts, exec_, host.P2P_BLOCKING)
        d_v = exec_.alloc(len(tails[rank]))
        exec_.copy_from_host(d_v, tails[rank])
        m.reverse_update(d_v)
        exec_.synchronize()
        assert np.array_equal(exec_.copy_to_host(d_v, len(tails[rank])), rev_ref[rank])
        exec_.free(d_v)
        m.close()

    tw.run(rank_body, gpu=True)


def test_symmetric_fem_matrix_with_long_rows_on_three_ranks():
    """create_matrix(..., symmetric = true) on a FEM-like matrix with a tail of
    long rows, distributed over three ranks (threads): every rank's local block
    takes the merged sliced jagged form with its long rows on the long-row
    kernels, the remote block is general -- update + mult equal the oracle's
    three-rank simulation of Matrix.cpp:337-349 + csr_kernels.cpp:26-40 bit for
    bit, for the blocking and the overlapping model, and cg() runs on it."""
    import scipy.sparse as sp
    from thread_world import ThreadWorld
    from util import lower_split
    world, N = 3, 45_000
    rp0, ci0, va0 = poisson.fem_like_csr(N, jitter=64, layer=900, tail_permille=20,
                                         tail_min=150, tail_max=900, tail_stride=5)
    lrp, lci, lva, dg = lower_split(rp0, ci0, va0)
    L = sp.csr_matrix((lva, lci, lrp), shape=(N, N))
    # strictly diagonally dominant: SPD
    dg = np.abs(L).sum(1).A1 + np.abs(L).sum(0).A1 + 1.0
    S = (L + L.T + sp.diags(dg)).tocsr()
    S.sort_indices()
    rp, ci, va = S.indptr.astype(np.int32), S.indices.astype(np.int64), S.data
    x = np.random.default_rng(31).uniform(-1, 1, N)
    ranges = oracle.owner_ranges(world, N)
    locs = [oracle.localise_rows(rp, ci, va, int(ranges[r]), int(ranges[r + 1]))
            for r in range(world)]
    refs = {cm: oracle.dist_spmv(world, rp, ci, va, x, True, cm)
            for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING)}
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    x_ref, k_ref, _ = oracle.dist_cg(world, rp, ci, va, b, 60, 1e-10, True,
                                     host.P2P_NONBLOCKING)
    tw = ThreadWorld(world, timeout=60.0)
    seen = []

    def rank_body(rank, comm, exec_):
        from spmv_amd import _lib
        _lib.call("spmv_hip_ctx_set_option", exec_.context, b"sj_min_nnz", 0)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        lrp_, lci_, lva_, ghosts = locs[rank]
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            A = host.Matrix.create_matrix(comm, exec_, lrp_, lci_, lva_, r1 - r0,
                                          r1 - r0, [], ghosts, True, cm)
            assert A.plan_get("sym_sj") == 1 and A.plan_get("sj_long_rows") > 0
            l2g = A.col_map()
            d_x = exec_.alloc(l2g.local_size() + l2g.num_ghosts())
            d_y = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_x, x[r0:r1])
            exec_.memset(d_y, 0xFF, 8 * (r1 - r0))
            l2g.update(d_x)
            A.mult(d_x, d_y)
            exec_.synchronize()
            y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
            assert np.array_equal(y, refs[cm]), (rank, cm)
            if cm == host.P2P_NONBLOCKING:
                d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
                exec_.copy_from_host(d_b, b[r0:r1])
                k, hist = host.cg(comm, exec_, A, d_b, d_s, 60, 1e-10)
                sol = tw.gather(rank, exec_.copy_to_host(d_s, r1 - r0))
                assert abs(k - k_ref) <= 1
                assert np.linalg.norm(sol - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
                exec_.free(d_b), exec_.free(d_s)
            A.close()
            exec_.free(d_x), exec_.free(d_y)
        seen.append(rank)

    tw.run(rank_body, gpu=True)
    assert sorted(seen) == list(range(world))


@pytest.mark.parametrize("symmetric", [False, True])
def test_cg_consumer_reductions_equal_reducer_kernels(exec_, comm, symmetric):
    """CgOptions::consumer_reductions (the one-rank default: the update kernels
    add the partials themselves) leaves every scalar, the iteration count and
    the solution bit-identical to the reducer-kernel form; also with early
    convergence far inside a long queue and with an odd vector length."""
    for n, kmax, rtol in ((11, 60, 1e-9), (9, 400, 1e-6), (16, 25, 1e-30)):
        N = n ** 3
        rp, ci, va = poisson.poisson3d_csr(n)
        b = oracle.gaussian_x_fast(N)
        A = host.Matrix.create_matrix(comm, exec_, rp, ci, va, N, N, [], [],
                                      symmetric, host.P2P_NONBLOCKING)
        d_b, d_x = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_b, b)
        out = []
        for consume in (True, False, True):
            k, hist, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, kmax, rtol,
                                       history=True,
                                       consumer_reductions=consume)
            out.append((k, hist.copy(), exec_.copy_to_host(d_x, N)))
        for other in out[1:]:
            if symmetric:  # (a bar from the atomic kernels' days; the default
                # symmetric kernels are bit-exact and pass it trivially)
                assert abs(out[0][0] - other[0]) <= 1
                m = min(out[0][0], other[0], 40)
                assert np.allclose(out[0][1][:m + 1], other[1][:m + 1], rtol=1e-7)
            else:
                assert out[0][0] == other[0]
                assert np.array_equal(out[0][1], other[1])
                assert np.array_equal(out[0][2], other[2])
        A.close()
        exec_.free(d_b), exec_.free(d_x)


# ---------------------------------------------------------------------------
# spmv::cg at PRODUCTION launch shapes (BASELINE configs[1]: 128^3 on one
# MI355X, HIP SpMV + CG, correctness vs CPU; demos/cg.cpp:64-65 kmax = 100,
# rtol = 1e-10).  From 128^3 on the grids are the persistent ones of the
# benchmark (2,048 dot partials, consumer-side reductions, LX form); 256^3
# crosses blas1_nt_min_elems = 2^24, so every BLAS-1 kernel runs its
# non-temporal path.  Bars: SURVEY 8d (|dk| <= 1 or both = kmax, residual
# history to 1e-6 over the first 50 iterations, ||dx|| <= 1e-8 ||x||).
# CG parity is unpinned by the reference itself (no reference test calls cg,
# its ddot comes from an unpinned BLAS): the oracle fixes ddot left to right.
# ---------------------------------------------------------------------------
def _cg_vs_oracle(exec_, comm, n, symmetric, rhs, consume_modes, threads=1,
                  kmax=100, rtol=1e-10, envelope_threads=0):
    """envelope_threads > 0: the oracle is run a second time with its dots
    summed in another order (that many OpenMP partials); where the oracle
    disagrees with ITSELF by more than a twentieth of a bar, the bar becomes
    20x that self-deviation.  Needed for the Gaussian right-hand side only:
    its tails are 1e-28 of its peak, the Krylov space is exhausted after
    ~20 iterations and from there any rounding difference grows 10x per
    iteration (oracle vs oracle: 1e-11 at k = 18, 1e-3 at k = 27, x to 6e-6),
    so no implementation -- the reference with another BLAS included -- can
    meet 1e-6 / 1e-8 on it."""
    N = n ** 3
    rp, ci, va = oracle.poisson3d(n)
    if rhs == "A*ones":
        b = oracle.csr_spmv(rp, ci, va, np.ones(N))
    else:
        b = oracle.gaussian_x_fast(N)
    if symmetric:
        rp, ci, va, dg = oracle.poisson3d_lower(n)
    else:
        dg = None
    x_ref, k_ref, hist_ref = oracle.cg(rp, ci, va, b, kmax, rtol, diagonal=dg,
                                       num_threads=threads)
    hist_bar = np.full(kmax + 1, 1e-6)
    x_bar = 1e-8
    if envelope_threads:
        # the alternative order always runs on the general storage
        x_alt, k_alt, hist_alt = oracle.cg(*oracle.poisson3d(n), b, kmax, rtol,
                                           num_threads=envelope_threads)
        m = min(k_ref, k_alt)
        self_dev = np.maximum.accumulate(
            np.abs(hist_alt[:m + 1] / hist_ref[:m + 1] - 1))
        hist_bar[:m + 1] = np.maximum(hist_bar[:m + 1], 20 * self_dev)
        x_bar = max(x_bar, 20 * np.linalg.norm(x_alt - x_ref)
                    / np.linalg.norm(x_ref))
        del x_alt
    del rp, ci, va, dg
    A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                     host.P2P_NONBLOCKING)
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_b, b)
    ws = host.CgWorkspace(exec_)
    for consume in consume_modes:
        k, hist, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, kmax, rtol, ws,
                                   history=True, consumer_reductions=consume)
        x = exec_.copy_to_host(d_x, N)
        what = (n, symmetric, rhs, consume)
        assert (k == k_ref == kmax) or abs(k - k_ref) <= 1, (what, k, k_ref)
        if k < kmax:
            assert hist[-1] / hist[0] < rtol, what
        m = min(k, k_ref, 50)
        dev = np.abs(hist[:m + 1] / hist_ref[:m + 1] - 1)
        assert np.all(dev <= hist_bar[:m + 1]), (what, dev.max(),
                                                 int(np.argmax(dev / hist_bar[:m + 1])))
        err = np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref)
        assert err <= x_bar, (what, err, x_bar)
    ws.close()
    A.close()
    exec_.free(d_b), exec_.free(d_x)


def test_cg_rejects_x_overlapping_b(exec_, comm):
    """cg.h: x is the iterate from the first kernel on, so it must not share
    memory with b (the reference tolerates x == b, it writes x once at the
    end)."""
    n = 8
    N = n ** 3
    A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_NONBLOCKING)
    d = exec_.alloc(2 * N)
    exec_.copy_from_host(d, np.ones(2 * N))
    with pytest.raises(Exception, match="overlaps"):
        host.cg(comm, exec_, A, d, d, 5, 1e-10)
    with pytest.raises(Exception, match="overlaps"):
        host.cg(comm, exec_, A, d, d + 8 * (N - 1), 5, 1e-10)
    k, _ = host.cg(comm, exec_, A, d, d + 8 * N, 5, 1e-10)  # adjacent: fine
    assert k == 5
    A.close()
    exec_.free(d)


def test_cg_workspace_reused_with_smaller_kmax(exec_, comm):
    """A workspace sized by a long solve serves shorter ones: the residual
    history on the device keeps the workspace's capacity, so the host copy
    must too (this overflowed a host vector once)."""
    n = 64
    N = n ** 3
    rp, ci, va = oracle.poisson3d(n)
    b = oracle.csr_spmv(rp, ci, va, np.ones(N))
    A = host.Matrix.create_poisson3d(comm, exec_, n, False, host.P2P_NONBLOCKING)
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_b, b)
    ws = host.CgWorkspace(exec_)
    k1, h1, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, 300, 1e-12, ws, history=True)
    for kmax in (7, 1, 40):
        k2, h2, _, _ = host.cg_ex(comm, exec_, A, d_b, d_x, kmax, 1e-12, ws,
                                  history=True)
        assert k2 == min(kmax, k1) and np.array_equal(h2, h1[:k2 + 1]), kmax
    ws.close()
    A.close()
    exec_.free(d_b), exec_.free(d_x)


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("rhs", ["A*ones", "gaussian"])
def test_cg_128_cubed_vs_oracle(exec_, comm, symmetric, rhs):
    """BASELINE configs[1], CG leg: 128^3, kmax 100, rtol 1e-10, both
    reduction forms."""
    _cg_vs_oracle(exec_, comm, 128, symmetric, rhs, (True, False),
                  envelope_threads=0 if rhs == "A*ones" else 3)


@pytest.mark.parametrize("symmetric", [False, True])
def test_cg_256_cubed_nontemporal_blas1_vs_oracle(exec_, comm, symmetric):
    """256^3 = 2^24 rows: the non-temporal path of every BLAS-1 kernel and
    the persistent SpMV grid, against the oracle (OpenMP restatement for the
    general storage to keep the CPU side short; sequential for symmetric)."""
    threads = 1 if symmetric else max(1, min(16, len(os.sched_getaffinity(0))))
    _cg_vs_oracle(exec_, comm, 256, symmetric, "A*ones", (True, False),
                  threads=threads, kmax=60)


def test_spmv_128_cubed_gaussian_bit_exact_default_path(exec_, comm):
    """BASELINE configs[1], SpMV leg: y = A x at 128^3 with the reference's
    input vector (demos/spmv.cpp:63-67) on the DEFAULT plans (general storage:
    the diagonal form behind the lattice analysis; symmetric storage: the
    symmetric diagonal form), every element identical to csr_kernels.cpp:41-51
    / :26-40.  The CSR-order kernels a matrix WITHOUT lattice structure gets
    are compared at this size by test_spmv_production_size_csr_order_kernels."""
    n = 128
    N = n ** 3
    rp, ci, va = oracle.poisson3d(n)
    x = oracle.gaussian_x_fast(N)
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    rpl, cil, val, dg = oracle.poisson3d_lower(n)
    y_sym_ref = oracle.csr_spmv_sym(rpl, cil, val, dg, x)
    from spmv_amd import _lib
    for symmetric in (False, True):
        # const = 1: the default (constant diagonals); 0: the same plans with the
        # values streamed by offset -- what a lattice matrix with varying
        # coefficients gets (the half diagonal form)
        for const in (1, 0):
            _lib.call("spmv_hip_ctx_set_option", exec_.context,
                      b"const_diagonals", const)
            try:
                A = host.Matrix.create_poisson3d(comm, exec_, n, symmetric,
                                                 host.P2P_NONBLOCKING)
            finally:
                _lib.call("spmv_hip_ctx_set_option", exec_.context,
                          b"const_diagonals", 1)
            assert A.plan_get("sdia") == 1
            assert A.plan_get("sdia_const") == const
            d_x, d_y = exec_.alloc(N), exec_.alloc(N)
            exec_.copy_from_host(d_x, x)
            exec_.memset(d_y, 0xFF, 8 * N)
            A.col_map().update(d_x)
            A.mult(d_x, d_y)
            y = exec_.copy_to_host(d_y, N)
            assert np.array_equal(y, y_sym_ref if symmetric else y_ref), (
                symmetric, const)
            A.close()
            exec_.free(d_x), exec_.free(d_y)


@pytest.mark.parametrize("world", [2, 3])
def test_async_transport_observes_stream_ordering(world):
    """The halo over an ASYNCHRONOUS transport (tests/thread_world.py: copies
    only enqueued, delayed by filler kernels, ranks ordered by events alone),
    as RCCL delivers it.  The blocking transports of the other multi-rank
    tests drain the stream inside the exchange, so a missing
    stream_wait_event in L2GMap / Matrix / cg would pass them; here it leaves
    NaN-poisoned ghosts in the remote block's input or lets the x/p update
    overwrite a send buffer the neighbour has not read yet."""
    from thread_world import ThreadWorld
    n = 20
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    x = oracle.gaussian_x_fast(N)
    b = oracle.csr_spmv(rp, ci.astype(np.int32), va, np.ones(N))
    ranges = oracle.owner_ranges(world, N)
    refs = {}
    for sym in (False, True):
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            refs[(sym, cm)] = (
                oracle.dist_spmv(world, rp, ci, va, x, sym, cm),
                oracle.dist_cg(world, rp, ci, va, b, 25, 1e-30, sym, cm))
    tw = ThreadWorld(world, timeout=60.0)

    def rank_body(rank, comm, exec_):
        from spmv_amd import _lib
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        for (sym, cm), (y_ref, (x_ref, k_ref, hist_ref)) in refs.items():
            A = host.Matrix.create_poisson3d(comm, exec_, n, sym, cm)
            l2g = A.col_map()
            ng = l2g.num_ghosts()
            d_x = exec_.alloc(r1 - r0 + ng)
            d_y = exec_.alloc(r1 - r0)
            for rep in range(3):  # back to back: the queues stay full
                exec_.memset(d_x, 0xFF, 8 * (r1 - r0 + ng))  # NaN ghosts
                exec_.copy_from_host(d_x, x[r0:r1])
                l2g.update(d_x)
                A.mult(d_x, d_y)
            y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
            assert np.array_equal(y, y_ref), (sym, cm)
            # the other direction: the PRODUCER is late (filler kernels, then a
            # device-side copy brings x in) and the transport is prompt -- if
            # the exchange did not wait for the compute stream it would ship
            # the NaN pattern
            tw.bar.wait()
            if rank == 0:
                tw.delay_launches = 0
            tw.bar.wait()
            d_stage = exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_stage, x[r0:r1])
            filler = exec_.alloc(1 << 22)
            exec_.memset(d_x, 0xFF, 8 * (r1 - r0 + ng))
            exec_.memset(d_y, 0xFF, 8 * (r1 - r0))
            exec_.synchronize()
            for _ in range(40):
                _lib.call("spmv_hip_fill_const_f64", exec_.context, 1 << 22, 1.0,
                          filler, None)
            _lib.call("spmv_hip_copy_d2d_async", exec_.context, d_x, d_stage,
                      8 * (r1 - r0), None)
            l2g.update(d_x)
            A.mult(d_x, d_y)
            y = tw.gather(rank, exec_.copy_to_host(d_y, r1 - r0))
            assert np.array_equal(y, y_ref), ("late producer", sym, cm)
            exec_.free(d_stage), exec_.free(filler)
            tw.bar.wait()
            if rank == 0:
                tw.delay_launches = 40
            tw.bar.wait()
            d_b, d_s = exec_.alloc(r1 - r0), exec_.alloc(r1 - r0)
            exec_.copy_from_host(d_b, b[r0:r1])
            k, hist = host.cg(comm, exec_, A, d_b, d_s, 25, 1e-30)
            xs = tw.gather(rank, exec_.copy_to_host(d_s, r1 - r0))
            assert k == 25 == k_ref
            assert np.allclose(hist, hist_ref, rtol=1e-7), (sym, cm)
            assert np.linalg.norm(xs - x_ref) <= 1e-9 * np.linalg.norm(x_ref)
            A.close()
            for p in (d_x, d_y, d_b, d_s):
                exec_.free(p)

    tw.run(rank_body, gpu=True, asynchronous=True)


def test_cg_mixed_precision(exec_, comm):
    """CgOptions::mixed: fp32 matrix values in the SpMV, fp64
    everywhere else.  (a) The Poisson matrix is exactly representable in fp32:
    without replacements the iteration is the fp64 one bit for bit; with them it
    still converges to the same tolerance.  (b) A diagonally scaled Poisson
    matrix (irrational values, same sparsity, SPD): the run must end with the
    TRUE residual under rtol -- residual replacement, and the fp64 correction
    solve when the fp32 values cannot get there -- and land on the fp64
    solution."""
    from spmv_amd import _lib
    n = 24
    N = n ** 3
    rp, ci, va = poisson.poisson3d_csr(n)
    ci32 = ci.astype(np.int32)
    rng = np.random.default_rng(41)
    d = rng.uniform(0.5, 2.0, N)
    rows = np.repeat(np.arange(N), np.diff(rp))
    va_scaled = d[rows] * va * d[ci]
    for name, vals in (("poisson", va), ("scaled", va_scaled)):
        b = oracle.csr_spmv(rp, ci32, vals, np.ones(N))
        A = host.Matrix.create_matrix(comm, exec_, rp, ci, vals, N, N, [], [],
                                      False, host.P2P_NONBLOCKING)
        d_b, d_x = exec_.alloc(N), exec_.alloc(N)
        exec_.copy_from_host(d_b, b)
        k64, h64 = host.cg(comm, exec_, A, d_b, d_x, 2000, 1e-10)
        x64 = exec_.copy_to_host(d_x, N)
        assert k64 < 2000
        for every in (0, 7, 50):
            k, h, st = host.cg_mixed(comm, exec_, A, d_b, d_x, 2000, 1e-10,
                                     replace_every=every)
            x = exec_.copy_to_host(d_x, N)
            true_rel = (np.linalg.norm(b - oracle.csr_spmv(rp, ci32, vals, x))
                        / np.linalg.norm(b))
            what = (name, every, k, st)
            if name == "poisson" and every == 0:
                assert k == k64 and np.array_equal(h, h64), what
                assert np.array_equal(x, x64), what
                assert st["continuation_iterations"] == 0
            assert true_rel < 1.001e-10, (what, true_rel)
            assert abs(st["final_true_rel_residual"] - true_rel) <= 0.05 * 1e-10 + 0.5 * true_rel, what
            assert np.linalg.norm(x - x64) <= 1e-7 * np.linalg.norm(x64), what
            assert st["replacements"] == (0 if every == 0 else k_loop(k, st) // every), what
            # the mixed switch was left off: a plain solve is the fp64 one again
        k2, h2 = host.cg(comm, exec_, A, d_b, d_x, 2000, 1e-10)
        assert k2 == k64 and np.array_equal(h2, h64)
        A.close()
        exec_.free(d_b), exec_.free(d_x)


def test_cg_mixed_precision_on_long_rows(exec_, comm):
    """CgOptions::mixed on a general matrix with more than 64 entries per row
    (the plan's VECTOR kernel): runs -- it threw ENOTSUP out of the first
    mult_dot once -- and meets rtol on the true residual."""
    N, half = 6000, 45
    offs = [d for d in range(-half, half + 1)]
    rows = np.repeat(np.arange(N), len(offs))
    cols = rows + np.tile(offs, N)
    ok = (cols >= 0) & (cols < N)
    rows, cols = rows[ok], cols[ok]
    d = np.abs(cols - rows)
    vals = np.where(d == 0, 12.0, -1.0 / np.maximum(d, 1)) * (1 + 0.25 * np.cos(rows + cols))
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int64)
    ci32 = cols.astype(np.int32)
    assert len(vals) / N > 64
    b = oracle.csr_spmv(rp.astype(np.int32), ci32, vals, np.ones(N))
    A = host.Matrix.create_matrix(comm, exec_, rp, cols.astype(np.int64), vals, N,
                                  N, [], [], False, host.P2P_NONBLOCKING)
    d_b, d_x = exec_.alloc(N), exec_.alloc(N)
    exec_.copy_from_host(d_b, b)
    k64, _ = host.cg(comm, exec_, A, d_b, d_x, 500, 1e-10)
    x64 = exec_.copy_to_host(d_x, N)
    k, h, st = host.cg_mixed(comm, exec_, A, d_b, d_x, 500, 1e-10,
                             replace_every=10)
    x = exec_.copy_to_host(d_x, N)
    true_rel = (np.linalg.norm(b - oracle.csr_spmv(rp.astype(np.int32), ci32,
                                                   vals, x)) / np.linalg.norm(b))
    assert k64 < 500 and k < 500, (k64, k, st)
    assert true_rel < 1.001e-10, (true_rel, st)
    assert np.linalg.norm(x - x64) <= 1e-7 * np.linalg.norm(x64)
    A.close()
    exec_.free(d_b), exec_.free(d_x)


def k_loop(k, st):
    """iterations of the mixed loop itself (the correction solve excluded)"""
    return k - st["continuation_iterations"]
