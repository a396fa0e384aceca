"""CPU tests (no GPU) of the oracle, the C ABI surface and the C++ host
logic, including the world_size>1 path over gloo."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
from spmv_amd import _lib, host, poisson
from util import (U, abs_bound, box_partition, lower_split, permute_csr,
                  random_csr)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
GOLDEN = os.path.join(ROOT, "tests", "golden")


def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------------------
# the oracle against the reference's golden vectors
# ---------------------------------------------------------------------------
def test_oracle_reproduces_kat():
    k = kat()
    x = oracle.gaussian_x(5)
    assert list(x) == k["x"]
    va = np.array(k["values"])
    y = oracle.csr_spmv(k["rowptr"], k["colind"], va, x)
    assert list(y) == k["y"]
    norm = float(np.sqrt(np.sum(y * y)))
    assert norm == k["norm_y"]
    # symmetric branch and OpenMP path agree bit for bit on the KAT (the
    # survey stage observed the same from the compiled reference)
    lrp, lci, lva, dg = lower_split(np.array(k["rowptr"]),
                                    np.array(k["colind"]), va)
    assert list(oracle.csr_spmv_sym(lrp, lci, lva, dg, x)) == k["y"]
    for nt in (1, 2):
        assert list(oracle.omp_spmv(k["rowptr"], k["colind"], va, x,
                                    num_threads=nt)) == k["y"]
        assert list(oracle.omp_spmv(lrp, lci, lva, x, diagonal=dg,
                                    num_threads=nt)) == k["y"]


@pytest.mark.parametrize("P", [1, 2, 3, 5])
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("cm", [0, 1, 2, 3])
def test_oracle_distributed_kat_norm(P, symmetric, cm):
    """The reference's own pass criterion (tests/test_spmv.cpp:20-23,159-160)
    at 1..5 simulated ranks, every supported communication model."""
    k = kat()
    y = oracle.dist_spmv(P, np.array(k["rowptr"]), np.array(k["colind"]),
                         np.array(k["values"]), np.array(k["x"]), symmetric, cm)
    a, b = float(np.sqrt(np.sum(y * y))), k["norm_y"]
    assert abs(a - b) <= min(abs(a), abs(b)) * np.finfo(float).eps


def test_oracle_halo_chain_fixture():
    """3-rank chain recorded from the compiled reference."""
    h = kat()["halo_chain"]
    P, n = h["ranks"], h["rows_per_rank"]
    plans = oracle.l2g_plans([n] * P, h["ghosts"])
    vecs = [np.concatenate([100.0 * r + np.arange(n), np.zeros(len(h["ghosts"][r]))])
            for r in range(P)]
    oracle.l2g_update(plans, vecs)
    for r in range(P):
        assert list(vecs[r][n:]) == h["ghost_tails"][r]


def test_oracle_poisson_matches_independent_generator():
    for n in (2, 3, 5, 8):
        rp, ci, va = poisson.poisson3d_csr(n)
        rp2, ci2, va2 = oracle.poisson3d_csr(n)  # scipy kron
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
        assert np.array_equal(va, va2)
        assert len(va) == poisson.poisson3d_nnz(n)


def test_oracle_variants_agree():
    rng = np.random.default_rng(1)
    rp, ci, va = random_csr(rng, 500, 500, 7)
    x = rng.uniform(-1, 1, 500)
    y = oracle.csr_spmv(rp, ci, va, x, 1.5, 0.25, np.ones(500))
    for nt in (1, 2, 3, 8):
        assert np.array_equal(y, oracle.omp_spmv(rp, ci, va, x, alpha=1.5,
                                                 beta=0.25, y=np.ones(500),
                                                 num_threads=nt))
        split = oracle.omp_row_split(rp, nt)
        assert split[0] == 0 and split[-1] == 500 and np.all(np.diff(split) >= 0)
    n = 7
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    x = oracle.gaussian_x_fast(n ** 3)
    y = oracle.csr_spmv(rp, ci, va, x)
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    bound = 16 * U * abs_bound(rp, ci, va, x)
    assert np.all(np.abs(oracle.csr_spmv_sym(lrp, lci, lva, dg, x) - y) <= bound)
    for nt in (2, 4):
        ys = oracle.omp_spmv(lrp, lci, lva, x, diagonal=dg, num_threads=nt)
        assert np.all(np.abs(ys - y) <= bound)
    for P in (2, 3, 4):
        for sym in (False, True):
            for cm in (0, 1):
                yd = oracle.dist_spmv(P, rp, ci, va, x, sym, cm)
                assert np.all(np.abs(yd - y) <= bound)


def test_oracle_cg():
    n = 8
    rp, ci, va = poisson.poisson3d_csr(n)
    ci = ci.astype(np.int32)
    b = oracle.csr_spmv(rp, ci, va, np.ones(n ** 3))
    x, k, hist = oracle.cg(rp, ci, va, b, 100, 1e-10)
    assert k < 100 and hist[-1] / hist[0] < 1e-10 and len(hist) == k + 1
    assert np.linalg.norm(x - 1) < 1e-8 * n ** 1.5
    lrp, lci, lva, dg = lower_split(rp, ci, va)
    xs, ks, _ = oracle.cg(lrp, lci, lva, b, 100, 1e-10, diagonal=dg)
    assert abs(ks - k) <= 1 and np.linalg.norm(xs - x) < 1e-9 * np.linalg.norm(x)
    xo, ko, _ = oracle.cg(rp, ci, va, b, 100, 1e-10, num_threads=3)
    assert abs(ko - k) <= 1 and np.linalg.norm(xo - x) < 1e-9 * np.linalg.norm(x)
    for P in (2, 3):
        xd, kd, hd = oracle.dist_cg(P, rp, ci, va, b, 100, 1e-10, False, 1)
        assert abs(kd - k) <= 1
        assert np.linalg.norm(xd - x) < 1e-9 * np.linalg.norm(x)
    # kmax reached
    _, k3, h3 = oracle.cg(rp, ci, va, b, 3, 1e-30)
    assert k3 == 3 and len(h3) == 4


# ---------------------------------------------------------------------------
# the C ABI surface (no compute: there is no GPU here)
# ---------------------------------------------------------------------------
def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    for header, prefix, lib in (("spmv_hip.h", "spmv_hip_", _lib.hip),
                                ("spmv_host_c.h", "spmvh_", host.lib)):
        names = [n for n in _declared(header, prefix)
                 if not n.endswith("_fn")]
        assert len(names) > 20
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {header} but not exported"
    # and the Python prototypes cover exactly the declared ABI
    assert sorted(_lib.HIP_SYMBOLS) == _declared("spmv_hip.h", "spmv_hip_")
    assert _lib.hip.spmv_hip_abi_version() == 5


def test_error_strings_and_loud_failure_without_gpu():
    assert _lib.error_string(0) == "success"
    assert "invalid argument" in _lib.error_string(-1)
    assert "RCCL" in _lib.error_string(10003)
    import ctypes as C
    n = C.c_int(-1)
    rc = _lib.hip.spmv_hip_device_count(C.byref(n))
    if rc != 0 or n.value == 0:
        # no device: creating an executor must fail with the HIP error text,
        # never fall back to a CPU path
        with pytest.raises(host.SpmvHostError):
            host.HipExecutor(0)
    # NULL handles are rejected before anything is launched
    assert _lib.hip.spmv_hip_synchronize(None) == -1
    assert _lib.hip.spmv_hip_csr_spmv_f64(None, None, 1, 1, 0, None, None, None,
                                          None, 1.0, None, 0.0, None, None,
                                          None) == -1


def test_host_executor_has_no_compute_path():
    assert host.host_executor_rejects_compute()
    assert b"no CPU compute path" in host.lib.spmvh_last_error()


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "spmv_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(base, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt
                assert "spmv_oracle" not in txt, f


# ---------------------------------------------------------------------------
# C++ host logic on one rank (SelfComm), no device
# ---------------------------------------------------------------------------
def test_split_rows_matches_oracle_all_ranks():
    rng = np.random.default_rng(3)
    mats = [poisson.poisson3d_csr(5)]
    N = 41
    dense = (rng.random((N, N)) < 0.15) | np.eye(N, dtype=bool)
    rp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    ci = np.nonzero(dense)[1].astype(np.int64)
    mats.append((rp, ci, rng.uniform(-1, 1, len(ci))))
    for rp, ci, va in mats:
        N = len(rp) - 1
        for P in (1, 2, 4):
            ranges = oracle.owner_ranges(P, N)
            for r in range(P):
                lrp, lci, lva, gh = oracle.localise_rows(rp, ci, va,
                                                         int(ranges[r]),
                                                         int(ranges[r + 1]))
                # shuffled ghost order on input: must be renumbered ascending
                perm = rng.permutation(len(gh))
                inv = np.argsort(perm)
                nloc = int(ranges[r + 1] - ranges[r])
                lci2 = lci.copy()
                g = lci >= nloc
                lci2[g] = nloc + inv[lci[g] - nloc]
                for sym in (False, True):
                    for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
                        A = oracle.create_matrix(r, ranges, ranges, lrp, lci,
                                                 lva, gh, sym, cm)
                        s = host.split_rows(lrp, lci2, lva, nloc, nloc,
                                            ranges[r], ranges[r], gh[perm], sym,
                                            cm)
                        assert s["nnz"] == A["nnz"]
                        assert np.array_equal(s["ghosts"], A["ghosts"])
                        for name in ("local", "remote"):
                            if A[name] is None:
                                assert len(s[name][2]) == 0
                            else:
                                for a, b in zip(s[name], A[name]):
                                    assert np.array_equal(a, b)
                        if sym:
                            assert np.array_equal(s["diagonal"], A["diagonal"])


def test_l2gmap_single_rank_and_errors():
    comm = host.Comm.self_comm()
    m = host.L2GMap(comm, 10, [], None)
    p = m.plan()
    assert len(p.neighbours) == 0 and len(p.indexbuf) == 0
    m.close()
    with pytest.raises(host.SpmvHostError, match="Ghosts must be sorted"):
        host.L2GMap(comm, 10, [12, 11], None)
    with pytest.raises(host.SpmvHostError, match="Ghost index in local range"):
        host.L2GMap(comm, 10, [3], None)
    for cm in (host.ONESIDED_PUT_ACTIVE, host.SHMEM, host.SHMEM_NODUP):
        m = host.L2GMap(comm, 10, [], None, cm)  # accepted, blocking semantics
        m.close()
    comm.close()


# ---------------------------------------------------------------------------
# world_size > 1 on CPU: one process per rank over gloo
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3])
def test_l2g_plan_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mp_plan_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("plan + split OK") == world


def test_ranks_as_threads_share_a_process():
    """Comm::ranks_share_a_process: threads of this process do (the library then
    refuses the peer reduction beside a one-sided halo), a single rank does not;
    the gloo workers above assert the process case."""
    from thread_world import ThreadWorld
    seen = [None] * 3

    def rank_body(rank, comm):
        seen[rank] = comm.ranks_share_a_process()
    ThreadWorld(3).run(rank_body)
    assert seen == [True, True, True]
    c = host.Comm.self_comm()
    assert c.ranks_share_a_process() is False
    c.close()


# ---------------------------------------------------------------------------
# PETSc binary ingest (spmv/read_petsc.cpp), host-only parse
# ---------------------------------------------------------------------------
def test_petsc_reader_matches_restatement(tmp_path):
    rng = np.random.default_rng(17)
    cases = []
    rp, ci, va = poisson.poisson3d_csr(4)
    cases.append((rp, ci, va, 64))
    rp, ci, va = random_csr(rng, 37, 53, 4)      # rectangular, empty rows
    cases.append((rp, ci, va, 53))
    for idx, (rp, ci, va, ncols) in enumerate(cases):
        f = tmp_path / f"m{idx}.dat"
        oracle.petsc_io.write_matrix(f, rp, ci, va, ncols)
        for size in (1, 2, 3, 5):
            for rank in range(size):
                got = host.read_petsc_binary_rows(f, rank, size)
                exp = oracle.petsc_io.read_matrix_rows(f, rank, size)
                for key in ("nrows", "ncols", "nnz", "row_begin", "row_end"):
                    assert got[key] == exp[key]
                for key in ("rowptr", "colind", "values", "col_ghosts"):
                    assert np.array_equal(got[key], exp[key]), key
            # the slices reassemble the matrix that was written
            rows = [host.read_petsc_binary_rows(f, r, size) for r in range(size)]
            assert sum(len(r["values"]) for r in rows) == len(va)
            assert np.array_equal(np.concatenate([r["values"] for r in rows]), va)
    bad = tmp_path / "bad.dat"
    bad.write_bytes(b"\x00" * 64)
    with pytest.raises(host.SpmvHostError, match="Bad signature"):
        host.read_petsc_binary_rows(bad, 0, 1)
    with pytest.raises(host.SpmvHostError, match="Could not open"):
        host.read_petsc_binary_rows(tmp_path / "missing.dat", 0, 1)
    trunc = tmp_path / "trunc.dat"
    trunc.write_bytes((tmp_path / "m0.dat").read_bytes()[:200])
    with pytest.raises(host.SpmvHostError, match="truncated"):
        host.read_petsc_binary_rows(trunc, 0, 1)


def test_tridiag_and_splitmix_generators():
    rp, ci, va = oracle.tridiag_csr(7)
    assert rp.tolist() == [0, 2, 5, 8, 11, 14, 17, 19]
    assert ci.tolist() == [0, 1, 0, 1, 2, 1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6]
    assert va[0] == 1.0 - 0.1 and va[1] == 0.1 and va[3] == 1.0 - 2.0 * 0.1
    assert va[-1] == 1.0 - 0.1 and va[-2] == 0.1
    # every row sums to one (demos/CreateA.cpp:41-58), so A.1 = 1 to rounding
    y = oracle.csr_spmv(rp, ci, va, np.ones(7))
    assert np.allclose(y, 1.0, rtol=0, atol=2e-16)
    # splitmix64 known answer: seed 0 -> first output 0xE220A8397B1DCDAF
    u = oracle.splitmix64_unit(3, seed=0)
    assert u[0] == (0xE220A8397B1DCDAF >> 11) * 2.0 ** -52 - 1.0
    u = oracle.splitmix64_unit(100000)
    assert u.min() >= -1.0 and u.max() < 1.0 and abs(u.mean()) < 0.01


def test_oracle_reverse_update_counts_and_conserves():
    """L2GMap.cpp:907-950: with zeros in the owned part and ones in every
    ghost tail, an owner ends up with the number of ranks ghosting each entry;
    in general the grand total of owned entries grows by the sum of all tails."""
    sizes = [6, 4, 7]
    ghosts = [np.array([6, 7, 12]), np.array([0, 5, 12, 16]), np.array([5, 7])]
    plans = oracle.l2g_plans(sizes, ghosts)
    vecs = [np.concatenate([np.zeros(n), np.ones(len(g))])
            for n, g in zip(sizes, ghosts)]
    out = oracle.l2g_reverse_update(plans, vecs)
    owned = np.concatenate([v[:n] for v, n in zip(out, sizes)])
    want = np.zeros(17)
    for g in ghosts:
        want[g] += 1
    assert np.array_equal(owned, want)
    assert all(np.all(v[n:] == 1.0) for v, n in zip(out, sizes))
    rng = np.random.default_rng(5)
    vecs = [rng.integers(-8, 8, n + len(g)).astype(float)
            for n, g in zip(sizes, ghosts)]
    before = sum(v[:n].sum() for v, n in zip(vecs, sizes))
    tails = sum(v[n:].sum() for v, n in zip(vecs, sizes))
    out = oracle.l2g_reverse_update(plans, [v.copy() for v in vecs])
    assert sum(v[:n].sum() for v, n in zip(out, sizes)) == before + tails
    # adjoint of the forward halo: <update(x), y> == <x, reverse(y)> on integers
    xs = [v.copy() for v in vecs]
    ys = [rng.integers(-8, 8, len(v)).astype(float) for v in vecs]
    fwd = oracle.l2g_update(plans, [v.copy() for v in xs])
    rev = oracle.l2g_reverse_update(plans, [v.copy() for v in ys])
    lhs = sum(np.dot(f, y) for f, y in zip(fwd, ys))
    # forward overwrites ghost tails, so compare against x with zeroed tails
    rhs = sum(np.dot(x[:n], r[:n]) for x, r, n in zip(xs, rev, sizes))
    assert lhs == rhs


# ---------------------------------------------------------------------------
# many ranks in one process (threads): random unstructured halos with up to
# P-1 neighbours per rank, uneven and EMPTY row ranges
# ---------------------------------------------------------------------------
def _random_global_csr(rng, N, density, symmetric_pattern):
    dense = rng.random((N, N)) < density
    if symmetric_pattern:
        dense = dense | dense.T
    dense |= np.eye(N, dtype=bool)
    rp = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    ci = np.nonzero(dense)[1].astype(np.int64)
    vals = rng.uniform(-1, 1, (N, N))
    vals = (vals + vals.T) / 2
    return rp, ci, vals[dense]


@pytest.mark.parametrize("world,N,density,seed", [
    (4, 41, 0.15, 0), (5, 64, 0.05, 1), (8, 97, 0.08, 2), (8, 40, 0.3, 3),
    (6, 9, 0.4, 4),   # fewer than two rows on most ranks
    (7, 5, 0.5, 5),   # some ranks own NO rows
])
def test_plan_and_split_many_ranks_threaded(world, N, density, seed):
    from thread_world import ThreadWorld
    rng = np.random.default_rng(seed)
    rp, ci, va = _random_global_csr(rng, N, density, symmetric_pattern=True)
    ranges = oracle.owner_ranges(world, N)
    sizes = np.diff(ranges)
    locs = [oracle.localise_rows(rp, ci, va, int(ranges[r]), int(ranges[r + 1]))
            for r in range(world)]
    plans = oracle.l2g_plans(sizes, [l[3] for l in locs])
    assert max(len(p["neighbours"]) for p in plans) >= min(3, world - 1) or N < 10

    def rank_body(rank, comm):
        lrp, lci, lva, ghosts = locs[rank]
        nloc = int(sizes[rank])
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING,
                   host.COLLECTIVE_BLOCKING):
            m = host.L2GMap(comm, nloc, ghosts, None, cm)
            got, exp = m.plan(), plans[rank]
            nn = len(exp["neighbours"])
            assert np.array_equal(got.neighbours, exp["neighbours"])
            assert np.array_equal(got.send_count, exp["send_count"][:nn])
            assert np.array_equal(got.recv_count, exp["recv_count"][:nn])
            assert np.array_equal(got.send_offset, exp["send_offset"][:nn + 1])
            assert np.array_equal(got.recv_offset, exp["recv_offset"][:nn + 1])
            assert np.array_equal(got.indexbuf, exp["indexbuf"])
            m.close()
        for sym in (False, True):
            for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
                A = oracle.create_matrix(rank, ranges, ranges, lrp, lci, lva,
                                         ghosts, sym, cm)
                s = host.split_rows(lrp, lci, lva, nloc, nloc, ranges[rank],
                                    ranges[rank], ghosts, sym, cm)
                assert s["nnz"] == A["nnz"]
                assert np.array_equal(s["ghosts"], A["ghosts"])
                for name in ("local", "remote"):
                    if A[name] is None:
                        assert len(s[name][2]) == 0
                        continue
                    for a, b in zip(s[name], A[name]):
                        assert np.array_equal(a, b), (name, sym, cm)
                if sym:
                    assert np.array_equal(s["diagonal"], A["diagonal"])

    ThreadWorld(world).run(rank_body)


@pytest.mark.parametrize("n,parts", [(8, (2, 2, 2)), (7, (3, 2, 2)), (6, (1, 2, 3))])
def test_box_partition_plan_many_ranks_threaded(n, parts):
    """The halo plan of the 3-D block partition: up to 6 face neighbours per
    rank, scattered (packed) sends."""
    from thread_world import ThreadWorld
    world = parts[0] * parts[1] * parts[2]
    perm, ranges = box_partition(n, parts)
    brp, bci, bva = permute_csr(*poisson.poisson3d_csr(n), perm)
    locs = [oracle.localise_rows(brp, bci, bva, int(ranges[r]), int(ranges[r + 1]))
            for r in range(world)]
    plans = oracle.l2g_plans(np.diff(ranges), [l[3] for l in locs])
    if parts == (2, 2, 2):
        assert all(len(p["neighbours"]) == 3 for p in plans)

    def rank_body(rank, comm):
        lrp, lci, lva, ghosts, off, box = host.poisson3d_box_rows(n, parts, rank)
        nloc = int(ranges[rank + 1] - ranges[rank])
        for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING):
            m = host.L2GMap(comm, nloc, ghosts, None, cm)
            got, exp = m.plan(), plans[rank]
            nn = len(exp["neighbours"])
            assert np.array_equal(got.neighbours, exp["neighbours"])
            assert np.array_equal(got.send_count, exp["send_count"][:nn])
            assert np.array_equal(got.recv_count, exp["recv_count"][:nn])
            assert np.array_equal(got.send_offset, exp["send_offset"][:nn + 1])
            assert np.array_equal(got.recv_offset, exp["recv_offset"][:nn + 1])
            assert np.array_equal(got.indexbuf, exp["indexbuf"])
            m.close()

    ThreadWorld(world).run(rank_body)


@pytest.mark.parametrize("world,N,seed,sym", [(4, 37, 10, False), (5, 53, 11, True),
                                              (8, 71, 12, False), (8, 90, 13, True)])
def test_row_ghost_assembly_many_ranks_threaded(world, N, seed, sym):
    """Matrix.cpp:188-292 with up to 7 ranks contributing pieces of a row."""
    from thread_world import ThreadWorld
    from util import assembled_inputs
    rng = np.random.default_rng(seed)
    _, ranges, inputs = assembled_inputs(rng, world, N, symmetric=sym)
    expected = {cm: oracle.create_matrices_with_row_ghosts(ranges, inputs, sym, cm)
                for cm in (host.P2P_BLOCKING, host.P2P_NONBLOCKING)}

    def rank_body(rank, comm):
        rp, ci, va, rg, cg = inputs[rank]
        nloc = int(ranges[rank + 1] - ranges[rank])
        for cm, exp_all in expected.items():
            exp = exp_all[rank]
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

    ThreadWorld(world).run(rank_body)


def test_plan_of_the_real_eight_way_512_cubed_slabs():
    """BASELINE configs[4]: the 512^3 matrix row-partitioned over 8 ranks.  The
    halo plan the host mirror builds for THAT partition (plan only, no matrix):
    16,777,216 local rows per rank, ghost planes of 262,144 entries (one on the
    edge ranks, two inside), indices up to 2^27 held exactly, every send list
    one contiguous run (the direct-send case: no pack kernel) -- array by
    array against oracle.l2g_plans (L2GMap.cpp:346-479)."""
    from thread_world import ThreadWorld
    n, world = 512, 8
    N, plane = n ** 3, n * n
    ranges = oracle.owner_ranges(world, N)
    sizes = np.diff(ranges)
    assert list(sizes) == [N // world] * world
    ghosts = []
    for r in range(world):
        lo, hi = int(ranges[r]), int(ranges[r + 1])
        g = []
        if r > 0:
            g.append(np.arange(lo - plane, lo, dtype=np.int64))
        if r < world - 1:
            g.append(np.arange(hi, hi + plane, dtype=np.int64))
        ghosts.append(np.concatenate(g))
    plans = oracle.l2g_plans(sizes, ghosts)

    def rank_body(rank, comm):
        for cm in (host.P2P_NONBLOCKING, host.P2P_BLOCKING):
            m = host.L2GMap(comm, int(sizes[rank]), ghosts[rank], None, cm)
            got, exp = m.plan(), plans[rank]
            nn = 1 if rank in (0, world - 1) else 2
            assert len(exp["neighbours"]) == nn
            assert np.array_equal(got.neighbours, exp["neighbours"])
            assert np.array_equal(got.send_count, exp["send_count"][:nn])
            assert np.array_equal(got.recv_count, exp["recv_count"][:nn])
            assert list(got.send_count) == [plane] * nn == list(got.recv_count)
            assert np.array_equal(got.send_offset, exp["send_offset"][:nn + 1])
            assert np.array_equal(got.recv_offset, exp["recv_offset"][:nn + 1])
            assert np.array_equal(got.indexbuf, exp["indexbuf"])
            assert got.indexbuf.dtype == np.int32 and len(got.indexbuf) == nn * plane
            # what goes to the lower neighbour is my first plane, to the upper
            # one my last plane: contiguous runs => sent straight from the vector
            runs = np.split(got.indexbuf, nn)
            for run in runs:
                assert np.array_equal(run, np.arange(run[0], run[0] + plane))
            assert got.packs is False
            m.close()

    ThreadWorld(world, timeout=120.0).run(rank_body)


@pytest.mark.parametrize("n,parts", [(6, (2, 2, 2)), (7, (3, 2, 1)),
                                     (5, (1, 1, 4)), (4, (4, 1, 1)),
                                     (9, (2, 3, 2)), (3, (1, 1, 1))])
def test_box_partition_rows_match_the_permuted_matrix(n, parts):
    """Matrix::poisson3d_box_rows (SURVEY 8f n4, 3-D block partition): every
    rank's rows, ghosts and numbering equal the slice the reference's test
    harness (tests/test_spmv.cpp:83-124, oracle.localise_rows) takes of the
    permuted global matrix, array for array."""
    rp, ci, va = poisson.poisson3d_csr(n)
    perm, ranges = box_partition(n, parts)
    assert np.array_equal(np.sort(perm), np.arange(n ** 3))
    brp, bci, bva = permute_csr(rp, ci, va, perm)
    P = parts[0] * parts[1] * parts[2]
    nnz = 0
    for rank in range(P):
        lrp, lci, lva, ghosts, off, box = host.poisson3d_box_rows(n, parts, rank)
        r0, r1 = int(ranges[rank]), int(ranges[rank + 1])
        assert off == r0 and box[0] * box[1] * box[2] == r1 - r0
        erp, eci, eva, eghosts = oracle.localise_rows(brp, bci, bva, r0, r1)
        assert np.array_equal(lrp, erp) and np.array_equal(lci, eci)
        assert np.array_equal(lva, eva) and np.array_equal(ghosts, eghosts)
        # the halo is the box's surface: at most one layer of points per face
        bx, by, bz = box
        assert len(ghosts) <= 2 * (bx * by + by * bz + bx * bz)
        nnz += len(lva)
    assert nnz == len(va)
    with pytest.raises(host.SpmvHostError):
        host.poisson3d_box_rows(n, (n + 1, 1, 1), 0)
    with pytest.raises(host.SpmvHostError):
        host.poisson3d_box_rows(n, parts, P)


def _zwalk_table(rows, plane_rows, grid, segments=0):
    import ctypes as C
    slots, segs = C.c_int64(), C.c_int()
    _lib.call("spmv_hip_zwalk_table", rows, plane_rows, grid, segments, None, 0,
              C.byref(slots), C.byref(segs))
    t = np.empty(slots.value, np.int32)
    _lib.call("spmv_hip_zwalk_table", rows, plane_rows, grid, segments,
              t.ctypes.data_as(C.c_void_p), len(t), C.byref(slots), C.byref(segs))
    return t, segs.value


@pytest.mark.parametrize("n,grid,segments", [
    (512, 1024, 0),   # the benchmark: one plane = 1024 row blocks = the grid
    (448, 1024, 0),   # 784 columns: runs along the plane axis balance the grid
    (384, 1024, 0), (216, 1024, 0),  # planes not a whole number of row blocks
    (128, 512, 3), (64, 64, 1), (33, 24, 2), (16, 1024, 5)])
def test_plane_walk_table(n, grid, segments):
    """spmv_hip_zwalk_table (what the lattice kernels walk): a permutation of
    the row blocks plus empty slots; inside a run a workgroup's consecutive
    blocks are one plane apart (exactly, when planes are whole row blocks: the
    plane chain of the kernels rests on it); 8 consecutive columns share an
    XCD; the work is balanced."""
    rows, d2 = n ** 3, n * n
    nrb = (rows + 255) // 256
    t, segs = _zwalk_table(rows, d2, grid, segments)
    assert len(t) % grid == 0
    used = t[t >= 0]
    assert np.array_equal(np.sort(used), np.arange(nrb))      # each block once
    if segments:
        assert 1 <= segs <= segments
    per_wg = t.reshape(-1, grid)                              # [step, workgroup]
    nz = (rows + d2 - 1) // d2
    L = -(-nz // segs)
    chained = total = 0
    for w in range(0, grid, max(1, grid // 64)):
        col = per_wg[:, w]
        for r in range(len(col) // L):
            run = col[r * L:(r + 1) * L].astype(np.int64)
            # consecutive STEPS that both have work (the last column of a
            # plane exists only in the planes with one row block more)
            both = (run[1:] >= 0) & (run[:-1] >= 0)
            d = (run[1:] - run[:-1])[both] * 256
            total += len(d)
            assert np.all(np.abs(d - d2) < 256 + 1), (w, r)   # one plane apart
            chained += int(np.sum(d == d2))
    if d2 % 256 == 0 and total:
        assert chained == total                               # exactly
    # balance: the busiest workgroup does at most 35 % more than the mean
    # (much less for the large cases)
    work = (per_wg >= 0).sum(0)
    if nrb >= 4 * grid and segments == 0:  # the builder's own choice of runs
        assert work.max() <= 1.35 * nrb / grid + 1
    if n == 512:
        assert segs == 1 and len(t) == nrb and work.min() == work.max() == 512
    if grid % 64 == 0 and nrb >= grid:
        # XCD = workgroup % 8 owns runs of 8 consecutive columns
        first = per_wg[0]
        cols = {w: int(first[w]) for w in range(64) if first[w] >= 0}
        for x in range(8):
            mine = sorted(c for w, c in cols.items() if w % 8 == x)
            if len(mine) == 8:
                assert mine == list(range(mine[0], mine[0] + 8))
    with pytest.raises(Exception):
        _zwalk_table(rows, 0, grid)


def test_synthetic_generators_numpy_twins():
    """spmv_amd/poisson.py: the 27-point operator against an independent
    scipy construction (Kronecker sums of 1-D neighbour matrices), and the
    seeded unstructured matrix's promises (shape, sorted rows, share of far
    entries, determinism) -- the GPU tests compare the device generators with
    these twins array for array."""
    import scipy.sparse as sp
    from spmv_amd import poisson
    for n in (3, 4, 7):
        rp, ci, va = poisson.stencil27_csr(n)
        A = sp.csr_matrix((va, ci, rp), shape=(n ** 3, n ** 3))
        t1 = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
        B = sp.kron(sp.kron(t1, t1), t1).tocsr()  # 1 wherever |dx|,|dy|,|dz| <= 1
        want = (27.0 * sp.identity(n ** 3) - B).tocsr()  # diag 26, neighbours -1
        assert abs(A - want).max() == 0
        assert len(va) == poisson.stencil27_nnz(n)
        assert np.all(np.diff(ci)[np.diff(np.repeat(np.arange(n ** 3),
                                                    np.diff(rp))) == 0] > 0)
    N = 20000
    rp, ci, va = poisson.unstructured_csr(N)
    rp2, ci2, va2 = poisson.unstructured_csr(N)
    assert np.array_equal(ci, ci2) and np.array_equal(va, va2)
    assert np.array_equal(rp, np.arange(N + 1) * 7) and len(ci) == 7 * N
    rows = ci.reshape(N, 7)
    assert np.all(np.diff(rows, axis=1) >= 0) and rows.min() >= 0 and rows.max() < N
    far = np.abs(rows - np.arange(N)[:, None]) > 2048
    assert 0.05 < far.mean() < 0.12 and -1 <= va.min() and va.max() < 1
    assert not np.array_equal(ci, poisson.unstructured_csr(N, seed=1)[1])



def test_fem_like_matrix_twin_properties_and_petsc_writer(tmp_path):
    """The numpy twin of the FEM-like test matrices (the ragged-row records of
    bench.py; the device generator is compared with it array by array in the
    GPU tests): row lengths in range and skewed to the short side, a 1 % tail
    of long rows where asked for, columns strictly ascending and in range, the
    diagonal in every row; tools/write_petsc.py writes it in the reference's
    file format (spmv/read_petsc.cpp:60-121: big-endian, magic 1211216) -- the
    oracle's restatement of the reader returns the same arrays."""
    import subprocess
    N = 60_000
    for kw, lo, hi in ((dict(), 5, 40), (dict(tail_permille=10), 5, 2000),
                       (dict(min_len=81, max_len=81), 81, 81)):
        rp, ci, va = poisson.fem_like_csr(N, **kw)
        lens = np.diff(rp)
        assert lens.max() <= hi and lens.min() >= 1
        interior = lens[2100:-2100]  # (edge rows lose a side cluster)
        if not kw.get("tail_permille"):
            assert interior.min() >= lo
        if kw == dict():
            assert 13 < lens.mean() < 17 and np.median(lens) < lens.mean()
        if kw.get("tail_permille"):
            long_rows = lens >= 200
            assert 0.005 < long_rows.mean() < 0.015
        first = np.zeros(len(ci), bool)
        first[rp[:-1][lens > 0]] = True
        assert np.all(np.diff(ci.astype(np.int64))[~first[1:]] > 0)
        assert ci.min() >= 0 and ci.max() < N
        rows = np.repeat(np.arange(N), lens)
        assert np.array_equal(np.bincount(rows[ci == rows], minlength=N),
                              np.ones(N, np.int64))
        assert np.array_equal(va[ci == rows], lens + 1.0)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa, fb = tmp_path / "A.dat", tmp_path / "b.dat"
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "write_petsc.py"),
                          "--kind", "fem_tail", "--rows", str(N), "--out", str(fa),
                          "--rhs", str(fb)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rp, ci, va = poisson.fem_like_csr(N, tail_permille=10)
    for size in (1, 3):
        parts = [oracle.petsc_io.read_matrix_rows(fa, r, size) for r in range(size)]
        assert sum(len(p["values"]) for p in parts) == len(va)
        assert np.array_equal(np.concatenate([p["values"] for p in parts]), va)
    one = oracle.petsc_io.read_matrix_rows(fa, 0, 1)
    assert np.array_equal(one["rowptr"], rp) and np.array_equal(one["colind"], ci)
    b = oracle.petsc_io.read_vector(fb, 0, 1)
    assert np.allclose(b, oracle.csr_spmv(rp, ci, va, np.ones(N)), rtol=1e-12)
