"""CPU emulation of march_probe.hip's kernel on a plan of march_build.py (small
matrices only): the same walk over units, steps, lanes and streams, in Python --
checks the BUILDER against the oracle before any GPU time is spent.

    python tools/probes/march/march_emulate.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import march_build  # noqa: E402


def run(p, x, N):
    y = np.full(N, np.nan)
    meta = p["meta"]
    B = march_build.B
    for u in range(p["nunits"]):
        s0, s1 = p["unit_step0"][u], p["unit_step0"][u + 1]
        pend = np.zeros(B)
        prow = -np.ones(B, dtype=np.int64)
        pmeta = np.zeros(B, dtype=np.uint64)
        ptile = -1
        for st in range(s0, s1):
            T = p["step_tile"][st]
            c0, c1 = p["step_chunk0"][st], p["step_chunk0"][st + 1]
            sx = np.zeros((c1 - c0) * 16)
            for i, ch in enumerate(p["chunks"][c0:c1]):
                lo = ch * 16
                n = min(16, N - lo)
                sx[i * 16:i * 16 + n] = x[lo:lo + n]
            stash = np.full(1 << 16, np.nan)
            s = np.zeros(B)
            row = -np.ones(B, dtype=np.int64)
            m = np.zeros(B, dtype=np.uint64)
            if T >= 0:
                m = meta[T * B:(T + 1) * B]

            def walk(stream, tile, mm, lenshift, acc, val, code, from_stash, to_stash,
                     x_own=None):
                sb = p["sb%d" % stream]
                for sl in range(march_build.SLICES):
                    base = int(sb[tile * march_build.SLICES + sl])
                    lens = [(int(mm[sl * 64 + l]) >> lenshift) & 0xff for l in range(64)]
                    k = 0
                    while any(k < ln for ln in lens):
                        for l in range(64):
                            if k < lens[l]:
                                v = sl * 64 + l
                                c = int(code[base])
                                xv = stash[c] if from_stash else sx[c]
                                pr = val[base] * xv if val is not None else xv
                                acc[v] = acc[v] + pr
                                if to_stash:
                                    stash[base - int(sb[tile * march_build.SLICES])] = val[base] * x_own[v]
                                base += 1
                        k += 1
            if T >= 0:
                x_own = np.zeros(B)
                for v in range(B):
                    if int(m[v]) >> 63:
                        loc = (int(m[v]) >> 40) & 0xffff
                        row[v] = p["tile_row0"][T] + loc
                        x_own[v] = sx[p["step_own0"][st] + loc]
                        s[v] = p["diag"][row[v]] * x_own[v]
                walk(0, T, m, 0, s, p["a_val"], p["a_code"], False, True, x_own)
                walk(1, T, m, 8, s, None, p["bc_code"], True, False)
                walk(2, T, m, 16, s, p["bs_val"], p["bs_code"], False, False)
            if ptile >= 0:
                walk(3, ptile, pmeta, 24, pend, None, p["cc_code"], True, False)
                walk(4, ptile, pmeta, 32, pend, p["cs_val"], p["cs_code"], False, False)
                ok = prow >= 0
                y[prow[ok]] = pend[ok]
            pend, prow, pmeta, ptile = s, row, m.copy(), T
    return y


def main():
    import oracle
    from spmv_amd import poisson
    from util import lower_split
    rng = np.random.default_rng(5)
    for N, layer, jit, lseg, sort, tile in ((30000, 5000, 512, 3, True, 1024),
                                            (30000, 5000, 512, 2, False, 1024),
                                            (26000, 4500, 300, 100, True, 1024),
                                            (30000, 5000, 512, 3, True, 512)):
        march_build.B, march_build.SLICES = tile, tile // 64
        rp, ci, va = poisson.fem_like_csr(N, layer=layer, jitter=jit)
        lrp, lci, lva, ldg = lower_split(rp, ci, va)
        S = march_build.far_offset(lrp, lci)
        p = march_build.build(lrp, lci, lva, ldg, S, lseg, 9216, 5632, sort_rows=sort)
        x = rng.uniform(-1, 1, N)
        y = run(p, x, N)
        y_ref = oracle.csr_spmv_sym(lrp, lci, lva, ldg, x)
        st = p["stats"]
        print(N, "S", S, "units", st["units"], "captured",
              (st["captured_near"] + st["captured_far"]) / st["stored"],
              "max chunks", st["max_chunks_per_step"], "max stash", st["max_stash"],
              "bit-equal:", bool(np.array_equal(y, y_ref)),
              "differ:", int(np.sum(y != y_ref)))
        assert np.array_equal(y, y_ref)


if __name__ == "__main__":
    main()
