import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spmv_amd import host
e = host.HipExecutor(0)
for rep in range(2):
    for gb in (0.5, 1, 2, 4, 4.6, 6, 7.5, 8, 12):
        n = int(gb * 2**30 / 8)
        e.synchronize(); t0 = time.perf_counter()
        p = e.alloc(n)
        e.synchronize(); t1 = time.perf_counter()
        e.memset(p, 0, 8 * n); e.synchronize(); t2 = time.perf_counter()
        e.free(p); e.synchronize(); t3 = time.perf_counter()
        print(rep, gb, "GB alloc %.1f ms memset %.1f ms free %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)
    # seven 1-GB pieces
    e.synchronize(); t0 = time.perf_counter()
    ps = [e.alloc(2**27) for _ in range(7)]
    e.synchronize(); t1 = time.perf_counter()
    for p in ps: e.free(p)
    print(rep, "7 x 1 GB alloc %.1f ms" % ((t1-t0)*1e3))
