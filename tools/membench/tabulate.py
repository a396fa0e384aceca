"""Pivot membench JSON lines: rows = test, columns = workgroups per CU
(0 = one workgroup per 16 KiB, -1 = runtime call)."""
import json
import sys

for path in sys.argv[1:]:
    rows = [json.loads(ln) for ln in open(path) if ln.startswith("{")]
    tests = []
    for r in rows:
        if r["test"] not in tests:
            tests.append(r["test"])
    wp = sorted({r["wg_per_cu"] for r in rows}, key=lambda x: (x <= 0, x))
    print(path, "GB/s; columns wg/cu:", wp)
    for t in tests:
        vals = [next((r["GB/s"] for r in rows
                      if r["test"] == t and r["wg_per_cu"] == w), 0) for w in wp]
        print(f"{t:28s}", " ".join(f"{v:7.0f}" if v else "      -" for v in vals))
