"""Kernel time against wall time of tools/rank_shape.py under rocprofv3:

    rocprofv3 --kernel-trace --output-format csv -d DIR -o rs -- \
        python3 tools/rank_shape.py --ranks 8 --steps 100
    python tools/rank_shape_report.py DIR/.../rs_kernel_trace.csv --steps 100

Takes the LAST `steps` iterations of the trace (an iteration begins at a
launch of the local block's SpMV kernel), sums the kernels' own durations and
compares with the span from the first kernel's start to the last kernel's end:
the rest is kernel boundaries (launch gaps, the cache actions between dependent
kernels, event waits).  Per kernel name: launches per iteration and average
duration."""
import argparse
import csv
import json
from collections import defaultdict

CG = ("csr_", "cg_", "reduce_partials", "peer_reduce", "put_")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--spmv", default=None,
                    help="substring of the local SpMV kernel's name (default: the "
                         "csr_ kernel with the longest total time)")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         r["Kernel_Name"]))
    rows.sort()
    rows = [r for r in rows if any(k in r[2] for k in CG)]
    tot = defaultdict(int)
    for s, e, n in rows:
        if "csr_" in n and "build" not in n:
            tot[n] += e - s
    spmv = args.spmv or max(tot, key=tot.get)
    starts = [i for i, r in enumerate(rows) if spmv in r[2]]
    first = starts[-args.steps]
    sel = rows[first:]
    # the run ends with the x / p update that follows its last SpMV launch
    after = starts[-1] - first
    last = min(i for i, r in enumerate(sel) if i > after and "cg_update_xp" in r[2])
    sel = sel[:last + 1]
    span = sel[-1][1] - sel[0][0]
    busy = sum(e - s for s, e, _ in sel)
    per = defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "")
        k = k.split("(")[0][:90]
        per[k][0] += 1
        per[k][1] += e - s
    out = {"iterations": args.steps, "kernels": len(sel),
           "launches_per_iteration": len(sel) / args.steps,
           "ms_per_iteration": span / args.steps / 1e6,
           "kernel_ms_per_iteration": busy / args.steps / 1e6,
           "boundary_share": 1.0 - busy / span,
           "per_kernel": {k: {"per_iteration": v[0] / args.steps,
                              "avg_us": v[1] / v[0] / 1e3}
                          for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
