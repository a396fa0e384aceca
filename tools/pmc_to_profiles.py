"""Turn one tools/pmc_passes.sh output directory into the committed evidence:
the per-pass CSVs (copied under profiles/ with a round prefix) and one entry
per SpMV kernel in profiles/rNN_pmc_summary.json, which bench.py reads for
`roofline.traffic`.

    python tools/pmc_to_profiles.py gpurun_out/pmc4 r02 --grid 512 --note "..."
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("round_tag")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--note", default="")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "profiles"),
                    help="where the CSVs and the summary go (on a GPU box: a "
                         "directory under gpurun_out/, copied to profiles/ later)")
    ap.add_argument("--record", default=None,
                    help="also file the (single) SpMV kernel of this directory "
                         "under this bench.py record name: what bench.py looks "
                         "up for the record's `traffic`")
    args = ap.parse_args()
    summ = json.loads(subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), args.pmc_dir]))
    os.makedirs(args.out_dir, exist_ok=True)
    out_path = os.path.join(args.out_dir, f"{args.round_tag}_pmc_summary.json")
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {"kernels": []}
    doc.setdefault("kernels", [])
    doc.setdefault("records", {})
    doc.setdefault("note", "fabric bytes per launch from rocprofv3 PMC passes (one "
                   "counter group per pass): reads = TCC_EA0_RDREQ_sum x 128 B "
                   "(calibrated on the streaming dot product in the same passes), "
                   "writes = WRITE_SIZE x 1 KiB; Infinity-Cache hits are counted")
    filed = False
    for tag, kernels in summ.items():
        files = []
        for f in sorted(glob.glob(os.path.join(args.pmc_dir, f"{tag}_*_counter_collection.csv"))):
            grp = os.path.basename(f).split("_")[-3]
            dst = f"{args.round_tag}_pmc_{tag}_{grp}_n{args.grid}.csv"
            shutil.copy(f, os.path.join(args.out_dir, dst))
            files.append(grp)
        for kname, c in kernels.items():
            if not kname.startswith("csr_"):
                continue
            if "TCC_EA0_RDREQ_sum" not in c or "WRITE_SIZE" not in c:
                continue
            rd = c["TCC_EA0_RDREQ_sum"] * 128.0
            wr = c["WRITE_SIZE"] * 1024.0
            rec = {
                "kernel_prefix": kname, "grid": args.grid,
                "fabric_read_bytes": rd, "write_bytes": wr,
                "fabric_bytes_per_launch": rd + wr,
                "FETCH_SIZE_KiB": c.get("FETCH_SIZE"),
                "WRITE_SIZE_KiB": c.get("WRITE_SIZE"),
                "TCC_EA0_RDREQ": c["TCC_EA0_RDREQ_sum"],
                "TCC_REQ": c.get("TCC_REQ_sum"), "TCC_HIT": c.get("TCC_HIT_sum"),
                "TCC_MISS": c.get("TCC_MISS_sum"),
                "avg_ea_read_latency_cycles":
                    (c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]
                     if "TCC_EA0_RDREQ_LEVEL_sum" in c else None),
                "ms_profiled": c["_ms_profiled"],
                "source": f"profiles/{args.round_tag}_pmc_{tag}_{{{','.join(files)}}}"
                          f"_n{args.grid}.csv" + (f" ({args.note})" if args.note else ""),
            }
            doc["kernels"] = [k for k in doc["kernels"]
                              if not (k["kernel_prefix"] == kname
                                      and k["grid"] == args.grid)] + [rec]
            if args.record:
                # a record whose product takes two launches (the sliced jagged
                # form with long rows): the bytes of both, the kernels named
                prev = doc["records"].get(args.record) if filed else None
                if prev:
                    rec = dict(rec)
                    for k in ("fabric_read_bytes", "write_bytes",
                              "fabric_bytes_per_launch", "TCC_EA0_RDREQ", "TCC_REQ",
                              "TCC_HIT", "TCC_MISS", "ms_profiled"):
                        if rec.get(k) is None or prev.get(k) is None:
                            continue
                        if isinstance(rec[k], dict):  # per-pass durations
                            rec[k] = {p_: rec[k][p_] + prev[k].get(p_, 0.0)
                                      for p_ in rec[k]}
                        else:
                            rec[k] = rec[k] + prev[k]
                    rec["kernel_prefix"] = prev["kernel_prefix"] + " + " + kname
                doc["records"][args.record] = rec
                filed = True
    json.dump(doc, open(out_path, "w"), indent=1)
    print("updated", out_path)


if __name__ == "__main__":
    main()
