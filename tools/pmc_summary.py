"""Per-launch means of the counters collected by tools/pmc_passes.sh for the
SpMV kernels (and the calibration kernels of tools/prof_spmv.py).

    python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN_pmc_xxx.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEEP = ("csr_", "dot_partial", "fill_const")


def short(name):
    # the two passes of symmetric storage in the sliced jagged form are launches
    # of one template (its last argument: 1 = lower block, 2 = transposed block)
    if "csr_sjds_kernel<" in name:
        args = name.split("csr_sjds_kernel<", 1)[1].split(">(", 1)[0].split(",")
        # <T, TV, WPB, E, DOT, MODE, SIG>
        mode = args[5].strip() if len(args) > 5 else "0"
        if mode == "1":
            return "csr_sjds_kernel sym lower"
        if mode == "2":
            return "csr_sjds_kernel sym transposed"
        if mode == "3":
            return "csr_sjds_kernel sym merged"
    for k in ("csr_sjds_longt_kernel", "csr_sjds_long_kernel", "csr_sjds_kernel", "csr_box27_half_kernel", "csr_box27_const_kernel", "csr_const_dia_tile_kernel", "csr_const_dia_kernel", "csr_lxw_kernel",
              "csr_wdia_kernel", "csr_lattice_kernel", "csr_sym_lattice_kernel", "csr_sym_dia_kernel", "csr_rowblock_lx_kernel",
              "csr_rowblock_kernel", "csr_symt_kernel", "csr_sym_window_kernel",
              "dot_partial_kernel", "fill_const_kernel"):
        if k in name:
            return k
    return name[:60]


def main():
    out = defaultdict(lambda: defaultdict(dict))
    for path in sorted(glob.glob(os.path.join(sys.argv[1], "*_counter_collection.csv"))):
        tag = os.path.basename(path).rsplit("_", 3)[0]
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                kn = row["Kernel_Name"]
                if not any(k in kn for k in KEEP):
                    continue
                k = short(kn)
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[(k, row["Dispatch_Id"])] = (int(row["End_Timestamp"])
                                                - int(row["Start_Timestamp"]))
        for k, counters in acc.items():
            for c, vals in counters.items():
                vals = vals[1:] if len(vals) > 2 else vals  # drop the cold launch
                out[tag][k][c] = sum(vals) / len(vals)
            ds = [d for (kk, _), d in dur.items() if kk == k]
            ds = ds[1:] if len(ds) > 2 else ds
            out[tag][k].setdefault("_ms_profiled", {})
            out[tag][k]["_ms_profiled"][os.path.basename(path).split("_")[-3]] = (
                sum(ds) / len(ds) / 1e6)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
