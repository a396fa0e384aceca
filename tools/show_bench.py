"""Prints the headline numbers of a bench.py run: give it the DETAIL file
(gpurun_out/bench_detail.json, the full record) or the log with the compact
line -- then the detail file named in the line is read."""
import json
import os
import sys

text = open(sys.argv[1]).read()
if text.lstrip().startswith("{") and "\n" in text.strip():
    d = json.loads(text)  # the detail file (indented JSON)
else:
    for ln in text.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
    if "detail" in d and "plan" not in d:  # the compact line: follow it
        path = d["detail"]
        if not os.path.isabs(path):
            path = os.path.join(os.path.dirname(os.path.dirname(
                os.path.abspath(__file__))), path)
        d = json.load(open(path))
print("value %.1f it/s  ms/step %.3f" % (d["value"], d["ms_per_step"]))
r = d["roofline"]
print("main: %.4f ms frac %.3f csr_eq %.3f traffic_frac %s" % (
    r["avg_launch_ms"], r["frac"], r["frac_csr_equivalent"], r.get("frac_traffic")))
if "time_to_solution" in d:
    t = d["time_to_solution"]
    print("tts: total %.1f ms (cg %.1f + plan %.1f), create %.1f" % (
        t["total_ms"], t["cg_ms"], t["plan_ms"], t["matrix_generate_and_plan_ms"]))
print("plan", d["plan"]["plan_ms"], d.get("matrix_create_ms"))
for k, v in d.items():
    if isinstance(v, dict) and "ms_per_apply" in v:
        print("%-26s rows %9d ms %.4f frac %.3f csr_eq %.3f plan %.1f ms %s %s" % (
            k, v["rows"], v["ms_per_apply"], v["frac"], v["frac_csr_equivalent"],
            v["plan_ms"], v.get("crosscheck", {}).get("bit_equal", ""),
            v["kernel"].split(" ")[0]))
if "symmetric" in d:
    s = d["symmetric"]
    print("symmetric: %.4f ms frac %.3f it/s %.1f plan %.1f" % (
        s["avg_launch_ms"], s["frac"], s["iters/s"], s["plan_ms"]))
if "value_stream_cg" in d:
    s = d["value_stream_cg"]
    print("values streamed: %.4f ms frac %.3f it/s %.1f plan %.1f" % (
        s["avg_launch_ms"], s["frac"], s["iters/s"], s["plan_ms"]))
if "mixed_precision_cg" in d:
    print("mixed speedup", d["mixed_precision_cg"]["speedup"])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu", c["value"], c.get("parity_checks"))
rg = d.get("roofline", {}).get("ragged", {})
for k, v in rg.items():
    if isinstance(v, dict):
        print("ragged %-18s %s" % (k, {a: (round(b, 4) if isinstance(b, float) else b)
                                        for a, b in v.items()
                                        if a in ("ms_per_apply", "frac", "iters/s",
                                                 "speedup_over_fp64")}))
