#!/usr/bin/env python3
"""Turns gpurun_out/profile/ (tools/archive/profile_round.sh) into the committed summary under profiles/<name>/."""
import collections, csv, glob, json, os, shutil, sys
src = "gpurun_out/profile"
name = sys.argv[1] if len(sys.argv) > 1 else "round1"
dst = os.path.join("profiles", name)
os.makedirs(dst, exist_ok=True)
out = {}
bench = [l for l in open(f"{src}/bench.json") if l.startswith("{")]
out["bench"] = json.loads(bench[-1])
for f in glob.glob(f"{src}/stats/*/*kernel_stats.csv"):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats_top"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage")} for r in rows[:4]]
pmc = collections.defaultdict(list)
for f in glob.glob(f"{src}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "aw_fused" in r["Kernel_Name"] or "aw_part" in r["Kernel_Name"]:
            pmc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
flat = {f"{k} :: {c}": sum(v) / len(v) for (k, c), v in sorted(pmc.items())}
out["pmc_avg_per_launch"] = flat
# HBM-side traffic of the dominant kernel per launch
def g(c):
    vals = [v for (k, cc), v in pmc.items() if cc == c and "aw_fused" in k]
    return sum(vals[0]) / len(vals[0]) if vals else None
r32, r64, r128, w64, wtot = g("TCC_EA0_RDREQ_32B_sum"), g("TCC_EA0_RDREQ_64B_sum"), g("TCC_EA0_RDREQ_128B_sum"), g("TCC_EA0_WRREQ_64B_sum"), g("TCC_EA0_WRREQ_sum")
if r128 is not None:
    rd = 32 * (r32 or 0) + 64 * (r64 or 0) + 128 * r128
    wr = 64 * (w64 or 0) + 32 * ((wtot or 0) - (w64 or 0))
    out["traffic_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr,
        "method": "TCC_EA0_RDREQ_{32,64,128}B x size + TCC_EA0_WRREQ (64B) ; FETCH_SIZE on gfx950 counts 128-B requests as 64 B",
        "fetch_size_kb": g("FETCH_SIZE"), "write_size_kb": g("WRITE_SIZE")}
json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "pmc_avg_per_launch"}, indent=1)[:3000])
