#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs under gpurun_out/pmc_*/ for the fused kernel."""
import collections, csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for d in sorted(glob.glob(f"{root}/pmc_*")):
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fused" in r["Kernel_Name"] or "part_" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{k:42s} {c:24s} n={len(v)} avg={sum(v)/len(v):.4g}")
