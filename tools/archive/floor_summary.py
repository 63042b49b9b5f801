#!/usr/bin/env python3
"""Condenses the output of tools/archive/floor_proof.sh into one table (floor_summary.json / .md in the same directory)."""
import json, os, re, statistics, sys
out = sys.argv[1]
rows = []
def smi(tag):
    p = os.path.join(out, f"smi_{tag}.txt")
    pw, sc = [], []
    if os.path.exists(p):
        for l in open(p):
            m = re.search(r"Power[^:]*:\s*([0-9.]+)", l)
            if m: pw.append(float(m.group(1)))
            m = re.search(r"sclk[^(]*\((\d+)Mhz\)", l)
            if m: sc.append(float(m.group(1)))
    f = lambda v: None if not v else {"median": statistics.median(v), "max": max(v), "samples": len(v)}
    return f(pw), f(sc)
for v in ("full", "nomem", "noload", "notab", "nostore", "nofft", "form8"):
    p = os.path.join(out, f"rows_{v}.txt")
    if not os.path.exists(p): continue
    t = open(p).read()
    m = re.search(r"best ([0-9.]+) ms, mean ([0-9.]+) ms, ([0-9.]+) TB/s", t)
    c = re.search(r"median ([0-9.]+) GHz, p10 ([0-9.]+), p90 ([0-9.]+)", t)
    sus = [float(x) for x in re.findall(r"sustained: \d+ launches, ([0-9.]+) ms each", t)]
    pw, sc = smi(v)
    rows.append({"variant": v, "best_ms": float(m.group(1)) if m else None, "mean_ms": float(m.group(2)) if m else None,
                 "sustained_ms": sus[-1] if sus else None, "in_kernel_clock_ghz": float(c.group(1)) if c else None,
                 "clock_p10_p90": [float(c.group(2)), float(c.group(3))] if c else None,
                 "mcycles": float(m.group(2)) * float(c.group(1)) if m and c else None, "smi_power_w": pw, "smi_sclk_mhz": sc})
p = os.path.join(out, "stream_probe.txt")
probe = []
if os.path.exists(p):
    t = open(p).read()
    for m in re.finditer(r"mode (\d).*?: ([0-9.]+) ms for ([0-9.]+) GB -> ([0-9.]+) TB/s\n\s*in-kernel clock: median ([0-9.]+)", t):
        probe.append({"mode": int(m.group(1)), "ms": float(m.group(2)), "GB": float(m.group(3)), "TBs": float(m.group(4)), "in_kernel_clock_ghz": float(m.group(5))})
pw, sc = smi("probe")
res = {"rows_kernel": rows, "stream_probe": probe, "stream_probe_smi_power_w": pw, "stream_probe_smi_sclk_mhz": sc,
       "what": "tools/archive/floor_proof.sh: rows kernel alone on cfg-3-shaped scratch (1024 stream-windows x 64 row pairs, 3.5 channel pairs) and a read-only probe of its access shape, "
               "each after 2.5 s of back-to-back launches; in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz per workgroup (median); mcycles = mean_ms x clock"}
json.dump(res, open(os.path.join(out, "floor_summary.json"), "w"), indent=1)
with open(os.path.join(out, "floor_summary.md"), "w") as f:
    f.write("| variant | best ms | mean ms | in-kernel clock GHz (p10-p90) | M cycles | rocm-smi power W (median / max) | rocm-smi sclk MHz (median) |\n|---|---|---|---|---|---|---|\n")
    for r in rows:
        pwv = r["smi_power_w"]; scv = r["smi_sclk_mhz"]
        f.write(f"| {r['variant']} | {r['best_ms']} | {r['mean_ms']} | {r['in_kernel_clock_ghz']} ({r['clock_p10_p90']}) | {None if r['mcycles'] is None else round(r['mcycles'], 2)} | "
                f"{None if not pwv else (pwv['median'], pwv['max'])} | {None if not scv else scv['median']} |\n")
    for q in probe:
        f.write(f"| stream_probe mode {q['mode']} | {q['ms']} | {q['GB']} GB | {q['in_kernel_clock_ghz']} | {q['TBs']} TB/s | {None if not pw else (pw['median'], pw['max'])} | {None if not sc else sc['median']} |\n")
print(open(os.path.join(out, "floor_summary.md")).read())
