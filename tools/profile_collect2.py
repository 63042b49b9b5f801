#!/usr/bin/env python3
"""Turns gpurun_out/profile_<name>/<workload>/ (tools/archive/profile_round2.sh) into the committed files under profiles/<name>/:
bench_<w>.json (the bench line), kernel_stats_<w>.csv (rocprofv3 --stats), traffic_<w>.json (per-kernel fabric-side bytes
per step from the TCC_EA0 request counters + SQ counters; bench.py reads it for roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airwave_amd.provenance import build_flags_digest, build_head, device_source_digest, host_source_digest
name, w = sys.argv[1], sys.argv[2]
src = f"gpurun_out/profile_{name}/{w}"
dst = os.path.join("profiles", name)
os.makedirs(dst, exist_ok=True)
bench = [l for l in open(f"{src}/bench.json") if l.startswith("{")]
line = json.loads(bench[-1])
json.dump(line, open(os.path.join(dst, f"bench_{w}.json"), "w"), indent=1)
if os.path.exists(f"{src}/bench_detail.json"):        # round 6: the stdout line is compact, the full result sits beside it
    shutil.copy(f"{src}/bench_detail.json", os.path.join(dst, f"bench_detail_{w}.json"))
for f in glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(dst, f"kernel_stats_{w}.csv"))
pmc = collections.defaultdict(list)                # (kernel, counter) -> per-dispatch values
for f in glob.glob(f"{src}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void awk::", "").replace("awk::", "")
        if k.startswith("aw_"):
            pmc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
steps = 3                                         # --steps 2 --warmup 1: every step launches every kernel once
kernels = sorted({k for k, _ in pmc})
by_kernel, tot_r, tot_w = {}, 0.0, 0.0
for k in kernels:
    def per_step(c):
        v = pmc.get((k, c), [])
        return sum(v) / steps if v else 0.0
    if k.startswith("aw_synth"):
        continue
    rd = 32 * per_step("TCC_EA0_RDREQ_32B_sum") + 64 * per_step("TCC_EA0_RDREQ_64B_sum") + 128 * per_step("TCC_EA0_RDREQ_128B_sum")
    w64 = per_step("TCC_EA0_WRREQ_64B_sum")
    wr = 64 * w64 + 32 * (per_step("TCC_EA0_WRREQ_sum") - w64)
    sq = {c: per_step(c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")}
    by_kernel[k] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr, "launches_per_step": len(pmc.get((k, "TCC_EA0_RDREQ_sum"), [])) / steps, "sq_per_step": sq}
    tot_r += rd; tot_w += wr
cfg = line["config"]
out = {"workload": w, "device_src_sha16": device_source_digest(), "host_src_sha16": host_source_digest(), "build_flags_sha16": build_flags_digest(), "git_head": build_head(), "streams_per_gpu": cfg["streams_per_gpu"], "frames_per_stream": cfg["frames_per_stream"], "input_channels": cfg["input_channels"],
       "total_bytes_per_step": tot_r + tot_w, "read_bytes_per_step": tot_r, "write_bytes_per_step": tot_w,
       "algorithmic_bytes_per_step": line["roofline"]["algorithmic_bytes_per_step"],
       "ratio_to_algorithmic": (tot_r + tot_w) / line["roofline"]["algorithmic_bytes_per_step"],
       "method": "rocprofv3 --pmc, separate passes: reads = TCC_EA0_RDREQ_{32,64,128}B x size (FETCH_SIZE counts a 128-B request as 64 B on gfx950), "
                 "writes = TCC_EA0_WRREQ_64B x 64 + others x 32; L2 <-> fabric requests, Infinity-Cache hits included; averaged over the 3 steps of "
                 "`bench.py --steps 2 --warmup 1`",
       "by_kernel": by_kernel}
json.dump(out, open(os.path.join(dst, f"traffic_{w}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "by_kernel"}, indent=1))
for k, v in by_kernel.items():
    print(f"{k:60s} R {v['read_bytes_per_step']/1e9:7.2f} GB  W {v['write_bytes_per_step']/1e9:7.2f} GB")
r = line["roofline"]
print("bench:", round(line["value"] / 1e9, 2), "G frames/s, frac", round(r["frac"], 4), "frac_of_measured", r.get("frac_of_measured"), r.get("measured"), r["stages_ms_per_step"])
if "secondary" in line:
    s2 = line["secondary"]; print("secondary:", round(s2["value"] / 1e9, 2), "G frames/s, frac", round(s2["roofline"]["frac"], 4), s2["roofline"].get("stages_ms_per_step", "(bench_detail.json)"))
if "cpu_baseline" in line:
    print("cpu:", line["cpu_baseline"])
