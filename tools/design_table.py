#!/usr/bin/env python3
"""Regenerates the "Where things stand" table of DESIGN.md and the current table of profiles/README.md from two profile directories
(CPU only):   python tools/design_table.py round6_v3 round6_v2"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
NAMES = {"cfg3": "**cfg 3 (headline)**: 7 speakers → 14 × 32 768-tap HRIR, 1024 streams × 10 s", "cfg3-14ch": "cfg 3, 14-channel-input reading",
         "cfg2": "cfg 2: 7.1 → RoomSH1.0 (4320 taps), 128 streams × 10 s", "cfg2-14ch": "cfg 2 with 14-channel input (north_star's literal layout)",
         "cfg4": "cfg 4: 7 speakers @ 96 kHz (8640 taps) + 10-band EQ, 512 streams/GPU", "cfg5": "cfg 5: 44.1 / 48 / 96 kHz mixed, 1024 streams/GPU",
         "cfg1": "cfg 1: one stereo stream (plumbing)"}
R5 = {"cfg3": "0.198 – 0.208", "cfg3-14ch": "0.184", "cfg2": "0.252", "cfg2-14ch": "0.199 (3.19 ×)", "cfg4": "0.151 (3.53 ×)", "cfg5": "0.2105", "cfg1": "—"}
ORDER = ("cfg3", "cfg3-14ch", "cfg2", "cfg2-14ch", "cfg4", "cfg5", "cfg1")


def load(d, w):
    return json.load(open(os.path.join(ROOT, "profiles", d, f"bench_{w}.json"))), json.load(open(os.path.join(ROOT, "profiles", d, f"traffic_{w}.json")))


rows, prow = [], []
for w in ORDER:
    (ba, ta), (bb, _) = load(a, w), load(b, w)
    r = ba["roofline"]
    st = {k: v for k, v in (r.get("stages_ms_per_step") or {}).items() if not k.startswith("aw_hist")}
    if w == "cfg5":
        kern = "44.1 / 48 kHz buckets: `aw_fused_ola_kernel<7, 4, 8>` / `<7, 4, 7>`; 96 kHz bucket: long-window " + " + ".join(f"{k.replace('aw_lw_', '').replace('_kernel', '')} {v:.2f}" for k, v in st.items()) + " ms"
    elif len(st) >= 2:
        kern = ba["config"]["path"].split(" (")[0] + ": " + " + ".join(f"{k.replace('aw_lw_', '').replace('_kernel', '')} {v:.2f}" for k, v in st.items()) + " ms"
    else:
        kern = ba["config"]["path"].split(" (")[0] + f": `{r['kernel']}` {r['kernel_avg_ms']:.3f} ms"
    rows.append(f"| {NAMES[w]} | {ba['value'] / 1e9:.1f} / {bb['value'] / 1e9:.1f} | **{r['frac']:.3f}** / {bb['roofline']['frac']:.3f} | {ta['ratio_to_algorithmic']:.2f} × | {R5[w]} | {kern} |")
    prow.append(f"| `{w}` | {ba['value'] / 1e9:.2f} | {r['frac']:.4f} | {r['step_ms']:.3f} | {ta['total_bytes_per_step'] / 1e9:.2f} GB = {ta['ratio_to_algorithmic']:.2f} × | `{r['kernel']}` {r['kernel_avg_ms']:.3f} ms |")
head = f"| workload | G stereo frames/s (`{a}` / `{b}`) | `roofline.frac` | traffic | round 5 | kernels of a step |\n|---|---|---|---|---|---|\n"
for path, table in ((os.path.join(ROOT, "DESIGN.md"), head + "\n".join(rows)),
                    (os.path.join(ROOT, "profiles", "README.md"), "| workload | G frames/s | `roofline.frac` | step ms | fabric traffic per step | dominant kernel (HIP events) |\n|---|---|---|---|---|---|\n" + "\n".join(prow))):
    s = open(path).read()
    s, n = re.subn(r"(<!-- table:current -->\n).*?(\n<!-- /table:current -->)", lambda m: m.group(1) + table + m.group(2), s, flags=re.S)
    assert n == 1, path
    open(path, "w").write(s)
print("tables written from", a, b)
