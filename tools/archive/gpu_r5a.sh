# round 5, GPU call 1: the whole GPU suite on the new runtime (scratch pool, chunked host entry, adapter reservation), the default bench
# with the measured ceilings / activation split / end-to-end legs, cfg 4 on one and two lanes, and the rows-kernel floor evidence.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r5a/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r5a/bench_default.json 2> gpurun_out/r5a/bench_default.err; echo "bench rc $?"
tail -3 gpurun_out/r5a/bench_default.err
for L in 1 2 3; do
  timeout 300 python bench.py --workload cfg4 --lanes $L --no-cpu-baseline > gpurun_out/r5a/bench_cfg4_lanes$L.json 2> gpurun_out/r5a/bench_cfg4_lanes$L.err; echo "cfg4 lanes $L rc $?"
done
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r5a/bench_cfg2.json 2> gpurun_out/r5a/bench_cfg2.err; echo "cfg2 rc $?"
python - <<'PY'
import json
def show(p):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(p, "unreadable", e); return
    r=d["roofline"]
    print(p.split("/")[-1], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"],3), "ms frac", round(r["frac"],4), "of measured", r.get("frac_of_measured"), "measured", {k: round(v) for k, v in (r.get("measured") or {}).items() if isinstance(v, float)},
          "stages", r["stages_ms_per_step"], "eq", r.get("eq_kernel_ms_per_step"), "act", d["config"].get("activation"), "warm", d["config"].get("activation_warm"), "parity", d.get("parity_spot_err"))
    if "secondary" in d: print("  secondary", round(d["secondary"]["value"]/1e9,2), d["secondary"]["roofline"]["frac"], d["secondary"]["config"].get("activation"), d["secondary"].get("parity_spot_err"))
    for e in d.get("secondary_end_to_end", []): print("  e2e", e["name"], round(e["value"]/1e9,3), "G/s", round(e["ms_per_batch"],2), "ms h2d", round(e["h2d_GBs"],1), "d2h", round(e["d2h_GBs"],1), "chunks", e["chunks"], "frac_of_pcie", e.get("frac_of_pcie"), "pcie", e.get("pcie_measured"), "pageable", round(e["pageable"]["value"]/1e9,3), "err", e.get("parity_spot_err"), "pin ms", e["pinned_alloc_ms"])
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for p in ["bench_default","bench_cfg4_lanes1","bench_cfg4_lanes2","bench_cfg4_lanes3","bench_cfg2"]: show(f"gpurun_out/r5a/{p}.json")
PY
bash tools/archive/floor_proof.sh gpurun_out/r5a/floor
