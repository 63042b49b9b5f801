# round 5, GPU call 5: balanced row tiles + steady-state defaults: GPU suite, every workload's bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r5e/pytest_gpu.txt
for W in cfg3 cfg2 cfg2-14ch cfg4 cfg5 cfg1; do
  X="--no-cpu-baseline"; [ $W = cfg3 ] && X=""
  timeout 600 python bench.py --workload $W $X > gpurun_out/r5e/bench_$W.json 2> gpurun_out/r5e/bench_$W.err; echo "$W rc $?"
done
python - <<'PY'
import json
for w in ["cfg3","cfg2","cfg2-14ch","cfg4","cfg5","cfg1"]:
    try: d=json.loads(open(f"gpurun_out/r5e/bench_{w}.json").read().strip().splitlines()[-1])
    except Exception as e: print(w, "unreadable", e); continue
    r=d["roofline"]
    print(w, round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"],4), "ms frac", round(r["frac"],4), "of copy", round(r.get("frac_of_measured") or 0,4), "of mix", round(r.get("frac_of_measured_mix") or 0,4), r["stages_ms_per_step"], "kern", round(r["kernel_avg_ms"],4), "act", d["config"]["activation"][0], "warm", d["config"].get("activation_warm"))
    if "secondary" in d: print("   sec", round(d["secondary"]["value"]/1e9,2), round(d["secondary"]["roofline"]["frac"],4), d["secondary"]["roofline"]["stages_ms_per_step"])
    for e in d.get("secondary_end_to_end", []): print("   e2e", e["name"], round(e["value"]/1e9,3), "G/s frac_of_pcie", round(e.get("frac_of_pcie",0),3), "pageable", round(e["pageable"]["value"]/1e9,3))
PY
