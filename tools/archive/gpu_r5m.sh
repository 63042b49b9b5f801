# round 5: what the driver runs at round end (the build check runs in the container): GPU suite, smoke, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r5m/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m/smoke.txt
SECONDS=0; python bench.py > gpurun_out/r5m/bench_default.json 2> gpurun_out/r5m/bench_default.err; echo "bench rc $? in ${SECONDS}s"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5m/bench_default.json").read().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print(round(d["value"]/1e9,2), "G/s frac", round(r["frac"],4), "of copy", round(r["frac_of_measured"],3), "mix", round(r["frac_of_measured_mix"],3), "traffic", r["traffic"], r["traffic_head"])
print(" act", d["config"]["activation"], d["config"]["activation_warm"], "mem", d["config"]["device_memory"])
print(" sec", round(d["secondary"]["value"]/1e9,2), d["secondary"]["roofline"]["frac"], d["secondary"]["roofline"]["traffic"])
c2=d["secondary_cfg2"]; print(" cfg2", round(c2["value"]/1e9,2), round(c2["roofline"]["frac"],4), c2["steps"], c2["warmup"], c2["parity_spot_err"], c2["roofline"]["traffic"])
for e in d["secondary_end_to_end"]: print(" e2e", e["name"], round(e["value"]/1e9,3), round(e["frac_of_pcie"],3), round(e["pageable"]["value"]/1e9,3), e.get("parity_spot_err"))
print(" cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "parity", d["parity_spot_err"], d["secondary"]["parity_spot_err"])
PY
