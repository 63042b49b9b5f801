cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4c/pytest_gpu.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r4c/bench_default.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4c/bench_default.json").read()); r=d["roofline"]
print("default bench:", round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"],3), "ms frac", round(r["frac"],4), "traffic", r["traffic"], "sec", round(d["secondary"]["value"]/1e9,2), d["secondary"]["roofline"]["frac"], d["secondary"]["roofline"]["traffic"], "parity", d.get("parity_spot_err"), d["secondary"].get("parity_spot_err"), "cpu", d["cpu_baseline"]["value"], d["config"]["activation_ms"])
PY
