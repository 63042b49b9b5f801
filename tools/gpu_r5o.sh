cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/realtime_latency.py > gpurun_out/r5o/realtime_latency.json 2> gpurun_out/r5o/rt.err; echo "rt rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5o/realtime_latency.json"))
for name, rows in d["surfaces"].items():
    for entry, lst in rows.items():
        print(name[:30], entry, [(r["callback_frames"], r["p50_us"], r["p99_us"], r["max_us"], r["over_budget"]) for r in lst])
PY
