cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_eq.py tests/test_effect_graph.py tests/test_abi_replay.py tests/test_gpu_reference_kats.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/realtime_latency.py > gpurun_out/r5p/realtime_latency.json 2> gpurun_out/r5p/rt.err; echo "rt rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5p/realtime_latency.json"))
for name, rows in d["surfaces"].items():
    for entry, lst in rows.items():
        print(name[:30], entry, [(r["callback_frames"], r["p50_us"], r["p99_us"], r["max_us"], r["over_budget"]) for r in lst])
PY
timeout 200 python tools/fuzz_eq.py 41 60 2>&1 | tail -2
