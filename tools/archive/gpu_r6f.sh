cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_eq_fold.py -x -q --tb=short 2>&1 | tail -30 | tee gpurun_out/r6f/pytest_fold.txt
python bench.py --workload cfg4 > gpurun_out/r6f/bench_cfg4.json 2> gpurun_out/r6f/bench_cfg4.err; echo "rc $?"; cp bench_detail.json gpurun_out/r6f/bench_detail_cfg4.json
tail -3 gpurun_out/r6f/bench_cfg4.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6f/bench_cfg4.json") if l.startswith("{")][-1]); r = d["roofline"]
print("cfg4", round(d["value"] / 1e9, 2), "G/s frac", round(r["frac"], 4), d["config"].get("equalizer"), d["config"]["path"], d["config"]["fft"], d["config"]["hop"], r.get("stages_ms_per_step"), "parity", d.get("parity_spot_err"))
c = d.get("secondary_eq_cascade")
if c: print(" cascade", round(c["value"] / 1e9, 2), c["roofline"]["frac"], c.get("parity_spot_err"))
print(" cpu", d.get("cpu_baseline"))
print(len(json.dumps(d)))
PY
