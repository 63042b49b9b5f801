# round 6: GPU parity of the overlap-add tile, then the whole GPU suite, then cfg2 / cfg2-14ch bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_ola.py -x -q 2>&1 | tail -15 | tee gpurun_out/r6c/pytest_ola.txt
timeout 1700 python -m pytest tests/ -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r6c/pytest_gpu.txt
for w in cfg2 cfg2-14ch; do python bench.py --workload $w --no-end-to-end > gpurun_out/r6c/bench_$w.json 2> gpurun_out/r6c/bench_$w.err; cp bench_detail.json gpurun_out/r6c/bench_detail_$w.json; done
python - <<'PY'
import json
for w in ("cfg2", "cfg2-14ch"):
    d = json.loads([l for l in open(f"gpurun_out/r6c/bench_{w}.json") if l.startswith("{")][-1]); r = d["roofline"]
    print(w, round(d["value"] / 1e9, 2), "G/s frac", round(r["frac"], 4), d["config"]["path"], d["config"]["hop"], r["kernel"], r["kernel_avg_ms"], "parity", d.get("parity_spot_err"))
PY
