# round 4: GPU check of the 16-point rows kernel through the library: parity, then A/B against the 8-point form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1200 python -m pytest tests/test_gpu_longwin.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -5 | tee gpurun_out/r4a/pytest_longwin.txt
STEPS=5 bash tools/ab_bench.sh "-" "-:AW_LW_ROWS_FORM=8" 2>&1 | tee gpurun_out/r4a/ab.txt
WORKLOAD=cfg4 STEPS=3 bash tools/ab_bench.sh "-" "-:AW_LW_ROWS_FORM=8" 2>&1 | tee gpurun_out/r4a/ab_cfg4.txt
for f in 16 8; do AW_LW_ROWS_FORM=$f python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r4a/bench_form$f.json; done
python - <<'PY'
import json
for f in (16, 8):
    d = json.loads(open(f"gpurun_out/r4a/bench_form{f}.json").read())
    print(f, d["value"]/1e9, d["ms_per_step"], d["roofline"]["frac"], d.get("secondary", {}).get("value", 0)/1e9, d.get("parity_spot_err"))
PY
