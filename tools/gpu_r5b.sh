# round 5, GPU call 2: device-built tables, cfg 4 role-split overlap probe, allocation timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_longwin.py tests/test_gpu_full_size.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r5b/pytest_lw.txt
timeout 300 python bench.py --no-end-to-end --no-cpu-baseline > gpurun_out/r5b/bench_default.json 2> gpurun_out/r5b/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5b/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e9,2), d["roofline"]["frac"], d["config"]["activation"], d["config"].get("activation_warm"), d.get("parity_spot_err"))
print(" sec", round(d["secondary"]["value"]/1e9,2), d["secondary"]["config"]["activation"], d["secondary"]["config"].get("activation_warm"), d["secondary"].get("parity_spot_err"))
PY
timeout 300 python tools/alloc_probe.py 2>&1 | tee gpurun_out/r5b/alloc_probe.txt
timeout 600 python tools/cfg4_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b/cfg4_overlap_probe.txt
