cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_gpu_host_entry.py tests/test_example_host.py -q -p no:cacheprovider --tb=short 2>&1 | tail -5
python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r6n/bench_cfg2.json 2>/dev/null; cp bench_detail.json gpurun_out/r6n/bench_detail_cfg2.json
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r6n/bench_cfg3.json 2>/dev/null; cp bench_detail.json gpurun_out/r6n/bench_detail_cfg3.json
python - <<'PY'
import json
for w in ("cfg2", "cfg3"):
    d = json.loads([l for l in open(f"gpurun_out/r6n/bench_{w}.json") if l.startswith("{")][-1])
    for e in d["secondary_end_to_end"]:
        print(w, e["name"], "pinned", round(e["value"] / 1e9, 3), "pageable", round(e["pageable_value"] / 1e9, 3), "ratio", round(e["pageable_value"] / e["value"], 3), "frac_of_pcie", e["frac_of_pcie"])
PY
