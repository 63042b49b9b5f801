cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
# the RCCL branch on the one GPU there is: a forced one-rank process group on the nccl backend
AW_BENCH_FORCE_PG=1 python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > gpurun_out/r4d/rccl_one_rank.txt 2> gpurun_out/r4d/rccl_one_rank.err; echo "rccl one-rank rc=$?"; tail -c 600 gpurun_out/r4d/rccl_one_rank.txt; grep -i 'bench rank\|nccl\|rccl' gpurun_out/r4d/rccl_one_rank.err | head -5
timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
for w in cfg3 cfg2-14ch cfg4 cfg5; do
  python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r4d/bench_$w.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4d/bench_$w.json").read()); r=d["roofline"]
print("$w", round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"],3), "ms frac", round(r["frac"],4), {k.replace("aw_","").replace("_kernel",""):round(v,2) for k,v in r.get("stages_ms_per_step",{}).items()}, "sec", round(d.get("secondary",{}).get("value",0)/1e9,2), "activation_ms", d["config"].get("activation_ms"), [l.get("path") for l in d["config"]["legs"]])
PY
done
