# same-box A/B of the round-2 build (commit 9eae087, its own bench.py and library under gpurun_r2/) against HEAD on cfg 2 and cfg2-14ch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_r2
for r in 1 2 3; do
  for w in cfg2 cfg2-14ch; do
    (cd gpurun_r2 && python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1) > gpurun_out/ab_r2/r2_${w}_$r.json
    AW_LW=0 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/ab_r2/head_${w}_$r.json
    python - <<PY
import json
a=json.loads(open("gpurun_out/ab_r2/r2_${w}_$r.json").read()); b=json.loads(open("gpurun_out/ab_r2/head_${w}_$r.json").read())
print("[$r] $w round-2 build:", round(a["value"]/1e9,2), "G/s", round(a["ms_per_step"],3), "ms | HEAD (fused tile, AW_LW=0):", round(b["value"]/1e9,2), "G/s", round(b["ms_per_step"],3), "ms")
PY
  done
done 2>&1 | tee gpurun_out/ab_r2/ab.txt
