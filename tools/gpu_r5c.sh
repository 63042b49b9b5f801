# round 5, GPU call 3: which workloads sit at the 1400 W cap?  power / sclk traces of the bench workloads (long runs of the timed loop)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
for W in cfg2 cfg3 cfg4 cfg2-14ch cfg1; do
  ST=300; [ $W = cfg2 ] && ST=2500; [ $W = cfg2-14ch ] && ST=1500; [ $W = cfg1 ] && ST=20000
  bash tools/power_trace.sh gpurun_out/r5c/power_$W python bench.py --workload $W --steps $ST --warmup 5 --no-cpu-baseline --no-secondary --no-end-to-end --no-ceiling --no-warm-activation
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5c/power_$W.out.txt").read().strip().splitlines()[-1]); r=d["roofline"]
print("$W", round(d["value"]/1e9,2), "G/s frac", round(r["frac"],4), r["stages_ms_per_step"], r.get("kernel_avg_ms"))
PY
done
