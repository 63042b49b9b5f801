cd $GRAFT_REPO_ROOT
O=gpurun_out/r2part2; mkdir -p $O
for v in "" d8 slp; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_$v.so
  for r in 1 2; do
  echo -n "variant '$v' run $r: "; AIRWAVE_HIP_LIBRARY=$lib python bench.py --workload cfg3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']/1e9,2),'Gframes/s', round(d['ms_per_step'],2),'ms/step')"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r2part2/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
