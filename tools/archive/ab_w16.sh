# A/B of 16384-window kernel variants: tools/archive/ab_w16.sh <suffix|base> ...   (cfg2 with AW_WINDOW=16384, then cfg4)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  AW_WINDOW=16384 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v cfg2/16384', round(d['value']/1e9,3), 'Gframes/s kernel', round(r['kernel_avg_ms'],4))"
  python bench.py --workload cfg4 --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v cfg4', round(d['value']/1e9,3), 'Gframes/s kernel', round(r['kernel_avg_ms'],4))"
done; done
