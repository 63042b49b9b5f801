# A/B of library variants on the secondary workloads: tools/archive/ab_cfg.sh <suffix|base> ...   (cfg4 and cfg5, 4 s per stream)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  for w in cfg4 cfg5; do
    python bench.py --workload $w --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v $w', round(d['value']/1e9,3), 'Gframes/s', round(d['ms_per_step'],3), 'ms/step')"
  done
done; done
