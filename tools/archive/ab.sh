# A/B bench of library variants: tools/archive/ab.sh <suffix> [<suffix> ...]   ("base" = the default library)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value']/1e9,3), 'Gframes/s', round(d['roofline']['kernel_avg_ms'],4), 'ms', round(d['roofline']['frac'],4))"
done; done
