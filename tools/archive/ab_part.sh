# A/B of partitioned-path variants on cfg4 (P=3) and cfg3 (P=8): tools/archive/ab_part.sh <suffix|base> ...
cd $GRAFT_REPO_ROOT
for w in cfg4 cfg3; do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  python bench.py --workload $w --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$w $v', round(d['value']/1e9,3),'Gframes/s', round(d['ms_per_step'],2),'ms/step cmac',round(r['kernel_avg_ms'],2))"
done; done
