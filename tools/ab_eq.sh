# A/B of EQ kernel variants: tools/ab_eq.sh <suffix|base> ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  echo -n "$v: "; python tools/eq_probe.py 512 960000 2>/dev/null | tail -1
done; done
