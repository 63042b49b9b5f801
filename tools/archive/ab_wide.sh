# A/B of library variants on the wide layouts: tools/archive/ab_wide.sh <suffix|base> ...   (CHANNELS=10,12,14,16)
cd $GRAFT_REPO_ROOT
export CHANNELS=${CHANNELS:-10,12,14,16}
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  echo -n "$v: "; python tools/wide_sweep.py 2>/dev/null | tail -1
done; done
