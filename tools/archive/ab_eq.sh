# A/B of EQ kernel variants (libairwave_hip_<suffix>.so): tools/archive/ab_eq.sh <suffix|base> ...   [AB_EQ_SPLIT=0|1 forces the form]
cd $GRAFT_REPO_ROOT
[ -n "$AB_EQ_SPLIT" ] && export AW_EQ_EAR_SPLIT=$AB_EQ_SPLIT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  [ $rep = 1 ] && { echo -n "$v tests: "; timeout 600 python -m pytest tests/test_gpu_eq.py -x -q 2>&1 | tail -1; }
  for S in 512 2048; do echo -n "$v streams=$S: "; python tools/eq_probe.py $S $((491520000/S)) 2>/dev/null | tail -1; done
done; done
