cd $GRAFT_REPO_ROOT
for v in base eqc8 eqc4; do for split in 0 1; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_$v.so; fi
  echo -n "$v split=$split: "; AW_EQ_EAR_SPLIT=$split python tools/eq_probe.py 512 960000 2>/dev/null | tail -1
done; done
