# A/B of frames per thread (AW_EQ_CHUNK builds: libairwave_hip_eqc<chunk>.so) on the per-ear kernel: tools/archive/ab_eq_chunk.sh 32 64
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" = "base" ]; then unset AIRWAVE_HIP_LIBRARY; else export AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_eqc$v.so; fi
  echo "== chunk $v"; AW_EQ_EAR_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_eq.py -x -q 2>&1 | tail -1
  for S in 128 512; do echo -n "split=1 streams=$S: "; AW_EQ_EAR_SPLIT=1 python tools/eq_probe.py $S 960000 2>/dev/null | tail -1; done
done
