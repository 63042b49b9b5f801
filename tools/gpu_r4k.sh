cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo -n "default (even: SLP): "; python tools/ols2_ab.py 4 6 8 2>&1 | tail -1
echo -n "even no SLP: "; AIRWAVE_HIP_LIBRARY=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_evns.so python tools/ols2_ab.py 4 6 8 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -2
