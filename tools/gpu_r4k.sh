cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo -n "before: "; WINDOW=8192 AIRWAVE_HIP_LIBRARY=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_h4.so python tools/ols2_ab.py 9 10 11 12 13 14 15 16 2>&1 | tail -1
echo -n "parts8 for 13,14: "; WINDOW=8192 python tools/ols2_ab.py 9 10 11 12 13 14 15 16 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
