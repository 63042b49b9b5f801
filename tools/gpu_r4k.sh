cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo -n "old8192: "; WINDOW=8192 AIRWAVE_HIP_LIBRARY=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_olsx.so python tools/ols2_ab.py 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 2>&1 | tail -1
echo -n "new8192: "; WINDOW=8192 python tools/ols2_ab.py 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 2>&1 | tail -1
done
echo -n "new16384: "; python tools/ols2_ab.py 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
