cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_kats.py tests/test_gpu_full_size.py -m gpu -q 2>&1 | tail -2
for w in cfg2 cfg5 cfg1; do bash tools/profile_round4.sh round4_v2 $w 2>&1 | tail -2; done
