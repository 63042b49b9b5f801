cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 120 tools/ubench/fft_core | head -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_kats.py -m gpu -q 2>&1 | tail -3
WORKLOAD=cfg2 STEPS=20 bash tools/ab_bench.sh olsx - 2>&1 | tee gpurun_out/r4f/ab_cfg2.txt
