cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6q
timeout 600 python -m pytest tests/test_gpu_ola.py tests/test_gpu_parity.py -q -p no:cacheprovider --tb=short 2>&1 | tail -6 | cut -c1-300
TAPS=2048,3000,3585,3969,4097,4098,4320,4609,4610,5121 timeout 900 python tools/ola_sweep.py 5 9 11 13 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6q/ola_sweep_odd.txt
