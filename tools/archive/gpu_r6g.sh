# round 6: GPU suite on the final tile, randomised parity sweeps (overlap-add tile included), EQ fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -8 | cut -c1-300 | tee gpurun_out/r6g/pytest_gpu.txt
for seed in 601 602 603; do timeout 400 python tools/fuzz_parity.py $seed 240 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-600; done | tee gpurun_out/r6g/fuzz_parity.txt
timeout 300 python tools/fuzz_eq.py 61 120 2>&1 | tail -3 | tee gpurun_out/r6g/fuzz_eq.txt
