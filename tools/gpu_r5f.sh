# round 5, GPU call 6: randomised parity sweeps on the new runtime (pool, chunked host entry, device-built tables) + EQ fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
for seed in 601 602 603 604; do timeout 400 python tools/fuzz_parity.py $seed 150 2>&1 | grep -v amdgpu.ids | tail -4; done | tee gpurun_out/r5f/fuzz.txt
timeout 200 python tools/fuzz_eq.py 31 120 2>&1 | tail -3 | tee gpurun_out/r5f/fuzz_eq.txt
