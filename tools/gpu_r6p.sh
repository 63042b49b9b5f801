# round 6: a longer randomised parity campaign on the final library (overlap-add tile included) + EQ fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
for seed in 611 612 613 614 615 616; do timeout 400 python tools/fuzz_parity.py $seed 300 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600; done | tee gpurun_out/r6p/fuzz_parity.txt
timeout 400 python tools/fuzz_eq.py 62 300 2>&1 | tail -2 | tee gpurun_out/r6p/fuzz_eq.txt
