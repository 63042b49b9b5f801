# round 6: a longer randomised parity campaign on the final library (overlap-add tile incl. the odd wide layouts) + EQ fuzz
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
for seed in 621 622 623 624; do timeout 400 python tools/fuzz_parity.py $seed 300 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-600; done | tee gpurun_out/r6p/fuzz_parity_final.txt
timeout 300 python tools/fuzz_eq.py 63 200 2>&1 | tail -2 | tee gpurun_out/r6p/fuzz_eq_final.txt
