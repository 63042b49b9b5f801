cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
for seed in 621 622 623 624 625; do timeout 400 python tools/fuzz_parity.py $seed 200 2>&1 | grep -v amdgpu.ids | tail -3; done | tee gpurun_out/r5s/fuzz.txt
timeout 300 python tools/fuzz_eq.py 51 150 2>&1 | tail -2 | tee gpurun_out/r5s/fuzz_eq.txt
