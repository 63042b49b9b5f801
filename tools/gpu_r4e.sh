cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
for seed in 401 402 403; do timeout 400 python tools/fuzz_parity.py $seed 240 2>&1 | grep -v amdgpu.ids | tail -4; done | tee gpurun_out/r4e/fuzz.txt
