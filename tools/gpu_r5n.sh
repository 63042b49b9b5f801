cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_process_path_contract.py tests/test_gpu_host_entry.py -m gpu -x -q 2>&1 | tail -5
for seed in 611 612 613; do timeout 400 python tools/fuzz_parity.py $seed 200 2>&1 | grep -v amdgpu.ids | tail -3; done | tee gpurun_out/r5n/fuzz.txt
