cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r4g/pytest_gpu.txt
for seed in 501 502; do timeout 400 python tools/fuzz_parity.py $seed 240 2>&1 | tail -3 | tee -a gpurun_out/r4g/fuzz.txt; done
for w in cfg2 cfg5 cfg1 cfg2-14ch; do WORKLOAD=$w STEPS=20 bash tools/ab_bench.sh olsx - 2>&1 | tee -a gpurun_out/r4g/ab.txt; done
