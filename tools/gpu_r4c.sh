cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r4c/pytest_gpu.txt
