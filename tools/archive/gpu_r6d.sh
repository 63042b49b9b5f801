cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
timeout 1700 python -m pytest tests/ -m gpu -q -p no:cacheprovider --tb=line > gpurun_out/r6d/pytest_gpu_full.txt 2>&1; echo "rc $?"
tail -30 gpurun_out/r6d/pytest_gpu_full.txt | cut -c1-400
