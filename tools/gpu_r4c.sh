cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 2400 python -m pytest tests/test_gpu_longwin.py -x -q -k "two_windows_then" 2>&1 | tail -12 | tee gpurun_out/r4c/pytest_gpu.txt
