cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
timeout 1500 python tools/ola_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6e/ola_sweep.txt
