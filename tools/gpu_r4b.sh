cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in 0 1 2 3 5 7 0; do echo "prio $v"; ./tools/ubench/tb_p$v; done
} 2>&1 | tee gpurun_out/r4b/out14.txt
