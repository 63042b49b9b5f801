cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in base deep base deep; do echo $v; ./tools/ubench/rb_$v; done
} 2>&1 | tee gpurun_out/r4b/out23.txt
