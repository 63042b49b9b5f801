cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in plain not1 new core new; do echo $v; ./tools/ubench/rb_$v; done
} 2>&1 | tee gpurun_out/r4b/out13.txt
