cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in base early plain ntst0; do echo $v; ./tools/ubench/rb_$v; done; ./tools/ubench/rb_early_st
} 2>&1 | tee gpurun_out/r4b/out10.txt
