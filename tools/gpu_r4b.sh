cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in f g h i; do echo $v; ./tools/ubench/rb_$v; done; echo j; ./tools/ubench/rb_j 4; echo k; ./tools/ubench/rb_k 2
./tools/ubench/rb_f_st
} 2>&1 | tee gpurun_out/r4b/out7.txt
