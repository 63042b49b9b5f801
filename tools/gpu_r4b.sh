cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
{
for v in st40 st40_nofft st40_nonly; do echo $v; ./tools/ubench/rb_$v; done
} 2>&1 | tee gpurun_out/r4b/out26.txt
