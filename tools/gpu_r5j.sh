cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
for v in 8 14x4 14x4noload 12x4 16x4 6x3 7x4; do echo -n "$v: "; ./tools/ubench/tb_$v 0.3; done 2>&1 | tee gpurun_out/r5j/tile_bench_narrow.txt
