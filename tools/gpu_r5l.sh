cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
for rep in 1 2 3; do for v in 8 8warm 6x3 6x3warm; do echo -n "$v: "; ./tools/ubench/tb_$v 0.3; done; done 2>&1 | tee gpurun_out/r5l/tile_bench_warm.txt
