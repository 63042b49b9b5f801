cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
for rep in 1 2 3; do for v in 8 8lanepair; do echo -n "$v: "; ./tools/ubench/tb_$v 0.3; done; done 2>&1 | tee gpurun_out/r5q/tile_bench_lanepair.txt
