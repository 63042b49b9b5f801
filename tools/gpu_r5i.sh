cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
for rep in 1 2; do for v in 14g8 14g4 14g16 14late 14lateg4 14lateg16; do echo -n "$v: "; ./tools/ubench/tb_$v 0.3; done; done 2>&1 | tee gpurun_out/r5i/tile_bench_14.txt
timeout 200 python -X faulthandler tools/fuzz_eq.py 31 20 2>&1 | tail -3; echo "fuzz_eq rc $?"
