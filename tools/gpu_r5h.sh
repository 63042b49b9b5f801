cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
for v in 8 8noload 8notab 14 14noload 14notab 14nocmac 12 16; do ./tools/ubench/tb_$v 0.3; done 2>&1 | tee gpurun_out/r5h/tile_bench.txt
timeout 200 python -X faulthandler tools/fuzz_eq.py 31 30 2>&1 | tail -3; echo "fuzz_eq rc $?"
