# round 6: overlap-add tile variants (tools/ubench/ola_bench.hip) against the overlap-save tile (tile_bench.hip); timing only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
{
for b in "$@"; do
  [ -x tools/ubench/$b ] && { echo -n "$b: "; timeout 120 tools/ubench/$b 0.3 | tail -1; }
done
} 2>&1 | tee gpurun_out/r6b/ola_variants.txt
