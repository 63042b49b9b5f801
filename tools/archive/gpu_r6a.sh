# round 6, first GPU call: the overlap-add tile against the overlap-save tile (microbenchmarks), then the default bench line (compact stdout line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
{
for b in tb_8 ola_8 ola_8e tb_14 ola_14 ola_14e tb_7x4 ola_7 ola_2; do
  [ -x tools/ubench/$b ] && { echo -n "$b: "; timeout 120 tools/ubench/$b 0.3 | tail -1; }
done
for b in ola_8 ola_14; do echo -n "$b again: "; timeout 120 tools/ubench/$b 0.3 | tail -1; done
} 2>&1 | tee gpurun_out/r6a/ola_bench.txt
SECONDS=0; python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench_default.json 2> gpurun_out/r6a/bench_default.err; echo "bench rc $? in ${SECONDS}s, line bytes $(wc -c < gpurun_out/r6a/bench_default.json)"
cp bench_detail.json gpurun_out/r6a/ 2>/dev/null
tail -c 600 gpurun_out/r6a/bench_default.err
