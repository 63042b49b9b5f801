# round 2: partitioned path with the marched CMAC kernel — parity tests, cfg3 bench, per-kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2part; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "long or partition or chunk or ragged or policy" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --workload cfg3 --seconds 4 --no-cpu-baseline > $O/cfg3_4s.json 2> $O/cfg3_4s.err; tail -c 600 $O/cfg3_4s.json; echo
python bench.py --workload cfg3 --no-cpu-baseline > $O/cfg3_10s.json 2> $O/cfg3_10s.err; tail -c 600 $O/cfg3_10s.json; echo
AW_PART_CMAC=group python bench.py --workload cfg3 --seconds 4 --no-cpu-baseline > $O/cfg3_4s_group.json 2> $O/cfg3_4s_group.err; tail -c 300 $O/cfg3_4s_group.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -12'
