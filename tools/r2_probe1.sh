# round 2, first GPU call: cache behaviour probe + today's cfg3 baseline + counter list
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p1
./tools/ubench/mall_probe > gpurun_out/r2p1/mall_probe.txt 2>&1
python bench.py --workload cfg3 --seconds 4 --no-cpu-baseline > gpurun_out/r2p1/cfg3_4s.json 2> gpurun_out/r2p1/cfg3_4s.err
python tools/copy_ceiling.py > gpurun_out/r2p1/copy.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r2p1/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r2p1/mall_probe.txt gpurun_out/r2p1/copy.txt
tail -c 1500 gpurun_out/r2p1/cfg3_4s.json
