# Round-5 profile (same recipe as rounds 2 to 4; the traffic file now records a digest of csrc/device and the build's git HEAD) of the default bench (cfg 3): tools/archive/profile_round5.sh <name> [workload]
#   1. the bench line itself (with CPU baseline + secondary)      -> profiles/<name>/bench_<workload>.json
#   2. rocprofv3 --kernel-trace --stats of the same command        -> profiles/<name>/kernel_stats_<workload>.csv
#   3. separate --pmc passes (TCC_EA0 read/write requests, SQ)     -> profiles/<name>/traffic_<workload>.json (tools/profile_collect2.py)
# Counter passes never share a run with --stats or a trace domain other than --kernel-trace.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=${1:-round5_v1}; W=${2:-cfg3}
OUT=gpurun_out/profile_$NAME/$W
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --workload $W > $OUT/bench.json 2> $OUT/bench.err
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-secondary --no-ceiling --no-end-to-end --no-warm-activation"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- $B > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1
B="$B --steps 2 --warmup 1"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_rd -- $B > $GRAFT_REPO_ROOT/$OUT/pmc_rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_wr -- $B > $GRAFT_REPO_ROOT/$OUT/pmc_wr.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq -- $B > $GRAFT_REPO_ROOT/$OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_grbm -- $B > $GRAFT_REPO_ROOT/$OUT/pmc_grbm.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/profile_collect2.py $NAME $W
