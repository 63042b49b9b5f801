# PMC passes (separate runs, counters only + kernel trace) of one bench workload: tools/archive/pmc_cfg.sh <workload> <outdir> [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cfg3}; O=${2:-gpurun_out/pmc_$W}; shift; shift
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline $*"
rm -rf $O; mkdir -p $O
i=0
cd /tmp
while read -r line; do
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$i -- $B > $GRAFT_REPO_ROOT/$O/pmc_$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
GRBM_GUI_ACTIVE
LIST
cd $GRAFT_REPO_ROOT
python3 tools/archive/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
