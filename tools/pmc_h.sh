# PMC passes for the two fused 8192-window kernels: tools/pmc_h.sh  (AW_KERNEL_H=0 then 1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for k in 0 1; do
export AW_KERNEL_H=$k
rm -rf gpurun_out/pmch$k; mkdir -p gpurun_out/pmch$k
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --pmc $line --kernel-trace --output-format csv -d gpurun_out/pmch$k/pmc_$i -- $B > gpurun_out/pmch$k/pmc_$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_CYCLES
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
LIST
python3 tools/pmc_summary.py gpurun_out/pmch$k > gpurun_out/pmch$k/summary.txt 2>&1
done
