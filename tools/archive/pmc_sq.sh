# SQ counters of the dominant kernels of one bench workload: WORKLOAD=cfg2 tools/archive/pmc_sq.sh [variant|-]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${WORKLOAD:-cfg2}; v=${1:--}
lib=""; [ "$v" != "-" ] && lib=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_$v.so
O=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_${W}_$v; rm -rf $O; mkdir -p $O
B="python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  (cd /tmp && AIRWAVE_HIP_LIBRARY=$lib rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$i -- $B > $O/log$i.txt 2>&1)
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void awk::","").replace("awk::","")
        if k.startswith("aw_part") or k.startswith("aw_fused") or k.startswith("aw_eq"): agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),vals in sorted(agg.items()):
    print(f"{k:44s} {c:26s} n={len(vals):3d} avg={sum(vals)/len(vals):16.0f}")
PY
