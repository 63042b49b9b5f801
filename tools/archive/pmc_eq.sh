#!/bin/bash
# PMC counters of the EQ cascade kernel (separate passes, counters only + kernel trace)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_eq; mkdir -p gpurun_out/pmc_eq
B="python3 tools/eq_probe.py 512 960000"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_eq/a -- $B > gpurun_out/pmc_eq/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/pmc_eq/b -- $B > gpurun_out/pmc_eq/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d gpurun_out/pmc_eq/c -- $B > gpurun_out/pmc_eq/c.log 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_eq/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "eq_cascade" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
