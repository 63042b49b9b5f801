# quick GPU iteration: parity tests + bench + two PMC passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_a -- $B > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT --kernel-trace --output-format csv -d gpurun_out/pmc_b -- $B > gpurun_out/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_c -- $B > gpurun_out/pmc_c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_d -- $B > gpurun_out/pmc_d.log 2>&1
