cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 300 python -X faulthandler tools/fuzz_eq.py 31 120 > gpurun_out/r5g/fuzz_eq31.txt 2>&1; echo "rc $?"
tail -30 gpurun_out/r5g/fuzz_eq31.txt
timeout 300 python -X faulthandler tools/fuzz_eq.py 21 60 > gpurun_out/r5g/fuzz_eq21.txt 2>&1; echo "rc $?"
tail -5 gpurun_out/r5g/fuzz_eq21.txt
for W in "" "AW_WIDE_TWO_PASS=1"; do
  env AW_LW=0 $W python bench.py --workload cfg2-14ch --no-cpu-baseline --no-end-to-end --no-ceiling --no-warm-activation 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg2-14ch fused tile AW_LW=0 $W:', round(d['value']/1e9,2), 'G/s frac', round(r['frac'],4), r['kernel'], round(r['kernel_avg_ms'],4))"
done
python bench.py --workload cfg2-14ch --no-cpu-baseline --no-end-to-end --no-ceiling --no-warm-activation 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg2-14ch default (long-window kernels):', round(d['value']/1e9,2), 'G/s frac', round(r['frac'],4), r['stages_ms_per_step'])"
timeout 600 python -m pytest tests/test_process_path_contract.py tests/test_gpu_longwin.py -m gpu -x -q 2>&1 | tail -3
