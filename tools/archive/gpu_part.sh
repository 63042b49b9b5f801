#!/bin/bash
# partitioned path check: parity tests touching it, then cfg4 / cfg3 / cfg5 benches (4 s per stream)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mixed_rate.py tests/test_gpu_reference_kats.py -x -q -m gpu 2>&1 | tail -3
for w in cfg4 cfg3 cfg5; do
  python bench.py --workload $w --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$w', round(d['value']/1e9,3),'Gframes/s', round(d['ms_per_step'],2),'ms/step kernels',round(r['kernel_avg_ms'],2), 'eq', r.get('eq_kernel_ms_per_step'))"
done
