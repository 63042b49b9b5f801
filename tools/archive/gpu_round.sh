#!/bin/bash
# Full GPU check: the gpu-marked tests, then every bench workload (short runs for cfg3-5).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for w in cfg2 cfg3 cfg4 cfg5; do
  extra=""; [ $w != cfg2 ] && extra="--seconds ${SECS:-4}"
  timeout 900 python bench.py --workload $w $extra 2>gpurun_out/bench_$w.err | tail -1 > gpurun_out/bench_$w.json
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/bench_{w}.json").read())
    r=d["roofline"]; c=d.get("cpu_baseline",{})
    print(w, f"{d['value']/1e9:.3f} Gframes/s  {d['ms_per_step']:.2f} ms/step  kernel {r['kernel_avg_ms']:.3f} ms  frac {r['frac']:.4f}", r.get("eq_kernel_ms_per_step"), [ (l['rate'],l['path'],l['partitions']) for l in d['config']['legs']], "cpu", f"{c.get('value',0)/1e6:.2f} Mframes/s x{c.get('cores')}")
except Exception as e:
    print(w, "FAILED", e); print(open(f"gpurun_out/bench_{w}.err").read()[-1500:])
PY
done
