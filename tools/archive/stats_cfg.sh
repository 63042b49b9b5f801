#!/bin/bash
# per-kernel time of one bench workload: tools/archive/stats_cfg.sh cfg4 [seconds]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
W=${1:-cfg4}; S=${2:-4}
rm -rf gpurun_out/stats_$W; mkdir -p gpurun_out/stats_$W
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$W -- python3 bench.py --workload $W --seconds $S --no-cpu-baseline > gpurun_out/stats_$W/log.txt 2>&1
python3 - "$W" <<'PY'
import csv,glob,sys
w=sys.argv[1]
for f in glob.glob(f"gpurun_out/stats_{w}/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
