# round 5, GPU call 4: how long does the clock governor take to settle?  same box, same binary, warm-up / step counts varied
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
run() { python bench.py --workload $1 --warmup $2 --steps $3 --no-cpu-baseline --no-secondary --no-end-to-end --no-ceiling --no-warm-activation 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 warmup $2 steps $3:', round(d['value']/1e9,2), 'G/s  frac', round(r['frac'],4), ' ms/step', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
for ws in "3 20" "20 50" "100 200" "400 400" "1000 1000"; do set -- $ws; run cfg2 $1 $2; done
for ws in "2 10" "10 30" "30 60" "100 100"; do set -- $ws; run cfg3 $1 $2; done
for ws in "1 5" "10 20" "40 40"; do set -- $ws; run cfg4 $1 $2; done
for ws in "3 20" "100 200" "400 400"; do set -- $ws; run cfg2-14ch $1 $2; done
for ws in "1 5" "10 20" "40 40"; do set -- $ws; run cfg5 $1 $2; done
done 2>&1 | tee gpurun_out/r5d/warmup_study.txt
