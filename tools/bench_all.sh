# every BASELINE workload at full duration on the current build -> gpurun_out/bench_all/<workload>.json (copy to profiles/<name>/other_workloads/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_all; rm -rf $O; mkdir -p $O
for w in cfg1 cfg2 cfg3-14ch cfg4 cfg5; do
  python bench.py --workload $w > $O/$w.json 2> $O/$w.err
  python -c "
import json,sys
d=json.loads(open('$O/$w.json').readlines()[-1]); r=d['roofline']
print('$w', round(d['value']/1e9,2),'G frames/s', round(d['ms_per_step'],3),'ms/step frac',round(r['frac'],4),'kernel_frac',round(r['kernel_frac'],4), 'cpu', round(d.get('cpu_baseline',{}).get('value',0)/1e6,1),'M/s')"
done
