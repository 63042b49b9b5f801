cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for e in "" "AW_EQ_EAR_SPLIT=0" "AW_EQ_EAR_SPLIT=1"; do
env $e python bench.py --workload cfg4 --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e:', round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],3),'ms', {k.replace('aw_','').replace('_kernel',''):round(v,2) for k,v in r['stages_ms_per_step'].items()})"
done 2>&1 | tee gpurun_out/r4b/out21.txt
