cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for sw in "10 2" "30 10" "10 2" "30 10" "60 20"; do set -- $sw
python bench.py --no-cpu-baseline --no-secondary --steps $1 --warmup $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps $1 warmup $2:', round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],3),'ms', {k.replace('aw_lw_','').replace('_kernel',''):round(v,2) for k,v in r['stages_ms_per_step'].items()})"
done 2>&1 | tee gpurun_out/r4b/out20.txt
