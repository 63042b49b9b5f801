cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
for w in cfg2-14ch cfg4; do for e in "" "AW_LW=128" "AW_LW=112" "AW_LW=96"; do
env $e python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$w $e:', round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],3),'ms', {k.replace('aw_','').replace('_kernel',''):round(v,2) for k,v in r['stages_ms_per_step'].items()}, d['config']['fft'])"
done; done 2>&1 | tee gpurun_out/r4b/out22.txt
