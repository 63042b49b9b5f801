# round 5: does a scratch that fits the 256 MiB Infinity Cache (the batch in stream chunks, split -> rows -> merge per chunk) pay on cfg 3?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5u
for mb in 0 200 400 800 1600 3200 6400; do
  X=""; [ $mb != 0 ] && X="AW_SPEC_SCRATCH_MB=$mb"
  env $X python bench.py --no-cpu-baseline --no-secondary --no-end-to-end --no-ceiling --no-warm-activation 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('scratch budget $mb MB (0 = default, one chunk):', round(d['value']/1e9,2), 'G/s', round(d['ms_per_step'],3), 'ms', r['stages_ms_per_step'], 'pool', d['config']['activation'][0]['scratch_bytes'])"
done 2>&1 | tee gpurun_out/r5u/scratch_sweep_cfg3.txt
