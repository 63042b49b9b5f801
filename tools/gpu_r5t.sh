cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(round(d['value']/1e9,2), d['roofline']['frac'], 'act', d['config']['activation'], 'warm', d['config']['activation_warm'])
print(' sec act', d['secondary']['config']['activation'], d['secondary']['config'].get('activation_warm'))"
python bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('cfg4', round(d['value']/1e9,2), d['roofline']['frac'], 'act', d['config']['activation'], 'warm', d['config']['activation_warm'])"
