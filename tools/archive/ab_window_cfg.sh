cd $GRAFT_REPO_ROOT
for w in 8192 16384; do for wl in cfg4 cfg5; do
  AW_WINDOW=$w python bench.py --workload $wl --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('window=$w $wl', round(d['value']/1e9,3), 'Gframes/s', round(d['ms_per_step'],3), 'ms/step kernels', round(r['kernel_avg_ms'],3), [(l['rate'],l['path'],l['hop']) for l in d['config']['legs']])"
done; done
