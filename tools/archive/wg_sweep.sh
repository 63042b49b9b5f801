# cfg 2 against the number of persistent workgroups (L2 footprint of the tiles in flight): tools/archive/wg_sweep.sh
cd $GRAFT_REPO_ROOT
for w in 256 248 240 224 192; do
  AW_PERSISTENT_WGS=$w python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgs $w', round(d['value']/1e9,2), 'Gframes/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['kernel_avg_ms'],4), 'ms kernel')"
done
