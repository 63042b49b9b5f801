# 8192- vs 16384-frame fused windows: parity subset under AW_WINDOW=16384, then the default bench with both.
cd $GRAFT_REPO_ROOT
AW_WINDOW=16384 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "channel_count or interior or ragged or full_size or planar or cfg1_full" 2>&1 | tail -3
for rep in 1 2; do for w in 8192 16384; do
  AW_WINDOW=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('window=$w', round(d['value']/1e9,3), 'Gframes/s', round(d['ms_per_step'],4), 'ms/step kernel', round(r['kernel_avg_ms'],4), 'frac', round(r['frac'],4), d['config']['hop'])"
done; done
