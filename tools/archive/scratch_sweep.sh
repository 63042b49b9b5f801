cd $GRAFT_REPO_ROOT
for mb in 6144 1024 256 128 64; do
  echo -n "scratch=$mb MB: "; AW_SPEC_SCRATCH_MB=$mb python bench.py --workload cfg4 --seconds 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(round(d['value']/1e9,3),'Gframes/s', round(d['ms_per_step'],2),'ms/step cmac',round(r['kernel_avg_ms'],2),'eq',round(r['eq_kernel_ms_per_step'],2))"
done
