cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
for s in 505 506 507; do timeout 400 python tools/fuzz_parity.py $s 240 2>&1 | tail -2 | tee -a gpurun_out/r4m/fuzz.txt; done
timeout 400 python tools/fuzz_eq.py 21 200 2>&1 | tail -2 | tee -a gpurun_out/r4m/fuzz_eq.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 > gpurun_out/r4m/bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4m/bench.json')); print(round(d['value']/1e9,2), d['roofline']['frac'], d['secondary']['value']/1e9, d.get('parity_spot_err'))"
