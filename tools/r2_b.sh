cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "long or partition or chunk or ragged or policy" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline --no-secondary > $O/cfg3.json 2> $O/cfg3.err; python -c "
import json; d=json.loads(open('$O/cfg3.json').readlines()[-1]); r=d['roofline']; print(round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],2),'ms frac',round(r['frac'],4),'kfrac',round(r['kernel_frac'],4), r['stages_ms_per_step'])"
AW_PART_HERM=0 python bench.py --no-cpu-baseline --no-secondary > $O/cfg3_noherm.json 2> $O/cfg3_noherm.err; python -c "
import json; d=json.loads(open('$O/cfg3_noherm.json').readlines()[-1]); r=d['roofline']; print('noherm', round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],2),'ms', r['stages_ms_per_step'])"
python -m pytest tests/test_gpu_full_size.py -q -x -m gpu > $O/pytest_full.log 2>&1; tail -5 $O/pytest_full.log
