cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_eq.py -x -q -m gpu 2>&1 | tail -1
for split in 0 1; do for n in "128 480000" "512 960000" "2048 240000"; do echo -n "split=$split $n: "; AW_EQ_EAR_SPLIT=$split python tools/eq_probe.py $n 2>/dev/null | tail -1; done; done
AW_EQ_EAR_SPLIT=1 python -m pytest tests/test_gpu_eq.py -x -q -m gpu 2>&1 | tail -1
