# EQ kernel: GPU tests + timing probe at the cfg 4 shape and two other batch sizes
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_eq.py -x -q 2>&1 | tail -3
python tools/eq_probe.py 512 960000 2>&1 | tail -1
python tools/eq_probe.py 256 960000 2>&1 | tail -1
python tools/eq_probe.py 128 960000 2>&1 | tail -1
python tools/eq_probe.py 2048 240000 2>&1 | tail -1
