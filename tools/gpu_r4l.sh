cd $GRAFT_REPO_ROOT
export AW_LW=0
TAPS=1500,1800,2000,2200,2500,3000 python tools/window_sweep.py 2
TAPS=1500,1800,2000,2300,2600,3000 python tools/window_sweep.py 3
TAPS=3000,3200,3500,3800,4100,4320 python tools/window_sweep.py 5
TAPS=4320,4500,4700,5000,5300,5600 python tools/window_sweep.py 7
TAPS=5600,5900,6145 python tools/window_sweep.py 6
TAPS=5600,5900,6145 python tools/window_sweep.py 8
