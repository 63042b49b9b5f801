cd $GRAFT_REPO_ROOT
export AW_LW=0
TAPS=1200,1500,1800,2000,2200 python tools/window_sweep.py 2
TAPS=1500,1800,2000,2300,2600 python tools/window_sweep.py 3
TAPS=3000,3200,3500,3800,4100 python tools/window_sweep.py 5
TAPS=4000,4320,4500,4700,5000 python tools/window_sweep.py 7
TAPS=4320,4700,5000,5300,5600,5900 python tools/window_sweep.py 4
TAPS=4320,4700,5000,5300,5600,5900 python tools/window_sweep.py 6
TAPS=4320,4700,5000,5300,5600,5900 python tools/window_sweep.py 8
unset AW_LW
echo "== lw sweep =="
TAPS=9000,10000,10500,11000,12000 python tools/lw_sweep.py 1
TAPS=8000,8500,9000,9500,10000 python tools/lw_sweep.py 2
TAPS=7600,8100,8600,9100,9600 python tools/lw_sweep.py 3
TAPS=6300,6800,7300,7800,8300 python tools/lw_sweep.py 5
TAPS=4700,5000,5300,5600,6000,6500,7000 python tools/lw_sweep.py 4 6 7 8
