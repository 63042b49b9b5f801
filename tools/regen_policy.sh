# Re-measures every crossover behind the path / window policy of aw_spatializer_create (runtime.cpp) on the box it runs on, in one
# command: bash tools/regen_policy.sh > gpurun_out/policy.txt   (≈5 GPU-minutes).  Compare with the thresholds in runtime.cpp.
cd $GRAFT_REPO_ROOT
echo "== 8192- vs 16384-frame windows by HRIR length (128 streams) =="
for c in 1 2 3 4 5 6 7 8; do python tools/window_sweep.py $c; done
echo "== 16384-frame windows vs partitioned path (AW_WINDOW=4096) at the long end =="
TAPS=6146,8000,9000,10000,11000,12289 WINS=16384,4096 python tools/path_sweep.py 1 2 3 4 5 6 7 8
echo "== 8192-frame windows vs partitioned path for 9-16 channels =="
TAPS=3000,4320,5300,5800,6145 WINS=8192,4096 python tools/path_sweep.py 9 10 12 14 16
echo "== small batches: 8192 vs 16384 windows by stream count =="
python tools/small_batch_sweep.py
echo "== long-window kernels against the fused default by HRIR length (path 0; feeds lw_fused_crossover_taps) =="
python tools/lw_sweep.py 1 2 3 4 5 6 7 8
TAPS=3000,4320,6145 python tools/lw_sweep.py 9 10 12 14 16
echo "== long HRIRs: partitioned kernels against the per-call policy by call length =="
python tools/lw_calls_sweep.py 7
S=16 python tools/lw_calls_sweep.py 2
