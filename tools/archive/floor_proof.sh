# The tracked evidence behind DESIGN §4.5b's "the rows kernel's memory part and compute part add up" (round-4 review: the claim rested on
# untracked scratch): every timing ablation of tools/ubench/rows_bench and the read-only probe of its access shape (tools/ubench/stream_probe),
# each after >= 2.5 s of back-to-back launches, with the IN-KERNEL clock of the last launch (s_memtime / s_memrealtime) and a rocm-smi
# power / sclk sample every 0.25 s while it runs.   usage (GPU box): bash tools/archive/floor_proof.sh <out dir>
#   build first (container): see the header of tools/ubench/rows_bench.hip; variants rb_full rb_nofft rb_noload rb_notab rb_nostore rb_nomem rb_form8
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/floor}
mkdir -p $OUT
SMI=$(command -v rocm-smi || echo /opt/rocm/bin/rocm-smi)
sample() {          # $1 = tag: power + clocks every 0.25 s until the file $OUT/.stop appears
    rm -f $OUT/.stop
    ( while [ ! -e $OUT/.stop ]; do echo "t $(date +%s.%N)"; $SMI --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" ; sleep 0.25; done ) > $OUT/smi_$1.txt 2>&1 &
    SAMPLER=$!
}
stop() { touch $OUT/.stop; wait $SAMPLER 2>/dev/null; rm -f $OUT/.stop; }
$SMI --showpower --showclocks --showperflevel --showmaxpower 2>&1 | head -40 > $OUT/smi_idle.txt
for v in full nomem noload notab nostore nofft form8; do
    sample $v
    timeout 120 ./tools/ubench/rb_$v 3 1024 2.5 > $OUT/rows_$v.txt 2>&1
    stop
    tail -2 $OUT/rows_$v.txt
done
sample probe
timeout 120 ./tools/ubench/stream_probe 3 1 2.5 > $OUT/stream_probe.txt 2>&1
stop
cat $OUT/stream_probe.txt
python3 tools/archive/floor_summary.py $OUT
