# rocm-smi power / sclk every 0.25 s while a command runs:  bash tools/archive/power_trace.sh <out prefix> <command ...>
# (the sampler is a separate process; nothing of it runs inside the measured command)
OUTP=$1; shift
SMI=$(command -v rocm-smi || echo /opt/rocm/bin/rocm-smi)
rm -f $OUTP.stop
( while [ ! -e $OUTP.stop ]; do echo "t $(date +%s.%N)"; $SMI --showpower --showclocks 2>&1 | grep -E "Power|sclk"; sleep 0.25; done ) > $OUTP.smi.txt 2>&1 &
S=$!
"$@" > $OUTP.out.txt 2> $OUTP.err.txt
RC=$?
touch $OUTP.stop; wait $S 2>/dev/null; rm -f $OUTP.stop
python3 - $OUTP <<'PY'
import re, statistics, sys
p = sys.argv[1]
pw = [float(m.group(1)) for m in re.finditer(r"Power[^:]*:\s*([0-9.]+)", open(p + ".smi.txt").read())]
sc = [float(m.group(1)) for m in re.finditer(r"sclk[^(]*\((\d+)Mhz\)", open(p + ".smi.txt").read())]
hi = [v for v in pw if v > 600]
print(f"{p}: samples {len(pw)}, under load {len(hi)}: power median {statistics.median(hi) if hi else None} W max {max(pw) if pw else None} W; sclk median under load {statistics.median([s for s, w in zip(sc, pw) if w > 600]) if hi else None} MHz")
PY
exit $RC
