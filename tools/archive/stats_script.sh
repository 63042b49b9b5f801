# per-kernel rocprofv3 stats of any python script: tools/archive/stats_script.sh <outdir> <script> [args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; rm -rf $O; mkdir -p $O
S=$GRAFT_REPO_ROOT/$1; shift
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $S "$@" > $O/log.txt 2>&1)
python3 - "$O" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  {r['Percentage']}%")
PY
