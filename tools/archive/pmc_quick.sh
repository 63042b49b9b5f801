# fabric-side read/write bytes per kernel for library variants on one bench workload: WORKLOAD=cfg2 tools/archive/pmc_quick.sh <variant|-> ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${WORKLOAD:-cfg3}
for v in "$@"; do
  lib=""; [ "$v" != "-" ] && lib=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_$v.so
  O=$GRAFT_REPO_ROOT/gpurun_out/pmcq_${W}_$v; rm -rf $O; mkdir -p $O
  (cd /tmp && AIRWAVE_HIP_LIBRARY=$lib rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-secondary --steps 2 --warmup 1 > $O/log.txt 2>&1)
  python3 - "$v" "$O" <<'PY'
import csv,glob,sys,collections
v,o=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(list)
for f in glob.glob(o+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void awk::","").replace("awk::","")
        if k.startswith("aw_part") or k.startswith("aw_fused") or k.startswith("aw_eq") or k.startswith("aw_lw"): agg[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),vals in sorted(agg.items()):
    print(f"{v:8s} {k:44s} {c:20s} n={len(vals):3d} {sum(vals)/3*(128 if 'RD' in c else 64)/1e9:8.3f} ~GB/step (approximate: every read request counted as 128 B, every write as 64 B, 3 steps assumed; tools/profile_collect2.py sizes requests by class)")
PY
done
