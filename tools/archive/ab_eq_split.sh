# A/B of one workgroup per stream (E = 2) against one per (stream, ear) (E = 1): tools/archive/ab_eq_split.sh [streams ...]
cd $GRAFT_REPO_ROOT
for S in ${@:-128 256 384 512 1024}; do for sp in 0 1; do
  echo -n "streams=$S split=$sp: "; AW_EQ_EAR_SPLIT=$sp python tools/eq_probe.py $S 960000 2>/dev/null | tail -1
done; done
