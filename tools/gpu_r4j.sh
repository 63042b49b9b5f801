cd $GRAFT_REPO_ROOT
for w in cfg3 cfg3-14ch cfg4 cfg2-14ch; do bash tools/profile_round4.sh round4_v2 $w 2>&1 | tail -2 | cut -c1-220; done
