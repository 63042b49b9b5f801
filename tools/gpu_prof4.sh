cd $GRAFT_REPO_ROOT
for w in cfg3 cfg3-14ch cfg4 cfg2-14ch cfg2; do
  bash tools/profile_round4.sh round4_v1 $w > gpurun_out/prof4_$w.log 2>&1
  tail -12 gpurun_out/prof4_$w.log
done
