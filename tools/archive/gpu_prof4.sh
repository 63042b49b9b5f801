cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-cfg3}; do
  bash tools/archive/profile_round4.sh round4_v1 $w > gpurun_out/prof4_$w.log 2>&1
  tail -8 gpurun_out/prof4_$w.log
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
