# round 5, GPU call: the round profile of every workload (bench line, rocprofv3 --kernel-trace --stats, separate --pmc passes)
cd $GRAFT_REPO_ROOT
for W in cfg3 cfg2 cfg4 cfg2-14ch cfg3-14ch cfg5 cfg1; do
  timeout 900 bash tools/profile_round5.sh round5_v1 $W 2>&1 | tail -12
done
mkdir -p gpurun_out/profiles_round5_v1 && cp -r profiles/round5_v1/* gpurun_out/profiles_round5_v1/
