# round 5, GPU call: the round profile of every workload (bench line, rocprofv3 --kernel-trace --stats, separate --pmc passes)
#   bash tools/archive/gpu_r5k.sh [name, default round5_v1]
cd $GRAFT_REPO_ROOT
NAME=${1:-round5_v1}
for W in cfg3 cfg2 cfg4 cfg2-14ch cfg3-14ch cfg5 cfg1; do
  timeout 900 bash tools/archive/profile_round5.sh $NAME $W 2>&1 | tail -12
done
mkdir -p gpurun_out/profiles_$NAME && cp -r profiles/$NAME/* gpurun_out/profiles_$NAME/
