# round 6, GPU call: the GPU suite, then the round profile of every workload (bench line, rocprofv3 --kernel-trace --stats, separate --pmc passes)
#   bash tools/gpu_r6k.sh [name, default round6_v6]
cd $GRAFT_REPO_ROOT
NAME=${1:-round6_v6}
mkdir -p gpurun_out/r6k
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -15 | cut -c1-300 | tee gpurun_out/r6k/pytest_gpu.txt
for W in cfg3 cfg2 cfg2-14ch cfg4 cfg3-14ch cfg5 cfg1; do
  timeout 900 bash tools/profile_round6.sh $NAME $W 2>&1 | tail -14 | cut -c1-400
done
mkdir -p gpurun_out/profiles_$NAME && cp -r profiles/$NAME/* gpurun_out/profiles_$NAME/
