# round 5: the N > 1 code path of bench.py on a one-GPU box (two ranks on one GPU over gloo; a forced one-rank RCCL group) — no scaling claim
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
AW_BENCH_DEVICE=0 AW_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --no-cpu-baseline > gpurun_out/r5r/two_ranks_one_gpu_gloo.json 2> gpurun_out/r5r/two_ranks.err; echo "two ranks rc $?"
tail -3 gpurun_out/r5r/two_ranks.err
AW_BENCH_FORCE_PG=1 timeout 600 python bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r5r/rccl_one_rank_nccl.json 2> gpurun_out/r5r/rccl_one_rank.err; echo "forced pg rc $?"
grep -i "nccl\|rccl" gpurun_out/r5r/rccl_one_rank.err | head -5
python - <<'PY'
import json
for f in ("two_ranks_one_gpu_gloo","rccl_one_rank_nccl"):
    d=json.loads(open(f"gpurun_out/r5r/{f}.json").read().strip().splitlines()[-1])
    print(f, "n_gpus", d["n_gpus"], "ranks_seen", d["ranks_seen"], round(d["value"]/1e9,2), "G/s", d["roofline"]["frac"], "secondary" in d, "e2e" in d and len(d.get("secondary_end_to_end",[])))
PY
