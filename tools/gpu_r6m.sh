# round 6: the N > 1 code path of the final bench.py on a one-GPU box (no scaling claim): two ranks on one GPU over gloo, weak and strong;
# a forced one-rank process group on the nccl (= RCCL) backend; then the driver's own command line.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
AW_BENCH_DEVICE=0 AW_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --streams 256 --steps 6 --warmup 2 > $O/two_ranks_one_gpu_gloo.json 2> $O/two_ranks_one_gpu_gloo.err; echo "weak rc $?"
AW_BENCH_DEVICE=0 AW_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling strong --streams 301 --steps 6 --warmup 2 --no-secondary > $O/two_ranks_one_gpu_gloo_strong.json 2> $O/two_ranks_one_gpu_gloo_strong.err; echo "strong rc $?"
AW_BENCH_FORCE_PG=1 timeout 600 python bench.py --steps 6 --warmup 2 > $O/rccl_one_rank_nccl.json 2> $O/rccl_one_rank_nccl.stderr.txt; echo "rccl rc $?"
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc $? in ${SECONDS}s, stdout bytes $(wc -c < $O/bench_driver_style.json)"
python - <<'PY'
import json
for n in ("two_ranks_one_gpu_gloo", "two_ranks_one_gpu_gloo_strong", "rccl_one_rank_nccl", "bench_driver_style"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r6m/{n}.json") if l.startswith("{")][-1])
        c = d["config"]
        print(n, "n_gpus", d["n_gpus"], "ranks", d.get("ranks_seen"), d["scaling"], "streams total / this rank", c.get("streams_total"), c.get("streams_this_rank"), round(d["value"] / 1e9, 2), "G/s frac", d["roofline"]["frac"],
              "traffic", d["roofline"].get("traffic"), {k: round(v["value"] / 1e9, 2) for k, v in d.items() if k.startswith("secondary") and isinstance(v, dict)})
    except Exception as e:
        print(n, "FAILED", e)
PY
