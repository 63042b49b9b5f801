"""World-size-2 (gloo, CPU) coverage of the N>1 path: stream sharding, global stream ids for the
synthetic input, and the final {sum frames, max elapsed} aggregate that bench.py performs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from airwave_amd.sharding import aggregate_throughput, plan_streams, shard_streams, weak_shard


def test_shard_partition_is_exact():
    for total in [0, 1, 7, 128, 1024, 8192]:
        for world in [1, 2, 3, 8]:
            parts = [shard_streams(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert weak_shard(128, 8, 3) == (384, 128)
    with pytest.raises(ValueError):
        shard_streams(10, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import airwave_oracle as orc
    first, count = weak_shard(3, world, rank)
    x = orc.synth_input(count, 64, 2, first_stream=first)        # global stream ids
    frames, elapsed, rate = aggregate_throughput(float(count * 64), 0.5 + rank)
    checksum = torch.tensor([float(np.abs(x).sum())], dtype=torch.float64)
    dist.all_reduce(checksum)
    q.put((rank, first, count, frames, elapsed, rate, float(checksum.item())))
    dist.destroy_process_group()


def test_two_ranks_aggregate_like_bench():
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 3]
    for r in res:
        assert r[3] == 2 * 3 * 64 and r[4] == 1.5 and abs(r[5] - 384 / 1.5) < 1e-9
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import airwave_oracle as orc
    whole = float(np.abs(orc.synth_input(6, 64, 2)).sum())
    assert abs(res[0][6] - whole) < 1e-6 * whole      # the two shards are exactly the single-process batch


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must itself start two ranks — here in --dry-run mode
    (no GPU in this container: rendezvous + shard + aggregate only, over gloo).  The line says it is no measurement."""
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--streams", "5")
    assert rc == 0, err
    assert line["dry_run"] is True and line["value"] == 0.0
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["self_launched"] is True
    assert line["frames_all_ranks"] == 2 * 5 * 1000          # both shards were aggregated


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    """Under a launcher the rank count must equal --gpus: never a line that claims N GPUs from a different world."""
    rc, line, err = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and line is None and "disagrees" in err
    rc, line, err = _bench("--gpus", "1", "--dry-run")
    assert rc == 0 and line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["self_launched"] is False


def test_strong_scaling_plan_partitions_every_rate_bucket():
    """--scaling strong: BASELINE's job totals split over the ranks, bucketed by rate first; the union over ranks is the whole batch
    exactly once and every rank gets its share of every rate (counts differ by at most one per bucket, by at most one in total)."""
    for total in [0, 1, 5, 10, 1024, 4096, 8192]:
        for rates in ([48000], [44100, 48000, 96000]):
            whole_ids, whole_rates = plan_streams(total, rates, 1, 0, "strong")
            assert whole_ids == list(range(total)) and sorted(whole_rates) == whole_rates
            for world in [1, 2, 3, 8]:
                plans = [plan_streams(total, rates, world, r, "strong") for r in range(world)]
                ids = sorted(i for p in plans for i in p[0])
                assert ids == list(range(total)), (total, rates, world)
                for p in plans:
                    assert all(whole_rates[i] == r for i, r in zip(*p))              # a stream keeps its rate whatever the world size
                    for r in rates:
                        mine = [i for i, rr in zip(*p) if rr == r]
                        assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True   # contiguous inside a bucket
                for r in rates:
                    per = [p[1].count(r) for p in plans]
                    assert max(per) - min(per) <= 1
                if total >= world * len(rates):
                    tot = [len(p[0]) for p in plans]
                    assert max(tot) - min(tot) <= 1, (total, rates, world, tot)
    # weak: the per-GPU count, rank-major global ids
    assert plan_streams(4, [44100, 48000, 96000], 2, 1, "weak") == ([4, 5, 6, 7], [44100, 44100, 48000, 96000])
    with pytest.raises(ValueError):
        plan_streams(4, [48000], 2, 0, "medium")


def test_bench_strong_scaling_dry_run_splits_the_baseline_totals():
    """`bench.py --gpus 2 --scaling strong --workload cfg5 --dry-run`: 8192 streams in three rate buckets over two ranks (uneven: 2731 /
    2731 / 2730), aggregated exactly; cfg 4: 4096 over two ranks; the line says which scaling it is."""
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--scaling", "strong", "--workload", "cfg5")
    assert rc == 0, err
    assert line["scaling"] == "strong" and line["streams_total"] == 8192 and sum(line["streams_by_rank"]) == 8192
    assert line["streams_by_rank"] == [4096, 4096] and line["id_sum"] == 8191 * 8192 // 2
    assert [sum(c) for c in zip(*line["streams_by_rank_and_rate"])] == [2731, 2731, 2730]
    assert line["frames_all_ranks"] == 8192 * 1000
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--scaling", "strong", "--workload", "cfg4")
    assert rc == 0 and line["streams_by_rank"] == [2048, 2048] and line["rates"] == [96000]
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--scaling", "strong", "--workload", "cfg3", "--streams", "7")
    assert rc == 0 and line["streams_by_rank"] == [4, 3] and line["first_id_by_rank"] == [0, 4] and line["frames_all_ranks"] == 7000
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--workload", "cfg3")          # weak stays the default
    assert rc == 0 and line["scaling"] == "weak" and line["streams_by_rank"] == [4, 4]
