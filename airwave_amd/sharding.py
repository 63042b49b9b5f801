"""Stream sharding and the throughput aggregate for multi-GPU runs.

Streams are independent, so N GPUs = N ranks with disjoint stream ranges and **no collective on the
data path**; the only exchange is one small all-reduce of {frames (sum), elapsed (max)} at the end
(RCCL over xGMI on GPUs, gloo in the CPU tests).  Global stream ids seed the synthetic input, so a
rank's data does not depend on the world size."""
from __future__ import annotations

from typing import Tuple


def shard_streams(total_streams: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous partition of [0, total) : returns (first_stream, count); counts differ by at most 1."""
    if world_size <= 0 or not (0 <= rank < world_size) or total_streams < 0:
        raise ValueError("bad shard request")
    q, r = divmod(total_streams, world_size)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def weak_shard(streams_per_gpu: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Weak scaling: every rank owns `streams_per_gpu` streams; global ids are rank-major."""
    if not (0 <= rank < world_size):
        raise ValueError("bad rank")
    return rank * streams_per_gpu, streams_per_gpu


def aggregate_throughput(frames_done: float, elapsed_s: float, device=None):
    """Whole-job stereo frames/s = sum(frames) / max(elapsed) over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    f = torch.tensor([frames_done], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():        # (also a one-rank group: bench.py's AW_BENCH_FORCE_PG debug run)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(f.item()), float(t.item()), float(f.item()) / float(t.item())


def rate_of_stream(i: int, total: int, rates) -> float:
    """BASELINE cfg 5 buckets the batch by sample rate: stream i of `total` gets rates[i * len(rates) // total] (contiguous buckets)."""
    return rates[i * len(rates) // total]


def plan_streams(streams: int, rates, world_size: int, rank: int, scaling: str = "weak"):
    """Which streams this rank owns, as (global_stream_ids, rates) in processing order.

    weak   — `streams` is the PER-GPU count: every rank owns that many, ids rank-major, rates bucketed within the rank's batch
             (per-GPU work fixed as N grows; bench.py's default and what the driver's N = 1, 2, 4, 8 runs measure).
    strong — `streams` is the JOB TOTAL (BASELINE cfg 3 / 4 / 5: 1024 / 4096 / 8192 streams over the node): the batch is bucketed by
             rate FIRST and every bucket is split across the ranks with shard_streams, so each rank gets an equal share of every
             rate (a rank that owned only 96 kHz streams would finish last); a bucket's remainder streams are dealt round robin across buckets, so rank totals differ by at most one.
    The union over ranks is exactly range(total) with no overlap; nothing here needs a collective."""
    rates = list(rates)
    if scaling == "weak":
        first, count = weak_shard(streams, world_size, rank)
        return [first + i for i in range(count)], [rate_of_stream(i, count, rates) for i in range(count)]
    if scaling != "strong":
        raise ValueError(f"scaling must be 'weak' or 'strong', not {scaling!r}")
    ids, out_rates = [], []
    bounds = [next((i for i in range(streams) if rate_of_stream(i, streams, rates) == r), streams) for r in rates] + [streams] if streams else [0] * (len(rates) + 1)
    extra = 0           # remainder streams handed out so far: the next bucket's go to the ranks after them (round robin), totals differ by <= 1
    for k, r in enumerate(rates):
        a, b = bounds[k], bounds[k + 1]
        first, count = shard_streams(b - a, world_size, (rank - extra) % world_size)
        extra += (b - a) % world_size
        ids += [a + first + i for i in range(count)]
        out_rates += [r] * count
    return ids, out_rates
