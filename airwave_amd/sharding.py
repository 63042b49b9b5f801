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
