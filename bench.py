#!/usr/bin/env python3
"""bench.py — stereo frames/s of the MI355X-native HRIR spatializer on BASELINE.json's workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (aw_spatializer_process through the C ABI) over one batch of
synthetic input already resident in HBM.  Default workload = BASELINE.json configs[1] (cfg 2):
128 streams x 10 s @ 48 kHz of 7.1 (8-ch) input -> RoomSH1.0 14-track HeSuVi HRIR -> stereo.
Streams are independent, so N GPUs = N ranks each owning its own 128-stream batch (weak scaling,
no data-path collective); RCCL carries only the final aggregate.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy ceiling)

WORKLOADS = {
    # name: (streams, channels, layout speakers or None (detect), hrir, seconds, description)
    "cfg1": dict(streams=1, channels=2, hrir="NeutralSH1.0.wav", taps=4320, seconds=10.0,
                 desc="cfg1: stereo 48 kHz -> NeutralSH1.0, 1 stream (plumbing)"),
    "cfg2": dict(streams=128, channels=8, hrir="RoomSH1.0.wav", taps=4320, seconds=10.0,
                 desc="cfg2: 7.1 (8ch) 48 kHz -> RoomSH1.0 14-track HeSuVi HRIR, 128-stream batch x 10 s"),
}


def load_hrir(name: str, taps: int):
    """The bundled HRIR fixture if present, else the seeded synthetic 14 x taps HRIR (timing is
    data independent).  Read through the product's own WAV loader."""
    import numpy as np
    import airwave_amd as aw
    path = os.path.join(ROOT, "tests", "golden", "hrtf", name)
    if os.path.exists(path):
        w = aw.WAVLoader.load(path)
        return w.audio_data, f"fixture {name}"
    rng = np.random.default_rng(1234)
    h = rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))
    return h.astype(np.float32), "synthetic 14-track exp-decay noise"


def measured_traffic(S: int, F: int, C: int):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC profile
    of this same workload (profiles/*/summary.json, made by tools/profile_round.sh: separate --pmc
    passes, read bytes from TCC_EA0_RDREQ_{32,64,128}B because FETCH_SIZE counts a 128-B request
    as 64 B on gfx950).  PMC counters cannot be collected from inside this process, so this is the
    last committed measurement, or None when the workload differs / no profile exists."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "summary.json"))):
        try:
            d = json.load(open(path))
            cfg = d["bench"]["config"]
            if (cfg["streams_per_gpu"], cfg["frames_per_stream"], cfg["input_channels"]) == (S, F, C) and "traffic_bytes_per_launch" in d:
                best = (path, d["traffic_bytes_per_launch"])
        except Exception:
            continue
    return best


def cpu_baseline(x_host, tracks, lt, rt, frames: int):
    """Times the CPU oracle (float32 restatement of the reference algorithm: B=512, one engine per
    (channel, ear), per-ear forward FFTs) on the host cores, on a bounded sample of the same input."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import airwave_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, x_host.shape[0]))
    orc.spatialize_f32(x_host[:1, :4096], tracks, lt, rt, threads=1)   # warm the library / page in
    t0 = time.perf_counter()
    orc.spatialize_f32(x_host, tracks, lt, rt, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": x_host.shape[0] * frames / dt, "unit": "stereo frames/s", "cores": threads, "kind": "port",
        "sample": f"{x_host.shape[0]} of the batch's streams x {frames} frames (same synthetic input), "
                  f"one stream per thread, oracle/airwave_oracle.c (reference algorithm, B=512); "
                  f"the Swift/vDSP reference itself cannot run on Linux",
        "seconds": dt,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--seconds", type=float, default=0.0, help="override seconds per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=48)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import airwave_amd as aw
    from airwave_amd.sharding import aggregate_throughput, weak_shard

    wl = dict(WORKLOADS[args.workload])
    S = args.streams or wl["streams"]
    C = wl["channels"]
    rate = 48000
    F = int(round((args.seconds or wl["seconds"]) * rate))
    tracks, hrir_src = load_hrir(wl["hrir"], wl["taps"])

    ctx = aw.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    layout = aw.InputLayout.detect(C)
    lt, rt = aw.HRIRChannelMap.hesuvi14Channel(layout).resolve(layout, tracks.shape[0])
    sp = aw.Spatializer(aw.HRIR(tracks, float(rate), ctx=ctx), lt, rt, n_streams=S, ctx=ctx)

    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    first_stream, _ = weak_shard(S, world, rank)                                  # stream ids are global
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=0xA17AE, first_stream=first_stream)
    torch.cuda.synchronize()

    def step():
        sp.process_device(x.data_ptr(), y.data_ptr(), F)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sp.set_profiling(True)            # HIP events around the dominant kernel, on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    n_launch, kernel_ms, kernel_name = sp.kernel_time()
    sp.set_profiling(False)

    # RCCL over xGMI: the only collective of the run (sum of frames, max of elapsed)
    frames_total, elapsed_max, _ = aggregate_throughput(float(S) * F * args.steps, elapsed, device="cuda")

    if rank == 0:
        finite = bool(torch.isfinite(y[:, -4096:]).all().item())
        bytes_per_frame = 4 * C + 8                      # PCM in + stereo out
        alg_bytes = bytes_per_frame * S * F              # per launch of the dominant kernel
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        info = sp.info()
        tr = measured_traffic(S, F, C)
        result = {
            "metric": "stereo frames/sec @48kHz, 14ch HeSuVi HRIR",
            "value": frames_total / elapsed_max,
            "unit": "stereo frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic U(-0.5,0.5) counter RNG seed 0xA17AE+stream, resident in HBM; HRIR: {hrir_src}",
            "config": {
                "workload": wl["desc"], "streams_per_gpu": S, "frames_per_stream": F, "sample_rate": rate,
                "input_channels": C, "hrir_tracks": int(tracks.shape[0]), "hrir_taps": int(tracks.shape[1]),
                "convolutions_per_stream": int(2 * (lt >= 0).sum()), "parallelism": f"streams sharded x{world}, no data-path collective",
                "fft": info["fft"], "hop": info["hop"], "path": "fused overlap-save" if info["path"] == 0 else "partitioned",
                "outputs_finite": finite,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": tr[1]["total"] if tr else None,
                "traffic_unit": "HBM-side bytes per launch (rocprofv3 PMC)", "traffic_source": os.path.relpath(tr[0], ROOT) if tr else None,
                "kernel": kernel_name, "kernel_avg_ms": kernel_ms, "launches_timed": n_launch,
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_frame": bytes_per_frame,
            },
        }
        if not args.no_cpu_baseline:
            ns = max(1, min(S, args.cpu_sample_streams))
            x_host = x[:ns].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(x_host, tracks, lt, rt, F)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
