#!/usr/bin/env python3
"""bench.py — stereo frames/s of the MI355X-native HRIR spatializer on BASELINE.json's workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (aw_spatializer_process through the C ABI) over one batch of
synthetic input already resident in HBM.  Default workload = BASELINE.json configs[1] (cfg 2):
128 streams x 10 s @ 48 kHz of 7.1 (8-ch) input -> RoomSH1.0 14-track HeSuVi HRIR -> stereo.
The other BASELINE configs are parity-test cases; `--workload cfg3|cfg4|cfg5` benches them on request
(long-tap partitioned path; 96 kHz + parametric EQ; mixed-rate buckets).
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
    # the other BASELINE configs (parity-test cases first; benched on request: --workload cfg3|cfg4|cfg5)
    "cfg3": dict(streams=1024, channels=7, hrir=None, taps=32768, seconds=10.0,
                 desc="cfg3: 7 speakers [FL,FR,FC,BL,BR,SL,SR] 48 kHz -> synthetic 14 x 32768-tap HRIR (seed 1234), 1024-stream batch x 10 s"),
    "cfg3-14ch": dict(streams=1024, channels=14, hrir=None, taps=32768, seconds=10.0, text_map="hesuvi14_custom_map.txt",
                      desc="cfg3, 14-ch-input reading: InputLayout.detect(14) custom channels through the committed parseHeSuViFormat "
                           "text map -> synthetic 14 x 32768-tap HRIR (seed 1234), 1024-stream batch x 10 s"),
    "cfg4": dict(streams=512, channels=7, hrir="StageSH1.0.wav", taps=4320, seconds=10.0, rates=[96000], eq=True,
                 desc="cfg4: 7 speakers 96 kHz -> StageSH1.0 resampled x2 (8640 taps) + 10-band parametric EQ, 512 streams/GPU x 10 s"),
    "cfg5": dict(streams=1024, channels=7, hrir="StageSH1.0.wav", taps=4320, seconds=10.0, rates=[44100, 48000, 96000],
                 desc="cfg5: 7 speakers, streams split evenly over 44.1/48/96 kHz -> StageSH1.0 resampled per rate, 1024 streams/GPU x 10 s"),
}
SPEAKERS7 = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]


def load_hrir(name: str, taps: int):
    """The bundled HRIR fixture if present, else the seeded synthetic 14 x taps HRIR (timing is
    data independent).  Read through the product's own WAV loader."""
    import numpy as np
    import airwave_amd as aw
    path = os.path.join(ROOT, "tests", "golden", "hrtf", name) if name else ""
    if name and os.path.exists(path):
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


def cpu_baseline(x_host, tracks, lt, rt, frames: int, eq_definition=None, rate: float = 48000.0):
    """Times the CPU oracle (float32 restatement of the reference algorithm: B=512, one engine per
    (channel, ear), per-ear forward FFTs) on the host cores, on a bounded sample of the same input."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import airwave_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, x_host.shape[0]))
    orc.spatialize_f32(x_host[:1, :4096], tracks, lt, rt, threads=1)   # warm the library / page in
    t0 = time.perf_counter()
    y = orc.spatialize_f32(x_host, tracks, lt, rt, threads=threads)
    if eq_definition is not None:      # the EQ that follows the spatializer in cfg 4 (one state per stream, sequential recurrence)
        od = orc.EqualizerDefinition(eq_definition.preampDB, [orc.EqualizerFilter(f.sourceLine, f.sourceNumber, f.isEnabled, f.type,
                                                                                    f.frequencyHz, f.gainDB, f.q) for f in eq_definition.filters])
        for s in range(y.shape[0]):
            orc.eq_prepare(od, rate).process(y[s, :, 0], y[s, :, 1])
    dt = time.perf_counter() - t0
    # (i) of SURVEY 8d: one thread = the reference's single audio thread, on one stream and at most 2 s of it
    f1 = min(frames, int(2 * rate))
    t1 = time.perf_counter()
    orc.spatialize_f32(x_host[:1, :f1], tracks, lt, rt, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": x_host.shape[0] * frames / dt, "unit": "stereo frames/s", "cores": threads, "kind": "port",
        "single_thread_value": f1 / dt1,
        "sample": f"{x_host.shape[0]} of the batch's streams x {frames} frames (same synthetic input), "
                  f"one stream per thread, oracle/airwave_oracle.c (reference algorithm, B=512); "
                  f"the Swift/vDSP reference itself cannot run on Linux",
        "seconds": dt,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (cfg1/cfg2), 3 (cfg3-5)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (cfg1/cfg2), 1 (cfg3-5)")
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
    # Debug overrides (a 1-GPU box can still exercise the N > 1 code path: AW_BENCH_DEVICE=0 AW_BENCH_BACKEND=gloo).
    if "AW_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["AW_BENCH_DEVICE"])
    backend = os.environ.get("AW_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import airwave_amd as aw
    from airwave_amd.sharding import aggregate_throughput, weak_shard

    wl = dict(WORKLOADS[args.workload])
    S = args.streams or wl["streams"]
    C = wl["channels"]
    seconds = args.seconds or wl["seconds"]
    rates = wl.get("rates", [48000])
    tracks, hrir_src = load_hrir(wl["hrir"], wl["taps"])
    if args.steps is None:
        args.steps = 20 if args.workload in ("cfg1", "cfg2") else 3
    if args.warmup is None:
        args.warmup = 3 if args.workload in ("cfg1", "cfg2") else 1

    ctx = aw.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    layout = aw.InputLayout.detect(C) if C != 7 else aw.InputLayout(SPEAKERS7, "7 speakers")
    first_stream, _ = weak_shard(S, world, rank)                                  # stream ids are global

    # One leg per sample rate (cfg 5 buckets streams by rate; every other workload has one leg).
    stream_rates = [rates[i * len(rates) // S] for i in range(S)] if len(rates) > 1 else [rates[0]] * S
    cmap = None
    if wl.get("text_map"):            # the 14-channel reading of cfg 3: custom channels mapped by a HeSuVi-style text map
        cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(ROOT, "tests", "golden", wl["text_map"])).read())
    batch = aw.MixedRateBatch(tracks, 48000.0, layout, stream_rates, hrirMap=cmap, ctx=ctx)
    eq_def = None
    if wl.get("eq"):
        eq_def = aw.EqualizerAPOParser.parse(open(os.path.join(ROOT, "tests", "golden", "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "CCA CRA ParametricEq.txt")
    legs = []
    for rate, b in batch.buckets.items():
        n, F = len(b.stream_ids), int(round(seconds * rate))
        x = torch.empty((n, F, C), dtype=torch.float32, device="cuda")
        y = torch.empty((n, F, 2), dtype=torch.float32, device="cuda")
        ctx.synth_fill(x.data_ptr(), n, F, C, seed=0xA17AE, first_stream=first_stream + b.stream_ids[0])
        eq = aw.ParametricEqualizerState(eq_def, float(rate), n_streams=n, ctx=ctx) if eq_def is not None else None
        legs.append(dict(rate=rate, n=n, F=F, x=x, y=y, sp=b.spatializer, eq=eq, taps=b.hrir_taps))
    torch.cuda.synchronize()
    eq_events = []

    def step(timed=False):
        for g in legs:
            g["sp"].process_device(g["x"].data_ptr(), g["y"].data_ptr(), g["F"])
            if g["eq"] is not None:
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                g["eq"].process_device(g["y"].data_ptr(), g["y"].data_ptr(), g["F"])     # in place, on the same stream
                if timed:
                    e1.record()
                    eq_events.append((e0, e1))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for g in legs:
        g["sp"].set_profiling(True)   # HIP events around the dominant kernel, on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    torch.cuda.synchronize()
    if world > 1:                     # the timed region ends, like it starts, with a barrier + synchronize on every rank
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    frames_step = sum(g["n"] * g["F"] for g in legs)

    # RCCL over xGMI: the only collective of the run (sum of frames, max of elapsed)
    frames_total, elapsed_max, _ = aggregate_throughput(float(frames_step) * args.steps, elapsed,
                                                             device="cuda" if backend == "nccl" else "cpu")

    if rank == 0:
        bytes_per_frame = 4 * C + 8                      # PCM in + stereo out
        kernel_ms, dom_frames, names, paths = 0.0, 0, [], []
        for g in legs:
            n_launch, ms, name = g["sp"].kernel_time()
            g["sp"].set_profiling(False)
            info = g["sp"].info()
            # a partitioned call may be chunked over streams: ms is the average per chunk launch
            launches_per_step = max(1, round(n_launch / args.steps))
            kernel_ms += ms * launches_per_step
            dom_frames += info["dominant_frames"] * launches_per_step if info["path"] == 0 else g["n"] * g["F"]
            names.append(name)
            paths.append({"rate": g["rate"], "streams": g["n"], "frames": g["F"], "taps": g["taps"], "fft": info["fft"], "hop": info["hop"],
                          "partitions": info["partitions"], "path": "fused overlap-save" if info["path"] == 0 else "partitioned"})
        finite = all(bool(torch.isfinite(g["y"][:, -4096:]).all().item()) for g in legs)
        # algorithmic bytes of the frames the timed (dominant) launches produced
        alg_bytes = bytes_per_frame * dom_frames
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        g0 = legs[0]
        tr = measured_traffic(g0["n"], g0["F"], C) if len(legs) == 1 else None
        lt, rt = batch.left_track, batch.right_track
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
            "dtype": "f32" if eq_def is None else "f32 (convolution) + f64 (EQ)",
            "data": f"synthetic U(-0.5,0.5) counter RNG seed 0xA17AE+stream, resident in HBM; HRIR: {hrir_src}",
            "config": {
                "workload": wl["desc"], "streams_per_gpu": S, "frames_per_stream": g0["F"], "sample_rate": g0["rate"],
                "input_channels": C, "hrir_tracks": int(tracks.shape[0]), "hrir_taps": g0["taps"],
                "convolutions_per_stream": int((lt >= 0).sum() + (rt >= 0).sum()), "parallelism": f"streams sharded x{world}, no data-path collective",
                "fft": paths[0]["fft"], "hop": paths[0]["hop"], "path": paths[0]["path"], "legs": paths,
                "outputs_finite": finite,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": tr[1]["total"] if tr else None,
                "traffic_unit": "HBM-side bytes per launch (rocprofv3 PMC)", "traffic_source": os.path.relpath(tr[0], ROOT) if tr else None,
                "kernel": ", ".join(sorted(set(names))), "kernel_avg_ms": kernel_ms, "launches_timed": args.steps,
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_frame": bytes_per_frame,
                "frames_per_launch": dom_frames,
            },
        }
        # Second roof (report which one binds): FP32 vector.  Algorithmic flops of the transforms this
        # implementation runs per output frame: (pairs + outputs) complex FFTs of 8192 points at 5 N log2 N, and
        # pairs x N x outputs complex multiply-accumulate pairs (16 flop) per tile of `hop` frames.
        if len(legs) == 1 and paths[0]["path"].startswith("fused"):
            n8 = 8192
            long_win = paths[0]["fft"] == 16384
            pairs = (2 * C + 1) // 2 if long_win else (C + 1) // 2
            outs = 2 if long_win else 1
            flops_per_frame = ((pairs + outs) * 5 * n8 * 13 + pairs * n8 * 16 * outs) / paths[0]["hop"]
            tfl = flops_per_frame * dom_frames / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
            result["fp32_roof"] = {"flops_per_frame": flops_per_frame, "achieved": tfl, "peak": 157.3, "unit": "TFLOP/s", "frac": tfl / 157.3,
                                   "intensity_flop_per_byte": flops_per_frame / bytes_per_frame, "ridge_flop_per_byte": 157.3e12 / (HBM_PEAK_GBS * 1e9),
                                   "note": "MI355X FP32 vector peak (MI355X_MICROARCH.md); " +
                                           ("the HBM roof binds" if flops_per_frame / bytes_per_frame < 0.95 * 157.3e12 / (HBM_PEAK_GBS * 1e9)
                                            else "the FP32 roof binds" if flops_per_frame / bytes_per_frame > 1.05 * 157.3e12 / (HBM_PEAK_GBS * 1e9)
                                            else "at this intensity both roofs bind within 5 %")}
        if eq_events:
            eq_ms = sum(a.elapsed_time(b) for a, b in eq_events) / args.steps
            result["roofline"]["eq_kernel_ms_per_step"] = eq_ms
            result["roofline"]["eq_achieved_GBs"] = 16.0 * frames_step / (eq_ms * 1e-3) / 1e9      # 8 B in + 8 B out per frame
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported at N = 1 only
            ns = max(1, min(g0["n"], args.cpu_sample_streams))
            Fc = g0["F"] if args.workload in ("cfg1", "cfg2") else min(g0["F"], int(g0["rate"]))     # long-tap configs: 1 s per stream
            x_host = g0["x"][:ns, :Fc].cpu().numpy()
            tr0 = aw.resample_tracks(tracks, 48000.0, float(g0["rate"]))
            result["cpu_baseline"] = cpu_baseline(x_host, tr0, lt, rt, Fc, eq_definition=eq_def, rate=float(g0["rate"]))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
