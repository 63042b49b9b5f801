#!/usr/bin/env python3
"""bench.py — stereo frames/s of the MI355X-native HRIR spatializer on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts the N ranks itself (one process per GPU through `python -m torch.distributed.run`, before
anything in this process touches the GPU); launched under torchrun (WORLD_SIZE set) it is one of the ranks:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (aw_spatializer_process through the C ABI) over one batch of synthetic input
already resident in HBM.  Default workload = the configuration BASELINE.json's metric is quoted on, cfg 3
(SURVEY.md §8d item 3): 1024 streams x 10 s @ 48 kHz, 14-track 32768-tap HeSuVi HRIR, primary reading C = 7 speakers
[FL,FR,FC,BL,BR,SL,SR] (all 14 tracks).  The 14-channel-input reading of the same configuration (C = 14 custom channels
through a parseHeSuViFormat text map) is measured in the same run and reported as `"secondary"` inside the one JSON line.
`--workload cfg1|cfg2|cfg2-14ch|cfg4|cfg5|cfg3-14ch` benches the other BASELINE configurations on request.  After the timed
region (untimed, in the CPU-baseline leg) two streams' head and tail of the last step are compared with the float64 oracle:
`parity_spot_err` (primary and secondary); above 1e-5 the exit code is 3.

Streams are independent, so N GPUs = N ranks each owning its own batch (weak scaling, no data-path collective); RCCL
carries only the final {frames (sum), elapsed (max)} aggregate.  One JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s) — what `roofline.frac` is priced on; the ceilings MEASURED on the
                               # box of this run (read-only / write-only / copy kernels, aw_context_bandwidth_probe) are reported next to it: roofline.measured
FP32_PEAK_TFLOPS = 157.3

# Warm-up: the chip's clock governor takes ~25 ms of a load to settle (profiles/round5_v1/warmup_study.txt: cfg 2 on one box, same binary,
# 3 + 20 steps of 1.3 ms: 47.0 G frames/s; 20 + 50: 50.9; 100 + 200 ... 1000 + 1000: 50.9 - 51.1).  Workloads whose step is a millisecond
# or less therefore warm up for >= 60 ms and time >= 0.2 s (rounds 1-4 timed cfg 2 over 3 + 20 steps, i.e. on a ramping clock: the same
# kernel read 0.2307 there and reads 0.2546 in steady state); steps of 10 ms and more settle within their first step.
WORKLOADS = {
    "cfg1": dict(streams=1, channels=2, hrir="NeutralSH1.0.wav", taps=4320, seconds=10.0, steps=5000, warmup=3000,
                 desc="cfg1: stereo 48 kHz -> NeutralSH1.0, 1 stream (plumbing)"),
    "cfg2": dict(streams=128, channels=8, hrir="RoomSH1.0.wav", taps=4320, seconds=10.0, steps=200, warmup=60,
                 desc="cfg2: 7.1 (8ch) 48 kHz -> RoomSH1.0 14-track HeSuVi HRIR, 128-stream batch x 10 s"),
    "cfg2-14ch": dict(streams=128, channels=14, hrir="RoomSH1.0.wav", taps=4320, seconds=10.0, steps=150, warmup=40, text_map="hesuvi14_custom_map.txt",
                      desc="cfg2, 14-ch-input reading (north_star's literal case): InputLayout.detect(14) custom channels through the committed "
                           "parseHeSuViFormat text map -> RoomSH1.0 14-track HeSuVi HRIR (4320 taps), 128-stream batch x 10 s"),
    # (cfg 3: 10 warm-up steps — the chip's clock governor needs ~30 ms of this load to settle: 1.6 -> 1.84 GHz over the first launches)
    "cfg3": dict(streams=1024, channels=7, hrir=None, taps=32768, seconds=10.0, steps=30, warmup=10,
                 desc="cfg3: 7 speakers [FL,FR,FC,BL,BR,SL,SR] 48 kHz -> synthetic 14 x 32768-tap HeSuVi HRIR (seed 1234), 1024-stream batch x 10 s"),
    "cfg3-14ch": dict(streams=1024, channels=14, hrir=None, taps=32768, seconds=10.0, steps=10, warmup=3, text_map="hesuvi14_custom_map.txt",
                      desc="cfg3, 14-ch-input reading: InputLayout.detect(14) custom channels through the committed parseHeSuViFormat "
                           "text map -> synthetic 14 x 32768-tap HRIR (seed 1234), 1024-stream batch x 10 s"),
    "cfg4": dict(streams=512, channels=7, hrir="StageSH1.0.wav", taps=4320, seconds=10.0, steps=5, warmup=1, rates=[96000], eq=True,
                 desc="cfg4: 7 speakers 96 kHz -> StageSH1.0 resampled x2 (8640 taps) + 10-band parametric EQ, 512 streams/GPU x 10 s"),
    "cfg5": dict(streams=1024, channels=7, hrir="StageSH1.0.wav", taps=4320, seconds=10.0, steps=5, warmup=1, rates=[44100, 48000, 96000],
                 desc="cfg5: 7 speakers, streams split evenly over 44.1/48/96 kHz -> StageSH1.0 resampled per rate, 1024 streams/GPU x 10 s"),
}
# --scaling strong: BASELINE.json quotes cfg 3 / 4 / 5 as JOB totals (1024 / 4096 / 8192 streams over the GPUs of one node); weak scaling (the
# default, what the driver's N = 1, 2, 4, 8 runs use) fixes the per-GPU count above instead
TOTAL_STREAMS = {"cfg1": 1, "cfg2": 128, "cfg2-14ch": 128, "cfg3": 1024, "cfg3-14ch": 1024, "cfg4": 4096, "cfg5": 8192}
SECONDARY = {"cfg3": "cfg3-14ch"}          # measured in the same run, reported inside the primary's JSON line
SPEAKERS7 = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]


# ---------------------------------------------------------------------------------------------- launch
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n: int, argv) -> int:
    """Starts the N ranks as children of this process (which has not touched the GPU) and returns their exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["AW_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------- inputs
def load_hrir(name, taps: int):
    """The bundled HRIR fixture if present, else the seeded synthetic 14 x taps HRIR (timing is data independent).
    Read through the product's own WAV loader."""
    import numpy as np
    import airwave_amd as aw
    path = os.path.join(ROOT, "tests", "golden", "hrtf", name) if name else ""
    if name and os.path.exists(path):
        w = aw.WAVLoader.load(path)
        return w.audio_data, f"fixture {name}"
    rng = np.random.default_rng(1234)
    h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
    # BASELINE cfg 3 reads "imported .wav HRIR (long tap)": the synthetic HRIR is written as a 14-track float32 WAV at setup and comes
    # back through the product's RIFF reader, the import route of HRIRManager.activatePreset (HRIRManager.swift:347-446, WAVLoader.swift:26-99)
    import struct, tempfile
    data = np.ascontiguousarray(h.T).astype("<f4").tobytes()                      # [frame][track]
    block = 14 * 4
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 14, 48000, 48000 * block, block, 32) + b"data" + struct.pack("<I", len(data)) + data
    with tempfile.NamedTemporaryFile(suffix=".wav", delete=False) as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        path = f.name
    try:
        w = aw.WAVLoader.load(path)
    finally:
        os.unlink(path)
    assert w.audio_data.shape == (14, taps) and np.array_equal(np.asarray(w.audio_data), h)
    return w.audio_data, "synthetic 14-track exp-decay noise, imported as a float32 .wav"


def committed_traffic(workload: str, S: int, F: int, C: int):
    """Fabric-side (L2 <-> Infinity Cache / HBM) bytes per step from the newest committed rocprofv3 PMC profile of this
    same workload (profiles/*/traffic_<workload>.json, made by tools/archive/profile_round5.sh: separate --pmc passes, read bytes
    from TCC_EA0_RDREQ_{32,64,128}B because FETCH_SIZE counts a 128-B request as 64 B on gfx950).  PMC counters cannot be
    collected from inside this process, so this is the last committed measurement — and only one made on THESE kernels: a
    profile records digests of csrc/device, of the host side of the library (runtime.cpp: launch policy, chunking; host/: table
    builders) and of the build's compiler flags at the time it was taken (and the git HEAD of that build), and a profile whose
    digests differ from the tree's is refused (returns (None, why)) instead of silently going stale."""
    import glob
    from airwave_amd.provenance import build_flags_digest, device_source_digest, host_source_digest
    digest = device_source_digest()
    ident = {"device_src_sha16": digest, "host_src_sha16": host_source_digest(), "build_flags_sha16": build_flags_digest()}
    best, why = None, "no committed profile of this workload"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"traffic_{workload}.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if (d["streams_per_gpu"], d["frames_per_stream"], d["input_channels"]) != (S, F, C):
            continue
        stale = [k for k, v in ident.items() if d.get(k) != v]          # kernels, launch policy / table builders, compiler flags: all three
        if stale:
            why = (f"newest profile of this shape ({os.path.relpath(path, ROOT)}, HEAD {d.get('git_head', 'unrecorded')}) was made on another build: "
                   + ", ".join(f"{k} {d.get(k, 'unrecorded')} != {ident[k]}" for k in stale) + ": re-profile")
            best = None
            continue
        best, why = (path, d), ""
    return best, why


def cpu_baseline(x_host, tracks, lt, rt, frames: int, eq_definition=None, rate: float = 48000.0):
    """Times the CPU oracle (float32 restatement of the reference algorithm: B=512, one engine per (channel, ear),
    per-ear forward FFTs) on the host cores, on a bounded sample of the same input."""
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


def oracle_spot_check(x_dev, y_dev, tracks, lt, rt, calls: int, streams=(0, -1), head: int = 4096, tail: int = 2048):
    """Part of the CPU-baseline leg (the oracle as the CHECKER of what was just timed, never the thing measured): the first `head`
    and last `tail` output frames of two streams of the LAST timed step against the float64 oracle (oracle.spatialize_f64, the
    mathematical definition the reference's ConvolutionEngine approximates; the reference holds no vector for real HRIRs).
    The steps continue one stream (no reset between them), so the head is checked with the previous step's tail as history.
    Returns the worst peak-relative error."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import airwave_oracle as orc
    L = tracks.shape[1]
    F = x_dev.shape[1]
    worst = 0.0

    def host(a):          # device tensors (the timed run) or host arrays (the end-to-end leg)
        return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    for s in streams:
        if calls >= 2 and F >= L - 1:          # history of the last step = the tail of the same input
            xin = np.concatenate([host(x_dev[s, F - (L - 1):]), host(x_dev[s, :head])])
            ref = orc.spatialize_f64(xin, tracks, lt, rt)[L - 1:]
        else:
            ref = orc.spatialize_f64(host(x_dev[s, :head]), tracks, lt, rt)
        worst = max(worst, orc.peak_rel_error(host(y_dev[s, :head]), ref))
        n0 = max(0, F - tail - (L - 1))
        xin = host(x_dev[s, n0:])
        if n0 == 0 and calls >= 2 and F >= L - 1:
            xin = np.concatenate([host(x_dev[s, F - (L - 1):]), xin])
            ref = orc.spatialize_f64(xin, tracks, lt, rt)[-tail:]
        else:
            ref = orc.spatialize_f64(xin, tracks, lt, rt)[-tail:]
        worst = max(worst, orc.peak_rel_error(host(y_dev[s, F - tail:]), ref))
    return worst


# ---------------------------------------------------------------------------------------------- one workload
def flops_per_frame(C: int, path: dict) -> float:
    """Algorithmic flops of the transforms THIS implementation runs per output frame (for the FP32-vector roof)."""
    n8 = 8192
    fft = 5 * n8 * 13
    if path["path"].startswith("long-window"):
        # per window of N frames: one complex FFT_N per channel pair (a real last channel: half of one), one inverse, and per bin pair
        # and channel pair four complex multiply-accumulates (2x2 in-lane CMAC)
        import math
        N, hop, frames = path["fft"], path["hop"], path["frames"]
        windows = -(-frames // hop)
        pairs_eff = C / 2.0
        per_window = (pairs_eff + 1) * 5 * N * math.log2(N) + (N / 2) * math.ceil(C / 2) * 4 * 8
        return windows * per_window / frames
    if path["path"].startswith("fused"):
        long_win = path["fft"] == 16384
        pairs = (2 * C + 1) // 2 if long_win else (C + 1) // 2
        outs = 2 if long_win else 1
        return ((pairs + outs) * fft + pairs * n8 * 16 * outs) / path["hop"]
    pairs = (C + 1) // 2                   # partitioned: per 4096-frame block, pairs forward + 1 inverse + P x pairs x N bin CMACs
    return ((pairs + 1) * fft + path["partitions"] * pairs * n8 * 16) / path["hop"]


def measured_ceilings(ctx) -> dict:
    """"confirm on the box and also quote a measured copy-kernel ceiling".  Read-only, write-only and copy kernels
    over 4 GiB each (aw_context_bandwidth_probe: 16 B per lane, plain and non-temporal forms, best of 3), on the box and in the
    process of this very run, before the timed region."""
    t0 = time.perf_counter()
    m = ctx.bandwidth_probe(4 << 30, 3)
    return {"read": m["read"], "write": m["write"], "copy": m["copy"], "unit": "GB/s", "bytes_per_kernel": 4 << 30,
            "copy_counts": "bytes read + bytes written", "probe_seconds": round(time.perf_counter() - t0, 3),
            "source": "aw_context_bandwidth_probe on this box, this process, before the timed region"}


def run_workload(name: str, args, ctx, world: int, rank: int, backend: str, with_cpu: bool, with_check: bool = None, measured: dict = None, eq_mode: str = None):
    import numpy as np
    import torch
    import torch.distributed as dist
    import airwave_amd as aw
    from airwave_amd import provenance as prov
    from airwave_amd.sharding import aggregate_throughput, plan_streams

    wl = dict(WORKLOADS[name])
    scaling = getattr(args, "scaling", "weak")
    rates = wl.get("rates", [48000])
    # weak: `streams` per GPU; strong: the job total split over the ranks, every rate bucket evenly (sharding.plan_streams)
    S_req = args.streams or (TOTAL_STREAMS[name] if scaling == "strong" else wl["streams"])
    global_ids, stream_rates = plan_streams(S_req, rates, world, rank, scaling)
    S = len(global_ids)
    streams_total = S_req if scaling == "strong" else S_req * world
    if S == 0:
        raise SystemExit(f"bench.py: rank {rank} of {world} owns no stream of {name} ({S_req} streams, strong scaling)")
    C = wl["channels"]
    seconds = args.seconds or wl["seconds"]
    steps = args.steps if args.steps is not None else wl["steps"]
    warmup = args.warmup if args.warmup is not None else wl["warmup"]
    n_lanes = max(1, min(args.lanes if args.lanes else wl.get("lanes", 1), S))
    tracks, hrir_src = load_hrir(wl["hrir"], wl["taps"])

    layout = aw.InputLayout.detect(C) if C != 7 else aw.InputLayout(SPEAKERS7, "7 speakers")
    # in + out bytes of this rank's batch must fit beside the scratch pool: refuse early instead of driving the box out of memory
    need = sum((4 * C + 8) * int(round(seconds * r)) for r in stream_rates)
    if need > 0.85 * torch.cuda.mem_get_info()[0]:
        raise SystemExit(f"bench.py: {name} with {S} streams on this rank needs {need / 2**30:.0f} GiB of PCM in HBM; use more GPUs or --streams")

    # One leg per (lane, sample rate).  cfg 5 buckets streams by rate; a workload with `lanes` > 1 (cfg 4) runs its batch as that many
    # chunks of streams, each on a context — a HIP stream — of its own, so that one chunk's EQ kernel (FP64-vector bound) can execute
    # beside another chunk's split / merge kernels (fabric bound); every other workload is one lane on the bench's context.
    cmap = None
    if wl.get("text_map"):            # the 14-channel reading of cfg 3: custom channels mapped by a HeSuVi-style text map
        cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(ROOT, "tests", "golden", wl["text_map"])).read())
    eq_def = None
    eq_mode = eq_mode or getattr(args, "eq", "auto")
    if wl.get("eq"):
        eq_def = aw.EqualizerAPOParser.parse(open(os.path.join(ROOT, "tests", "golden", "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "CCA CRA ParametricEq.txt")
    mem_free0, mem_total = torch.cuda.mem_get_info()
    main_stream = torch.cuda.current_stream()                                    # torch's current stream IS the bench context's stream
    lane_ctx = [ctx] + [aw.Context(ctx.device) for _ in range(n_lanes - 1)]
    lane_stream = [main_stream] + [torch.cuda.ExternalStream(c.stream) for c in lane_ctx[1:]]
    legs, batches = [], []
    activation = []                   # per leg: what aw_spatializer_reserve spent where (table build on host threads / upload / scratch pool)
    for li in range(n_lanes):
        lo, hi = li * S // n_lanes, (li + 1) * S // n_lanes
        # the equalizer of cfg 4 follows the spatializer (AudioEffectGraph.swift:195-211); --eq auto / fold: folded into the HRIR at activation
        # (aw_eq_fold_hrir: one pass over the audio), --eq cascade: the Float64 biquad kernel after the spatializer, in place
        batch = aw.MixedRateBatch(tracks, 48000.0, layout, stream_rates[lo:hi], hrirMap=cmap, ctx=lane_ctx[li], equalizer=eq_def,
                                  fold_equalizer=None if eq_mode == "auto" else eq_mode == "fold")
        batches.append(batch)
        for rate, b in batch.buckets.items():
            n, F = len(b.stream_ids), int(round(seconds * rate))
            x = torch.empty((n, F, C), dtype=torch.float32, device="cuda")
            y = torch.empty((n, F, 2), dtype=torch.float32, device="cuda")
            lane_ctx[li].synth_fill(x.data_ptr(), n, F, C, seed=0xA17AE, first_stream=global_ids[lo + b.stream_ids[0]])       # (a bucket's ids are contiguous)
            t_act = time.perf_counter()
            b.spatializer.reserve(F)  # every internal buffer is sized here (and the long-window tables built): process never allocates
            torch.cuda.synchronize()
            total_ms = (time.perf_counter() - t_act) * 1e3
            inf = b.spatializer.info()
            activation.append({"total_ms": round(total_ms, 1), "tables_ms": round(inf["reserve_tables_ms"], 1), "upload_ms": round(inf["reserve_upload_ms"], 1),
                               "scratch_alloc_ms": round(inf["reserve_scratch_ms"], 1), "scratch_bytes": inf["scratch_bytes"], "warm_context": False})
            legs.append(dict(lane=li, rate=rate, n=n, F=F, x=x, y=y, sp=b.spatializer, eq=b.equalizer, taps=b.hrir_taps, first=lo + b.stream_ids[0],
                             eq_response_taps=b.eq_response_taps, eq_tail_bound=b.eq_tail_bound))
    batch = batches[0]
    torch.cuda.synchronize()
    mem_free1 = torch.cuda.mem_get_info()[0]
    eq_events = []

    def step(timed=False):
        for g in legs:
            st = lane_stream[g["lane"]]
            g["sp"].process_device(g["x"].data_ptr(), g["y"].data_ptr(), g["F"])
            if g["eq"] is not None:
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                g["eq"].process_device(g["y"].data_ptr(), g["y"].data_ptr(), g["F"])     # in place, on the lane's stream
                if timed:
                    e1.record(st)
                    eq_events.append((e0, e1))

    def fork():        # the other lanes' streams start after everything queued on the main stream ...
        if n_lanes > 1:
            e = torch.cuda.Event()
            e.record(main_stream)
            for st in lane_stream[1:]:
                st.wait_event(e)

    def join():        # ... and the main stream continues after everything queued on them
        for st in lane_stream[1:]:
            e = torch.cuda.Event()
            e.record(st)
            main_stream.wait_event(e)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    for g in legs:
        g["sp"].set_profiling(True)   # HIP events around every kernel launch, on the launch stream
    if world > 1 or dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(main_stream)
    fork()
    for _ in range(steps):
        step(timed=True)
    join()
    ev1.record(main_stream)
    torch.cuda.synchronize()
    if world > 1 or dist.is_initialized():        # the timed region ends, like it starts, with a barrier + synchronize on every rank
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / steps
    frames_step = sum(g["n"] * g["F"] for g in legs)

    # RCCL over xGMI: the only collective of the run (sum of frames, max of elapsed)
    frames_total, elapsed_max, _ = aggregate_throughput(float(frames_step) * steps, elapsed,
                                                         device="cuda" if backend == "nccl" else "cpu")
    result = None
    if rank == 0:
        bytes_per_frame = 4 * C + 8                      # PCM in + stereo out
        stages, paths = {}, []
        dom_ms, dom_frames, dom_names = 0.0, 0, []
        for g in legs:
            n_launch, ms, kname = g["sp"].kernel_time()
            for sname, total_ms, launches in g["sp"].stage_times():
                st = stages.setdefault(sname, {"ms_per_step": 0.0, "launches_per_step": 0.0})
                st["ms_per_step"] += total_ms / steps
                st["launches_per_step"] += launches / steps
            g["sp"].set_profiling(False)
            info = g["sp"].info()
            launches_per_step = max(1, round(n_launch / steps))
            dom_ms += ms * launches_per_step
            dom_frames += info["dominant_frames"] * launches_per_step if info["path"] == 0 else g["n"] * g["F"]
            dom_names.append(kname)
            lw = info.get("long_window_rows", 0)
            if lw:        # the long-window kernels ran (tile_lw.hpp): windows of lw x 4096 frames, hop = window - history
                paths.append({"lane": g["lane"], "rate": g["rate"], "streams": g["n"], "frames": g["F"], "taps": g["taps"], "fft": lw * 4096, "hop": lw * 4096 - info["history"],
                              "partitions": 1, "path": "long-window overlap-save (four-step FFT: split / rows / merge)"})
            elif info.get("overlap_add_rows", 0):      # the overlap-add tile ran (tile_ola.hpp): blocks of 512 H frames, one launch for the whole call
                paths.append({"lane": g["lane"], "rate": g["rate"], "streams": g["n"], "frames": g["F"], "taps": g["taps"], "fft": info["fft"], "hop": 512 * info["overlap_add_rows"],
                              "partitions": 1, "path": "fused overlap-add"})
            else:
                paths.append({"lane": g["lane"], "rate": g["rate"], "streams": g["n"], "frames": g["F"], "taps": g["taps"], "fft": info["fft"], "hop": info["hop"],
                              "partitions": info["partitions"], "path": "fused overlap-save" if info["path"] == 0 else "partitioned"})
        finite = all(bool(torch.isfinite(g["y"][:, -4096:]).all().item()) for g in legs)
        g0 = legs[0]
        multi_kernel = all(not pp["path"].startswith("fused") for pp in paths) and bool(stages)
        if multi_kernel:
            # multi-kernel pipeline: every stage's launches cover all frames of the step; the dominant kernel is the longest stage
            kname = max(stages, key=lambda k: stages[k]["ms_per_step"])
            dom_ms, dom_frames, dom_names = stages[kname]["ms_per_step"], frames_step, [kname]
        # step level: every launch of the step (all kernels, boundary tiles, history carry), HIP events on the launch stream
        step_alg = bytes_per_frame * frames_step
        step_gbs = step_alg / (step_ms * 1e-3) / 1e9
        # kernel level: the dominant kernel alone, priced on the frames its launches produced
        kern_alg = bytes_per_frame * dom_frames
        kern_gbs = kern_alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        tr, tr_why = committed_traffic(name, S, g0["F"], C) if len(rates) == 1 else (None, "several legs")
        lt, rt = batch.left_track, batch.right_track
        roof = {
            "bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": step_gbs / HBM_PEAK_GBS,
            "frac_scope": "whole step: algorithmic bytes of every output frame / HIP-event time of all launches of the step",
            "step_ms": step_ms, "algorithmic_bytes_per_step": step_alg, "bytes_per_frame": bytes_per_frame, "frames_per_step": frames_step,
            "kernel": ", ".join(sorted(set(dom_names))), "kernel_avg_ms": dom_ms,
            "algorithmic_bytes_per_launch": kern_alg, "frames_per_launch": dom_frames, "launches_timed": steps,
            "stages_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "traffic": tr[1]["total_bytes_per_step"] if tr else None,
            "traffic_unit": "L2 <-> fabric (Infinity Cache / HBM) bytes per step, all kernels, rocprofv3 PMC (Infinity-Cache hits are counted)",
            "traffic_by_kernel": tr[1].get("by_kernel") if tr else None,
            "traffic_source": os.path.relpath(tr[0], ROOT) if tr else None,
            "traffic_head": tr[1].get("git_head") if tr else None,
            "traffic_device_src_sha16": tr[1].get("device_src_sha16") if tr else None,
            "traffic_note": ("committed profile of these very device sources, made on another box by tools/archive/profile_round5.sh; not measured in this run"
                             if tr else tr_why),
        }
        if multi_kernel:
            # (a share of the step's time, NOT a roofline fraction: every launch of a multi-kernel step covers every frame, so pricing the
            # step's bytes on one of them would flatter it — round-4 review)
            roof["dominant_kernel_share"] = dom_ms / step_ms if step_ms > 0 else None
        else:
            roof["kernel_frac"] = kern_gbs / HBM_PEAK_GBS
            roof["kernel_frac_scope"] = "the single fused kernel's interior launch: algorithmic bytes of the frames it covers / its HIP-event time"
        if measured:
            # the mix ceiling: this workload's algorithmic bytes are 4C read + 8 written per frame; at the measured read-only and
            # write-only rates they cannot move faster than bytes / (read_bytes / read + write_bytes / write)
            mix = bytes_per_frame / (4 * C / measured["read"] + 8 / measured["write"])
            roof["measured"] = dict(measured, mix=mix, mix_scope=f"{4 * C} B read at `read` + 8 B written at `write` per frame")
            roof["frac_of_measured"] = step_gbs / measured["copy"]
            roof["frac_of_measured_scope"] = "achieved / measured copy-kernel ceiling of this box (SURVEY 8d); frac_of_measured_mix = achieved / measured.mix"
            roof["frac_of_measured_mix"] = step_gbs / mix
        result = {
            "metric": "stereo frames/sec @48kHz, 14ch HeSuVi HRIR",
            "value": frames_total / elapsed_max,
            "unit": "stereo frames/s",
            "n_gpus": world,
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if eq_def is None else ("f32 (convolution; the equalizer folded into the HRIR in f64 at activation)" if all(g["eq"] is None for g in legs)
                                                    else "f32 (convolution) + f64 (EQ)"),
            "data": f"synthetic U(-0.5,0.5) counter RNG seed 0xA17AE+stream, resident in HBM; HRIR: {hrir_src}",
            "config": {
                "workload": wl["desc"], "name": name, "streams_per_gpu": S, "streams_total": streams_total, "streams_this_rank": S, "frames_per_stream": g0["F"], "sample_rate": g0["rate"],
                "input_channels": C, "hrir_tracks": int(tracks.shape[0]), "hrir_taps": g0["taps"],
                "convolutions_per_stream": int((lt >= 0).sum() + (rt >= 0).sum()),
                "parallelism": f"streams sharded x{world}, no data-path collective" + (f"; {n_lanes} stream chunks per GPU on {n_lanes} HIP streams" if n_lanes > 1 else ""),
                "lanes": n_lanes,
                "fft": paths[0]["fft"], "hop": paths[0]["hop"], "path": paths[0]["path"], "legs": paths,
                "outputs_finite": finite,
                **({"equalizer": {"filters": len(eq_def.filters), "mode": "folded into the HRIR (aw_eq_fold_hrir)" if g0["eq"] is None else "cascade kernel after the spatializer",
                                  "response_taps": g0["eq_response_taps"], "tail_bound": g0["eq_tail_bound"]}} if eq_def is not None else {}),
                # creation-time cost (not in the timed region; the reference does its HRIR partition FFTs at engine init too,
                # ConvolutionEngine.swift:143-182): aw_spatializer_reserve = table build in float64 on host threads + upload + scratch pool
                # per-rank HBM footprint of this workload (input + output + history + tables + scratch pool), of the device's total
                "device_memory": {"used_by_workload_bytes": int(mem_free0 - mem_free1), "free_before_bytes": int(mem_free0), "total_bytes": int(mem_total)},
                "activation": activation,
                "activation_ms": [a["total_ms"] for a in activation],
                "device_src_sha16": prov.device_source_digest(), "host_src_sha16": prov.host_source_digest(),
                "build_flags_sha16": prov.build_flags_digest(), "build_flags_env": prov.build_flags().get("env", {}),
                "build_head": prov.build_head(),
            },
            "roofline": roof,
        }
        if len(legs) == 1:            # second roof (report which one binds): FP32 vector
            fpf = flops_per_frame(C, paths[0])
            tfl = fpf * frames_step / (step_ms * 1e-3) / 1e12
            ridge = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            result["fp32_roof"] = {"flops_per_frame": fpf, "achieved": tfl, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP32_PEAK_TFLOPS,
                                   "intensity_flop_per_byte": fpf / bytes_per_frame, "ridge_flop_per_byte": ridge,
                                   "note": "MI355X FP32 vector peak (MI355X_MICROARCH.md); step level; " +
                                           ("the HBM roof binds" if fpf / bytes_per_frame < 0.95 * ridge
                                            else "the FP32 roof binds" if fpf / bytes_per_frame > 1.05 * ridge else "both roofs bind within 5 %")}
        if eq_events:
            eq_ms = sum(a.elapsed_time(b) for a, b in eq_events) / steps
            result["roofline"]["eq_kernel_ms_per_step"] = eq_ms           # (sum over lanes: with several lanes the EQ launches overlap other lanes' kernels)
            result["roofline"]["eq_achieved_GBs"] = 16.0 * frames_step / (eq_ms * 1e-3) / 1e9      # 8 B in + 8 B out per frame
        if with_check is None:
            with_check = with_cpu
        if with_check and world == 1 and eq_def is not None and len(legs) == 1:
            # parity of the equalized configuration: the recurrence starts at the stream's first frame, so the streams are reset and one
            # more (untimed) call of the same buffers runs; two streams' first 6144 frames against the oracle — float64 convolution, then
            # the sequential Float64 cascade of ParametricEqualizerState.process on its float32 result, as the reference orders them
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import airwave_oracle as orc
            batch.reset()
            step()
            torch.cuda.synchronize()
            tr0 = aw.resample_tracks(tracks, 48000.0, float(g0["rate"]))
            od = orc.EqualizerDefinition(eq_def.preampDB, [orc.EqualizerFilter(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in eq_def.filters])
            err = 0.0
            for s_ in (0, g0["n"] - 1):
                ref = orc.spatialize_f64(g0["x"][s_, :6144].cpu().numpy(), tr0, lt, rt).astype(np.float32)
                el, er = orc.eq_prepare(od, float(g0["rate"])).process(np.ascontiguousarray(ref[:, 0]), np.ascontiguousarray(ref[:, 1]))
                err = max(err, orc.peak_rel_error(g0["y"][s_, :6144].cpu().numpy(), np.stack([el, er], axis=1)))
            result["parity_spot_err"] = err
            result["parity_spot"] = ("max peak-relative error of streams 0 and n-1, first 6144 output frames of one more call after a reset, against the float64 "
                                     "convolution followed by the oracle's sequential Float64 biquad cascade; tolerance 1e-5")
        if with_check and world == 1 and eq_def is None and len(legs) == 1:
            # parity of the TIMED configuration (same buffers, same scratch chunking, same kernels): two streams' head and tail
            tr0 = aw.resample_tracks(tracks, 48000.0, float(g0["rate"]))
            err = oracle_spot_check(g0["x"], g0["y"], tr0, lt, rt, calls=warmup + steps)
            result["parity_spot_err"] = err
            result["parity_spot"] = ("max peak-relative error of streams 0 and n-1, first 4096 and last 2048 output frames of the last timed step, "
                                     "against the float64 oracle (direct-form linear convolution; the reference pins only delta-HRIR KATs); tolerance 1e-5")
        if world == 1 and len(legs) == 1 and not args.no_warm_activation:
            # activation on a WARM context (a preset change: HRIRManager.activatePreset builds the next renderer network while the old one
            # plays): a second spatializer of the same shape on the same context finds the scratch pool already there
            t_act = time.perf_counter()
            b2 = aw.MixedRateBatch(tracks, 48000.0, layout, stream_rates, hrirMap=cmap, ctx=ctx)
            t_create = time.perf_counter()
            sp2 = next(iter(b2.buckets.values())).spatializer
            sp2.reserve(g0["F"])
            torch.cuda.synchronize()
            t_res = time.perf_counter()
            inf = sp2.info()
            result["config"]["activation_warm"] = {
                "create_ms": round((t_create - t_act) * 1e3, 1), "reserve_ms": round((t_res - t_create) * 1e3, 1),
                "tables_ms": round(inf["reserve_tables_ms"], 1), "upload_ms": round(inf["reserve_upload_ms"], 1),
                "scratch_alloc_ms": round(inf["reserve_scratch_ms"], 1), "warm_context": True}
            del b2, sp2
        if with_cpu and world == 1:      # the CPU baseline is reported at N = 1 only
            ns = max(1, min(g0["n"], args.cpu_sample_streams))
            Fc = g0["F"] if name in ("cfg1", "cfg2") else min(g0["F"], int(4 * g0["rate"]))     # long-tap configs: 4 s per stream
            x_host = g0["x"][:ns, :Fc].cpu().numpy()
            tr0 = aw.resample_tracks(tracks, 48000.0, float(g0["rate"]))
            result["cpu_baseline"] = cpu_baseline(x_host, tr0, lt, rt, Fc, eq_definition=eq_def, rate=float(g0["rate"]))
    # release the device buffers before the next workload of this run
    del legs, batch, batches
    torch.cuda.empty_cache()
    return result


def end_to_end(name: str, args, ctx, streams: int = 0, pcie: dict = None):
    """SURVEY.md 8d, secondary: the PCIe-inclusive rate of the same hot path when the host hands over HOST buffers
    (aw_spatializer_process_host: the batch crosses PCIe in chunks of streams, H2D of chunk k+1 || kernels of chunk k || D2H of chunk
    k-1).  Page-locked buffers (aw_host_alloc_pinned) and, for comparison, pageable ones.  Never `value`."""
    import numpy as np
    import torch
    import airwave_amd as aw
    wl = dict(WORKLOADS[name])
    S = streams or wl["streams"]
    C, rate = wl["channels"], 48000
    F = int(round((args.seconds or wl["seconds"]) * rate))
    tracks, _ = load_hrir(wl["hrir"], wl["taps"])
    layout = aw.InputLayout.detect(C) if C != 7 else aw.InputLayout(SPEAKERS7, "7 speakers")
    cmap = None
    if wl.get("text_map"):
        cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(ROOT, "tests", "golden", wl["text_map"])).read())
    batch = aw.MixedRateBatch(tracks, 48000.0, layout, [rate] * S, hrirMap=cmap, ctx=ctx)
    sp = batch.buckets[float(rate)].spatializer
    x_dev = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x_dev.data_ptr(), S, F, C, seed=0xA17AE, first_stream=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x_pin, y_pin = ctx.pinned_empty((S, F, C)), ctx.pinned_empty((S, F, 2))
    pin_ms = (time.perf_counter() - t0) * 1e3
    ctx.d2h(x_pin, x_dev.data_ptr())
    del x_dev
    torch.cuda.empty_cache()
    sp.reserve_host(F)
    ctx.synchronize()
    in_b, out_b = x_pin.nbytes, y_pin.nbytes

    def timed(x, y, reps):
        best = 1e30
        for _ in range(reps):
            t = time.perf_counter()
            sp.process_host_into(x, y)          # synchronous
            best = min(best, time.perf_counter() - t)
        return best
    timed(x_pin, y_pin, 1)                       # warm-up (first touch of the pinned pages by the DMA engines, clocks)
    calls = 1
    t_pin = timed(x_pin, y_pin, 3); calls += 3
    # parity of what just crossed PCIe both ways: two streams' head and tail of the last call against the float64 oracle
    err = oracle_spot_check(x_pin, y_pin, np.asarray(tracks), batch.left_track, batch.right_track, calls=calls) if not args.no_cpu_baseline else None
    x_page, y_page = np.array(x_pin), np.empty_like(y_pin)             # pageable copies
    t_page = timed(x_page, y_page, 5)          # (pageable host memory is the noisy leg: page placement, the copy threads' scheduling — best of five)
    chunk = sp.info()["host_chunk_streams"]
    res = {"workload": wl["desc"] + (f" — first {S} of {wl['streams']} streams (the PCIe-bound rate does not depend on the batch size)" if S != wl["streams"] else ""),
           "name": name, "streams": S, "frames_per_stream": F, "input_channels": C,
           "value": S * F / t_pin, "unit": "stereo frames/s, host buffers in -> host buffers out (PCIe inclusive)", "pinned": True,
           "ms_per_batch": t_pin * 1e3, "h2d_GBs": in_b / t_pin / 1e9, "d2h_GBs": out_b / t_pin / 1e9,
           "bytes_in": in_b, "bytes_out": out_b, "streams_per_chunk": chunk, "chunks": -(-S // chunk) if chunk else 1,
           "pipeline": "H2D of chunk k+1 || kernels of chunk k || D2H of chunk k-1 on three HIP streams; two staged chunks each way",
           "pageable": {"value": S * F / t_page, "ms_per_batch": t_page * 1e3, "pinned": False,
                        "note": "same entry on pageable numpy arrays: bounced through page-locked chunks by the context's host copy threads (input and output directions on pools of their own)"},
           "pinned_alloc_ms": round(pin_ms, 1)}
    if err is not None:
        res["parity_spot_err"] = err
    if pcie:
        serial = in_b / (pcie["h2d"] * 1e9) + out_b / (pcie["d2h"] * 1e9)
        res["pcie_measured"] = dict(pcie, unit="GB/s", source="aw_context_pcie_probe: 1 GiB page-locked hipMemcpyAsync each way, and both at once")
        res["pcie_only_ms"] = serial * 1e3
        res["frac_of_pcie"] = serial / t_pin
        res["frac_of_pcie_scope"] = ("time of a page-locked H2D of the input followed by a D2H of the output at the measured rates (no kernels) "
                                     "/ time of the host entry; above 1.0 = the two directions overlapped")
    del batch, sp, x_pin, y_pin, x_page, y_page
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------- the stdout line
LINE_LIMIT = 8000          # bytes: the driver keeps an 8 KB tail of stdout; a line that does not fit in it whole is a line nobody parsed (round 5)
DETAIL_FILE = "bench_detail.json"


def _r(v, nd=6):
    """Numbers on the line carry what a reader compares, not 17 digits."""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}")
    return v


def _pick(d: dict, keys, nd=6) -> dict:
    return {k: _r(d[k], nd) for k in keys if k in d and d[k] is not None}


def compact_roofline(r: dict, full: bool) -> dict:
    keys = ["bound", "achieved", "peak", "unit", "frac", "step_ms", "kernel", "kernel_avg_ms", "kernel_frac", "dominant_kernel_share"]
    out = _pick(r, keys)
    out["traffic"] = r.get("traffic")              # bytes per step (PMC, committed profile of these sources) or null — the key is always there
    if r.get("traffic") is not None and r.get("algorithmic_bytes_per_step"):
        out["traffic_over_algorithmic"] = _r(r["traffic"] / r["algorithmic_bytes_per_step"], 4)
    if full:
        out.update(_pick(r, ["algorithmic_bytes_per_step", "bytes_per_frame", "frames_per_step", "launches_timed", "traffic_source",
                             "frac_of_measured", "frac_of_measured_mix", "eq_kernel_ms_per_step"]))
        if r.get("traffic") is None and r.get("traffic_note"):
            out["traffic_note"] = r["traffic_note"][:200]
        out["stages_ms_per_step"] = r.get("stages_ms_per_step", {})
        if "measured" in r:
            out["measured"] = _pick(r["measured"], ["read", "write", "copy", "mix", "unit"], 5)
    else:
        out.update(_pick(r, ["frac_of_measured_mix"]))
    return out


def compact_config(c: dict, full: bool) -> dict:
    keys = ["workload", "name", "streams_per_gpu", "frames_per_stream", "sample_rate", "input_channels", "hrir_tracks", "hrir_taps",
            "convolutions_per_stream", "parallelism", "fft", "hop", "path"]
    if full:
        keys += ["equalizer", "streams_total", "streams_this_rank", "lanes", "outputs_finite", "activation_ms", "device_src_sha16", "host_src_sha16", "build_flags_sha16", "build_head"]
    if not full:          # a secondary: what identifies the workload and the kernels that ran (its description is in bench_detail.json)
        keys = ["name", "streams_per_gpu", "frames_per_stream", "input_channels", "hrir_taps", "fft", "hop", "path"]
    out = _pick(c, keys)
    if len(c.get("legs", [])) > 1:
        out["legs"] = [_pick(l, ["rate", "streams", "frames", "taps", "fft", "hop", "partitions"]) for l in c["legs"]]
    return out


def compact_secondary(r: dict) -> dict:
    out = _pick(r, ["value", "steps", "warmup", "ms_per_step", "parity_spot_err", "note"])
    out["config"] = compact_config(r["config"], full=False)
    out["roofline"] = compact_roofline(r["roofline"], full=False)
    if "fp32_roof" in r:
        out["fp32_frac"] = _r(r["fp32_roof"]["frac"], 4)
    return out


def compact_line(result: dict) -> dict:
    """The ONE stdout line: the contract keys, `roofline`, `cpu_baseline`, the parity spot check and a few numbers per secondary.
    Everything else this run measured (per-kernel PMC counters, activation breakdown, PCIe detail, scopes and notes) is in
    bench_detail.json next to bench.py.  tests/test_bench_line.py holds the length bound on a canned worst case."""
    head = ["metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    line = _pick(result, head, 9)
    line["vs_baseline"] = result.get("vs_baseline")          # null is part of the contract
    line["config"] = compact_config(result["config"], full=True)
    line["roofline"] = compact_roofline(result["roofline"], full=True)
    if "fp32_roof" in result:
        line["fp32_roof"] = _pick(result["fp32_roof"], ["achieved", "peak", "unit", "frac", "intensity_flop_per_byte", "ridge_flop_per_byte"], 5)
    if "cpu_baseline" in result:
        line["cpu_baseline"] = _pick(result["cpu_baseline"], ["value", "unit", "cores", "kind", "single_thread_value", "sample", "seconds"])
    if "parity_spot_err" in result:
        line["parity_spot_err"] = _r(result["parity_spot_err"], 4)
        line["parity_tolerance"] = 1e-5
    for k in ("secondary", "secondary_cfg2", "secondary_cfg2_14ch", "secondary_eq_cascade"):
        if k in result:
            line[k] = compact_secondary(result[k])
    if "secondary_end_to_end" in result:
        line["secondary_end_to_end"] = [dict(_pick(e, ["name", "streams", "value", "ms_per_batch", "h2d_GBs", "d2h_GBs", "frac_of_pcie", "parity_spot_err"]),
                                             pinned=True, pageable_value=_r(e["pageable"]["value"]), unit="stereo frames/s, host buffers in and out (PCIe inclusive)")
                                        for e in result["secondary_end_to_end"]]
    line["detail"] = DETAIL_FILE
    return line


def emit(result: dict) -> str:
    """Writes the full result to bench_detail.json (next to bench.py, and under gpurun_out/ when that exists) and returns the compact
    stdout line.  A line over LINE_LIMIT is a bug: secondaries are dropped from it (they stay in the detail file) rather than printed."""
    full = json.dumps(result, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    line = compact_line(result)
    text = json.dumps(line, separators=(", ", ": "))
    for k in ("secondary_end_to_end", "secondary_eq_cascade", "secondary_cfg2_14ch", "secondary_cfg2", "secondary"):
        if len(text) < LINE_LIMIT:
            break
        if line.pop(k, None) is not None:
            line["dropped_for_length"] = line.get("dropped_for_length", []) + [k]
        text = json.dumps(line, separators=(", ", ": "))
    return text


# ---------------------------------------------------------------------------------------------- main
def dry_run(args, world: int, rank: int) -> None:
    """GPU-less rehearsal of the launch / rendezvous / aggregate path (tests/test_multi_rank.py): every rank "processes"
    its shard for a fixed time.  The line it prints says so and carries no measurement."""
    import torch
    import torch.distributed as dist
    from airwave_amd.sharding import aggregate_throughput, plan_streams
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = WORKLOADS[args.workload]
    rates = wl.get("rates", [48000])
    S_req = args.streams or (TOTAL_STREAMS[args.workload] if args.scaling == "strong" else 4)
    ids, stream_rates = plan_streams(S_req, rates, world, rank, args.scaling)
    if world > 1:
        dist.barrier()
    frames, elapsed, _ = aggregate_throughput(float(len(ids) * 1000), 0.25, device="cpu")
    # every rank's shard, gathered for the line (tests/test_multi_rank.py asserts the split): counts, first / last global id, id checksum
    mine = torch.tensor([len(ids), ids[0] if ids else -1, ids[-1] if ids else -1, sum(ids)] + [stream_rates.count(r) for r in rates], dtype=torch.int64)
    shards = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(shards, mine)
    else:
        shards = [mine]
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (no GPU work, no measurement)", "dry_run": True, "value": 0.0, "unit": "stereo frames/s",
                          "n_gpus": world, "ranks_seen": dist.get_world_size() if world > 1 else 1, "frames_all_ranks": frames,
                          "scaling": args.scaling, "streams_total": S_req if args.scaling == "strong" else S_req * world,
                          "streams_by_rank": [int(t[0]) for t in shards], "first_id_by_rank": [int(t[1]) for t in shards],
                          "last_id_by_rank": [int(t[2]) for t in shards], "id_sum": int(sum(int(t[3]) for t in shards)),
                          "streams_by_rank_and_rate": [[int(v) for v in t[4:]] for t in shards], "rates": rates,
                          "self_launched": os.environ.get("AW_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default per workload (cfg3: 10)")
    ap.add_argument("--warmup", type=int, default=None, help="default per workload (cfg3: 2)")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU (weak) or the job total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): per-GPU batch fixed as N grows; strong: BASELINE's job-total stream counts (cfg3 1024 / cfg4 4096 / cfg5 8192) split over the ranks")
    ap.add_argument("--seconds", type=float, default=0.0, help="override seconds per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the second reading of the workload (cfg3: the 14-channel-input run)")
    ap.add_argument("--cpu-sample-streams", type=int, default=48)
    ap.add_argument("--dry-run", action="store_true", help="no GPU: rehearse launch + rendezvous + aggregate only (tests)")
    ap.add_argument("--lanes", type=int, default=0, help="run the batch as this many chunks of streams on HIP streams of their own (default per workload; cfg4: see WORKLOADS)")
    ap.add_argument("--eq", default="auto", choices=["auto", "fold", "cascade"],
                    help="cfg4's equalizer: folded into the HRIR at activation (auto / fold) or the Float64 biquad kernel after the spatializer (cascade)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the measured read / write / copy ceiling probe (roofline.measured)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive secondary legs (secondary_end_to_end)")
    ap.add_argument("--no-warm-activation", action="store_true", help="skip the second, warm-context activation (config.activation_warm)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # N > 1 and not yet under a launcher: become the launcher.  Nothing above has touched the GPU (no torch.cuda call).
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.dry_run and os.environ.get("AW_BENCH_BACKEND", "nccl") == "nccl":
            import torch
            have = torch.cuda.device_count()          # counting devices does not initialise the GPU
            if have < args.gpus:
                print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
                return 2
        return self_launch(args.gpus, sys.argv[1:])

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}", file=sys.stderr)
        return 2
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        dry_run(args, world, rank)
        return 0

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # Debug overrides (a 1-GPU box can still exercise the N > 1 code path: AW_BENCH_DEVICE=0 AW_BENCH_BACKEND=gloo).
    if "AW_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["AW_BENCH_DEVICE"])
    backend = os.environ.get("AW_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    torch.cuda.set_device(local_rank)
    # AW_BENCH_FORCE_PG=1 (debug, one GPU): a ONE-rank process group on the chosen backend, so that the RCCL branch below — group
    # creation with device_id, the all-reduce on device tensors — executes on hardware that has a single GPU.  No scaling claim.
    force_pg = world == 1 and os.environ.get("AW_BENCH_FORCE_PG") == "1"
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or force_pg:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"rendezvous produced {dist.get_world_size()} ranks for --gpus {args.gpus}")

    import airwave_amd as aw
    print(f"[bench rank {rank}/{world}] device {local_rank}: {torch.cuda.get_device_name(local_rank)}, backend {backend if world > 1 or force_pg else 'none'}"
          + (" (forced one-rank process group)" if force_pg else ""), file=sys.stderr, flush=True)
    ctx = aw.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    measured = measured_ceilings(ctx) if rank == 0 and not args.no_ceiling else None       # before any timed region, on this box
    result = run_workload(args.workload, args, ctx, world, rank, backend, with_cpu=not args.no_cpu_baseline, measured=measured)
    sec = SECONDARY.get(args.workload)
    if sec and not args.no_secondary and not args.streams and not args.seconds:
        r2 = run_workload(sec, args, ctx, world, rank, backend, with_cpu=False, with_check=not args.no_cpu_baseline, measured=measured)
        if rank == 0:
            result["secondary"] = {k: r2[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "fp32_roof", "parity_spot_err") if k in r2}
            result["secondary"]["note"] = "same configuration read as 14-channel INPUT (north_star: 'synthetic 48 kHz 14-ch input'); value is never the headline"
    if args.workload == "cfg3" and not args.no_secondary and not args.streams and not args.seconds:
        # the on-chip family beside the headline: cfg 2 (BASELINE configs[1]) with its own steady-state step counts, whatever --steps /
        # --warmup say (they are the headline's: 25 steps of 1.3 ms would time a clock that is still ramping, WORKLOADS above)
        a2 = argparse.Namespace(**vars(args))
        a2.steps = a2.warmup = None
        r3 = run_workload("cfg2", a2, ctx, world, rank, backend, with_cpu=False, with_check=not args.no_cpu_baseline, measured=measured)
        if rank == 0:
            result["secondary_cfg2"] = {k: r3[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "fp32_roof", "parity_spot_err") if k in r3}
            result["secondary_cfg2"]["note"] = "BASELINE configs[1] (7.1 -> RoomSH1.0, 128 streams x 10 s, all state on chip: the fused 8192-frame tile); value is never the headline"
        # ... and the same configuration with 14-channel input, north_star's literal layout ("synthetic 48 kHz 14-ch input"), on chip too since round 6
        r5 = run_workload("cfg2-14ch", a2, ctx, world, rank, backend, with_cpu=False, with_check=not args.no_cpu_baseline, measured=measured)
        if rank == 0:
            result["secondary_cfg2_14ch"] = {k: r5[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "fp32_roof", "parity_spot_err") if k in r5}
            result["secondary_cfg2_14ch"]["note"] = "cfg 2 with 14-channel input through the parseHeSuViFormat text map (the overlap-add tile); value is never the headline"
    if args.workload == "cfg4" and args.eq != "cascade" and not args.no_secondary and not args.streams and not args.seconds:
        # the same configuration with the equalizer as a SEPARATE pass (the reference's structure: AudioEffectGraph runs two effects)
        r4 = run_workload("cfg4", args, ctx, world, rank, backend, with_cpu=False, with_check=not args.no_cpu_baseline, measured=measured, eq_mode="cascade")
        if rank == 0:
            result["secondary_eq_cascade"] = {k: r4[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "parity_spot_err") if k in r4}
            result["secondary_eq_cascade"]["note"] = "cfg 4 with the equalizer as its own Float64 biquad-cascade pass over the stereo output (round 5's form); value is never the headline"
    e2e_of = {"cfg3": [("cfg3", 128), ("cfg2", 0)], "cfg2": [("cfg2", 0)], "cfg2-14ch": [("cfg2-14ch", 0)], "cfg1": [("cfg1", 0)]}
    if rank == 0 and world == 1 and not args.no_end_to_end and not args.streams and not args.seconds and args.workload in e2e_of:
        # SURVEY 8d "end-to-end incl. PCIe as secondary": the same hot path fed from host memory; never `value`
        pcie = ctx.pcie_probe(1 << 30, 2)
        result["secondary_end_to_end"] = [end_to_end(n, args, ctx, streams=k, pcie=pcie) for n, k in e2e_of[args.workload]]
    rc = 0
    if rank == 0:
        print(emit(result), flush=True)
        errs = [e for e in [result.get("parity_spot_err"), result.get("secondary", {}).get("parity_spot_err"), result.get("secondary_cfg2", {}).get("parity_spot_err"), result.get("secondary_cfg2_14ch", {}).get("parity_spot_err"),
                           result.get("secondary_eq_cascade", {}).get("parity_spot_err")]
                + [r.get("parity_spot_err") for r in result.get("secondary_end_to_end", [])] if e is not None]
        if any(not (e < 1e-5) for e in errs):
            print(f"bench.py: parity spot check FAILED: {errs} (tolerance 1e-5)", file=sys.stderr)
            rc = 3
    if dist.is_initialized():
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
