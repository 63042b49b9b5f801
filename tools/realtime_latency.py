"""Wall time per callback of the plug-in shaped (host-buffer, one-stream) entries — the reference's own perf workloads
(AirwaveTests/RealtimeAudioProcessorTests.swift:128-166 and ParametricEqualizerProcessorTests.swift:317-357: 10 s of stereo
at 48 kHz in callbacks of 128 / 512 / 1024 frames) — against the callback period.  Each call is a synchronous
host -> device copy, kernel launches and a device -> host copy.  Run on the GPU box:

    python tools/realtime_latency.py [--seconds 10] > gpurun_out/realtime_latency.json
"""
import argparse, ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw
from airwave_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=10.0)
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = _capi.load()
fp = ctypes.POINTER(ctypes.c_float)
rate = 48000.0


def run(call, cb, total):
    n = int(total * rate) // cb
    rng = np.random.default_rng(1)
    il = (rng.random(cb, dtype=np.float32) - 0.5); ir = (rng.random(cb, dtype=np.float32) - 0.5)
    ol = np.zeros(cb, np.float32); orr = np.zeros(cb, np.float32)
    a = [x.ctypes.data_as(fp) for x in (il, ir, ol, orr)]
    for _ in range(50):
        call(*a, cb)
    t = np.empty(n)
    for i in range(n):
        t0 = time.perf_counter_ns()
        rc = call(*a, cb)
        t[i] = time.perf_counter_ns() - t0
    assert rc == 0 and np.isfinite(ol).all()
    t /= 1e3
    period = cb / rate * 1e6
    return {"callback_frames": cb, "callbacks": n, "period_us": round(period, 1), "p50_us": round(float(np.percentile(t, 50)), 1),
            "p99_us": round(float(np.percentile(t, 99)), 1), "max_us": round(float(t.max()), 1), "mean_us": round(float(t.mean()), 1),
            "p99_over_period": round(float(np.percentile(t, 99)) / period, 3), "over_budget": int((t > period).sum())}


ctx = aw.Context(0)
wav = aw.WAVLoader.load(os.path.join(ROOT, "tests", "golden", "hrtf", "NeutralSH1.0.wav"))
out = {"what": "wall time per callback (perf_counter around the C call), 10 s of stereo @48 kHz per callback size; budget = callback period",
       "surfaces": {}}
for name, taps_src in (("bundled NeutralSH1.0 (4320 taps)", wav.audio_data), ("reference perf workload: 1-tap gains (RealtimeAudioProcessorTests.swift:8-28)", None)):
    if taps_src is None:
        tr = np.zeros((14, 1), np.float32); tr[0, 0] = 1.0; tr[1, 0] = 2.0; tr[8, 0] = 1.0; tr[7, 0] = 2.0
    else:
        tr = np.asarray(taps_src, np.float32)
    hrir = aw.HRIR(tr, ctx=ctx)
    rt = aw.RealtimeAudioProcessor(hrir, [(0, 1), (8, 7)], blockSize=512, maxFramesPerCallback=4096)
    sp = aw.Spatializer(hrir, np.array([0, 8], np.int32), np.array([1, 7], np.int32), n_streams=1, ctx=ctx)
    sp.reserve(4096)
    rows = {}
    rows["aw_realtime_process"] = [run(lambda a, b, c, d, n: lib.aw_realtime_process(rt._h, a, b, c, d, n), cb, args.seconds) for cb in (128, 512, 1024)]
    rows["aw_spatializer_process_planar"] = [run(lambda a, b, c, d, n: lib.aw_spatializer_process_planar(sp._h, a, b, c, d, n), cb, args.seconds) for cb in (128, 512, 1024)]
    out["surfaces"][name] = rows
d = aw.EqualizerAPOParser.parse(open(os.path.join(ROOT, "tests", "golden", "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "CCA CRA ParametricEq.txt")
eq = aw.ParametricEqualizerProcessor(rate, 4096, ctx=ctx)
eq.setTarget(d)
out["surfaces"]["10-band fixture EQ (ParametricEqualizerProcessorTests.swift:317-357)"] = {
    "aw_eq_process_planar": [run(lambda a, b, c, d2, n: lib.aw_eq_process_planar(eq._h, a, b, c, d2, n), cb, args.seconds) for cb in (128, 512, 1024)]}
print(json.dumps(out, indent=1))
