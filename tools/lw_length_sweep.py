"""Throughput of the long-window policy by call length: 2 ... 60 s in 1-s steps at 48 kHz (the reference's cost per frame does not
depend on the stream length, ConvolutionEngine.swift:93,232-367; this path's depends on how well the call fills its windows).
    python tools/lw_length_sweep.py [channels]      (env: S=1024 TAPS=32768 STEP=1 MAXSEC=60)
Prints G frames/s, the windows chosen, and at the end the spread over all lengths >= 5 s."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S = int(os.environ.get("S", "1024")); taps = int(os.environ.get("TAPS", "32768"))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
maxsec = int(os.environ.get("MAXSEC", "60"))
Fmax = maxsec * 48000
x = torch.empty((S * Fmax * C,), device="cuda"); y = torch.empty((S * Fmax * 2,), device="cuda")
ctx.synth_fill(x.data_ptr(), S, Fmax, C)
rates = {}
for sec in range(2, maxsec + 1, int(os.environ.get("STEP", "1"))):
    F = sec * 48000
    reps = max(2, int(1.5e9 // (S * F)))
    for _ in range(2): sp.process_device(x.data_ptr(), y.data_ptr(), F)         # (builds the tables of new window lengths)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    g = S * F * reps / (time.perf_counter() - t0) / 1e9
    i = sp.info()
    rates[sec] = g
    print(f"C={C} taps {taps} S={S} {sec:3d} s: {g:6.2f} G frames/s  windows: rows {i['long_window_rows']:3d} + rest {i['long_window_rows_rest']:3d}", flush=True)
sel = [v for k, v in rates.items() if k >= 5]
print(f"lengths >= 5 s: best {max(sel):.2f}, worst {min(sel):.2f} G frames/s: worst / best = {min(sel) / max(sel):.3f}")
