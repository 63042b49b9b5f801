"""Per-kernel times of the 8192-frame kernels by channel count (128 streams x 4 s, 4320 taps): CHANNELS=7,8,14 python tools/archive/wide_stages.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for C in tuple(int(c) for c in os.environ.get("CHANNELS", "7,8,14").split(",")):
    S, F, taps = 128, int(os.environ.get("FRAMES", 192000)), 4320
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    for _ in range(2):
        sp.process_device(x.data_ptr(), y.data_ptr(), F)
    sp.set_profiling(True)
    N = 4
    for _ in range(N):
        sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    st = sp.stage_times()
    tot = sum(ms for _, ms, _ in st) / N
    print(f"C={C}: {S * F / tot / 1e6:.2f} G frames/s, {tot:.3f} ms/step: " + "; ".join(f"{n.split('<')[0][-24:]}<{n.split('<')[1] if '<' in n else ''} {ms / N:.3f} ms x{k // N}" for n, ms, k in st))
    del sp, x, y
