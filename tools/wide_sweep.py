"""Wide layouts (9-16 channels) on the 8192-frame kernels: frames/s by channel count.  Run on the GPU box, per library variant."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
out = []
for C in (tuple(int(c) for c in os.environ["CHANNELS"].split(",")) if "CHANNELS" in os.environ else (9, 10, 11, 13, 14, 15, 16, 12)):
    S, F, taps = 128, 192000, 4320
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    for _ in range(2):
        sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    out.append(f"C={C}: {S * F * 4 / (time.perf_counter() - t0) / 1e9:.2f}")
    del sp, x, y
print("  ".join(out))
