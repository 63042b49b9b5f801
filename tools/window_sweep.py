"""Fused window choice: frames/s of the 8192- and 16384-frame window kernels against HRIR length (8 channels,
128 streams x 4 s, synthetic HRIR).  Run on the GPU box; AW_WINDOW is read at spatializer creation."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S, F, C = 128, 192000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
ctx.synth_fill(x.data_ptr(), S, F, C)
lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
rng = np.random.default_rng(1)
for taps in [int(t) for t in os.environ.get("TAPS", "512,1024,2048,3000,3600,4320,5000,5600,6145").split(",")]:
    h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
    row = []
    for win in (8192, 16384):
        os.environ["AW_WINDOW"] = str(win)
        sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        for _ in range(2):
            sp.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sp.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        row.append(S * F * 5 / (time.perf_counter() - t0) / 1e9)
        del sp
    print(f"C={C} taps {taps:5d}: 8192-window {row[0]:6.2f}  16384-window {row[1]:6.2f} Gframes/s  -> {'16384' if row[1] > row[0] else '8192'}")
