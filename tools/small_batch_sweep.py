"""Small batches: 8192- against 16384-frame windows by stream count (10 s per stream, 4320 taps).  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
taps, F = 4320, 480000
h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
for C in (1, 2, 3, 5):
    for S in (1, 2, 4, 8, 12, 16, 24):
        x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
        ctx.synth_fill(x.data_ptr(), S, F, C)
        lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
        row = []
        for win in ("8192", "16384"):
            os.environ["AW_WINDOW"] = win
            sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
            for _ in range(3):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            row.append(S * F * 10 / (time.perf_counter() - t0) / 1e9)
            del sp
        print(f"C={C} S={S:2d}: 8192 {row[0]:7.2f}  16384 {row[1]:7.2f}  -> {'16384' if row[1] > row[0] else '8192'}", flush=True)
        del x, y
