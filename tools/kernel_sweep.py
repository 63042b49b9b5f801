"""8192-frame window kernels: one workgroup per CU (tile_ols.hpp) against the sibling-workgroup cut (tile_olsh.hpp,
AW_KERNEL_H=2), frames/s by channel count and HRIR length (synthetic HRIR).  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

os.environ["AW_WINDOW"] = "8192"
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for C in (2, 4, 6, 8, 12, 14, 16):
    S, F = (512 if C <= 4 else 256), 96000
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    for taps in (1024, 4320):
        h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
        row = []
        for k in ("0", "2"):
            os.environ["AW_KERNEL_H"] = k
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
        print(f"C={C:2d} taps {taps:5d}: one-per-CU {row[0]:7.2f}  siblings {row[1]:7.2f} Gframes/s  -> {'siblings' if row[1] > row[0] else 'one-per-CU'}", flush=True)
    del x, y
