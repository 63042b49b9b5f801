"""Path choice by channel count and HRIR length: two AW_WINDOW settings side by side (default WINS=16384,4096: the 16384-frame
window kernels against the partitioned path, which AW_WINDOW=4096 forces; WINS=8192,4096 for the end of the 8192-frame
window's range).  TAPS=a,b,c overrides the HRIR lengths.  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for C in (int(a) for a in (sys.argv[1:] or ["9", "12", "14", "16"])):
    S, F = 128, 192000
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    for taps in [int(a) for a in os.environ.get("TAPS", "6146,8640,12289").split(",")]:
        h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
        row = []
        for win in os.environ.get("WINS", "16384,4096").split(","):
            os.environ["AW_WINDOW"] = win
            sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
            for _ in range(2):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            row.append(S * F * 4 / (time.perf_counter() - t0) / 1e9)
            del sp
        print(f"C={C:2d} taps {taps:5d}: first {row[0]:6.2f}  second {row[1]:6.2f} Gframes/s (WINS order)  -> {'first' if row[0] > row[1] else 'second'}", flush=True)
    del x, y
