"""Per-call choice for long HRIRs (path-1 spatializers): partitioned kernels (AW_LW=0) against the policy's choice (lw_choose in
runtime.cpp) by call length.   python tools/lw_calls_sweep.py [channels]      (env: S=128 TAPS=32768)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S = int(os.environ.get("S", "128")); taps = int(os.environ.get("TAPS", "32768"))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
for F in (8192, 16384, 32768, 49152, 65536, 98304, 131072, 196608, 262144, 480000):
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    row = []
    for lw in ("0", None, "32", "64", "128"):
        if lw is None: os.environ.pop("AW_LW", None)
        else: os.environ["AW_LW"] = lw
        sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        sp.reserve(F)
        reps = max(3, int(2e8 // (S * F)))
        for _ in range(2): sp.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): sp.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        row.append((S * F * reps / (time.perf_counter() - t0) / 1e9, sp.info()["long_window_rows"]))
        del sp
    print(f"C={C} taps {taps} S={S} frames {F:7d}: partitioned {row[0][0]:6.2f} | policy {row[1][0]:6.2f} (rows {row[1][1]:3d}) | forced 32: {row[2][0]:6.2f}  64: {row[3][0]:6.2f}  128: {row[4][0]:6.2f} G frames/s", flush=True)
    del x, y
    torch.cuda.empty_cache()
