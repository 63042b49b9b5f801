"""Long-window kernels (tile_lw.hpp) against the default choice of aw_spatializer_create for path-0 layouts (HRIRs one fused
window can hold), by HRIR length: G stereo frames/s, S streams x SECONDS s at 48 kHz, synthetic HRIR.  Feeds
lw_fused_crossover_taps() in runtime.cpp.   python tools/lw_sweep.py [channels ...]     (env: S=128 SECONDS=10 TAPS=...)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S = int(os.environ.get("S", "128"))
F = int(float(os.environ.get("SECONDS", "10")) * 48000)
TAPS = [int(t) for t in os.environ.get("TAPS", "1024,2048,3000,4320,6145,8640,12288").split(",")]
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for C in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8]:
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    for taps in TAPS:
        h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
        row, what = [], []
        for lw in ("0", "128"):
            os.environ["AW_LW"] = lw
            sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
            sp.reserve(F)
            for _ in range(2):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            row.append(S * F * 5 / (time.perf_counter() - t0) / 1e9)
            i = sp.info()
            what.append(f"lw{i['long_window_rows']}" if i["long_window_rows"] else (f"fused{i['fft']}" if i["path"] == 0 else "partitioned"))
            del sp
        print(f"C={C:2d} taps {taps:6d} S={S}: {what[0]:12s} {row[0]:7.2f}   {what[1]:6s} {row[1]:7.2f} Gframes/s  -> {'long-window' if row[1] > row[0] else what[0]}", flush=True)
    del x, y
    torch.cuda.empty_cache()
