"""The overlap-add tile (tile_ola.hpp) against what served the layout before it (the overlap-save tiles on 8192- or 16384-frame windows,
whichever the round-4 policy picks: AW_OLA=0) and against the long-window kernels (AW_LW=128), by layout and HRIR length: G stereo
frames/s, S streams x SECONDS s at 48 kHz, synthetic HRIR.  Feeds ola_policy() and lw_fused_crossover_taps() in runtime.cpp.
    python tools/ola_sweep.py [channels ...]     (env: S=128 SECONDS=10 TAPS=...)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S = int(os.environ.get("S", "128"))
F = int(float(os.environ.get("SECONDS", "10")) * 48000)
TAPS = [int(t) for t in os.environ.get("TAPS", "2048,3000,3585,3969,4097,4098,4320,4609,4610,5121,5122,5633").split(",")]
os.environ["AW_OLA_MIN_BLOCKS"] = "0"
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
for C in [int(a) for a in sys.argv[1:]] or [4, 6, 7, 8, 10, 12, 14, 16]:
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    for taps in TAPS:
        h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
        row, what = [], []
        for ola, lw in (("0", "0"), ("1", "0"), ("0", "128")):
            os.environ["AW_OLA"] = ola; os.environ["AW_LW"] = lw
            sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
            sp.reserve(F)
            for _ in range(3):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                sp.process_device(x.data_ptr(), y.data_ptr(), F)
            torch.cuda.synchronize()
            row.append(S * F * 6 / (time.perf_counter() - t0) / 1e9)
            i = sp.info()
            what.append(f"lw{i['long_window_rows']}" if i["long_window_rows"] else f"ola{i['overlap_add_rows']}" if i["overlap_add_rows"] else (f"ols{i['fft']}" if i["path"] == 0 else "partitioned"))
            del sp
        best = max(range(3), key=lambda k: row[k])
        print(f"C={C:2d} taps {taps:6d} S={S}: {what[0]:9s} {row[0]:7.2f}   {what[1]:9s} {row[1]:7.2f}   {what[2]:6s} {row[2]:7.2f} Gframes/s  -> {what[best]}  (ola/old {row[1] / row[0]:.3f}, ola/lw {row[1] / row[2]:.3f})", flush=True)
    del x, y
    torch.cuda.empty_cache()
