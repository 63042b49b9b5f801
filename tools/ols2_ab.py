"""A/B helper: G stereo frames/s of the 16384-frame-window kernels by layout (S streams x 10 s, synthetic HRIR of TAPS taps).
python tools/ols2_ab.py [channels ...]   (env AIRWAVE_HIP_LIBRARY selects the library variant)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AW_LW"] = "0"
import airwave_amd as aw
S, F = int(os.environ.get("S", "128")), 480000
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
out = []
for C in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6, 7, 8]:
    taps = int(os.environ.get("TAPS", "4320"))
    x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
    h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
    os.environ["AW_WINDOW"] = os.environ.get("WINDOW", "16384")
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    for _ in range(3): sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    out.append(f"C={C}: fft {sp.info()['fft']} {S * F * 10 / (time.perf_counter() - t0) / 1e9:.1f}")
    del sp, x, y
print("  ".join(out))
