#!/usr/bin/env python3
"""Diagnostic: phase breakdown of the 16384-window tile kernel (tile_ols2.hpp) from s_memtime stamps of one
wave per workgroup.  Run with AIRWAVE_HIP_LIBRARY=airwave_amd/libairwave_hip_stamps.so AW_WINDOW=16384 and
AW_STAMP_THREAD=<thread index> (0 = wave 0, 256 = wave 4 ...).  Shares, not absolute speed."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import airwave_amd as aw
from airwave_amd import _capi

S, F, C = 128, 240000, 8
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
w = aw.WAVLoader.load(os.path.join(ROOT, "tests/golden/hrtf/RoomSH1.0.wav"))
lay = aw.InputLayout.detect(C)
lt, rt = aw.HRIRChannelMap.hesuvi14Channel(lay).resolve(lay, 14)
sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
assert sp.info()["fft"] == 16384, "run with AW_WINDOW=16384"
x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
ctx.synth_fill(x.data_ptr(), S, F, C)
for _ in range(3):
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
torch.cuda.synchronize()
nwg = S * ((F + sp.info()["hop"] - 1) // sp.info()["hop"])
buf = np.zeros((nwg, 32), dtype=np.uint64)
n = ctypes.c_int64()
st = _capi.load().aw_spatializer_debug_stamps(sp._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size, ctypes.byref(n))
assert st == 0, _capi.load().aw_last_error_message()
t = buf.astype(np.int64)
t = t[t[:, 31] > 0]
print("workgroups with stamps:", len(t))
tot = t[:, 31] - t[:, 0]
print(f"tile cycles (last tile of each workgroup): median {np.median(tot):.0f} mean {tot.mean():.0f}")
def seg(a, b, name):
    d = t[:, b] - t[:, a]
    print(f"  [{a:2d}->{b:2d}] {name:44s} median {np.median(d):8.0f}  ({np.median(d)/np.median(tot)*100:4.1f}%)")
for b in range(4):
    o = 4 * b
    seg(o, o + 1, f"batch {b}: pass 1 of both pairs (waits frames)")
    seg(o + 1, o + 2, f"batch {b}: barrier")
    seg(o + 2, o + 3, f"batch {b}: first pair sub-FFT + 2x CMAC")
    if b < 3:
        seg(o + 3, o + 4, f"batch {b}: prefetch + second pair + barrier")
seg(15, 30, "batch 3: second pair")
seg(30, 31, "2 inverse sub-FFTs, prefetch, barrier, final, store")
print("detail of the last forward pair / inverse (stamps 16..29 as in tools/archive/stamps.py):")
seg(16, 17, "fwd pass A"); seg(17, 18, "fwd exchange A"); seg(18, 19, "fwd pass B"); seg(19, 20, "fwd exchange B"); seg(20, 21, "fwd pass C")
seg(21, 22, "to end of sub-FFT"); seg(22, 23, "publish + 4x (table load, partner read, CMAC)")
seg(24, 25, "inv pass A"); seg(25, 26, "inv exchange A"); seg(26, 27, "inv pass B"); seg(27, 28, "inv exchange B"); seg(28, 29, "inv pass C")
