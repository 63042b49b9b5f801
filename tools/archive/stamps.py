#!/usr/bin/env python3
"""Diagnostic: per-phase cycle breakdown of the fused tile kernel from in-kernel s_memtime stamps.
Run on the GPU box with AIRWAVE_HIP_LIBRARY=airwave_amd/libairwave_hip_stamps.so (build:
python airwave_amd/build.py --stamps).  Reads SHARES, not absolute speed (stamps perturb)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import airwave_amd as aw
from airwave_amd import _capi

S, F, C = 128, 96000, 8
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
w = aw.WAVLoader.load(os.path.join(ROOT, "tests/golden/hrtf/RoomSH1.0.wav"))
lay = aw.InputLayout.detect(C)
lt, rt = aw.HRIRChannelMap.hesuvi14Channel(lay).resolve(lay, 14)
sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
ctx.synth_fill(x.data_ptr(), S, F, C)
for _ in range(3):
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
torch.cuda.synchronize()
nwg = S * ((F + sp.info()["hop"] - 1) // sp.info()["hop"])   # buffer is sized for all tiles; interior ones are filled
buf = np.zeros((nwg, 32), dtype=np.uint64)
n = ctypes.c_int64()
st = _capi.load().aw_spatializer_debug_stamps(sp._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size, ctypes.byref(n))
assert st == 0, _capi.load().aw_last_error_message()
t = buf.astype(np.int64)
t = t[t[:, 13] > 0]      # interior workgroups only (boundary launch records nothing)
nwg = len(t)
names = ["load issue+tw copy", "pass1 A (waits raw)", "barrier A1", "subfft+cmac p0", "tab/raw issue + subfft+cmac p1",
         "barrier B0", "pass1 B", "barrier B1", "subfft+cmac p2", "subfft+cmac p3", "inverse subfft", "barrier inv", "final radix16+store"]
d = np.diff(t[:, :14], axis=1)
tot = (t[:, 13] - t[:, 0])
print(f"workgroups {nwg}; tile total cycles: median {np.median(tot):.0f} mean {tot.mean():.0f} p90 {np.percentile(tot,90):.0f}")
for i, nm in enumerate(names):
    print(f"  {i:2d} {nm:34s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  share {d[:, i].mean()/tot.mean()*100:5.1f}%")
# gap between consecutive workgroups on one CU cannot be seen here; compare sum with kernel time
print("sum of tile cycles / 256 CUs:", tot.sum() / 256)

inner = ["fwd: z read (before stamp16 not shown)", "fwd passA fft8x2+tw", "fwd exchA write/sync/read", "fwd passB", "fwd exchB", "fwd passC",
         "", "tab load + publish (to 22..)", "", "inv passA", "inv exchA", "inv passB", "inv exchB", "inv passC"]
def seg(a, b, name):
    d2 = t[:, b] - t[:, a]
    print(f"  [{a:2d}->{b:2d}] {name:40s} median {np.median(d2):8.0f}")
print("last forward pair (p3), detail (half-wave form: 16 -> fft16 -> 17 -> row twiddles -> 18 -> radix 2 -> 19 -> transpose -> 20 -> fft16 -> 21):")
seg(16, 17, "fwd radix 16 (waits for the row reads)"); seg(17, 18, "fwd row twiddles (15 LDS reads)"); seg(18, 19, "fwd radix 2 across lane bit 4")
seg(19, 20, "fwd 16 x 16 transpose through LDS"); seg(20, 21, "fwd radix 16"); seg(21, 22, "table loads issue (+wait?)"); seg(22, 23, "publish + partner read + CMAC")
