"""PCIe-inclusive rate of the host-buffer entry (aw_spatializer_process_host) on the cfg 2 batch: pageable numpy buffers
in, host buffers out.  Never the bench value (inputs resident in HBM there); DESIGN.md section 4 quotes it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import airwave_amd as aw

S, F, C = 128, 480000, 8
w = aw.WAVLoader.load(os.path.join(ROOT, "tests/golden/hrtf/RoomSH1.0.wav"))
lay = aw.InputLayout.detect(C)
lt, rt = aw.HRIRChannelMap.hesuvi14Channel(lay).resolve(lay, 14)
sp = aw.Spatializer(aw.HRIR(w.audio_data), lt, rt, n_streams=S)
x = np.random.default_rng(0).uniform(-0.5, 0.5, (S, F, C)).astype(np.float32)
sp.process(x[:, :48000])
best = 1e9
for _ in range(3):
    sp.reset()
    t0 = time.perf_counter()
    y = sp.process(x)
    best = min(best, time.perf_counter() - t0)
print(f"host entry, {S} x {F} x {C}ch ({x.nbytes / 1e9:.2f} GB in, {y.nbytes / 1e9:.2f} GB out): {best * 1e3:.1f} ms -> {S * F / best / 1e9:.2f} G frames/s PCIe-inclusive")
