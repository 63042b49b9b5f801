import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
import airwave_amd as aw
S, F, C = 128, 480000, 8
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
w = aw.WAVLoader.load(os.path.join(ROOT, "tests/golden/hrtf/RoomSH1.0.wav"))
lay = aw.InputLayout.detect(C); lt, rt = aw.HRIRChannelMap.hesuvi14Channel(lay).resolve(lay, 14)
sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
x = torch.empty((S, F, C), device="cuda"); y = torch.empty((S, F, 2), device="cuda")
ctx.synth_fill(x.data_ptr(), S, F, C); torch.cuda.synchronize()
ts = []
for i in range(60):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); sp.process_device(x.data_ptr(), y.data_ptr(), F); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(" ".join(f"{t:.2f}" for t in ts))
