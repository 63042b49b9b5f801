"""Timing probe of the EQ cascade kernel (GPU box): frames/s and effective HBM rate on a cfg-4 shaped batch."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

S, F, fs = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), (int(sys.argv[2]) if len(sys.argv) > 2 else 960000), 96000.0
d = aw.EqualizerAPOParser.parse(open("tests/golden/eq/CCA CRA ParametricEq.txt", "rb").read(), "f.txt")
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
x = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
ctx.synth_fill(x.data_ptr(), S, F, 2, seed=1)
st = aw.ParametricEqualizerState(d, fs, n_streams=S, ctx=ctx)
for _ in range(2):
    st.process_device(x.data_ptr(), x.data_ptr(), F)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 5
for _ in range(N):
    st.process_device(x.data_ptr(), x.data_ptr(), F)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print(f"eq cascade: {S} streams x {F} frames, 10 filters: {ms:.3f} ms/launch, {S*F/ms/1e6:.2f} Gframes/s, {S*F*16/ms/1e6:.1f} GB/s algorithmic")
