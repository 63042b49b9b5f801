"""Probe: do the long-window path's bandwidth-bound (split, merge) and latency-bound (rows) kernels overlap when two half-batches
run on disjoint halves of the CUs (hipExtStreamCreateWithCUMask), one chain a phase behind the other?  Prints G frames/s of
(a) one context, all CUs, 1024 streams and (b) two contexts on complementary CU masks, 512 streams each, launches interleaved.
Run on the GPU box: python tools/archive/cu_mask_probe.py"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

hip = ctypes.CDLL("libamdhip64.so")
S, F, C, L = 1024, 480000, 7, 32768
rng = np.random.default_rng(1234)
h = (rng.standard_normal((14, L)) * np.exp(-np.arange(L) / (L / 6.0))).astype(np.float32)
lt = np.array([0, 8, 6, 4, 12, 2, 10], np.int32); rt = np.array([1, 7, 13, 5, 11, 3, 9], np.int32)


def masked_stream(mask_words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(mask_words), arr)
    assert rc == 0, rc
    return st


def make(ctx, n):
    import airwave_amd as aw
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=n, ctx=ctx)
    sp.reserve(F)
    x = torch.empty((n, F, C), dtype=torch.float32, device="cuda"); y = torch.empty((n, F, 2), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), n, F, C)
    return sp, x, y


import airwave_amd as aw
steps = 6
# (a) baseline
ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
sp, x, y = make(ctx, S)
for _ in range(2): sp.process_device(x.data_ptr(), y.data_ptr(), F)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): sp.process_device(x.data_ptr(), y.data_ptr(), F)
torch.cuda.synchronize(); base = S * F * steps / (time.perf_counter() - t0) / 1e9
print(f"one context, all CUs: {base:.2f} G frames/s", flush=True)
del sp, x, y
torch.cuda.empty_cache()
# (b) two masked halves, chain B delayed by a filler kernel so that its bandwidth-bound phases meet chain A's latency-bound one
sa, sb = masked_stream([0x00FF00FF] * 8), masked_stream([0xFF00FF00] * 8)
os.environ["AW_PERSISTENT_WGS"] = "128"
ca, cb = aw.Context(0, stream=sa.value), aw.Context(0, stream=sb.value)
os.environ.pop("AW_PERSISTENT_WGS", None)
A, B = make(ca, S // 2), make(cb, S // 2)
scratch = torch.empty((S // 2, F, C), dtype=torch.float32, device="cuda")
for delay in (0, 100, 200, 300, 400, 500):
    for _ in range(2):
        A[0].process_device(A[1].data_ptr(), A[2].data_ptr(), F); B[0].process_device(B[1].data_ptr(), B[2].data_ptr(), F)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A[0].process_device(A[1].data_ptr(), A[2].data_ptr(), F)
    if delay: cb.synth_fill(scratch.data_ptr(), delay, F, C)          # filler on stream B only
    B[0].process_device(B[1].data_ptr(), B[2].data_ptr(), F)
    for _ in range(steps - 1):
        A[0].process_device(A[1].data_ptr(), A[2].data_ptr(), F); B[0].process_device(B[1].data_ptr(), B[2].data_ptr(), F)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"two contexts on complementary CU masks, chain B delayed by a {delay}-stream fill: {steps} steps in {dt * 1e3:.1f} ms = {S * F * steps / dt / 1e9:.2f} G frames/s (baseline {steps * S * F / base / 1e9 * 1e3:.1f} ms)", flush=True)
