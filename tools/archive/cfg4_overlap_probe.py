"""cfg 4 (96 kHz, 8640-tap HRIR, 10-band EQ): can the EQ kernel (FP64-vector bound) execute beside the split / rows / merge kernels of the
NEXT chunk of streams (fabric / LDS bound)?  Round-4 review, next #2: "CU-mask split or plain concurrency, both measured".
Plain concurrency of whole lanes is `bench.py --workload cfg4 --lanes N`.  This probe separates the ROLES: one HIP stream runs the
convolution of chunk k while another runs the EQ of chunk k-1, (a) both on all CUs, (b) on complementary CU masks
(hipExtStreamCreateWithCUMask) with E CUs for the EQ and 256 - E for the convolution, persistent grids sized to their share.
Reference order: spatial first, then EQ (AudioEffectGraph.swift:195-211); the streams are independent, so chunks are.
Run on the GPU box: python tools/archive/cfg4_overlap_probe.py"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw

hip = ctypes.CDLL("libamdhip64.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, F, C, RATE = 512, 960000, 7, 96000.0
tracks = np.asarray(aw.WAVLoader.load(os.path.join(ROOT, "tests", "golden", "hrtf", "StageSH1.0.wav")).audio_data)
tr96 = aw.resample_tracks(tracks, 48000.0, RATE)
lt = np.array([0, 8, 6, 4, 12, 2, 10], np.int32); rt = np.array([1, 7, 13, 5, 11, 3, 9], np.int32)
eq_def = aw.EqualizerAPOParser.parse(open(os.path.join(ROOT, "tests", "golden", "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "CCA CRA ParametricEq.txt")


def stream(mask_cus=None):
    """A HIP stream, optionally restricted to the CUs listed (0..255: bit i of the mask = CU i in the driver's numbering)."""
    st = ctypes.c_void_p()
    if mask_cus is None:
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        return st
    words = [0] * 8
    for cu in mask_cus:
        words[cu // 32] |= 1 << (cu % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr) == 0
    return st


def context(st, wgs):
    os.environ["AW_PERSISTENT_WGS"] = str(wgs)
    try:
        return aw.Context(0, stream=st.value)
    finally:
        os.environ.pop("AW_PERSISTENT_WGS", None)


def run(label, conv_stream, eq_stream, conv_wgs, n_chunks, steps=4):
    cc = context(conv_stream, conv_wgs)
    ce = cc if eq_stream is None else context(eq_stream, 256)
    n = S // n_chunks
    sps, eqs, xs, ys = [], [], [], []
    for k in range(n_chunks):
        sp = aw.Spatializer(aw.HRIR(tr96, RATE, ctx=cc), lt, rt, n_streams=n, ctx=cc)
        sp.reserve(F)
        x = torch.empty((n, F, C), dtype=torch.float32, device="cuda"); y = torch.empty((n, F, 2), dtype=torch.float32, device="cuda")
        cc.synth_fill(x.data_ptr(), n, F, C, first_stream=k * n)
        sps.append(sp); xs.append(x); ys.append(y)
        eqs.append(aw.ParametricEqualizerState(eq_def, RATE, n_streams=n, ctx=ce))
    sc = torch.cuda.ExternalStream(conv_stream.value)
    se = sc if eq_stream is None else torch.cuda.ExternalStream(eq_stream.value)
    done_eq = [None] * n_chunks

    def step():
        for k in range(n_chunks):
            if done_eq[k] is not None and se is not sc:
                sc.wait_event(done_eq[k])                      # y_k is free again once its EQ has run
            sps[k].process_device(xs[k].data_ptr(), ys[k].data_ptr(), F)
            if se is not sc:
                e = torch.cuda.Event(); e.record(sc); se.wait_event(e)
            eqs[k].process_device(ys[k].data_ptr(), ys[k].data_ptr(), F)
            if se is not sc:
                done_eq[k] = torch.cuda.Event(); done_eq[k].record(se)
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    ok = bool(torch.isfinite(ys[-1][:, -1000:]).all().item())
    print(f"{label:90s} {dt * 1e3:7.3f} ms per step  {S * F / dt / 1e9:6.2f} G frames/s  finite {ok}", flush=True)
    del sps, eqs, xs, ys
    torch.cuda.empty_cache()


run("one stream, one chunk (the bench's schedule)", stream(), None, 256, 1)
run("one stream, two chunks", stream(), None, 256, 2)
run("two streams on all CUs: convolution of chunk k || EQ of chunk k-1, two chunks", stream(), stream(), 256, 2)
run("two streams on all CUs, four chunks", stream(), stream(), 256, 4)
for e_cus in (32, 64, 96):
    # CU numbering: give the EQ the same share of every XCD (CU i belongs to XCD i % 8 in the mask's numbering is NOT assumed: the share is
    # spread evenly over the 256 bits)
    eq_set = [i for i in range(256) if (i * e_cus) // 256 != ((i + 1) * e_cus) // 256]
    conv_set = [i for i in range(256) if i not in set(eq_set)]
    run(f"CU masks: EQ on {len(eq_set)} CUs, convolution on {len(conv_set)} (persistent grids sized to the share), two chunks",
        stream(conv_set), stream(eq_set), len(conv_set), 2)
