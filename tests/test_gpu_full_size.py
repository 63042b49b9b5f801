"""BASELINE cfg 3, cfg 4 and cfg 5 at their full per-GPU batch size (1024 / 512 / 1024 streams x 10 s) through device
buffers, checked by size-independent properties (modelled on test_gpu_parity.py::test_full_size_properties_cfg2):
sampled streams against the float64 truth on a head and a tail window, linearity on the whole batch, and per-stream
energies that are all different (no stream skipped, duplicated or written twice).  Tolerance: 1e-5 of the peak."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5
SPEAKERS7 = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]


@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


def _pin_long_window_plan(sp, F):
    """The long-window kernels ran the last call (tile_lw.hpp / tile_lw16.hpp), on windows that the call fills: returns the rows of
    the window groups.  (Which kernel set a full-size test exercises is asserted, not inferred from the policy.)"""
    i = sp.info()
    Ra, Rb, hist = i["long_window_rows"], i["long_window_rows_rest"], i["history"]
    assert Ra > 0, i
    hopa, hopb = Ra * 4096 - hist, (Rb * 4096 - hist if Rb else 0)
    n = -(-(F - hopb) // hopa)
    assert n >= 1 and n * hopa + hopb >= F, i
    assert (n * Ra + Rb) * 4096 <= 1.15 * F, i               # padded length of the windows against the frames of the call
    return Ra, Rb


def _energies_distinct(y, S):
    e = (y.double() ** 2).sum(dim=(1, 2))
    assert float(e.min()) > 0.25 * float(e.max())
    assert len(set(e.cpu().numpy().tolist())) == S          # float64 sums of 10^6 squares: equal only for duplicated streams


@pytest.mark.parametrize("kernels,scratch_mb", [("long-window", None), ("long-window", "8192"), ("partitioned", None), ("partitioned", "16384")])
def test_full_size_properties_cfg3(aw, oracle, monkeypatch, kernels, scratch_mb):
    """cfg 3: 1024 streams x 10 s x 7 speakers, 14 x 32768-tap HRIR, on both kernel sets a path-1 spatializer owns:
    the long-window kernels (one 524288-frame window per stream; what bench.py times, with the default scratch budget = ONE
    stream chunk) and the partitioned ones (AW_LW=0: 8 partitions of 4096), each also with a lowered scratch budget so that the
    batch runs as several stream chunks; sampled streams include every chunk edge."""
    import torch
    if scratch_mb:
        monkeypatch.setenv("AW_SPEC_SCRATCH_MB", scratch_mb)
    if kernels == "partitioned":
        monkeypatch.setenv("AW_LW", "0")
    S, F, C, L = 1024, 480000, 7, 32768
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    h = oracle.synth_hrir(14, L, seed=1234)
    layout = aw.InputLayout(SPEAKERS7, "7 speakers")
    lt, rt = aw.HRIRChannelMap.hesuvi14Channel(layout).resolve(layout, 14)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    info = sp.info()
    assert info["path"] == 1 and info["partitions"] == 8 and info["hop"] == 4096
    sp.reserve(F)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert sp.info()["long_window_rows"] == (128 if kernels == "long-window" else 0)
    assert torch.isfinite(y).all()
    if kernels == "long-window":     # one window of 128 x 4096 frames per stream: rows 3.5 pairs x N + s1/s2 N complex values
        per_stream = (7 * 524288 // 2 + 524288) * 8
    else:                            # 124 windows x 4 pairs + 117 blocks of 8192 bins
        per_stream = ((F + 4095) // 4096 + 7) * 4 * 8192 * 8 + ((F + 4095) // 4096) * 8192 * 8
    chunk = min(S, ((int(scratch_mb) << 20) // per_stream) if scratch_mb else S)
    assert (chunk < S) == bool(scratch_mb)
    edges = sorted({0, S - 1} | {e for k in range(chunk, S, chunk) for e in (k - 1, k)})
    head, tail = 6000, 2000
    for s in edges:
        xs = x[s, :head].cpu().numpy()
        assert np.array_equal(xs, oracle.synth_input(1, head, C, first_stream=s)[0])
        assert oracle.peak_rel_error(y[s, :head].cpu().numpy(), oracle.spatialize_f64(xs, h, lt, rt)) < TOL, s
        tail_in = x[s, F - tail - (L - 1):].cpu().numpy()
        ref_tail = oracle.spatialize_f64(tail_in, h, lt, rt)[L - 1:]
        assert oracle.peak_rel_error(y[s, F - tail:].cpu().numpy(), ref_tail) < TOL, s
    # linearity on the whole batch
    sp.reset()
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    sp.process_device(x.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())
    _energies_distinct(y, S)
    del x, y, y2
    torch.cuda.empty_cache()


def test_full_size_properties_cfg3_14_channel_input(aw, oracle, golden_dir):
    """The 14-channel-input reading of cfg 3 (bench.py's "secondary") at its full batch: InputLayout.detect(14) custom channels
    through the committed parseHeSuViFormat text map, 28 convolutions per stream; the long-window kernels over two groups of
    channels (8 + 6)."""
    import torch
    S, F, C, L = 1024, 480000, 14, 32768
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    h = oracle.synth_hrir(14, L, seed=1234)
    layout = aw.InputLayout.detect(C)
    cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(golden_dir, "hesuvi14_custom_map.txt")).read())
    lt, rt = cmap.resolve(layout, 14)
    assert (np.asarray(lt) >= 0).sum() + (np.asarray(rt) >= 0).sum() == 28
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    sp.reserve(F)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert sp.info()["long_window_rows"] == 128
    assert torch.isfinite(y).all()
    head, tail = 6000, 2000
    for s in (0, 511, S - 1):
        xs = x[s, :head].cpu().numpy()
        assert oracle.peak_rel_error(y[s, :head].cpu().numpy(), oracle.spatialize_f64(xs, h, lt, rt)) < TOL, s
        tail_in = x[s, F - tail - (L - 1):].cpu().numpy()
        assert oracle.peak_rel_error(y[s, F - tail:].cpu().numpy(), oracle.spatialize_f64(tail_in, h, lt, rt)[L - 1:]) < TOL, s
    sp.reset()
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    sp.process_device(x.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())
    _energies_distinct(y, S)
    del x, y, y2
    torch.cuda.empty_cache()


def test_full_size_properties_cfg4(aw, oracle, golden_dir):
    """cfg 4 per GPU: 512 streams x 10 s at 96 kHz, StageSH1.0 resampled x2 (8640 taps: a layout whose short calls run on 16384-frame
    fused windows and whose long calls — this one, reserved like bench.py does — on the long-window kernels), then the 10-band
    parametric EQ fixture in place — the spatial -> EQ order of the effect graph."""
    import torch
    S, fs, C = 512, 96000.0, 7
    F = int(10 * fs)
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=fs)            # the oracle resamples the HRIR itself
    layout = aw.InputLayout(SPEAKERS7, "7 speakers")
    batch = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, layout, [fs] * S, ctx=ctx)
    b = batch.buckets[fs]
    assert b.hrir_taps == 8640 and b.spatializer.info()["path"] == 0 and b.spatializer.info()["fft"] == 16384      # the fused window short calls take
    b.spatializer.reserve(F)
    d = aw.EqualizerAPOParser.parse(open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "f.txt")
    od = oracle.EqualizerDefinition(d.preampDB, [oracle.EqualizerFilter(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in d.filters])
    eq = aw.ParametricEqualizerState(d, fs, n_streams=S, ctx=ctx)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    b.spatializer.process_device(x.data_ptr(), y.data_ptr(), F)
    _pin_long_window_plan(b.spatializer, F)
    ysp_tail = {s: y[s, F - 3000:].cpu().numpy() for s in (0, 255, 511)}               # spatializer output before the in-place EQ
    eq.process_device(y.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    head = 20000
    for s in (0, 255, 511):
        xs = x[s, :head].cpu().numpy()
        sp64 = oracle.spatialize_f64(xs, tracks, lt, rt)
        el, er = oracle.eq_prepare(od, fs).process(sp64[:, 0].astype(np.float32), sp64[:, 1].astype(np.float32))
        got = y[s, :head].cpu().numpy()
        assert oracle.peak_rel_error(got[:, 0], el) < TOL and oracle.peak_rel_error(got[:, 1], er) < TOL, s
        # tail window of the convolution stage (the EQ is recursive from frame 0: its tail is covered by linearity below)
        tail_in = x[s, F - 3000 - 8639:].cpu().numpy()
        assert oracle.peak_rel_error(ysp_tail[s], oracle.spatialize_f64(tail_in, tracks, lt, rt)[8639:]) < TOL, s
    # linearity of the chain on the whole batch
    b.spatializer.reset()
    eq2 = aw.ParametricEqualizerState(d, fs, n_streams=S, ctx=ctx)
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    b.spatializer.process_device(x.data_ptr(), y2.data_ptr(), F)
    eq2.process_device(y2.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())
    _energies_distinct(y, S)


def test_full_size_properties_cfg5(aw, oracle, golden_dir):
    """cfg 5 per GPU: 1024 streams split evenly over 44.1 / 48 / 96 kHz, 10 s each, StageSH1.0 resampled per rate; every
    rate bucket has its own renderer network: fused 8192-frame windows at 44.1 / 48 kHz (3969 / 4320 taps), the long-window kernels
    for the 96 kHz bucket (8640 taps) — asserted per bucket."""
    import torch
    S, C = 1024, 7
    rates = [44100.0, 48000.0, 96000.0]
    stream_rates = [rates[i * 3 // S] for i in range(S)]
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    layout = aw.InputLayout(SPEAKERS7, "7 speakers")
    batch = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, layout, stream_rates, ctx=ctx)
    assert sorted(batch.buckets) == rates and sum(len(b.stream_ids) for b in batch.buckets.values()) == S
    for rate in rates:
        b = batch.buckets[rate]
        n, F = len(b.stream_ids), int(10 * rate)
        taps = b.hrir_taps
        tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=rate)
        assert tracks.shape[1] == taps
        x = torch.empty((n, F, C), dtype=torch.float32, device="cuda")
        ctx.synth_fill(x.data_ptr(), n, F, C, seed=oracle.SYNTH_SEED, first_stream=b.stream_ids[0])      # global stream ids
        y = torch.empty((n, F, 2), dtype=torch.float32, device="cuda")
        b.spatializer.reserve(F)
        b.spatializer.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        if rate == 96000.0:
            _pin_long_window_plan(b.spatializer, F)
        else:
            assert b.spatializer.info()["long_window_rows"] == 0 and b.spatializer.info()["fft"] == 8192, b.spatializer.info()
        for k in (0, n // 2, n - 1):
            xs = x[k, :12000].cpu().numpy()
            assert np.array_equal(xs, oracle.synth_input(1, 12000, C, first_stream=b.stream_ids[k])[0])
            assert oracle.peak_rel_error(y[k, :12000].cpu().numpy(), oracle.spatialize_f64(xs, tracks, lt, rt)) < TOL, (rate, k)
            tail_in = x[k, F - 3000 - (taps - 1):].cpu().numpy()
            ref_tail = oracle.spatialize_f64(tail_in, tracks, lt, rt)[taps - 1:]
            assert oracle.peak_rel_error(y[k, F - 3000:].cpu().numpy(), ref_tail) < TOL, (rate, k)
        b.spatializer.reset()
        x.mul_(-0.5)
        y2 = torch.empty_like(y)
        b.spatializer.process_device(x.data_ptr(), y2.data_ptr(), F)
        torch.cuda.synchronize()
        assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())
        _energies_distinct(y, n)
        del x, y, y2
        torch.cuda.empty_cache()
