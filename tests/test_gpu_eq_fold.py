"""The equalizer folded into the HRIR (aw_eq_fold_hrir + the convolution kernels) against the reference's two-effect order: float64
convolution (the mathematical definition ConvolutionEngine approximates), rounded to float32 as the spatial effect's output is, then
the oracle's sequential Float64 biquad cascade (ParametricEqualizerProcessor.swift:58-91).  Tolerance: 1e-5 of the peak."""
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


def _defs(aw, oracle, golden_dir, name):
    d = aw.EqualizerAPOParser.parse(open(os.path.join(golden_dir, "eq", name), "rb").read(), name)
    od = oracle.EqualizerDefinition(d.preampDB, [oracle.EqualizerFilter(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in d.filters])
    return d, od


def _reference(oracle, od, rate, x, tracks, lt, rt):
    s = oracle.spatialize_f64(x, tracks, lt, rt).astype(np.float32)
    el, er = oracle.eq_prepare(od, rate).process(np.ascontiguousarray(s[:, 0]), np.ascontiguousarray(s[:, 1]))
    return np.stack([el, er], axis=1)


@pytest.mark.parametrize("name", ["CCA CRA ParametricEq.txt", "Bass Reducer.txt", "Treble Booster.txt"])
@pytest.mark.parametrize("rate", [48000.0, 96000.0])
def test_folded_equalizer_matches_the_two_effect_order(aw, oracle, golden_dir, name, rate):
    """Three streams, two calls (the state that carries over is the spatializer's input history: the folded filter is one FIR), then a
    reset.  48 kHz: 4320 + ~6300 taps; 96 kHz: 8640 + ~12 000 taps — both beyond the fused tiles, on whatever kernels the policy picks."""
    d, od = _defs(aw, oracle, golden_dir, name)
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=rate)
    batch = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, aw.InputLayout(SPEAKERS7, "7 speakers"), [rate] * 3, equalizer=d, fold_equalizer=True)
    b = batch.buckets[rate]
    assert b.equalizer is None and b.eq_response_taps > 0 and b.eq_tail_bound <= 1e-7 and b.hrir_taps == tracks.shape[1] + b.eq_response_taps - 1
    x = oracle.synth_input(3, 60000, 7, seed=int(rate) % 97)
    y1 = b.spatializer.process(x[:, :41000])
    y2 = b.spatializer.process(x[:, 41000:])
    y = np.concatenate([y1, y2], axis=1)
    for s in range(3):
        assert oracle.peak_rel_error(y[s], _reference(oracle, od, rate, x[s], tracks, lt, rt)) < TOL, s
    batch.reset()
    assert np.array_equal(b.spatializer.process(x[:, :41000]), y1)


def test_automatic_choice_and_the_cascade_fallback(aw, oracle, golden_dir):
    """fold_equalizer=None: the 96 kHz bucket (8640 taps, already on the long-window kernels) takes the fold; the 48 kHz bucket keeps
    its 4320-tap HRIR on the on-chip tile and runs the cascade kernel after it (the fold would add 6300 taps); a narrow band at 20 Hz
    cannot be folded at all.  Every variant matches the same reference."""
    d, od = _defs(aw, oracle, golden_dir, "CCA CRA ParametricEq.txt")
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    layout = aw.InputLayout(SPEAKERS7, "7 speakers")
    rates = [48000.0, 96000.0, 48000.0, 96000.0]
    batch = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, layout, rates, equalizer=d)
    assert batch.buckets[96000.0].equalizer is None and batch.buckets[96000.0].eq_response_taps > 11000
    assert batch.buckets[48000.0].equalizer is not None and batch.buckets[48000.0].eq_response_taps == 0 and batch.buckets[48000.0].hrir_taps == 4320
    xs = [oracle.synth_input(1, 30000, 7, seed=20 + i)[0] for i in range(4)]
    ys = batch.process(xs, rates)
    for i, rate in enumerate(rates):
        tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=rate)
        assert oracle.peak_rel_error(ys[i], _reference(oracle, od, rate, xs[i], tracks, lt, rt)) < TOL, i
    ringing = aw.EqualizerDefinition(-1.0, [aw.EqualizerFilter(1, 1, True, 0, 20.0, 9.0, 30.0)])
    oringing = oracle.EqualizerDefinition(-1.0, [oracle.EqualizerFilter(1, 1, True, 0, 20.0, 9.0, 30.0)])
    with pytest.raises(aw.EqualizerNotFoldable):
        aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, layout, [96000.0], equalizer=ringing, fold_equalizer=True)
    b2 = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, layout, [96000.0], equalizer=ringing)
    assert b2.buckets[96000.0].equalizer is not None
    tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=96000.0)
    y = b2.process([xs[1]], [96000.0])[0]
    assert oracle.peak_rel_error(y, _reference(oracle, oringing, 96000.0, xs[1], tracks, lt, rt)) < TOL


def test_full_size_cfg4_with_the_equalizer_folded(aw, oracle, golden_dir):
    """cfg 4 per GPU as bench.py runs it by default: 512 streams x 10 s at 96 kHz, 7 speakers, StageSH1.0 resampled x2 with the 10-band
    fixture folded in (20 673 taps): one pass of the long-window kernels, no equalizer kernel.  Heads against the two-effect
    reference, the END of the 10 s against a reference run over the whole stream, linearity and distinct streams on the whole batch."""
    import torch
    S, fs, C = 512, 96000.0, 7
    F = int(10 * fs)
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    d, od = _defs(aw, oracle, golden_dir, "CCA CRA ParametricEq.txt")
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(w, SPEAKERS7, target_rate=fs)
    batch = aw.MixedRateBatch(np.asarray(w.audio_data), 48000.0, aw.InputLayout(SPEAKERS7, "7 speakers"), [fs] * S, ctx=ctx, equalizer=d)
    b = batch.buckets[fs]
    assert b.equalizer is None and b.hrir_taps == 8640 + b.eq_response_taps - 1 and 11000 < b.eq_response_taps < 13000
    b.spatializer.reserve(F)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    b.spatializer.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert b.spatializer.info()["long_window_rows"] > 0 and torch.isfinite(y).all()
    head = 20000
    for s in (0, 255, 511):
        assert oracle.peak_rel_error(y[s, :head].cpu().numpy(), _reference(oracle, od, fs, x[s, :head].cpu().numpy(), tracks, lt, rt)) < TOL, s
    # the last 4000 frames of one stream: the recursive reference needs the whole stream before them
    ref = _reference(oracle, od, fs, x[300].cpu().numpy(), tracks, lt, rt)
    assert oracle.peak_rel_error(y[300, F - 4000:].cpu().numpy(), ref[F - 4000:]) < TOL
    assert oracle.peak_rel_error(y[300, F // 2: F // 2 + 4000].cpu().numpy(), ref[F // 2: F // 2 + 4000]) < TOL
    b.spatializer.reset()
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    b.spatializer.process_device(x.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())
    e = (y.double() ** 2).sum(dim=(1, 2)).cpu().numpy()
    assert len(np.unique(np.round(e / e.max(), 9))) == S                      # every stream is its own
