"""cfg 5 host row: streams bucketed by sample rate, one renderer network per rate with the HRIR resampled to
that rate (HRIRManager.swift:389-403).  CPU part: bucketing + resampled-HRIR assembly against the oracle's
restatement; GPU part: every bucket's output against the float64 truth on the resampled HRIR."""
import os

import numpy as np
import pytest

import airwave_amd as aw


def test_bucketing_is_stable_and_complete():
    rates = [48000, 44100, 48000, 96000, 44100, 48000]
    b = aw.bucket_by_rate(rates)
    assert list(b) == [48000.0, 44100.0, 96000.0]
    assert b[48000.0] == [0, 2, 5] and b[44100.0] == [1, 4] and b[96000.0] == [3]
    assert sorted(i for ids in b.values() for i in ids) == list(range(len(rates)))


def test_resampled_tracks_match_oracle(oracle, golden_dir):
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tr = np.asarray(w.audio_data)
    for rate in (44100.0, 48000.0, 96000.0):
        got = aw.resample_tracks(tr, 48000.0, rate)
        exp = np.stack([oracle.resample_intended(t, 48000.0, rate) for t in tr]) if rate != 48000.0 else tr
        assert got.shape == exp.shape and got.shape[1] == {44100.0: 3968, 48000.0: 4320, 96000.0: 8640}[rate]
        assert np.max(np.abs(got - exp)) <= 1e-7


@pytest.mark.gpu
def test_mixed_rate_batch_matches_truth_per_bucket(oracle, golden_dir):
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tr = np.asarray(w.audio_data)
    layout = aw.InputLayout(["FL", "FR", "FC", "BL", "BR", "SL", "SR"], "7 speakers")
    rates = [44100.0, 48000.0, 96000.0, 48000.0, 96000.0, 44100.0]
    batch = aw.MixedRateBatch(tr, 48000.0, layout, rates)
    assert {r: (b.hrir_taps, b.spatializer.info()["fft"]) for r, b in batch.buckets.items()} == {
        44100.0: (3968, 8192), 48000.0: (4320, 8192), 96000.0: (8640, 16384)}
    frames = {44100.0: 9000, 48000.0: 10000, 96000.0: 12000}
    xs = [oracle.synth_input(1, frames[r], 7, seed=100 + i)[0] for i, r in enumerate(rates)]
    ys = batch.process(xs, rates)
    spk = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]
    for i, r in enumerate(rates):
        tracks, lt, rt = oracle.assemble_tracks(w, spk, target_rate=r)      # the oracle resamples the HRIR itself
        truth = oracle.spatialize_f64(xs[i], tracks, lt, rt)
        assert oracle.peak_rel_error(ys[i], truth) < 1e-5, (i, r)
    batch.reset()
    again = batch.process(xs, rates)
    assert all(np.array_equal(a, b) for a, b in zip(again, ys))


@pytest.mark.parametrize("to_rate", [44100.0, 96000.0, 88200.0, 32000.0])
def test_literal_vgenp_resampler_matches_oracle_and_differs_from_intended(oracle, golden_dir, to_rate):
    """SURVEY 8f-2's optional variant: the call the reference literally makes (vDSP_vramp + vDSP_vgenp as documented).
    Product == oracle restatement; and it is NOT the intended interpolation: it evaluates at n / stride instead of n * stride."""
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    x = np.asarray(w.audio_data)[3]
    lit = aw.Resampler.resampleHighQuality(x, 48000.0, to_rate, literal_vgenp=True)
    exp = oracle.resample_vgenp_literal(x, 48000.0, to_rate)
    intended = aw.Resampler.resampleHighQuality(x, 48000.0, to_rate)
    assert lit.shape == exp.shape == intended.shape
    assert np.max(np.abs(lit - exp)) <= 1e-7 * np.max(np.abs(x))
    assert np.max(np.abs(lit - intended)) > 0.05 * np.max(np.abs(x))           # the two are different filters
    stride = 48000.0 / to_rate
    if to_rate == 96000.0:
        # up by 2: every second INPUT sample, then the last sample held (a 2x time-COMPRESSED response with a flat tail)
        assert np.max(np.abs(lit[:x.size // 2] - x[0:2 * (x.size // 2):2])) <= 1e-7 * np.max(np.abs(x))     # a + (b - a) * 1 rounds
        assert np.all(lit[(x.size + 1) // 2:] == x[-1])
    else:
        # in general: lerp(input, n / stride) (intended: n * stride)
        pos = np.arange(lit.size) / stride
        ok = pos <= x.size - 1
        ref = np.interp(pos[ok], np.arange(x.size), x.astype(np.float64))
        assert np.max(np.abs(lit[ok] - ref)) <= 2e-4 * np.max(np.abs(x))


@pytest.mark.gpu
def test_preset_activation_with_the_literal_resampler(oracle, golden_dir):
    """aw_context_set_resampler(ctx, 1): activatePreset at 96 kHz builds the renderer network on the literally resampled HRIR."""
    ctx = aw.Context(0)
    ctx.set_resampler(literal_vgenp=True)
    layout = aw.InputLayout(["FL", "FR", "FC", "BL", "BR", "SL", "SR"], "7 speakers")
    sp = aw.HRIRManager(ctx=ctx).activatePreset(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"), 96000.0, layout)
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"))
    tracks = np.stack([oracle.resample_vgenp_literal(t, 48000.0, 96000.0) for t in np.asarray(w.audio_data)])
    _, lt, rt = oracle.assemble_tracks(w, ["FL", "FR", "FC", "BL", "BR", "SL", "SR"])
    x = oracle.synth_input(1, 12000, 7, seed=3)
    assert oracle.peak_rel_error(sp.process(x)[0], oracle.spatialize_f64(x[0], tracks, lt, rt)) < 1e-5
