"""aw_eq_fold_hrir (host only: no device work): the equalizer that follows the spatializer in the reference's graph
(AudioEffectGraph.swift:195-211), folded into the HRIR tracks — EQ(x * h) = x * (h * g).  Checked against the oracle's restatement of
ParametricEqualizerState.process (ParametricEqualizerProcessor.swift:58-91) run over the zero-extended tracks, which is what the fold
is defined to be, and against an explicit convolution with the oracle's impulse response."""
import os

import numpy as np
import pytest

import airwave_amd as aw


def _definition(golden_dir, name):
    return aw.EqualizerAPOParser.parse(open(os.path.join(golden_dir, "eq", name), "rb").read(), name)


def _oracle_state(oracle, golden_dir, name, rate):
    return oracle.eq_prepare(oracle.eq_parse(open(os.path.join(golden_dir, "eq", name), "rb").read(), name), rate)


@pytest.mark.parametrize("name", ["CCA CRA ParametricEq.txt", "Bass Booster.txt", "Treble Reducer.txt", "Vocal Booster.txt"])
@pytest.mark.parametrize("rate", [44100.0, 96000.0])
def test_folded_tracks_are_the_reference_recurrence_over_the_zero_extended_tracks(oracle, golden_dir, name, rate):
    h = oracle.synth_hrir(3, 700, seed=11)
    f = aw.fold_equalizer(h, _definition(golden_dir, name), rate)
    assert f.tracks.shape == (3, 700 + f.responseTaps - 1) and f.tracks.dtype == np.float32
    assert 0.0 <= f.tailBound <= 1e-7
    for t in range(3):
        x = np.zeros(f.tracks.shape[1], np.float32)
        x[:700] = h[t]
        want, _ = _oracle_state(oracle, golden_dir, name, rate).process(x, x.copy())
        assert np.array_equal(f.tracks[t], want)               # the same Float64 recurrence, the same rounding to float32
    # ... which is the convolution with the equalizer's impulse response, cut after responseTaps samples
    imp = np.zeros(1 << 17, np.float32)
    imp[0] = 1.0
    g, _ = _oracle_state(oracle, golden_dir, name, rate).process(imp, imp.copy())
    conv = np.convolve(h[1].astype(np.float64), g.astype(np.float64))[: f.tracks.shape[1]]
    assert np.abs(conv - f.tracks[1]).max() < 2e-6 * np.abs(conv).max()
    # the response length is the smallest one whose remaining tail is within the tolerance
    a = np.abs(g.astype(np.float64))
    tail = np.cumsum(a[::-1])[::-1]
    assert tail[f.responseTaps] <= 1.01e-7 * a.max() + 1e-12 and tail[max(f.responseTaps - 2, 0)] > 0.9e-7 * a.max()


def test_response_lengths_of_the_bundled_presets(golden_dir):
    """What the fold costs: taps added to the HRIR (tolerance 1e-7).  Low shelves at 100 Hz ring for thousands of frames, treble
    presets for dozens; twice the rate, twice the frames."""
    lengths = {}
    h = np.ones((1, 8), np.float32)
    for name in ("CCA CRA ParametricEq.txt", "Bass Booster.txt", "Treble Booster.txt", "Vocal Booster.txt"):
        lengths[name] = [aw.fold_equalizer(h, _definition(golden_dir, name), r).responseTaps for r in (48000.0, 96000.0)]
    assert 5000 < lengths["CCA CRA ParametricEq.txt"][0] < 7000 and 11000 < lengths["CCA CRA ParametricEq.txt"][1] < 13000
    assert lengths["Treble Booster.txt"][1] < 200 and lengths["Bass Booster.txt"][0] > 3000
    for a, b in lengths.values():
        assert 1.8 * a < b < 2.2 * a


def test_unity_and_disabled_filters_leave_the_tracks_alone():
    h = np.arange(12, dtype=np.float32).reshape(2, 6) - 5
    f = aw.fold_equalizer(h, None, 48000.0)
    assert f.responseTaps == 1 and np.array_equal(f.tracks, h) and f.tailBound == 0.0
    d = aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(1, 1, False, 0, 100.0, 6.0, 1.0)])           # present but disabled
    assert np.array_equal(aw.fold_equalizer(h, d, 48000.0).tracks, h)
    d = aw.EqualizerDefinition(-6.0, [])                                                            # preamp only: a gain
    f = aw.fold_equalizer(h, d, 48000.0)
    assert f.responseTaps == 1 and np.allclose(f.tracks, h * 10 ** (-6 / 20), rtol=1e-7)


def test_a_response_that_does_not_decay_in_time_is_refused():
    """A narrow band at 20 Hz rings for seconds: not foldable within 65 536 taps — the host runs the cascade after the spatializer.
    With room for it, it folds."""
    h = np.ones((1, 100), np.float32)
    d = aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(1, 1, True, 0, 20.0, 12.0, 30.0)])
    with pytest.raises(aw.EqualizerNotFoldable) as e:
        aw.fold_equalizer(h, d, 48000.0)
    assert e.value.status == 17 and "decay" in str(e.value)
    with pytest.raises(aw.EqualizerNotFoldable):
        aw.fold_equalizer(h, _eq(100.0, 3.0, 0.7), 48000.0, maxTaps=600)              # an ordinary shelf, but no room
    f = aw.fold_equalizer(h, d, 48000.0, maxTaps=1 << 21)
    assert f.responseTaps > 200000 and f.tailBound <= 1e-7


def _eq(freq, gain, q):
    return aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(1, 1, True, 1, freq, gain, q)])


def test_validation_errors_are_the_preparation_errors(golden_dir):
    """Same checks, same codes and texts as ParametricEqualizerProcessor.prepare (:168-212)."""
    h = np.ones((1, 10), np.float32)
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        aw.fold_equalizer(h, _eq(100.0, 3.0, 0.7), 0.0)
    assert e.value.status == 13
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        aw.fold_equalizer(h, _eq(30000.0, 3.0, 0.7), 48000.0)                        # above Nyquist
    assert e.value.status == 16 and "Filter 1 is invalid" in str(e.value)
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        aw.fold_equalizer(h, aw.EqualizerDefinition(float("inf"), []), 48000.0)
    assert e.value.status == 14
    many = aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(i + 1, i + 1, True, 0, 1000.0 + i, 1.0, 1.0) for i in range(65)])
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        aw.fold_equalizer(h, many, 48000.0)
    assert e.value.status == 15


def test_the_batch_folds_only_where_that_is_the_cheaper_form(monkeypatch):
    """batching._fold_pays: an HRIR already beyond the fused tiles takes the fold (cfg 4: 8640 taps at 96 kHz); a short HRIR that the
    fold would push off the on-chip tile keeps the cascade; one that stays on it takes the fold."""
    from airwave_amd.batching import _fold_pays
    assert _fold_pays(8640, 20673) and _fold_pays(4320, 4500) and not _fold_pays(4320, 10400) and _fold_pays(5122, 65536)
