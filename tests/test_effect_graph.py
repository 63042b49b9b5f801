"""AudioEffectGraph ordering and passthrough rules (AirwaveTests/AudioEffectGraphTests.swift re-expressed).
The spy cases run on the CPU (the graph is host glue); the production-effect cases need the GPU."""
import os

import numpy as np
import pytest

import airwave_amd as aw


class SpatialSpy:
    def __init__(self, isReady, offset=0.0):
        self.isReady, self.offset, self.processCount = isReady, np.float32(offset), 0

    def process(self, l, r=None):
        self.processCount += 1
        l = np.asarray(l, np.float32)
        r = l if r is None else np.asarray(r, np.float32)
        return l + self.offset, r + self.offset


class EqualizerSpy:
    def __init__(self, multiplier=1.0):
        self.multiplier, self.processCount, self.preparedSampleRates, self.error = np.float32(multiplier), 0, [], None
        self.setTargetError = None

    def prepare(self, definition, sampleRate):
        self.preparedSampleRates.append(sampleRate)
        if self.error is not None:
            raise self.error

    def setTarget(self, definition):
        if self.setTargetError is not None:
            raise self.setTargetError

    def process(self, l, r=None):
        self.processCount += 1
        l = np.asarray(l, np.float32)
        r = l if r is None else np.asarray(r, np.float32)
        return l * self.multiplier, r * self.multiplier


def test_neither_effect_copies_stereo_and_duplicates_mono():
    g = aw.AudioEffectGraph(SpatialSpy(False), EqualizerSpy(), 8)                     # :5-23
    assert g.prepare(48000.0, None).noEffectCanRun
    l, r = g.process([1, 2], [3, 4])
    assert l.tolist() == [1, 2] and r.tolist() == [3, 4]
    l, r = g.process([5, 6], None)
    assert l.tolist() == [5, 6] and r.tolist() == [5, 6]


def test_spatial_only_uses_spatial_effect():
    sp, eq = SpatialSpy(True, 10), EqualizerSpy()                                     # :25-37
    g = aw.AudioEffectGraph(sp, eq, 8)
    g.prepare(48000.0, None)
    l, r = g.process([1], [2])
    assert l.tolist() == [11] and r.tolist() == [12] and sp.processCount == 1 and eq.processCount == 0


def test_equalizer_only_runs_after_input_passthrough():
    eq = EqualizerSpy(2)                                                              # :39-54
    g = aw.AudioEffectGraph(SpatialSpy(False), eq, 8)
    res = g.prepare(44100.0, aw.EqualizerDefinition(3.0))
    assert res.runnableEffects == {"equalizer"} and eq.preparedSampleRates == [44100.0]
    l, r = g.process([1], None)
    assert l.tolist() == [2] and r.tolist() == [2] and eq.processCount == 1


def test_both_effects_run_in_spatial_then_equalizer_order():
    sp, eq = SpatialSpy(True, 10), EqualizerSpy(2)                                    # :56-73
    g = aw.AudioEffectGraph(sp, eq, 8)
    assert g.prepare(96000.0, aw.EqualizerDefinition(3.0)).runnableEffects == {"spatial", "equalizer"}
    l, r = g.process([1], [2])
    assert l.tolist() == [22] and r.tolist() == [24] and sp.processCount == 1 and eq.processCount == 1


def test_preparation_returns_nonfatal_line_specific_warning():
    eq = EqualizerSpy()                                                               # :75-91
    eq.error = aw.EqualizerAudioEffectError("invalidFilter", "frequency is above Nyquist", 17)
    g = aw.AudioEffectGraph(SpatialSpy(True), eq, 8)
    res = g.prepare(44100.0, aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(17, None, True, 0, 1000.0, 1.0, 1.0)]))
    assert res.runnableEffects == {"spatial"} and not res.noEffectCanRun
    assert res.equalizerWarning.filterLine == 17 and "Nyquist" in res.equalizerWarning.reason


def test_update_keeps_equalizer_in_path_and_callback_guard():
    eq = EqualizerSpy(2)
    g = aw.AudioEffectGraph(SpatialSpy(False), eq, 8)
    g.prepare(48000.0, None)
    assert g.process([1], None)[0].tolist() == [1]
    assert g.updateEqualizer(None).noEffectCanRun                                      # :143-156: active stays true for the unity ramp
    assert g.process([1], None)[0].tolist() == [2]
    with pytest.raises(ValueError):
        g.process(np.zeros(9, np.float32))
    with pytest.raises(ValueError):
        aw.AudioEffectGraph(SpatialSpy(False), eq, 4097)


def test_invalid_live_target_keeps_equalizer_in_callback_for_unity_crossfade():
    eq = EqualizerSpy(2)                                                              # :165-188
    g = aw.AudioEffectGraph(SpatialSpy(True), eq, 8)
    g.prepare(48000.0, aw.EqualizerDefinition(3.0))
    g.process([1], [1])
    assert eq.processCount == 1
    eq.setTargetError = aw.EqualizerAudioEffectError("invalidFilter", "frequency is above Nyquist", 31)
    res = g.updateEqualizer(aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(31, None, True, 0, 30000.0, 1.0, 1.0)]))
    assert not res.noEffectCanRun and res.runnableEffects == {"spatial"}
    assert res.equalizerWarning.filterLine == 31
    g.process([1], [1])
    assert eq.processCount == 2


@pytest.mark.gpu
def test_production_equalizer_can_reenable_after_none_selection():
    g = aw.AudioEffectGraph(SpatialSpy(False), aw.EqualizerRuntimeEffect(), 4096)     # :113-163
    a, b = aw.EqualizerDefinition(6.0), aw.EqualizerDefinition(-6.0)
    pos, neg = np.float32(10 ** (6 / 20)), np.float32(10 ** (-6 / 20))
    ones = np.ones(960, np.float32)
    last = lambda: g.process(ones, ones)[0][-1]
    assert g.prepare(48000.0, a).runnableEffects == {"equalizer"}
    assert abs(last() - pos) < 1e-5
    assert g.updateEqualizer(None).noEffectCanRun
    assert abs(last() - 1) < 1e-5
    assert g.updateEqualizer(a).runnableEffects == {"equalizer"}
    assert abs(last() - pos) < 1e-5
    g.updateEqualizer(None)
    last()
    assert g.prepare(48000.0, a).runnableEffects == {"equalizer"}
    assert abs(last() - pos) < 1e-5
    assert g.updateEqualizer(b).runnableEffects == {"equalizer"}
    assert abs(last() - neg) < 1e-5


@pytest.mark.gpu
def test_production_equalizer_uses_output_rate_and_rejects_nyquist():
    g = aw.AudioEffectGraph(SpatialSpy(False), aw.EqualizerRuntimeEffect(), 8)        # :93-115
    flt = aw.EqualizerFilter(31, None, True, 0, 23000.0, 1.0, 1.0)
    bad = g.prepare(44100.0, aw.EqualizerDefinition(0.0, [flt]))
    assert bad.noEffectCanRun and bad.equalizerWarning.filterLine == 31
    ok = g.prepare(96000.0, aw.EqualizerDefinition(3.0, [flt]))
    assert ok.runnableEffects == {"equalizer"} and ok.equalizerWarning is None


@pytest.mark.gpu
def test_production_graph_spatial_then_eq_matches_oracle(oracle, golden_dir):
    """Real effects end to end: HRIRManager (stereo layout, NeutralSH1.0) then a 6 dB preamp; 512-frame callbacks."""
    mgr = aw.HRIRManager()
    mgr.activatePreset(os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav"), 48000.0, aw.InputLayout.detect(2))
    g = aw.AudioEffectGraph(mgr, aw.EqualizerRuntimeEffect(), 512)
    assert g.prepare(48000.0, aw.EqualizerDefinition(6.0)).runnableEffects == {"spatial", "equalizer"}
    x = oracle.synth_input(1, 4096, 2, seed=5)[0]
    outs = [g.process(x[i:i + 512, 0], x[i:i + 512, 1]) for i in range(0, 4096, 512)]
    yl, yr = np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(w, ["FL", "FR"])
    sp = oracle.spatialize_f64(x, tracks, lt, rt)
    p = oracle.ParametricEqualizerProcessor(48000.0)
    p.set_target(oracle.EqualizerDefinition(6.0))
    el, er = [], []
    for i in range(0, 4096, 512):
        a, b = p.process(sp[i:i + 512, 0].astype(np.float32), sp[i:i + 512, 1].astype(np.float32))
        el.append(a); er.append(b)
    assert oracle.peak_rel_error(yl, np.concatenate(el)) < 1e-5 and oracle.peak_rel_error(yr, np.concatenate(er)) < 1e-5
