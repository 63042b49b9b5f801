"""Pins the EQ oracle against the reference's golden numbers (AirwaveTests/ParametricEqualizerProcessorTests.swift,
AirwaveTests/EqualizerAPOParserTests.swift).  The HIP EQ runs the same cases in tests/test_gpu_eq.py."""
import math
import os

import numpy as np
import pytest

PK, LSC, HSC = 0, 1, 2

GOLDEN_COEFFS = [  # ParametricEqualizerProcessorTests.swift:6-44
    (PK, 6, 1000, 0.707, 44100, [1.066059044304402, -1.848333006078428, 0.801193953602049, -1.848333006078428, 0.867252997906451]),
    (PK, 6, 1000, 0.707, 48000, [1.061051079218484, -1.861255902473044, 0.816265527066576, -1.861255902473044, 0.877316606285061]),
    (PK, 6, 1000, 0.707, 96000, [1.031556835547465, -1.932439513787206, 0.905029057291346, -1.932439513787206, 0.936585892838811]),
    (LSC, 4, 250, 0.8, 44100, [1.005181131876713, -1.959818685223499, 0.956203632826288, -1.960107660288434, 0.961095789638066]),
    (LSC, 4, 250, 0.8, 48000, [1.004757001839771, -1.963119655421762, 0.959686684133658, -1.963363967297150, 0.964199374098040]),
    (LSC, 4, 250, 0.8, 96000, [1.002369381638864, -1.981663998355715, 0.979628621963737, -1.981725629447349, 0.981936372510967]),
    (HSC, -5, 6000, 0.8, 44100, [0.659738038304301, -0.493423574823573, 0.211192786614601, -1.024348043481364, 0.401855293576692]),
    (HSC, -5, 6000, 0.8, 48000, [0.651371052565336, -0.549995923363222, 0.224963798271964, -1.105037860095793, 0.431376787569872]),
    (HSC, -5, 6000, 0.8, 96000, [0.605207918981539, -0.855707120775878, 0.345827037126246, -1.558782199620635, 0.654110034952544]),
]


def magnitude_db(c, f, fs):
    w = 2 * math.pi * f / fs
    z1, z2 = complex(math.cos(-w), math.sin(-w)), complex(math.cos(-2 * w), math.sin(-2 * w))
    return 20 * math.log10(abs((c[0] + c[1] * z1 + c[2] * z2) / (1 + c[3] * z1 + c[4] * z2)))


def mk(oracle, t, f, g, q, enabled=True):
    return oracle.EqualizerFilter(1, None, enabled, t, f, g, q)


def test_golden_coefficients(oracle):
    for t, g, f, q, fs, exp in GOLDEN_COEFFS:                    # :46-59, accuracy 1e-12
        got = oracle.biquad_make(t, g, f, q, fs)
        assert np.max(np.abs(np.array(got) - np.array(exp))) < 1e-12


def test_golden_magnitudes(oracle):
    for t, g, f, q, fs, exp in [(PK, 6, 1000, 0.707, 48000, [0, 6, 0]), (LSC, 4, 250, 0.8, 48000, [4, 2, 0]),
                                (HSC, -5, 6000, 0.8, 48000, [0, -2.5, -5])]:                 # :62-85
        c = oracle.biquad_make(t, g, f, q, fs)
        for tf, e in zip([0, f, fs / 2 - 1], exp):
            assert abs(magnitude_db(c, tf, fs) - e) < 1e-9


def test_unity_and_preamp(oracle):
    unity = oracle.eq_prepare(None, 48000)                       # :87-108
    pre = oracle.eq_prepare(oracle.EqualizerDefinition(6.0, []), 48000)
    l = np.array([0.25, -0.5, 1], np.float32)
    r = np.array([-0.75, 0.5, 0.125], np.float32)
    ul, ur = unity.process(l, r)
    assert np.array_equal(ul, l) and np.array_equal(ur, r)
    pl, pr = pre.process(l, r)
    g = np.float32(10.0 ** (6.0 / 20.0))
    assert abs(pl[0] - l[0] * g) < 1e-6 and abs(pr[2] - r[2] * g) < 1e-6


def test_known_impulse_response(oracle):
    st = oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [mk(oracle, PK, 1000, 6, 0.707), mk(oracle, PK, 3000, -3, 1.1)]), 48000)
    l, r = st.process(np.array([1, 0, 0, 0, 0, 0], np.float32), np.zeros(6, np.float32))       # :110-133
    exp = np.array([1.007962105198731, 0.026656172367575, 0.046848317472827, 0.062845911221200, 0.072328817552935,
                    0.074696369241889])
    assert np.max(np.abs(l - exp)) < 1e-6 and np.all(r == 0)


def test_disabled_filters_and_subnormal_flush(oracle):
    st = oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [mk(oracle, PK, 1000, 12, 0.7, enabled=False)]), 48000)   # :135-152
    l, r = st.process(np.array([1, 0], np.float32), np.array([1, 0], np.float32))
    assert l.tolist() == [1, 0] and r.tolist() == [1, 0]
    act = oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [mk(oracle, PK, 1000, 6, 0.707)]), 48000)
    tiny = np.float32(1.4e-45)
    sl, _ = act.process(np.array([tiny, 0], np.float32), np.zeros(2, np.float32))
    assert sl[0] != 0 and sl[1] == 0


def test_preparation_rejections(oracle):
    with pytest.raises(oracle.EqualizerPreparationError):        # :192-212
        oracle.eq_prepare(None, 0)
    for bad in [mk(oracle, PK, 24000, 1, 1), mk(oracle, PK, 1000, 1, 0)]:
        with pytest.raises(oracle.EqualizerPreparationError):
            oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [bad]), 48000)
    with pytest.raises(oracle.EqualizerPreparationError):
        oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [mk(oracle, PK, 500 + i, 1, 1) for i in range(65)]), 48000)


def run(p, n, lv=1.0, rv=1.0):
    return p.process(np.full(n, lv, np.float32), np.full(n, rv, np.float32))


def test_crossfade_ramp_across_callbacks(oracle):
    for fs in (44100.0, 48000.0, 96000.0):                      # :214-232
        p = oracle.ParametricEqualizerProcessor(fs)
        g = np.float32(10.0 ** (6.0 / 20.0))
        p.set_target(oracle.EqualizerDefinition(6.0))
        length = max(1, int(round(fs * 0.020)))
        first = max(1, length // 2)
        a = run(p, first)
        b = run(p, length - first)
        assert abs(a[0][0] - (1 + (g - 1) / np.float32(length))) < 1e-5
        assert abs(b[0][-1] - g) < 1e-5 and abs(b[1][-1] - g) < 1e-5


def test_transitions_to_and_from_unity(oracle):
    p = oracle.ParametricEqualizerProcessor(48000.0)             # :234-247
    p.set_target(oracle.EqualizerDefinition(6.0))
    run(p, 960)
    p.set_target(None)
    res = run(p, 960)
    g = np.float32(10.0 ** (6.0 / 20.0))
    assert abs(res[0][0] - (g - (g - 1) / np.float32(960))) < 1e-5 and abs(res[0][-1] - 1) < 1e-5


def test_rapid_publication_queues_newest(oracle):
    p = oracle.ParametricEqualizerProcessor(48000.0)             # :249-266
    pos, neg = np.float32(10 ** (6 / 20)), np.float32(10 ** (-6 / 20))
    p.set_target(oracle.EqualizerDefinition(6.0)); run(p, 480)
    p.set_target(oracle.EqualizerDefinition(-6.0))
    assert abs(run(p, 480)[0][-1] - pos) < 1e-5
    assert abs(run(p, 960)[0][-1] - neg) < 1e-5


def test_retirement_pressure(oracle):
    p = oracle.ParametricEqualizerProcessor(48000.0)             # :268-293
    g1, g2, g3 = (np.float32(10 ** (x / 20)) for x in (6, -6, 12))
    p.set_target(oracle.EqualizerDefinition(6.0)); run(p, 960)
    p.set_target(oracle.EqualizerDefinition(-6.0)); second = run(p, 960)
    assert abs(second[0][-1] - g2) < 1e-5
    p.set_target(oracle.EqualizerDefinition(12.0)); held = run(p, 960)
    assert abs(held[0][-1] - g2) < 1e-5
    p.drain_retired_states()
    assert abs(run(p, 960)[0][-1] - g3) < 1e-5
    assert abs(second[0][0] - (g1 + (g2 - g1) / np.float32(960))) < 1e-5


def test_reset_clears_histories(oracle):
    p = oracle.ParametricEqualizerProcessor(48000.0)             # :318-329
    p.set_target(oracle.EqualizerDefinition(0.0, [mk(oracle, PK, 1000, 6, 0.707)])); run(p, 960)
    p.reset(); p.set_target(None); run(p, 960)
    a = run(p, 1, 0, 0)
    assert a[0].tolist() == [0] and a[1].tolist() == [0]


def test_reference_fixture_curve(oracle, golden_dir):
    data = open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read()       # :359-394
    d = oracle.eq_parse(data, "CCA CRA ParametricEq.txt")
    assert d.preamp_db == -2.56 and len([f for f in d.filters if f.is_enabled]) == 10
    assert [f.frequency_hz for f in d.filters] == [105.0, 65.3, 180.0, 625.7, 894.2, 1431.5, 3020.2, 6165.4, 9079.1, 10000.0]
    assert [f.gain_db for f in d.filters] == [-2.8, 1.0, -2.2, 0.6, 2.0, -1.5, 2.5, 2.3, 1.2, -5.2]
    assert [f.q for f in d.filters] == [0.70, 1.68, 1.08, 1.07, 1.24, 1.77, 2.25, 5.37, 2.75, 0.70]
    assert d.filters[0].type == LSC and d.filters[-1].type == HSC
    fs, n, skip = 48000.0, 48000, 24000
    for f, exp in [(20, -5.3379478445), (1000, -0.9694887656), (10000, -4.2646888095)]:
        st = oracle.eq_prepare(d, fs)
        x = np.sin(2 * np.pi * f * np.arange(n) / fs).astype(np.float32)
        y, yr = st.process(x, x)
        db = 20 * np.log10(np.sqrt(np.mean(y[skip:].astype(np.float64) ** 2)) / np.sqrt(np.mean(x[skip:].astype(np.float64) ** 2)))
        assert abs(db - exp) < 0.03 and np.all(np.isfinite(y)) and np.all(np.isfinite(yr))


def test_parser_cases(oracle):
    src = "# comment\nPreamp: -2.5 dB\nFilter 7: ON PK Fc 1000 Hz Gain 3.25 dB Q 1.20\nFilter: off LSC Fc 80 Hz Gain -1 dB Q 0.7\nFilter 9: ON HSC Fc 10000 Hz Gain -2 dB Q 0.70\n"
    d = oracle.eq_parse(src.encode(), "curve.txt")               # EqualizerAPOParserTests.swift:29-47
    assert d.preamp_db == -2.5 and [f.source_line for f in d.filters] == [3, 4, 5]
    assert [f.source_number for f in d.filters] == [7, None, 9] and [f.is_enabled for f in d.filters] == [True, False, True]
    d = oracle.eq_parse("﻿  pReAmP : 1e0 dB\r\n\t# ignored\r\n fIlTeR 1 : oN pK Fc 440 Hz gAiN 2 dB q 1\r\n".encode(), "m.txt")   # :49-57
    assert d.preamp_db == 1 and len(d.filters) == 1 and d.filters[0].gain_db == 2
    with pytest.raises(oracle.EqualizerParseError) as e:
        oracle.eq_parse(b"Filter 1: OFF PK Fc 440 Hz Gain 2 dB Q 1", "x.txt")           # :59-63
    assert any("effective" in r for _, r in e.value.issues)
    with pytest.raises(oracle.EqualizerParseError) as e:
        oracle.eq_parse(b"Preamp: 1 dB\nPreamp: 2 dB\nGraphicEQ: 1 2\nFilter 1: ON XX Fc 1 Hz Gain 1 dB Q 1\nFilter 2: ON PK Fc -5 Hz Gain nan dB Q 0\n", "y.txt")
    reasons = [r for _, r in e.value.issues]
    assert "duplicate Preamp directive" in reasons and "unsupported directive" in reasons and "malformed Filter directive" in reasons
    assert "frequency must be positive" in reasons and "gain must be a finite number" in reasons and "Q must be positive" in reasons


def test_in_place_processing_preserves_canaries(oracle):
    import ctypes
    st = oracle.eq_prepare(oracle.EqualizerDefinition(0.0, [mk(oracle, HSC, 6000, -5, 0.8)]), 48000)     # :154-190
    size, canary = 4096, np.float32(12345)
    left = np.full(size + 2, canary, np.float32)
    right = np.full(size + 2, canary, np.float32)
    i = np.arange(size)
    left[1:-1] = (i % 17).astype(np.float32) / 17
    right[1:-1] = -(i % 13).astype(np.float32) / 13
    fp = ctypes.POINTER(ctypes.c_float)
    lp = ctypes.cast(left.ctypes.data + 4, fp)
    rp = ctypes.cast(right.ctypes.data + 4, fp)
    oracle._eq_lib().orc_eq_state_process(st._h, lp, rp, lp, rp, size)            # in place, one sample into the buffers
    assert left[0] == canary and left[-1] == canary and right[0] == canary and right[-1] == canary
    assert np.isfinite(left).all() and np.isfinite(right).all()


def test_ten_filter_workload_stays_finite_across_callback_sizes(oracle):
    fl = [mk(oracle, PK if i % 2 == 0 else HSC, 250 + i * 1000, (i % 3) - 1, 0.8) for i in range(10)]   # :317-357 (1 s per size)
    for size in (128, 512, 1024):
        st = oracle.eq_prepare(oracle.EqualizerDefinition(-3.0, fl), 48000)
        x = np.full(size, 0.25, np.float32)
        for _ in range(48000 // size):
            l, r = st.process(x, x)
        assert np.isfinite(l).all() and np.isfinite(r).all()
