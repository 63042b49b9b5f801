"""Host side of the EQ row through the C ABI (no GPU needed): aw_biquad_make and aw_eq_parse against the
reference's golden numbers and against the oracle's restatement on the same inputs."""
import random

import numpy as np
import pytest

import airwave_amd as aw
from test_oracle_eq_kats import GOLDEN_COEFFS, magnitude_db


def test_golden_coefficients_through_abi():
    for t, g, f, q, fs, exp in GOLDEN_COEFFS:                    # ParametricEqualizerProcessorTests.swift:46-59
        got = aw.BiquadCoefficientBuilder.make(t, g, f, q, fs)
        assert np.max(np.abs(np.array(got) - np.array(exp))) < 1e-12


def test_coefficients_bit_equal_oracle(oracle):
    rng = random.Random(7)
    for _ in range(300):
        t, g = rng.randrange(3), rng.uniform(-24, 24)
        fs = rng.choice([44100.0, 48000.0, 96000.0, 192000.0])
        f, q = rng.uniform(10, fs / 2 - 1), rng.uniform(0.1, 10)
        assert aw.BiquadCoefficientBuilder.make(t, g, f, q, fs) == oracle.biquad_make(t, g, f, q, fs)


def test_magnitudes_through_abi():
    c = aw.BiquadCoefficientBuilder.make(0, 6, 1000, 0.707, 48000)                  # :62-85
    for f, e in zip([0, 1000, 23999], [0, 6, 0]):
        assert abs(magnitude_db(c, f, 48000) - e) < 1e-9


@pytest.mark.parametrize("args,kind", [((0, 1, 1000, 1, 0), "invalidSampleRate"), ((0, 1, 1000, 1, float("nan")), "invalidSampleRate"),
                                       ((0, float("inf"), 1000, 1, 48000), "nonFiniteInput"), ((0, 1, 24000, 1, 48000), "invalidFrequency"),
                                       ((0, 1, 0, 1, 48000), "invalidFrequency"), ((0, 1, 1000, 0, 48000), "invalidQ"),
                                       ((1, 1, 1000, -2, 48000), "invalidQ")])
def test_coefficient_errors(args, kind, oracle):
    with pytest.raises(aw.BiquadCoefficientError) as e:                            # BiquadCoefficientBuilder.swift:36-47
        aw.BiquadCoefficientBuilder.make(*args)
    assert e.value.kind == kind
    with pytest.raises(oracle.BiquadCoefficientError) as o:
        oracle.biquad_make(*args)
    assert o.value.kind == kind


def test_reference_fixture(golden_dir):
    data = open(f"{golden_dir}/eq/CCA CRA ParametricEq.txt", "rb").read()          # EqualizerAPOParserTests.swift:7-26
    d = aw.EqualizerAPOParser.parse(data, "CCA CRA ParametricEq.txt")
    assert d.preampDB == -2.56 and len(d.filters) == 10 and all(f.isEnabled for f in d.filters)
    assert d.filters[0].type == aw.eq.LOW_SHELF and d.filters[-1].type == aw.eq.HIGH_SHELF
    assert [f.frequencyHz for f in d.filters] == [105.0, 65.3, 180.0, 625.7, 894.2, 1431.5, 3020.2, 6165.4, 9079.1, 10000.0]
    assert [f.gainDB for f in d.filters] == [-2.8, 1.0, -2.2, 0.6, 2.0, -1.5, 2.5, 2.3, 1.2, -5.2]
    assert [f.q for f in d.filters] == [0.70, 1.68, 1.08, 1.07, 1.24, 1.77, 2.25, 5.37, 2.75, 0.70]


def test_parser_reference_cases():
    src = "# comment\nPreamp: -2.5 dB\nFilter 7: ON PK Fc 1000 Hz Gain 3.25 dB Q 1.20\nFilter: off LSC Fc 80 Hz Gain -1 dB Q 0.7\nFilter 9: ON HSC Fc 10000 Hz Gain -2 dB Q 0.70\n"
    d = aw.EqualizerAPOParser.parse(src.encode(), "curve.txt")                      # :29-47
    assert d.preampDB == -2.5 and [f.sourceLine for f in d.filters] == [3, 4, 5]
    assert [f.sourceNumber for f in d.filters] == [7, None, 9] and [f.isEnabled for f in d.filters] == [True, False, True]
    assert [f.type for f in d.filters] == [0, 1, 2] and [f.frequencyHz for f in d.filters] == [1000, 80, 10000]
    d = aw.EqualizerAPOParser.parse("﻿  pReAmP : 1e0 dB\r\n\t# ignored\r\n fIlTeR 1 : oN pK Fc 440 Hz gAiN 2 dB q 1\r\n".encode(), "m.txt")   # :49-57
    assert d.preampDB == 1 and len(d.filters) == 1 and d.filters[0].gainDB == 2
    with pytest.raises(aw.EqualizerParseError) as e:                                # :59-63
        aw.EqualizerAPOParser.parse(b"Filter 1: OFF PK Fc 440 Hz Gain 2 dB Q 1", "t.txt")
    assert any("effective" in r for _, r in e.value.issues)
    d = aw.EqualizerAPOParser.parse(b"Filter 1: ON PK Fc 440 Hz Gain 2 dB Q 1", "t.txt")
    assert d.preampDB == 0 and d.filters[0].isEnabled
    with pytest.raises(aw.EqualizerParseError) as e:                                # :71-87
        aw.EqualizerAPOParser.parse(b"Preamp: 1 dB\nPreamp: 2 dB\nFilter 1: ON PK Fc 440 Hz Gain 2 dB\nInclude: other.txt", "bad.txt")
    assert e.value.filename == "bad.txt" and e.value.errorDescription.startswith("Could not read bad.txt: line 2: duplicate")
    iss = e.value.issues
    assert any(l == 2 and "duplicate" in r for l, r in iss) and any(l == 3 and "malformed" in r for l, r in iss)
    assert any(l == 4 and "unsupported" in r for l, r in iss)
    with pytest.raises(aw.EqualizerParseError) as e:                                # :89-103
        aw.EqualizerAPOParser.parse(b"Preamp: NaN dB\nFilter 1: ON PK Fc 0 Hz Gain inf dB Q -1", "t.txt")
    rs = [r for _, r in e.value.issues]
    assert any("finite" in r for r in rs) and any("frequency" in r for r in rs) and any("Q" in r for r in rs)
    many = "\n".join(f"Filter {i}: ON PK Fc {i} Hz Gain 1 dB Q 1" for i in range(1, 66))
    with pytest.raises(aw.EqualizerParseError) as e:
        aw.EqualizerAPOParser.parse(many.encode(), "t.txt")
    assert any("64" in r for _, r in e.value.issues)
    with pytest.raises(aw.EqualizerParseError) as e:                                # :105-110
        aw.EqualizerAPOParser.parse(b" " * (1_048_576 + 1), "large.txt")
    assert e.value.filename == "large.txt" and any("1 MiB" in r for _, r in e.value.issues)
    with pytest.raises(aw.EqualizerParseError) as e:
        aw.EqualizerAPOParser.parse(b"Preamp: 1 dB\xff", "t.txt")
    assert any("UTF-8" in r for _, r in e.value.issues)


def _same_parse(oracle, text: bytes):
    try:
        o = oracle.eq_parse(text, "f.txt")
        o = (o.preamp_db, [(f.source_line, f.source_number, f.is_enabled, f.type, f.frequency_hz, f.gain_db, f.q) for f in o.filters])
    except oracle.EqualizerParseError as e:
        o = ("error", e.issues)
    try:
        p = aw.EqualizerAPOParser.parse(text, "f.txt")
        p = (p.preampDB, [(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in p.filters])
    except aw.EqualizerParseError as e:
        p = ("error", e.issues)
    assert o == p, (text, o, p)


def test_parser_matches_oracle_on_mutated_presets(oracle):
    """Token-level fuzz: every line is assembled from valid and invalid pieces; the HIP library's parser and
    the oracle's regex restatement must agree on the definition or on the exact issue list."""
    rng = random.Random(11)
    heads = ["Filter", "filter 3", "Filter  12", "FILTER 4 ", "Filter1", "Filter 2x", "Filterx", "Filter 99999999999999999999"]
    colon = [":", " :", ": ", " : ", "", ":\t"]
    onoff = ["ON", "off", "On", "OFFF", "yes", ""]
    types = ["PK", "lsc", "HSC", "LP", "pk "]
    nums = ["1000", "-3.5", "1e3", "0x10", ".5", "5.", "+2", "nan", "inf", "-inf", "1e999", "abc", "1,5", "1_0", "0", "-0", "1e-400", "٣"]
    seps = [" ", "  ", "\t", " ", ""]
    def s():
        return rng.choice(seps[:3]) if rng.random() < 0.9 else rng.choice(seps)
    eols = ["\n", "\r\n", "\r", " ", "\n\n"]
    for _ in range(400):
        lines = []
        for _ in range(rng.randrange(1, 6)):
            r = rng.random()
            if r < 0.25:
                lines.append(rng.choice(["Preamp", "preamp", "PREAMP ", "Preampx"]) + rng.choice(colon) + rng.choice(nums) + s() + rng.choice(["dB", "db", "DB", "d B", ""]))
            elif r < 0.85:
                lines.append(rng.choice(heads) + rng.choice(colon) + s() + rng.choice(onoff) + s() + rng.choice(types) + s() + "Fc" + s() +
                             rng.choice(nums) + s() + rng.choice(["Hz", "hz", "kHz"]) + s() + "Gain" + s() + rng.choice(nums) + s() + "dB" + s() +
                             rng.choice(["Q", "q", "BW"]) + s() + rng.choice(nums) + rng.choice(["", " ", " x"]))
            elif r < 0.92:
                lines.append(rng.choice(["# note", "", "   ", "GraphicEQ: 1 2", "Include: a.txt", "Channel: L"]))
            else:
                lines.append("Filter 1: ON PK Fc 1000 Hz Gain 3 dB Q 1")
        text = "".join(l + rng.choice(eols) for l in lines)
        if rng.random() < 0.1:
            text = "﻿" + text
        _same_parse(oracle, text.encode("utf-8"))


def test_parser_follows_icu_whitespace_and_case_folding(oracle):
    """NSRegularExpression is ICU: \\s is [\\t\\n\\f\\r\\p{Z}] — U+001C-001F are NOT white space (Python's str.isspace() says they are) —, and
    .caseInsensitive compares simple case foldings: KELVIN SIGN matches k, LONG S matches s, dotted / dotless i match nothing
    (EqualizerAPOParser.swift:27-34).  captures[2].uppercased() then turns a long s into S (a valid shelf) and leaves the Kelvin
    sign alone ("unsupported filter type", :91-99).  trimmingCharacters(in: .whitespacesAndNewlines) (:60) trims Z* and TAB only."""
    ok = "Filter 1: ON PK Fc 100 Hz Gain 3 dB Q 1"
    for sp in ["\u00a0", "\u1680", "\u2003", "\u202f", "\u205f", "\u3000", "\t"]:
        d = aw.EqualizerAPOParser.parse(ok.replace(" ", sp).encode(), "f.txt")
        assert len(d.filters) == 1 and d.filters[0].frequencyHz == 100.0, repr(sp)
    for cc in ["\x1c", "\x1d", "\x1e", "\x1f", "\u200b", "\ufeff"]:
        with pytest.raises(aw.EqualizerParseError) as e:
            aw.EqualizerAPOParser.parse(ok.replace("Fc 100", "Fc" + cc + "100").encode(), "f.txt")
        assert e.value.issues == [(1, "malformed Filter directive")], repr(cc)
        if cc == "\ufeff":                                              # one leading BOM is dropped (:49-51)
            assert aw.EqualizerAPOParser.parse((cc + "Preamp: 3 dB").encode(), "f.txt").preampDB == 3.0
            continue
        with pytest.raises(aw.EqualizerParseError) as e:                 # not trimmed either: the line does not start with "Preamp"
            aw.EqualizerAPOParser.parse((cc + "Preamp: 3 dB").encode(), "f.txt")
        assert e.value.issues == [(1, "unsupported directive")], repr(cc)
    d = aw.EqualizerAPOParser.parse("Filter 1: ON L\u017fC Fc 100 Hz Gain 3 dB Q 1".encode(), "f.txt")
    assert d.filters[0].type == 1
    d = aw.EqualizerAPOParser.parse("filter 2: on h\u017fc fc 1e3 hz gain -2 db q .7\u3000".encode(), "f.txt")
    assert d.filters[0].type == 2 and d.filters[0].sourceNumber == 2
    with pytest.raises(aw.EqualizerParseError) as e:
        aw.EqualizerAPOParser.parse("Filter 1: ON P\u212a Fc 100 Hz Gain 3 dB Q 1".encode(), "f.txt")
    assert e.value.issues == [(1, "unsupported filter type")]
    for bad in ["F\u0131lter 1: ON PK Fc 100 Hz Gain 3 dB Q 1", "Filter 1: ON PK Fc 100 Hz Ga\u0131n 3 dB Q 1", "Filter 1: ON PK Fc 100 Hz Ga\u0130n 3 dB Q 1"]:
        with pytest.raises(aw.EqualizerParseError):
            aw.EqualizerAPOParser.parse(bad.encode(), "f.txt")
    for text in [ok, ok.replace(" ", "\u2003"), ok.replace("Fc 100", "Fc\x1d100"), "Filter 1: ON L\u017fC Fc 100 Hz Gain 3 dB Q 1",
                 "Filter 1: ON P\u212a Fc 100 Hz Gain 3 dB Q 1", "\x1fPreamp: 3 dB", "Preamp: 3 dB\x1f", "Preamp:\u00a03\u00a0dB\u2029Filter: ON PK Fc 1 Hz Gain 1 dB Q 1"]:
        _same_parse(oracle, text.encode())


def test_eq_needs_a_device_or_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aw.AirwaveError) as e:
        aw.ParametricEqualizerProcessor(48000.0)
    assert e.value.name == "NO_DEVICE"


def test_invalid_filter_is_reported_with_index_kind_and_source_line():
    """aw_last_eq_filter_error: the structured form of ParametricEqualizerPreparationError.invalidFilter(index:error:) plus the sourceLine
    EqualizerRuntimeEffect.map looks up (EqualizerRuntimeEffect.swift:80-100) — the index counts ENABLED filters only.  Through the host-only
    entry aw_eq_fold_hrir (the preparation checks are the same as aw_eq_state_create's, which needs a device)."""
    import ctypes
    import numpy as np
    from airwave_amd import _capi
    lib = _capi.load()
    i, k, l = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
    d = aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(11, 1, True, 0, 1000.0, 3.0, 1.0),
                                     aw.EqualizerFilter(12, 2, False, 0, -5.0, 3.0, 1.0),         # disabled: never validated, not counted
                                     aw.EqualizerFilter(13, 3, True, 1, 100.0, 3.0, 0.0),         # Q = 0: invalidQ, second ENABLED filter
                                     aw.EqualizerFilter(14, 4, True, 0, 99000.0, 3.0, 1.0)])
    h = np.ones((1, 8), np.float32)
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        aw.fold_equalizer(h, d, 48000.0)
    assert e.value.status == 16 and "Filter 2 is invalid: Q must be finite and positive." in str(e.value)
    assert lib.aw_last_eq_filter_error(ctypes.byref(i), ctypes.byref(k), ctypes.byref(l)) == 1
    assert (i.value, k.value, l.value) == (1, 3, 13)
    # any other failing call on the thread clears it
    with pytest.raises(aw.ParametricEqualizerPreparationError):
        aw.fold_equalizer(h, None, -1.0)
    assert lib.aw_last_eq_filter_error(ctypes.byref(i), ctypes.byref(k), ctypes.byref(l)) == 0
    assert lib.aw_last_eq_filter_error(None, None, None) == 0
