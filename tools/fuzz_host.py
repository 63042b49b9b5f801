#!/usr/bin/env python3
"""Byte-level fuzz of everything the library parses or computes on the HOST (no GPU needed): the RIFF/WAVE reader, the HeSuVi text map
parser, the Equalizer APO parser, layouts / channel maps / resolve, the resampler and the biquad builder.

Meant for the sanitizer build (tools/asan_host.sh: the host sources under AddressSanitizer + UBSan): the pass criterion is "no
sanitizer report, no crash, no exception other than the documented error categories", plus agreement with the oracle where the
oracle defines the answer (decode results, parser outputs).  tests/ hold the structure-level fuzzers; this one breaks structure.

    python tools/fuzz_host.py [--seconds 60] [--seed 1]
"""
from __future__ import annotations

import argparse
import os
import random
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import airwave_amd as aw                                    # noqa: E402
from airwave_amd import eq as aweq                          # noqa: E402
from oracle import airwave_oracle as orc                    # noqa: E402  (checker only)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WAV_ERRS = {"emptyFile": "WAV_EMPTY_FILE", "unsupportedFormat": "WAV_UNSUPPORTED_FORMAT", "fileReadError": "WAV_FILE_READ",
            "invalidChannelCount": "INVALID_CHANNEL_COUNT"}


def mutate(b: bytes, rng: random.Random, region: int) -> bytes:
    b = bytearray(b)
    for _ in range(rng.randrange(1, 6)):
        kind = rng.randrange(6)
        pos = rng.randrange(0, min(len(b), region)) if b else 0
        if kind == 0 and b:
            b[pos] = rng.randrange(256)
        elif kind == 1 and len(b) >= pos + 4:
            b[pos:pos + 4] = struct.pack("<I", rng.choice([0, 1, 2, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, rng.randrange(1 << 32), len(b), len(b) - pos]))
        elif kind == 2 and b:
            del b[pos:pos + rng.randrange(1, 9)]
        elif kind == 3:
            b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        elif kind == 4 and b:
            del b[rng.randrange(0, len(b)):]
        elif kind == 5 and len(b) >= pos + 2:
            b[pos:pos + 2] = struct.pack("<H", rng.choice([0, 1, 3, 0xFFFE, 0xFFFF, 8, 16, 24, 32, 64, rng.randrange(1 << 16)]))
    return bytes(b)


def small_wavs(rng: random.Random):
    out = []
    for tag, bits in [(1, 8), (1, 16), (1, 24), (1, 32), (3, 32), (3, 64)]:
        for ch in (1, 2, 7, 14):
            frames = rng.randrange(1, 20)
            align = ch * bits // 8
            payload = bytes(rng.randrange(256) for _ in range(frames * align))
            fmt = struct.pack("<HHIIHH", tag, ch, 48000, 48000 * align, align, bits)
            if rng.random() < 0.5:
                fmt = struct.pack("<HHIIHH", 0xFFFE, ch, 48000, 48000 * align, align, bits) + struct.pack("<HHI", 22, bits, 0) + \
                    struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
            body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + \
                b"data" + struct.pack("<I", len(payload)) + payload
            out.append(b"RIFF" + struct.pack("<I", len(body)) + body)
    return out


def fuzz_wav(rng, path, seeds, stats):
    blob = mutate(rng.choice(seeds), rng, 96 if rng.random() < 0.8 else 1 << 30)
    with open(path, "wb") as f:
        f.write(blob)
    try:
        ow, oerr = orc.wav_load(path), None
    except ValueError as e:
        ow, oerr = None, str(e).split(":")[0]
    except Exception:                         # the oracle itself gave up on the bytes (struct.error, …): no expectation
        ow, oerr = None, "?"
    try:
        w, perr = aw.WAVLoader.load(path), None
    except aw.WAVError as e:
        w, perr = None, e.name
    except aw.AirwaveError as e:
        w, perr = None, e.name
    stats["wav"] += 1
    if oerr == "?":
        return
    if oerr is not None:
        assert perr is not None, ("library accepted what the oracle refuses", oerr, blob[:64].hex())
        if oerr in WAV_ERRS and perr != WAV_ERRS[oerr]:
            stats["wav_category_differs"] += 1          # both refuse; which category wins on doubly-broken files is not pinned
    elif perr is None:
        assert (w.sample_rate, w.channel_count, w.frame_count) == (ow.sample_rate, ow.channel_count, ow.frame_count), blob[:64].hex()
        assert np.array_equal(w.audio_data, ow.audio_data, equal_nan=True), blob[:64].hex()
        stats["wav_decoded"] += 1
    else:
        stats["wav_library_stricter"] += 1


def odd_scalars(raw: bytes, rng) -> bytes:
    """Scalars on which ICU, Foundation and a naive ASCII restatement part ways, put at random places."""
    try:
        t = list(raw.decode("utf-8"))
    except UnicodeDecodeError:
        return raw
    for _ in range(rng.randrange(1, 4)):
        pos = rng.randrange(0, len(t) + 1)
        if rng.random() < 0.5 and pos < len(t):
            t[pos] = rng.choice(ODD)
        else:
            t.insert(pos, rng.choice(ODD))
    return "".join(t).encode("utf-8")


def fuzz_text_map(rng, seeds, stats):
    raw = mutate(rng.choice(seeds).encode(), rng, 1 << 30)
    if rng.random() < 0.6:
        raw = odd_scalars(raw, rng)
    text = raw.decode("utf-8", "replace").replace("\x00", " ")
    try:
        m = aw.HRIRChannelMap.parseHeSuViFormat(text)
    except aw.AirwaveError:
        stats["map_refused"] += 1
        return
    om = orc.parse_hesuvi_format(text)
    assert len(m) == len(om), (text, om)
    clamp = lambda v: max(-2 ** 31, min(2 ** 31 - 1, v))
    for k, v in om.items():
        assert m.getIndices(k) == (clamp(v[0]), clamp(v[1])), (text, k, v)
    lay = aw.InputLayout.detect(rng.choice([1, 2, 6, 8, 14]))
    try:
        m.resolve(lay, rng.randrange(0, 20))
    except aw.AirwaveError:
        pass
    stats["map"] += 1


ODD = ["\x1c", "\x1d", "\x1e", "\x1f", "\x0b", "\x0c", "\x85", "\u00a0", "\u1680", "\u2003", "\u2028", "\u2029", "\u202f", "\u205f", "\u3000", "\u200b", "\ufeff",
       "\u212a", "\u017f", "\ufb02", "\u0130", "\u0131", "\u0301", "\u0660", "\uff11", "\x00", "\x7f"]


def fuzz_apo(rng, seeds, stats):
    raw = mutate(rng.choice(seeds), rng, 1 << 30)
    if rng.random() < 0.6:
        raw = odd_scalars(raw, rng)
    try:
        o = orc.eq_parse(raw, "f.txt")
        o = (o.preamp_db, [(f.source_line, f.source_number, f.is_enabled, f.type, f.frequency_hz, f.gain_db, f.q) for f in o.filters])
    except orc.EqualizerParseError as e:
        o = ("error", e.issues)
    try:
        d = aweq.EqualizerAPOParser.parse(raw, "f.txt")
        d = (d.preampDB, [(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in d.filters])
    except aweq.EqualizerParseError as e:
        d = ("error", e.issues)
    assert o == d, (raw, o, d)
    stats["apo"] += 1


def fuzz_numeric(rng, stats):
    n = rng.randrange(0, 300)
    x = np.asarray([rng.uniform(-1, 1) for _ in range(n)], dtype=np.float32)
    fr, to = rng.choice([8000.0, 44100.0, 48000.0, 96000.0, 0.0, -1.0, 1e-9, 1e12, float("nan"), float("inf")]), \
        rng.choice([44100.0, 48000.0, 96000.0, 192000.0, 0.0, float("nan"), 1.0])
    try:
        aw.Resampler.resampleHighQuality(x, fr, to, literal_vgenp=rng.random() < 0.5)
    except (aw.AirwaveError, ValueError):
        pass
    vals = [0.0, -1.0, 1.0, 20.0, 1e3, 2.4e4, 4.8e4, 1e-300, 1e300, float("nan"), float("inf"), -float("inf"), rng.uniform(-1e5, 1e5)]
    try:
        aweq.BiquadCoefficientBuilder.make(rng.randrange(-2, 12), rng.choice(vals), rng.choice(vals), rng.choice(vals), rng.choice(vals))
    except (aw.AirwaveError, ValueError, TypeError):
        pass
    names = ["FL", "fr", "Fc ", "LFE", "", "bl", "BR", "SL", "sr", "x" * 300, "TFL", "é", "FL"]
    try:
        lay = aw.InputLayout([rng.choice(names) for _ in range(rng.randrange(0, 20))], name=rng.choice(names))
        for fn in (aw.HRIRChannelMap.hesuvi14Channel, aw.HRIRChannelMap.hesuvi7Channel, aw.HRIRChannelMap.interleavedPairs, aw.HRIRChannelMap.splitBlocks):
            fn(lay).resolve(lay, rng.randrange(0, 40))
    except (aw.AirwaveError, ValueError):
        pass
    stats["numeric"] += 1


def fuzz_fold(rng, stats):
    """aw_eq_fold_hrir (round 6): random definitions, track shapes, tolerances and length limits.  Either the documented refusals /
    preparation errors, or folded tracks that equal the oracle's recurrence over the zero-extended tracks bit for bit."""
    stats["fold"] += 1
    vals = [20.0, 60.0, 105.0, 1000.0, 9000.0, 0.0, -3.0, 0.7, 1.41, 12.0, 30.0, 23999.0, 24000.0, 1e9, float("nan"), float("inf"), 1e-300]
    n_f = rng.choice([0, 1, 2, 3, 10, 64, 65])
    filters = [aweq.EqualizerFilter(i + 1, i + 1, rng.random() < 0.85, rng.randrange(0, 3), rng.choice(vals) if rng.random() < 0.03 else rng.uniform(20, 20000),
                                    rng.uniform(-12, 12), rng.choice(vals) if rng.random() < 0.03 else rng.uniform(0.3, 6)) for i in range(n_f)]
    d = None if rng.random() < 0.05 else aweq.EqualizerDefinition(rng.choice([0.0, -6.0, 3.5, float("nan"), 1e4, -1e4]) if rng.random() < 0.1 else rng.uniform(-12, 6), filters)
    rate = rng.choice([44100.0, 48000.0, 96000.0, 48000.0, 96000.0, 8000.0, 0.0, float("nan")])
    n, taps = rng.randrange(1, 4), rng.choice([1, 2, 7, 64, 300])
    h = np.asarray([[rng.uniform(-1, 1) for _ in range(taps)] for _ in range(n)], dtype=np.float32)
    tol = rng.choice([1e-7, 1e-5, 1e-3, 1e-10])
    max_taps = rng.choice([taps, taps + 1, 400, 4096, 65536])
    try:
        f = aw.fold_equalizer(h, d, rate, tailTolerance=tol, maxTaps=max_taps)
    except aw.EqualizerNotFoldable:
        stats["fold_refused"] += 1
        return
    except aweq.ParametricEqualizerPreparationError:
        stats["fold_invalid"] += 1
        return
    assert f.tracks.shape == (n, taps + f.responseTaps - 1) and f.tracks.shape[1] <= max_taps and 0.0 <= f.tailBound <= tol, (f.tracks.shape, f.tailBound)
    od = None if d is None else orc.EqualizerDefinition(d.preampDB, [orc.EqualizerFilter(x.sourceLine, x.sourceNumber, x.isEnabled, x.type, x.frequencyHz, x.gainDB, x.q) for x in d.filters])
    x = np.zeros(f.tracks.shape[1], np.float32)
    x[:taps] = h[0]
    want, _ = orc.eq_prepare(od, rate).process(x, x.copy())
    assert np.array_equal(f.tracks[0], want), "folded track differs from the oracle's recurrence"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    wav_seeds = small_wavs(rng)
    with open(os.path.join(GOLDEN, "hrtf", "NeutralSH1.0.wav"), "rb") as f:
        wav_seeds.append(f.read())
    map_seeds = ["FL = 0, 1\nFR = 1, 0\nFC = 6, 13\nLFE=6,13\nBL = 4,5\nBR= 5 ,4\nSL = 2,3\nSR = 3,2\n",
                 "# comment\nfl=0,7\n\nFR = 8 , 1 # trailing\nTFL = 10,11\nbogus line\nFC=x,1\n", "FL=4294967296,1\nFR=-1,2\n"]
    apo_seeds = [open(os.path.join(GOLDEN, "eq", n), "rb").read() for n in sorted(os.listdir(os.path.join(GOLDEN, "eq"))) if n.endswith(".txt")] or \
                [b"Preamp: -6.2 dB\nFilter 1: ON PK Fc 100 Hz Gain -3.5 dB Q 1.41\nFilter 2: ON LSC Fc 105 Hz Gain 5 dB Q 0.7\n"]
    stats = {k: 0 for k in ("wav", "wav_decoded", "wav_library_stricter", "wav_category_differs", "map", "map_refused", "apo", "numeric", "fold", "fold_refused", "fold_invalid")}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "f.wav")
        while time.time() - t0 < args.seconds:
            fuzz_wav(rng, path, wav_seeds, stats)
            fuzz_text_map(rng, map_seeds, stats)
            fuzz_apo(rng, apo_seeds, stats)
            fuzz_numeric(rng, stats)
            fuzz_fold(rng, stats)
    print("fuzz_host:", " ".join(f"{k}={v}" for k, v in stats.items()), f"library={os.environ.get('AIRWAVE_HIP_LIBRARY', 'default')}", "OK")


if __name__ == "__main__":
    main()
