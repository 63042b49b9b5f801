"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/airwave_hip.h declares,
fails loudly without a GPU, and its host-side data model agrees with the oracle restatement."""
import ctypes
import os
import re

import numpy as np
import pytest

import airwave_amd as aw
from airwave_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "airwave_hip.h")).read()
    return sorted(set(re.findall(r"AW_API[^;]*?\b(aw_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 60
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/airwave_hip.h but not exported"
    assert sorted(_capi.SIGNATURES) == names, "ctypes table and header disagree"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aw.AirwaveError) as ei:
        aw.Context(0)
    assert ei.value.name == "NO_DEVICE"


def test_product_package_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "airwave_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "airwave_oracle" not in src.replace("oracle/airwave_oracle.h: orc_synth_value", "").replace(
                    "oracle/airwave_oracle.h:orc_synth_value", ""), f


def test_layouts_and_maps_match_oracle(oracle):
    for n in [1, 2, 3, 6, 7, 8, 12, 14]:
        lay = aw.InputLayout.detect(n)
        assert lay.channels == oracle.layout_detect(n)
        for ctor, ofn in [(aw.HRIRChannelMap.hesuvi14Channel, oracle.map_hesuvi14), (aw.HRIRChannelMap.hesuvi7Channel, oracle.map_hesuvi7),
                          (aw.HRIRChannelMap.interleavedPairs, oracle.map_interleaved_pairs), (aw.HRIRChannelMap.splitBlocks, oracle.map_split_blocks)]:
            m, om = ctor(lay), ofn(lay.channels)
            assert len(m) == len(om)
            for s in lay.channels:
                assert m.getIndices(s) == om.get(s)
    assert aw.InputLayout.detect(8).name == "7.1 Surround" and aw.InputLayout.detect(5).name == "5 Channel"


def test_parse_text_matches_oracle(oracle, golden_dir):
    texts = ["# c\n; c\n\nL = 0, 1\n  r=8 ,7\nSUB = 6, 13\nRL = 4, 5\nbogus\nX = 1\nY = 1, 2, 3\nMine = 2, 3\nC = a, 1\nL = 5, 6\r\nTFL=1,+2\rsl = -1, 3",
             open(os.path.join(golden_dir, "hesuvi14_custom_map.txt")).read()]
    for t in texts:
        m, om = aw.HRIRChannelMap.parseHeSuViFormat(t), oracle.parse_hesuvi_format(t)
        assert len(m) == len(om)
        for k, v in om.items():
            assert m.getIndices(k) == v


def test_resolve_rules(oracle, golden_dir):
    lay = aw.InputLayout.detect(12)
    lt, rt = aw.HRIRChannelMap.hesuvi14Channel(lay).resolve(lay, 14)
    assert lt.tolist() == [0, 8, 6, 6, 4, 12, 2, 10, -1, -1, -1, -1] and rt.tolist() == [1, 7, 13, 13, 5, 11, 3, 9, -1, -1, -1, -1]
    with pytest.raises(aw.HRIRError) as e1:        # HRIRError.invalidChannelMapping
        aw.HRIRChannelMap.hesuvi14Channel(aw.InputLayout.detect(2)).resolve(aw.InputLayout.detect(2), 2)
    assert e1.value.name == "INVALID_CHANNEL_MAPPING" and "(8, 7) out of range for 2 channels" in str(e1.value)
    with pytest.raises(aw.HRIRError) as e2:        # convolutionSetupFailed("No valid renderers created")
        lay3 = aw.InputLayout.detect(3)
        aw.HRIRChannelMap.hesuvi14Channel(lay3).resolve(lay3, 14)
    assert e2.value.name == "CONVOLUTION_SETUP_FAILED"


def test_wav_loader_matches_oracle(oracle, golden_dir, tmp_path):
    for name in ["NeutralSH1.0.wav", "RoomSH1.0.wav", "StageSH1.0.wav"]:
        p = os.path.join(golden_dir, "hrtf", name)
        w, ow = aw.WAVLoader.load(p), oracle.wav_load(p)
        assert (w.sample_rate, w.channel_count, w.frame_count) == (ow.sample_rate, ow.channel_count, ow.frame_count)
        assert np.array_equal(w.audio_data, ow.audio_data)
    from test_oracle_data_model import _write_wav
    p = str(tmp_path / "x.wav")
    cases = [(1, 16, 2, np.array([[0, 16384], [-32768, 32767], [1, -1]], dtype="<i2").tobytes(), False),
             (1, 16, 8, np.arange(40, dtype="<i2").tobytes(), True),
             (1, 32, 1, np.array([2 ** 30, -2 ** 31], dtype="<i4").tobytes(), False),
             (1, 24, 1, bytes([0, 0, 0x40, 0, 0, 0x80, 0xFF, 0xFF, 0x7F]), False),
             (3, 64, 2, np.array([0.25, -0.75, 1e-3, 3.0], dtype="<f8").tobytes(), False),
             (1, 8, 1, bytes([0, 128, 255]), False)]
    for tag, bits, ch, payload, ext in cases:
        _write_wav(p, tag, bits, ch, payload, extensible=ext)
        w, ow = aw.WAVLoader.load(p), oracle.wav_load(p)
        assert np.array_equal(w.audio_data, ow.audio_data) and w.channel_count == ow.channel_count
    _write_wav(p, 3, 32, 1, b"")
    with pytest.raises(aw.WAVError) as e:
        aw.WAVLoader.load(p)
    assert e.value.name == "WAV_EMPTY_FILE"
    _write_wav(p, 2, 4, 1, b"\x00" * 64)     # ADPCM
    with pytest.raises(aw.WAVError) as e:
        aw.WAVLoader.load(p)
    assert e.value.name in ("WAV_UNSUPPORTED_FORMAT", "WAV_EMPTY_FILE")
    with pytest.raises(aw.WAVError) as e:
        aw.WAVLoader.load(str(tmp_path / "missing.wav"))
    assert e.value.name == "WAV_FILE_READ"


def test_resampler_matches_oracle(oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(4320).astype(np.float32)
    for fr, to in [(48000, 96000), (48000, 44100), (44100, 48000), (96000, 48000), (48000, 48000.004), (48000, 192000)]:
        y, oy = aw.Resampler.resampleHighQuality(x, fr, to), oracle.resample_intended(x, fr, to)
        assert y.size == oy.size == oracle.resample_output_count(x.size, fr, to) or abs(fr - to) < 0.01
        assert np.array_equal(y, oy)


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/airwave_hip.h must compile as strict C99 (what cgo / Swift's clang importer /
    a ctypes generator see), with no C++ or HIP types in any signature."""
    import subprocess
    src = tmp_path / "abi_check.c"
    src.write_text('#include "airwave_hip.h"\nint main(void) { aw_context *c = 0; aw_eq *e = 0; (void)c; (void)e; return AW_OK; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    "-c", str(src), "-o", str(tmp_path / "abi_check.o")], check=True)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "airwave_hip.h")).read(), flags=re.S)   # comments may name them
    for banned in ("hipStream_t", "torch", "std::", "template", "class "):
        assert banned not in text, banned


def test_hesuvi_text_parser_matches_oracle_on_mutated_maps(oracle):
    """Token-level fuzz of parseHeSuViFormat (VirtualSpeaker.swift:301-346): names with aliases and odd case, comments,
    missing / extra fields, signs, non-numeric and overflowing indices, every line-ending flavour — the library's
    parser and the oracle's restatement must build the same map."""
    import random
    rng = random.Random(5)
    names = ["L", "R", "C", "FL", "fr", "Fc", "LFE", "SUB", "BL", "RL", "rr", "SL", "sr", "TFL", "Mine", "Ch7", "", "A B", "#x", ";y", "FL "]
    nums = ["0", "13", " 7", "+2", "-1", "007", "a", "1.5", "", "99999999999999999999", "1e2", "0x3", "٣", "5000000000",
            "-3000000000", "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809"]
    clamp = lambda v: max(-2**31, min(2**31 - 1, v))          # the ABI carries indices as int32 (out of range either way)
    seps = ["=", " = ", "\t=\t", "==", ":", ""]
    eols = ["\n", "\r\n", "\r"]
    for _ in range(300):
        lines = []
        for _ in range(rng.randrange(1, 8)):
            r = rng.random()
            if r < 0.75:
                k = rng.choice([2, 2, 2, 1, 3])
                lines.append(rng.choice(["", " ", "\t"]) + rng.choice(names) + rng.choice(seps) + rng.choice([",", " , ", ",\t"]).join(rng.choice(nums) for _ in range(k)))
            else:
                lines.append(rng.choice(["# comment", "; note", "", "   ", "garbage", "L = 1, 2 = 3"]))
        text = "".join(l + rng.choice(eols) for l in lines)
        m, om = aw.HRIRChannelMap.parseHeSuViFormat(text), oracle.parse_hesuvi_format(text)
        assert len(m) == len(om), (text, om)
        for k, v in om.items():
            assert m.getIndices(k) == (clamp(v[0]), clamp(v[1])), (text, k, v)


def test_hesuvi_text_parser_uses_foundation_character_sets(oracle):
    """parseHeSuViFormat (VirtualSpeaker.swift:301-346) splits at CharacterSet.newlines (U+000A-000D, U+0085, U+2028, U+2029), trims
    CharacterSet.whitespaces (Zs + TAB — not U+001C-001F, not ZERO WIDTH SPACE) and matches aliases on String.uppercased(), the full
    Unicode mapping (a long s reaches "SL" / "SUB", the "fl" ligature "FL" / "TFL")."""
    m = aw.HRIRChannelMap.parseHeSuViFormat("FL=0,1\x0bFR=1,0\x0cFC=2,3\x85BL=4,5\u2028BR=5,4\u2029SL=6,7\r\nSR=7,6")
    assert len(m) == 7 and m.getIndices("FC") == (2, 3) and m.getIndices("BR") == (5, 4) and m.getIndices("SR") == (7, 6)
    m = aw.HRIRChannelMap.parseHeSuViFormat("\u00a0FL\u2003=\u30000\u202f,\u205f1\u1680")
    assert m.getIndices("FL") == (0, 1)
    m = aw.HRIRChannelMap.parseHeSuViFormat("FL\x1d=0,1\nFR=\u200b2,3\nFC=4,\x1f5")
    assert len(m) == 1 and m.getIndices("FL\x1d") == (0, 1) and m.getIndices("FL") is None     # a custom speaker; the other lines have one index
    m = aw.HRIRChannelMap.parseHeSuViFormat("\u017fl=0,1\n\u017fub=2,3\n\ufb02=4,5\nt\ufb02=6,7\nP\u212a=8,9")
    assert m.getIndices("SL") == (0, 1) and m.getIndices("LFE") == (2, 3) and m.getIndices("FL") == (4, 5) and m.getIndices("TFL") == (6, 7)
    assert m.getIndices("P\u212a") == (8, 9) and len(m) == 5
    for text in ["FL=0,1\x0bFR=1,0\x85C = 2 , 3", "\u00a0FL\u2003=\u30000\u202f,\u205f1\u1680", "FL\x1d=0,1\nFR=\u200b2,3", "\u017fl=0,1\n\ufb02=4,5\nt\ufb02=6,7",
                 "# c\u2028; d\u2029 rl = 3,2 \r\n\rsub=+6,-0"]:
        om, m = oracle.parse_hesuvi_format(text), aw.HRIRChannelMap.parseHeSuViFormat(text)
        assert len(m) == len(om) and all(m.getIndices(k) == v for k, v in om.items()), (text, om)


def test_wav_reader_matches_oracle_on_generated_files(oracle, tmp_path):
    """Structure-level fuzz of the RIFF/WAVE reader: every sample format the loader knows, plain and EXTENSIBLE headers,
    1..16 channels, extra chunks (odd sizes with their pad byte) before, between and after fmt/data, truncated data.
    The library and the oracle's restatement of the decode contract (WAVLoader.swift:26-99) must agree bit for bit."""
    import random
    import struct
    rng = random.Random(17)
    nprng = np.random.default_rng(17)

    def chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\x00" if len(body) & 1 else b"")

    fmts = [(1, 8), (1, 16), (1, 24), (1, 32), (3, 32), (3, 64)]
    for it in range(120):
        tag, bits = rng.choice(fmts)
        ch = rng.choice([1, 2, 3, 7, 8, 14, 16])
        frames = rng.randrange(1, 40)
        rate = rng.choice([8000, 44100, 48000, 96000, 192000])
        n = frames * ch
        if tag == 3:
            vals = nprng.uniform(-1.5, 1.5, n)
            payload = vals.astype("<f4" if bits == 32 else "<f8").tobytes()
        elif bits == 8:
            payload = bytes(nprng.integers(0, 256, n, dtype=np.uint8))
        elif bits == 16:
            payload = nprng.integers(-32768, 32768, n).astype("<i2").tobytes()
        elif bits == 24:
            payload = b"".join(int(v).to_bytes(3, "little", signed=True) for v in nprng.integers(-2 ** 23, 2 ** 23, n))
        else:
            payload = nprng.integers(-2 ** 31, 2 ** 31, n).astype("<i4").tobytes()
        align = ch * bits // 8
        if rng.random() < 0.4:      # WAVE_FORMAT_EXTENSIBLE: real tag in the SubFormat GUID
            fmt = struct.pack("<HHIIHH", 0xFFFE, ch, rate, rate * align, align, bits) + struct.pack("<HHI", 22, bits, 0) + \
                  struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        else:
            fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits)
        if rng.random() < 0.2:
            payload = payload[: max(0, len(payload) - rng.randrange(1, align + 1))]     # truncated last frame(s)
        extras = [chunk(b"LIST", bytes(rng.randrange(0, 9))), chunk(b"fact", struct.pack("<I", frames)),
                  chunk(b"PEAK", bytes(rng.randrange(1, 24))), chunk(b"junk", b"x" * rng.randrange(0, 5))]
        body = b"WAVE"
        parts = [chunk(b"fmt ", fmt), chunk(b"data", payload)]
        for e in extras:
            if rng.random() < 0.5:
                parts.insert(rng.randrange(0, len(parts) + 1), e)
        if parts.index(chunk(b"fmt ", fmt)) > parts.index(chunk(b"data", payload)) and rng.random() < 0.7:
            parts.remove(chunk(b"fmt ", fmt)); parts.insert(0, chunk(b"fmt ", fmt))         # mostly fmt first, sometimes not
        body += b"".join(parts)
        path = str(tmp_path / f"g{it}.wav")
        open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
        try:
            ow = oracle.wav_load(path)
            oerr = None
        except ValueError as e:
            ow, oerr = None, str(e).split(":")[0]
        try:
            w = aw.WAVLoader.load(path)
            perr = None
        except aw.WAVError as e:
            w, perr = None, e.name
        if oerr is not None:
            assert perr == {"emptyFile": "WAV_EMPTY_FILE", "unsupportedFormat": "WAV_UNSUPPORTED_FORMAT", "fileReadError": "WAV_FILE_READ",
                            "invalidChannelCount": "INVALID_CHANNEL_COUNT"}[oerr], (it, oerr, perr)
        else:
            assert perr is None, (it, perr)
            assert (w.sample_rate, w.channel_count, w.frame_count) == (ow.sample_rate, ow.channel_count, ow.frame_count)
            assert np.array_equal(w.audio_data, ow.audio_data), it
