"""Oracle restatements of the host-side data model (VirtualSpeaker.swift, HRIRManager.swift assembly
rules, Resampler.swift, WAVLoader.swift).  The reference has no tests for these, so
the cases below are derived from the source lines cited."""
import os
import struct

import numpy as np
import pytest


def test_layout_detect(oracle):
    # VirtualSpeaker.swift:88-99
    assert oracle.layout_detect(2) == ["FL", "FR"]
    assert oracle.layout_detect(6) == ["FL", "FR", "FC", "LFE", "BL", "BR"]
    assert oracle.layout_detect(8) == ["FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR"]
    assert oracle.layout_detect(12)[8:] == ["TFL", "TFR", "TBL", "TBR"]
    assert oracle.layout_detect(3) == ["Ch0", "Ch1", "Ch2"]


def test_hesuvi14_table(oracle, golden_dir):
    # VirtualSpeaker.swift:270-297; track order documented :253-269
    m = oracle.map_hesuvi14(oracle.layout_detect(8))
    assert m == {"FL": (0, 1), "FR": (8, 7), "FC": (6, 13), "LFE": (6, 13), "BL": (4, 5), "BR": (12, 11),
                 "SL": (2, 3), "SR": (10, 9)}
    assert oracle.map_hesuvi14(["TFL", "Ch0"]) == {}   # unmapped speakers are skipped, not errors
    rows = np.load(os.path.join(golden_dir, "channel_map_tables.npz"))["rows"]
    for n, kind, i, l, r in rows.tolist():
        spk = oracle.layout_detect(n)
        m = (oracle.map_hesuvi14 if kind == 14 else oracle.map_hesuvi7)(spk)
        assert m.get(spk[i], (-1, -1)) == (l, r)


def test_hesuvi7_table(oracle):
    # VirtualSpeaker.swift:224-250
    m = oracle.map_hesuvi7(oracle.layout_detect(8))
    assert m["FL"] == (0, 1) and m["FR"] == (1, 0) and m["FC"] == (2, 2) and m["LFE"] == (2, 2)
    assert m["BL"] == (3, 4) and m["BR"] == (4, 3) and m["SL"] == (5, 6) and m["SR"] == (6, 5)


def test_interleaved_pairs_and_split_blocks(oracle):
    # VirtualSpeaker.swift:126-159, :200-209
    m = oracle.map_interleaved_pairs(["FL", "FR", "FC"])
    assert m == {"FL": (0, 1), "FR": (3, 2), "FC": (4, 5)}
    assert oracle.map_split_blocks(["FL", "FR", "FC"]) == {"FL": (0, 3), "FR": (1, 4), "FC": (2, 5)}


def test_parse_hesuvi_format(oracle):
    # VirtualSpeaker.swift:301-346
    text = "# comment\n; also comment\n\nL = 0, 1\n  r=8 ,7\nSUB = 6, 13\nRL = 4, 5\nbogus line\nX = 1\nY = 1, 2, 3\nMine = 2, 3\nC = a, 1\n"
    m = oracle.parse_hesuvi_format(text)
    assert m == {"FL": (0, 1), "FR": (8, 7), "LFE": (6, 13), "BL": (4, 5), "Mine": (2, 3)}


def test_assemble_rules(oracle, golden_dir):
    # HRIRManager.swift:355-360 map choice, :370-372 skip, :375-379 bounds, :420-422 empty
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(wav, oracle.layout_detect(12))
    assert lt.tolist() == [0, 8, 6, 6, 4, 12, 2, 10, -1, -1, -1, -1]
    assert rt.tolist() == [1, 7, 13, 13, 5, 11, 3, 9, -1, -1, -1, -1]
    seven = oracle.WAVData(48000.0, 7, 16, np.zeros((7, 16), dtype=np.float32))
    _, lt7, rt7 = oracle.assemble_tracks(seven, ["FL", "FR"])
    assert lt7.tolist() == [0, 1] and rt7.tolist() == [1, 0]
    two = oracle.WAVData(48000.0, 2, 16, np.zeros((2, 16), dtype=np.float32))
    with pytest.raises(oracle.InvalidChannelMapping):
        oracle.assemble_tracks(two, ["FL", "FR"])      # FR -> (8,7) out of range for 2 tracks
    with pytest.raises(oracle.ConvolutionSetupFailed):
        oracle.assemble_tracks(wav, ["Ch0", "Ch1", "Ch2"])


def test_resampler_intended(oracle):
    # Resampler.swift:31-68: identity within 0.01 Hz, length floor(count * to/from), lerp at i*from/to
    x = np.arange(10, dtype=np.float32)
    assert np.array_equal(oracle.resample_intended(x, 48000, 48000.005), x)
    up = oracle.resample_intended(x, 48000, 96000)
    assert up.size == 20 and np.allclose(up[:19], np.arange(19) * 0.5) and up[19] == 9.0
    down = oracle.resample_intended(x, 96000, 48000)
    assert down.size == 5 and np.allclose(down, [0, 2, 4, 6, 8])
    assert oracle.resample_output_count(4320, 48000, 44100) == 3968   # Int(4320 / (48000/44100)) truncates 3968.999.. (IEEE double, as Swift)
    assert oracle.resample_output_count(4320, 48000, 96000) == 8640


def _write_wav(path, tag, bits, ch, frames_bytes, rate=44100, extensible=False):
    if extensible:
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits, 22, bits, 0)
        fmt += struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00"
    body += b"data" + struct.pack("<I", len(frames_bytes)) + frames_bytes
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_loader_formats(oracle, tmp_path):
    # WAVLoader.swift:66-91: float passthrough, Int16/32768, Int32/2^31; planar [channels][frames]
    p = str(tmp_path / "a.wav")
    i16 = np.array([[0, 16384], [-32768, 32767], [1, -1]], dtype="<i2")
    _write_wav(p, 1, 16, 2, i16.tobytes())
    w = oracle.wav_load(p)
    assert (w.channel_count, w.frame_count, w.sample_rate) == (2, 3, 44100.0)
    assert np.array_equal(w.audio_data, (i16.T.astype(np.float32) / 32768.0))
    _write_wav(p, 1, 16, 8, np.zeros((5, 8), dtype="<i2").tobytes(), extensible=True)
    assert oracle.wav_load(p).channel_count == 8
    i32 = np.array([[2 ** 30], [-2 ** 31]], dtype="<i4")
    _write_wav(p, 1, 32, 1, i32.tobytes())
    assert oracle.wav_load(p).audio_data.tolist() == [[0.5, -1.0]]
    i24 = bytes([0x00, 0x00, 0x40, 0x00, 0x00, 0x80])
    _write_wav(p, 1, 24, 1, i24)
    assert oracle.wav_load(p).audio_data.tolist() == [[0.5, -1.0]]
    _write_wav(p, 3, 64, 1, np.array([0.25, -0.75], dtype="<f8").tobytes())
    assert oracle.wav_load(p).audio_data.tolist() == [[0.25, -0.75]]
    _write_wav(p, 3, 32, 1, b"")
    with pytest.raises(ValueError, match="emptyFile"):
        oracle.wav_load(p)
    with open(p, "wb") as f:
        f.write(b"not a wav file at all")
    with pytest.raises(ValueError, match="fileReadError"):
        oracle.wav_load(p)
