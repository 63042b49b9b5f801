"""GPU parity of the long-window path (airwave_amd/csrc/device/tile_lw.hpp: split -> rows -> merge kernels) through the
C ABI against the float64 oracle: long HRIRs (cfg 3's 32768 taps) on calls long enough for windows of 32/64/128 x 4096
frames.  The reference's semantics are those of ConvolutionEngine.process (ConvolutionEngine.swift:232-367) summed over
speakers (RealtimeAudioProcessor.swift:141-172): exact streaming linear convolution, state = the convolution tail."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


TOL = 1e-5          # north_star: <= 1e-5 max rel error (peak-relative per stream and ear)


def _maps(channels):
    lt = np.resize(np.array([0, 8, 6, 6, 4, 12, 2, 10], dtype=np.int32), channels)
    rt = np.resize(np.array([1, 7, 13, 13, 5, 11, 3, 9], dtype=np.int32), channels)
    return lt, rt


@pytest.mark.parametrize("rows", [32, 64, 128])
@pytest.mark.parametrize("channels", [7, 8, 2, 1, 5, 14, 9, 12, 13, 16])
def test_long_window_kernels_match_truth(aw, oracle, monkeypatch, rows, channels):
    monkeypatch.setenv("AW_LW", str(rows))                   # force the window length (automatic choice: next test)
    taps, S = 32768, 2
    F = {32: 150000, 64: 200000, 128: 60000}[rows]           # 32: two windows; 64: one; 128: a short call in a long window
    h = oracle.synth_hrir(14, taps, seed=77)
    lt, rt = _maps(channels)
    x = oracle.synth_input(S, F, channels, seed=5)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    assert sp.info()["path"] == 1
    y = sp.process(x)
    assert sp.info()["long_window_rows"] == rows
    assert not np.isnan(y).any()
    for s in range(S):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL


def test_long_window_state_carries_across_calls_and_kernel_sets(aw, oracle, monkeypatch):
    """Any split of the timeline into calls gives the same samples — also when short calls run on the partitioned kernels
    (4096-frame blocks) and long ones on the long-window kernels: both keep the same history (reset clears it)."""
    monkeypatch.setenv("AW_LW", "32")
    taps, S, C = 20000, 3, 7
    h = oracle.synth_hrir(14, taps, seed=9)
    lt, rt = _maps(C)
    lt[2] = -1                                               # an unmapped speaker is skipped (HRIRManager.swift:370-372)
    F = 260000
    x = oracle.synth_input(S, F, C, seed=11)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    whole = sp.process(x)
    assert sp.info()["long_window_rows"] == 32
    for s in range(S):
        assert oracle.peak_rel_error(whole[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL
    sp.reset()
    monkeypatch.delenv("AW_LW")
    sp2 = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)    # automatic choice per call
    parts, pos, used = [], 0, []
    for n in [5000, 120001, 777, 130000, 4222]:
        parts.append(sp2.process(np.ascontiguousarray(x[:, pos:pos + n])))
        used.append(sp2.info()["long_window_rows"])
        pos += n
    assert pos == F
    assert used[0] == 0 and used[2] == 0 and used[4] == 0     # short calls: the partitioned kernels
    assert np.max(np.abs(np.concatenate(parts, axis=1) - whole)) <= 3e-6 * np.abs(whole).max()


def test_long_window_stream_chunks(aw, oracle, monkeypatch):
    monkeypatch.setenv("AW_LW", "32")
    monkeypatch.setenv("AW_SPEC_SCRATCH_MB", "20")           # one (stream, window) needs 4.5 MB: several stream chunks
    taps, S, C = 16000, 7, 4
    h = oracle.synth_hrir(14, taps, seed=3)
    lt, rt = _maps(C)
    x = oracle.synth_input(S, 110000, C, seed=2)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    y = sp.process(x)
    assert sp.info()["long_window_rows"] == 32
    for s in (0, 3, 6):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL


def test_reserved_spatializer_never_reallocates(oracle, monkeypatch):
    """aw_spatializer_reserve(max) then shorter calls: the scratch stays as allocated whichever kernel set runs — the stream
    chunk is clamped to the held buffer (a 1-block call would otherwise pick a larger chunk whose rounding exceeds what a
    10-block reserve allocated: 64 MiB budget, 1 pair, 2 partitions, 400 streams: 65 472 KiB against 64 512 KiB)."""
    import torch
    import airwave_amd as aw
    monkeypatch.setenv("AW_SPEC_SCRATCH_MB", "64")
    monkeypatch.setenv("AW_WINDOW", "4096")                  # the partitioned kernels for an HRIR one fused window could hold
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    taps, S, C, F = 8000, 400, 2, 10 * 4096
    h = oracle.synth_hrir(14, taps, seed=4)
    lt, rt = _maps(C)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    assert sp.info()["path"] == 1 and sp.info()["partitions"] == 2
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C)
    sp.reserve(F)
    held = sp.info()["scratch_bytes"]
    assert held > 0
    sizes = (4096, F, 100, 3 * 4096 + 5)
    for n in sizes:
        sp.process_device(x.data_ptr(), y.data_ptr(), n)
        assert sp.info()["scratch_bytes"] == held, n
    torch.cuda.synchronize()
    flat = x.view(-1, C)                                     # a call of n frames reads streams packed with stride n
    s = 399
    ref = oracle.spatialize_f64(np.concatenate([flat[s * n:(s + 1) * n].cpu().numpy() for n in sizes]), h, lt, rt)
    n = sizes[-1]
    assert oracle.peak_rel_error(y.view(-1, 2)[s * n:(s + 1) * n].cpu().numpy(), ref[-n:]) < TOL


@pytest.mark.parametrize("fmt", ["float32", "pcm16"])
def test_imported_long_tap_wav_through_preset_activation(aw, oracle, tmp_path, fmt):
    """BASELINE cfg 3 reads "HeSuVi 14ch 48kHz -> imported .wav HRIR (long tap)": a 14-track 32768-frame WAV written here goes through
    the product's own reader and activatePreset (HRIRManager.swift:347-446: load -> hesuvi14 map -> engines), and a long call of the
    activated renderer network runs on the long-window kernels; the oracle reads the same file."""
    import struct
    taps, C = 32768, 14
    h = oracle.synth_hrir(C, taps, seed=1234)
    h = (h / np.abs(h).max() * 0.9).astype(np.float32)
    inter = np.ascontiguousarray(h.T)                                  # [frame][track]
    if fmt == "float32":
        data, tag, bits = inter.astype("<f4").tobytes(), 3, 32
    else:
        data, tag, bits = np.round(inter * 32767.0).astype("<i2").tobytes(), 1, 16
    block = C * bits // 8
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, tag, C, 48000, 48000 * block, block, bits) + b"data" + struct.pack("<I", len(data)) + data
    path = str(tmp_path / f"long_{fmt}.wav")
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    speakers = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]
    layout = aw.InputLayout(speakers, "7 speakers")
    mgr = aw.HRIRManager()
    sp = mgr.activatePreset(path, 48000.0, layout, n_streams=2)
    assert mgr.isReady and sp.info()["path"] == 1
    x = oracle.synth_input(2, 200000, 7, seed=4)
    y = sp.process(x)
    assert sp.info()["long_window_rows"] == 64
    tracks, lt, rt = oracle.assemble_tracks(oracle.wav_load(path), speakers)
    assert tracks.shape == (14, taps)
    for s in range(2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], tracks, lt, rt)) < TOL
