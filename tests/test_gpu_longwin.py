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


@pytest.mark.parametrize("rows,channels", [(40, 7), (48, 2), (56, 5), (72, 4), (80, 8), (96, 14), (112, 3), (120, 2), (40, 13), (56, 1), (120, 16), (72, 7)])
def test_window_lengths_that_are_not_powers_of_two(aw, oracle, monkeypatch, rows, channels):
    """R = 8 RA with RA = 5, 6, 7, 9, 10, 12, 14, 15: odd-prime in-register DFTs in the split / merge kernels, host tables from a
    mixed-radix transform.  Two windows per call."""
    monkeypatch.setenv("AW_LW", str(rows))
    taps, S = 20000, 2
    hop = rows * 4096 - 5 * 4096                              # history of a 20000-tap path-1 spatializer: 5 partitions of 4096
    F = hop + 30000
    h = oracle.synth_hrir(14, taps, seed=21)
    lt, rt = _maps(channels)
    x = oracle.synth_input(S, F, channels, seed=6)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    assert sp.info()["path"] == 1 and sp.info()["history"] == 5 * 4096
    y = sp.process(x)
    assert sp.info()["long_window_rows"] == rows and sp.info()["long_window_rows_rest"] == 0
    assert not np.isnan(y).any()
    for s in range(S):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL


def test_one_call_mixes_window_lengths(aw, oracle):
    """The automatic choice (lw_choose): a call a little longer than what the largest window holds runs as two groups of windows of
    different lengths that it fills, instead of two half-empty 128-row ones; the split of the timeline into calls still does not matter."""
    taps, S, C = 32768, 2, 7
    h = oracle.synth_hrir(14, taps, seed=31)
    lt, rt = _maps(C)
    hop128 = 128 * 4096 - 8 * 4096
    F = hop128 + 50000                                        # 11.3 s at 48 kHz
    x = oracle.synth_input(S, F, C, seed=8)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    y = sp.process(x)
    info = sp.info()
    Ra, Rb = info["long_window_rows"], info["long_window_rows_rest"]
    assert 0 < Ra < 128 and Rb < Ra, info                    # (two 128-row windows would be 1.94 x the call)
    hopa, hopb = Ra * 4096 - 32768, (Rb * 4096 - 32768 if Rb else 0)
    n = -(-(F - hopb) // hopa)
    assert (n * Ra + Rb) * 4096 < 1.25 * F, info
    for s in range(S):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL
    # the carried state after a two-group call: a second call continues the convolution
    x2 = oracle.synth_input(S, 70000, C, seed=9)
    y2 = sp.process(x2)
    both = np.concatenate([x, x2], axis=1)
    for s in range(S):
        ref = oracle.spatialize_f64(both[s], h, lt, rt)[F:]
        assert oracle.peak_rel_error(y2[s], ref) < TOL


def test_two_window_groups_on_a_batch(aw, oracle):
    """With a batch that amortises the second group's launches (lw_choose's group penalty) a 14 s call takes two window lengths; parity
    of streams from both ends of the batch, linearity over the whole batch."""
    import torch
    taps, S, C = 32768, 96, 7
    h = oracle.synth_hrir(14, taps, seed=33)
    lt, rt = _maps(C)
    F = 14 * 48000
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    info = sp.info()
    assert info["long_window_rows"] > 0 and info["long_window_rows_rest"] > 0, info
    edge = info["long_window_rows"] * 4096 - 32768           # first frame of the second group (one window in the first)
    for s in (0, S - 1):
        xs = x[s].cpu().numpy()
        ref = oracle.spatialize_f64(xs[:60000], h, lt, rt)
        assert oracle.peak_rel_error(y[s, :60000].cpu().numpy(), ref) < TOL
        lo = edge - 3000 - (taps - 1)
        seg = oracle.spatialize_f64(xs[lo:edge + 3000], h, lt, rt)[taps - 1:]          # across the seam of the two groups
        assert oracle.peak_rel_error(y[s, edge - 3000:edge + 3000].cpu().numpy(), seg) < TOL
        tail = oracle.spatialize_f64(xs[F - 3000 - (taps - 1):], h, lt, rt)[taps - 1:]
        assert oracle.peak_rel_error(y[s, F - 3000:].cpu().numpy(), tail) < TOL
    sp.reset()
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    sp.process_device(x.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * float(y.abs().max())


def test_two_windows_then_a_shorter_one(aw, oracle):
    """25 s: two windows of one length, then one of another (the first group has more than one window); parity at the head, across
    both seams and at the tail."""
    import torch
    taps, S, C = 32768, 40, 7
    h = oracle.synth_hrir(14, taps, seed=35)
    lt, rt = _maps(C)
    F = 25 * 48000
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    i = sp.info()
    Ra, Rb = i["long_window_rows"], i["long_window_rows_rest"]
    hopa = Ra * 4096 - 32768
    assert Ra > 0 and Rb > 0 and 2 * hopa < F <= 2 * hopa + Rb * 4096 - 32768, i      # two windows of Ra rows, one of Rb
    for s in (0, S - 1):
        xs = x[s].cpu().numpy()
        assert oracle.peak_rel_error(y[s, :30000].cpu().numpy(), oracle.spatialize_f64(xs[:30000], h, lt, rt)) < TOL
        for edge in (hopa, 2 * hopa, F - 2000):
            lo, hi = edge - 2000, min(F, edge + 2000)
            seg = oracle.spatialize_f64(xs[lo - (taps - 1):hi], h, lt, rt)[taps - 1:]
            assert oracle.peak_rel_error(y[s, lo:hi].cpu().numpy(), seg) < TOL, (s, edge)


def test_window_fill_does_not_depend_on_the_call_length(aw, oracle):
    """Policy pin: for every call length from 5 s to 30 s (48 kHz, 32768 taps) the padded length of the chosen windows stays within
    30 % of the frames processed (one window length per call and three lengths: up to 1.64)."""
    taps, C = 32768, 7
    h = oracle.synth_hrir(14, taps, seed=31)
    lt, rt = _maps(C)
    import torch
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=64, ctx=ctx)
    hist = sp.info()["history"]
    worst = 0.0
    for sec in range(5, 31):
        F = sec * 48000
        x = torch.zeros((64, F, C), dtype=torch.float32, device="cuda")
        y = torch.empty((64, F, 2), dtype=torch.float32, device="cuda")
        sp.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        i = sp.info()
        R, Rb = i["long_window_rows"], i["long_window_rows_rest"]
        assert R > 0, (sec, i)
        hop = R * 4096 - hist
        n = -(-F // hop) if Rb == 0 else -(-(F - (Rb * 4096 - hist)) // hop)
        padded = n * R * 4096 + Rb * 4096
        assert n * hop + (Rb * 4096 - hist if Rb else 0) >= F
        worst = max(worst, padded / F)
        del x, y
    assert worst < 1.30, worst


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


def test_reserve_builds_the_tables_and_process_builds_none(aw, oracle):
    """aw_spatializer_reserve(max) builds the long-window tables of the plan max implies (creation may block, process must not
    allocate: ConvolutionEngine allocates everything in init, ConvolutionEngine.swift:97-138); calls within the reserved length then
    choose among the window lengths whose tables exist — no table set and no scratch is added on the process path (device-buffer
    entry; the host-buffer entry's staging is reserved for single-stream, plug-in shaped spatializers only).  A spatializer that was
    never reserved builds them inside its first long call and gives the same samples."""
    import torch
    taps, S, C = 32768, 2, 7
    h = oracle.synth_hrir(14, taps, seed=41)
    lt, rt = _maps(C)
    F = 14 * 48000
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    assert sp.info()["long_window_table_sets"] == 0
    sp.reserve(F)
    built, held = sp.info()["long_window_table_sets"], sp.info()["scratch_bytes"]
    assert built >= 1 and held > 0
    i0 = sp.info()
    assert i0["reserve_tables_ms"] > 0 and i0["reserve_tables_ms"] + i0["reserve_upload_ms"] + i0["reserve_scratch_ms"] < 5000      # what bench.py prints as config.activation
    allocs, copies = i0["device_allocs"], i0["sync_copies"]
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    for n in (F, 200000, 4096, F):                           # long, shorter (an existing window length or the partitioned kernels), short, long
        sp.reset()
        sp.process_device(x.data_ptr(), y.data_ptr(), n)     # (a call of n frames reads the streams packed with stride n)
        torch.cuda.synchronize()
        assert sp.info()["long_window_table_sets"] == built and sp.info()["scratch_bytes"] == held, (n, sp.info())
        # the library's own count: a reserved spatializer makes no hipMalloc and no blocking hipMemcpy on its process path
        assert (sp.info()["device_allocs"], sp.info()["sync_copies"]) == (allocs, copies), (n, sp.info())
    assert sp.info()["long_window_rows"] > 0
    xs = x[1].cpu().numpy()
    assert oracle.peak_rel_error(y[1].cpu().numpy()[:50000], oracle.spatialize_f64(xs[:50000], h, lt, rt)) < TOL
    lazy = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    y2 = torch.empty_like(y)
    lazy.process_device(x.data_ptr(), y2.data_ptr(), F)      # never reserved: the first long call builds what it needs
    torch.cuda.synchronize()
    assert lazy.info()["long_window_table_sets"] >= 1
    assert float((y2 - y).abs().max()) <= 3e-6 * float(y.abs().max())


def test_path0_spatializer_takes_the_long_window_kernels_by_itself(aw, oracle, monkeypatch):
    """An HRIR one fused window could hold (7 channels x 8640 taps: path 0, cfg 4's layout), AW_LW unset: the policy sends a long call of a batch to the
    long-window kernels and a single stream to the fused tiles; a small scratch budget (AW_SPEC_SCRATCH_MB now applies to every
    spatializer that can take the long-window kernels) runs the batch as several stream chunks; any split of the timeline into calls
    gives the same samples."""
    monkeypatch.delenv("AW_LW", raising=False)
    monkeypatch.setenv("AW_SPEC_SCRATCH_MB", "96")
    taps, S, C, F = 8640, 24, 7, 300000
    h = oracle.synth_hrir(14, taps, seed=12)
    lt, rt = _maps(C)
    x = oracle.synth_input(S, F, C, seed=13)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    assert sp.info()["path"] == 0
    y = sp.process(x)
    i = sp.info()
    assert i["long_window_rows"] > 0, i
    per_stream = (i["long_window_rows"] + i["long_window_rows_rest"]) * 4096 * 9 // 2 * 8            # rows of 3.5 pairs + s1/s2, bytes
    assert per_stream * S > (96 << 20)                       # more than the budget: several stream chunks
    for s in (0, 11, S - 1):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL
    sp.reset()
    parts, pos = [], 0
    for n in (3000, 150000, 147000):                         # fused tiles, long-window kernels, long-window kernels
        parts.append(sp.process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    assert np.max(np.abs(np.concatenate(parts, axis=1) - y)) <= 3e-6 * np.abs(y).max()
    one = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=1)
    y1 = one.process(np.ascontiguousarray(x[:1]))
    assert one.info()["long_window_rows"] == 0               # a single stream: the fused tiles
    assert oracle.peak_rel_error(y1[0], oracle.spatialize_f64(x[0], h, lt, rt)) < TOL


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


@pytest.mark.parametrize("rows,channels,taps,unmap", [(64, 7, 32768, None), (40, 14, 20000, None), (128, 2, 32768, None), (56, 5, 20000, 2), (96, 16, 20000, 9), (32, 1, 9000, None)])
def test_device_built_tables_match_the_host_builder(aw, oracle, monkeypatch, rows, channels, taps, unmap):
    """The HRIR-prep kernels (device/prep_kernels.hip: float64 on the GPU, the analogue of the partition FFTs of ConvolutionEngine.init,
    ConvolutionEngine.swift:141-175) against the float64 host builder (AW_LW_TABLES=host, the builder the CPU emulation tests run on):
    same samples to float32 rounding of the tables, and both within tolerance of the float64 oracle.  Odd channel counts (the folded real
    last channel), 14 and 16 channels, an unmapped speaker, window lengths that are not powers of two."""
    monkeypatch.setenv("AW_LW", str(rows))
    S = 2
    h = oracle.synth_hrir(14, taps, seed=31)
    lt, rt = _maps(channels)
    if unmap is not None:
        lt, rt = lt.copy(), rt.copy()
        lt[unmap] = -1                                        # a speaker without a mapping is skipped (HRIRManager.swift:370-372)
    ys, x = {}, None
    for mode in ("host", "gpu"):
        monkeypatch.setenv("AW_LW_TABLES", mode)             # read once, at context creation
        ctx = aw.Context(0)
        sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        if x is None:
            F = rows * 4096 - sp.info()["history"] + 20000   # two windows
            x = oracle.synth_input(S, F, channels, seed=8)
        ys[mode] = sp.process(x)
        assert sp.info()["long_window_rows"] == rows, sp.info()
    ref = oracle.spatialize_f64(x[1], h, lt, rt)
    peak = np.max(np.abs(ref))
    assert np.max(np.abs(ys["gpu"][1] - ys["host"][1])) <= 2e-6 * peak
    for mode in ys:
        assert oracle.peak_rel_error(ys[mode][1], ref) < TOL, mode
