"""Parity of the HIP spatializer (through the C ABI) with the float64 goldens and the float32 CPU
oracle.  Tolerance: BASELINE.json north_star, <= 1e-5 max error relative to the signal peak."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


def _aligned(hop):
    """runtime.cpp align_hop: tiles start on 64-frame boundaries of the timeline."""
    return hop - hop % 64 if hop > 1024 else hop


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def wav(oracle, golden_dir, name):
    return oracle.wav_load(os.path.join(golden_dir, "hrtf", name))


@pytest.mark.parametrize("gold,wavname,speakers", [
    ("cfg1_neutral_stereo.npz", "NeutralSH1.0.wav", 2),
    ("cfg2_room_71.npz", "RoomSH1.0.wav", 8),
    ("cfg3_stage_7spk.npz", "StageSH1.0.wav", ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]),
])
def test_goldens_via_preset_activation(aw, oracle, golden_dir, gold, wavname, speakers):
    g = load(golden_dir, gold)
    layout = aw.InputLayout.detect(speakers) if isinstance(speakers, int) else aw.InputLayout(speakers, "custom")
    mgr = aw.HRIRManager()
    sp = mgr.activatePreset(os.path.join(golden_dir, "hrtf", wavname), 48000.0, layout)
    # 4320 taps, one stream: 8192-frame windows for every layout (stereo moves to 16384-frame windows from 16 streams)
    assert mgr.isReady and sp.info()["path"] == 0 and sp.info()["hop"] == _aligned(8192 - 4319)
    x = oracle.synth_input(1, int(g["frames"]), len(layout.channels), seed=int(g["seed"]))
    y = sp.process(x)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], g["expected"][:, ear]) < TOL


def test_custom_14ch_text_map(aw, oracle, golden_dir):
    g = load(golden_dir, "cfg3_stage_14ch_custom.npz")
    cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(golden_dir, "hesuvi14_custom_map.txt")).read())
    sp = aw.HRIRManager().activatePreset(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"), 48000.0,
                                        aw.InputLayout.detect(14), hrirMap=cmap)
    x = oracle.synth_input(1, int(g["frames"]), 14, seed=int(g["seed"]))
    assert oracle.peak_rel_error(sp.process(x)[0], g["expected"]) < TOL


@pytest.mark.parametrize("channels", [1, 2, 3, 5, 6, 7, 8, 12])
def test_every_channel_count_matches_f32_oracle_and_truth(aw, oracle, golden_dir, channels):
    w = wav(oracle, golden_dir, "RoomSH1.0.wav")
    spk = oracle.layout_detect(channels)
    if channels in (1, 3, 5, 7):
        spk = oracle.layout_detect(8)[:channels]
    tracks, lt, rt = oracle.assemble_tracks(w, spk)
    frames = 9000
    x = oracle.synth_input(2, frames, channels, seed=77)
    sp = aw.Spatializer(aw.HRIR(tracks), lt, rt, n_streams=2)
    y = sp.process(x)
    yo = oracle.spatialize_f32(x, tracks, lt, rt)
    for s in range(2):
        ref = oracle.spatialize_f64(x[s], tracks, lt, rt)
        assert oracle.peak_rel_error(y[s], ref) < TOL
        assert oracle.peak_rel_error(y[s], yo[s]) < TOL


@pytest.mark.parametrize("channels", [2, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_interior_kernels_of_every_vector_layout(aw, oracle, golden_dir, channels):
    """Long enough for interior tiles (whole-frame vector loads; 6/7/14-channel frames are not whole float4s and
    read into the next frame: zero tables must cancel the stray lanes; 10-16 channels run two passes, 4 pairs then the rest
    accumulated into the output), odd frame count, two calls."""
    w = wav(oracle, golden_dir, "StageSH1.0.wav")
    tracks = np.asarray(w.audio_data)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    frames = 21001
    x = oracle.synth_input(3, frames, channels, seed=channels)
    sp = aw.Spatializer(aw.HRIR(tracks), lt, rt, n_streams=3)
    y = np.concatenate([sp.process(x[:, :15000]), sp.process(x[:, 15000:])], axis=1)
    assert not np.isnan(y).any()
    for s in (0, 2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], tracks, lt, rt)) < TOL


@pytest.mark.parametrize("taps", [1, 2, 511, 512, 513, 4097, 6145])
def test_hrir_lengths_on_the_fused_path(aw, oracle, taps):
    h = oracle.synth_hrir(4, taps, seed=taps)
    x = oracle.synth_input(1, 2 * (8192 - taps + 1) + 5, 2, seed=9)
    sp = aw.Spatializer(aw.HRIR(h), [0, 2], [1, 3])
    assert sp.info()["path"] == 0 and sp.info()["hop"] == _aligned(8192 - (taps - 1))
    y = sp.process(x)
    ref = oracle.spatialize_f64(x[0], h, [0, 2], [1, 3])
    assert oracle.peak_rel_error(y[0], ref) < TOL


@pytest.mark.parametrize("channels,taps,streams,fft", [(2, 4320, 1, 8192), (2, 4320, 23, 8192), (2, 4320, 24, 16384), (2, 1799, 64, 8192), (2, 1800, 64, 16384), (1, 4320, 7, 8192), (1, 4320, 8, 16384),
                                                       (8, 4320, 128, 8192), (8, 5499, 16, 8192), (8, 5500, 16, 16384), (8, 6100, 4, 8192), (2, 6146, 1, 16384), (4, 5299, 128, 8192), (4, 5300, 128, 16384), (6, 4699, 128, 8192), (6, 4700, 128, 16384),
                                                       (1, 512, 16, 16384), (3, 1799, 128, 8192), (3, 1800, 128, 16384), (3, 4320, 128, 16384), (3, 4320, 11, 8192), (5, 4320, 32, 16384), (5, 3399, 128, 8192), (5, 3400, 128, 16384),
                                                       (7, 4320, 128, 8192), (7, 4399, 16, 8192), (7, 4400, 16, 16384), (12, 6145, 16, 8192), (9, 6145, 16, 8192), (13, 6000, 16, 8192), (16, 5900, 16, 8192)])
def test_window_policy(aw, oracle, channels, taps, streams, fft):
    """runtime.cpp: the measured crossover of the two fused kernels by layout and HRIR length, 8192-frame windows for
    batches too small to fill the chip with 16384-frame tiles, 16384 whenever one 8192-frame window cannot hold the HRIR."""
    h = oracle.synth_hrir(14, taps, seed=1)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=streams)
    assert sp.info()["path"] == 0 and sp.info()["fft"] == fft
    x = oracle.synth_input(streams, 20011, channels, seed=3)
    y = sp.process(x)
    s = streams - 1
    assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL


def test_ragged_call_sizes_carry_state_exactly(aw, oracle, golden_dir):
    """Any split of the timeline into calls gives the same samples as one call: the tail state is
    carried like consecutive ConvolutionEngine.process calls (overlap + FDL, ConvolutionEngine.swift:237-264)."""
    w = wav(oracle, golden_dir, "NeutralSH1.0.wav")
    tracks, lt, rt = oracle.assemble_tracks(w, ["FL", "FR"])
    x = oracle.synth_input(3, 12000, 2, seed=5)
    hrir = aw.HRIR(tracks)
    whole = aw.Spatializer(hrir, lt, rt, n_streams=3).process(x)
    sp = aw.Spatializer(hrir, lt, rt, n_streams=3)
    parts, pos = [], 0
    for n in [1, 7, 512, 3873, 3874, 100, 2000, 1633]:
        parts.append(sp.process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    assert pos == 12000
    chunked = np.concatenate(parts, axis=1)
    assert np.max(np.abs(chunked - whole)) <= 2e-6 * np.abs(whole).max()
    ref = oracle.spatialize_f64(x[1], tracks, lt, rt)
    assert oracle.peak_rel_error(chunked[1], ref) < TOL
    # reset() == fresh engine (ConvolutionEngine.swift:397-407)
    sp.reset()
    again = sp.process(np.ascontiguousarray(x[:, :5000]))
    assert np.max(np.abs(again - whole[:, :5000])) <= 2e-6 * np.abs(whole).max()


def test_unmapped_channels_are_skipped_and_errors_match(aw, oracle):
    h = oracle.synth_hrir(4, 100, seed=1)
    hrir = aw.HRIR(h)
    x = oracle.synth_input(1, 3000, 3, seed=3)
    y = aw.Spatializer(hrir, [0, -1, 2], [1, -1, 3]).process(x)          # HRIRManager.swift:370-372
    ref = oracle.spatialize_f64(x[0], h, [0, -1, 2], [1, -1, 3])
    assert oracle.peak_rel_error(y[0], ref) < TOL
    with pytest.raises(aw.HRIRError) as e:
        aw.Spatializer(hrir, [0, 4], [1, 2])                             # :375-379
    assert e.value.name == "INVALID_CHANNEL_MAPPING"
    with pytest.raises(aw.HRIRError) as e:
        aw.Spatializer(hrir, [-1, -1], [-1, -1])                         # :420-422
    assert e.value.name == "CONVOLUTION_SETUP_FAILED"


def test_planar_plugin_entry_and_passthrough(aw, oracle, golden_dir):
    mgr = aw.HRIRManager()
    l = oracle.synth_input(1, 700, 1, seed=1)[0, :, 0]
    r = oracle.synth_input(1, 700, 1, seed=2)[0, :, 0]
    pl, pr = mgr.process(l, r)                                           # passthrough, HRIRManager.swift:550-559
    assert np.array_equal(pl, l) and np.array_equal(pr, r)
    ml, mr = mgr.process(l, None)
    assert np.array_equal(ml, l) and np.array_equal(mr, l)
    mgr.activatePreset(os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav"), 48000.0, aw.InputLayout.detect(2))
    w = wav(oracle, golden_dir, "NeutralSH1.0.wav")
    tracks, lt, rt = oracle.assemble_tracks(w, ["FL", "FR"])
    ol, orr = mgr.process(l, r)
    ref = oracle.spatialize_f64(np.stack([l, r], 1), tracks, lt, rt)
    assert oracle.peak_rel_error(np.stack([ol, orr], 1), ref) < TOL
    mgr.resetConvolutionState()
    ol2, orr2 = mgr.process(l, None)                                     # mono duplication
    ref2 = oracle.spatialize_f64(np.stack([l, l], 1), tracks, lt, rt)
    assert oracle.peak_rel_error(np.stack([ol2, orr2], 1), ref2) < TOL


def test_resampled_hrir_activation(aw, oracle, golden_dir):
    # cfg 4/5: HRIR resampled to the device rate (HRIRManager.swift:389-403)
    w = wav(oracle, golden_dir, "StageSH1.0.wav")
    for rate in (44100.0, 96000.0):
        sp = aw.HRIRManager().activatePreset(os.path.join(golden_dir, "hrtf", "StageSH1.0.wav"), rate, aw.InputLayout.detect(2))
        tracks, lt, rt = oracle.assemble_tracks(w, ["FL", "FR"], target_rate=rate)
        x = oracle.synth_input(1, 6000, 2, seed=4)
        ref = oracle.spatialize_f64(x[0], tracks, lt, rt)
        assert oracle.peak_rel_error(sp.process(x)[0], ref) < TOL


def test_device_synth_fill_is_the_oracle_generator(aw, oracle):
    ctx = aw.default_context()
    S, F, C = 3, 1000, 7
    d = ctx.alloc(S * F * C * 4)
    ctx.synth_fill(d, S, F, C, seed=oracle.SYNTH_SEED, first_stream=5)
    got = np.zeros((S, F, C), dtype=np.float32)
    ctx.d2h(got, d)
    ctx.free(d)
    assert np.array_equal(got, oracle.synth_input(S, F, C, seed=oracle.SYNTH_SEED, first_stream=5))


def test_full_size_properties_cfg2(aw, oracle, golden_dir):
    """BASELINE cfg 2 at full size (128 streams x 10 s x 8 ch) through device buffers:
    (a) sampled streams match the float64 truth on a window, (b) linearity: spatialize(a*x) == a*spatialize(x),
    (c) every stream is processed (checksum of checksums vs per-stream recomputation of 4 streams)."""
    import torch
    S, F, C = 128, 480000, 8
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    w = wav(oracle, golden_dir, "RoomSH1.0.wav")
    tracks, lt, rt = oracle.assemble_tracks(w, oracle.layout_detect(8))
    sp = aw.Spatializer(aw.HRIR(tracks, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    # (a) truth on the first 12000 frames and on the last 3000 frames of three streams
    for s in (0, 63, 127):
        xs = x[s, :12000].cpu().numpy()
        assert np.array_equal(xs, oracle.synth_input(1, 12000, C, first_stream=s)[0])
        ref = oracle.spatialize_f64(xs, tracks, lt, rt)
        assert oracle.peak_rel_error(y[s, :12000].cpu().numpy(), ref) < TOL
        tail_in = x[s, F - 3000 - 4319:].cpu().numpy()
        ref_tail = oracle.spatialize_f64(tail_in, tracks, lt, rt)[4319:]
        assert oracle.peak_rel_error(y[s, F - 3000:].cpu().numpy(), ref_tail) < TOL
    # (b) linearity on the whole batch
    sp.reset()
    x.mul_(-0.5)
    y2 = torch.empty_like(y)
    sp.process_device(x.data_ptr(), y2.data_ptr(), F)
    torch.cuda.synchronize()
    peak = float(y.abs().max())
    assert float((y2 + 0.5 * y).abs().max()) <= 2e-6 * peak
    # (c) per-stream energy is non-trivial everywhere (no stream skipped or duplicated)
    e = (y.double() ** 2).sum(dim=(1, 2))
    assert float(e.min()) > 0.25 * float(e.max())
    assert len(set(np.round(e.cpu().numpy(), 6).tolist())) == S


def test_long_tap_partitioned_path(aw, oracle, golden_dir):
    """cfg 3: 32768-tap synthetic HRIR -> partitioned path (8 partitions of 4096 frames), golden truth."""
    g = load(golden_dir, "cfg3_longtap_7spk.npz")
    h = oracle.synth_hrir(14, int(g["hrir_taps"]), seed=int(g["hrir_seed"]))
    sp = aw.Spatializer(aw.HRIR(h), g["left_track"], g["right_track"])
    info = sp.info()
    assert info["path"] == 1 and info["partitions"] == 8 and info["hop"] == 4096
    x = oracle.synth_input(1, int(g["frames"]), 7, seed=int(g["seed"]))
    assert oracle.peak_rel_error(sp.process(x)[0], g["expected"]) < TOL


@pytest.mark.parametrize("taps,channels,path,fft", [(6146, 2, 0, 16384), (7800, 8, 0, 16384), (8640, 8, 0, 16384), (11200, 8, 0, 16384), (11201, 8, 1, 8192), (8640, 7, 0, 16384), (11500, 7, 0, 16384), (12288, 5, 0, 16384), (8500, 6, 0, 16384),
                                                    (12288, 7, 1, 8192), (10500, 4, 0, 16384), (10501, 4, 1, 8192), (12288, 6, 0, 16384), (12290, 6, 1, 8192), (12290, 5, 1, 8192), (6100, 12, 0, 8192), (6146, 12, 1, 8192), (6145, 9, 0, 8192), (6000, 16, 1, 8192), (20000, 3, 1, 8192),
                                                    (20000, 1, 1, 8192), (40000, 2, 1, 8192), (70000, 7, 1, 8192)])
def test_long_hrir_paths_chunks_and_state(aw, oracle, taps, channels, path, fft, monkeypatch):
    """HRIRs beyond one 8192-frame window: up to 12288 taps (cfg 4: 4320 taps resampled x2 = 8640) run fused on
    16384-frame windows where that measures faster (runtime.cpp: per channel count), longer ones — and the long end
    of the wider layouts — on the partitioned path (with stream chunking of its scratch)."""
    monkeypatch.setenv("AW_SPEC_SCRATCH_MB", "3")          # forces several stream chunks
    h = oracle.synth_hrir(14, taps, seed=taps)
    lt = np.resize(np.array([0, 8, 6, 6, 4, 12, 2, 10], dtype=np.int32), channels)
    rt = np.resize(np.array([1, 7, 13, 13, 5, 11, 3, 9], dtype=np.int32), channels)
    S, F = 3, 30000
    x = oracle.synth_input(S, F, channels, seed=21)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    assert sp.info()["path"] == path and sp.info()["fft"] == fft
    whole = sp.process(x)
    for s in range(S):
        assert oracle.peak_rel_error(whole[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL
    sp.reset()
    parts, pos = [], 0
    for n in [1, 4095, 4097, 1000, 807, 20000]:
        parts.append(sp.process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    assert pos == F
    assert np.max(np.abs(np.concatenate(parts, axis=1) - whole)) <= 3e-6 * np.abs(whole).max()


def test_cfg4_shaped_chain_properties(aw, oracle, golden_dir):
    """BASELINE cfg 4 shape at a reduced batch (96 kHz, 7 speakers, StageSH1.0 resampled x2 = 8640 taps ->
    fused path on 16384-frame windows, then the 10-band EQ in place on the same stream): spot streams against the oracle chain
    (float64 convolution truth -> sequential Float64 EQ), split-call invariance of the whole chain, linearity."""
    import torch
    S, F, C, fs = 96, 200001, 7, 96000.0
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    w = wav(oracle, golden_dir, "StageSH1.0.wav")
    spk = ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]
    tracks, lt, rt = oracle.assemble_tracks(w, spk, target_rate=fs)
    assert tracks.shape[1] == 8640
    d = aw.EqualizerAPOParser.parse(open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "f.txt")
    od = oracle.eq_parse(open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "f.txt")

    def chain():
        sp = aw.Spatializer(aw.HRIR(tracks, fs, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        return sp, aw.ParametricEqualizerState(d, fs, n_streams=S, ctx=ctx)

    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=4)
    sp, eq = chain()
    assert sp.info()["path"] == 0 and sp.info()["fft"] == 16384 and sp.info()["hop"] == _aligned(16384 - 8640)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    eq.process_device(y.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    n = 30000
    for s in (0, 47, 95):
        xs = x[s, :n].cpu().numpy()
        conv = oracle.spatialize_f64(xs, tracks, lt, rt).astype(np.float32)
        el, er = oracle.eq_prepare(od, fs).process(conv[:, 0], conv[:, 1])
        got = y[s, :n].cpu().numpy()
        assert oracle.peak_rel_error(got, np.stack([el, er], axis=1)) < TOL
    # the same timeline in two ragged calls continues both stages exactly
    sp2, eq2 = chain()
    cut = 77777
    a, b = x[:, :cut].contiguous(), x[:, cut:].contiguous()
    ya = torch.empty((S, cut, 2), dtype=torch.float32, device="cuda")
    yb = torch.empty((S, F - cut, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for xi, yi, ni in ((a, ya, cut), (b, yb, F - cut)):
        sp2.process_device(xi.data_ptr(), yi.data_ptr(), ni)
        eq2.process_device(yi.data_ptr(), yi.data_ptr(), ni)
    torch.cuda.synchronize()
    peak = float(y.abs().max())
    assert float((torch.cat([ya, yb], 1) - y).abs().max()) <= 2e-6 * peak
    # linearity of the chain
    sp3, eq3 = chain()
    x.mul_(-0.5)
    y3 = torch.empty_like(y)
    torch.cuda.synchronize()
    sp3.process_device(x.data_ptr(), y3.data_ptr(), F)
    eq3.process_device(y3.data_ptr(), y3.data_ptr(), F)
    torch.cuda.synchronize()
    assert float((y3 + 0.5 * y).abs().max()) <= 2e-6 * peak


def test_cfg1_full_length_against_the_cpu_reference_port(aw, oracle, golden_dir):
    """BASELINE cfg 1 (the reference's own CPU-runnable case) at its full size: 1 stream, stereo, 10 s at 48 kHz
    through NeutralSH1.0 — every one of the 480 000 output frames against the float32 port of the reference
    algorithm (B = 512 blocks) and against the float64 truth."""
    w = wav(oracle, golden_dir, "NeutralSH1.0.wav")
    tracks, lt, rt = oracle.assemble_tracks(w, ["FL", "FR"])
    x = oracle.synth_input(1, 480000, 2)
    sp = aw.Spatializer(aw.HRIR(tracks), lt, rt, n_streams=1)
    y = sp.process(x)[0]
    port = oracle.spatialize_f32(x, tracks, lt, rt)[0]
    truth = oracle.spatialize_f64(x[0], tracks, lt, rt)
    assert oracle.peak_rel_error(y, port) < TOL and oracle.peak_rel_error(y, truth) < TOL
    assert oracle.peak_rel_error(port, truth) < TOL          # the port itself sits inside the same budget


@pytest.mark.parametrize("taps,channels,seed", [(700, 3, 1), (4320, 2, 2), (4320, 8, 3), (9000, 7, 4), (15000, 4, 5)])
def test_random_call_splits_on_every_path(aw, oracle, taps, channels, seed):
    """Any split of a timeline into calls gives the same samples as one call, on all three paths (8192-frame windows,
    16384-frame windows, partitioned), with call sizes around the hop/window/block boundaries."""
    rng = np.random.default_rng(seed)
    h = oracle.synth_hrir(14, taps, seed=taps)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 5) % 14).astype(np.int32)
    S, F = 2, 60000
    x = oracle.synth_input(S, F, channels, seed=seed)
    sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
    info = sp.info()
    whole = sp.process(x)
    assert oracle.peak_rel_error(whole[1], oracle.spatialize_f64(x[1], h, lt, rt)) < TOL
    sp.reset()
    parts, pos = [], 0
    specials = [1, 2, info["hop"] - 1, info["hop"], info["hop"] + 1, info["fft"], info["history"] + 1, 4096, 4097]
    while pos < F:
        n = int(min(F - pos, rng.choice(specials + [int(rng.integers(1, 20000))])))
        parts.append(sp.process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    assert np.max(np.abs(np.concatenate(parts, axis=1) - whole)) <= 3e-6 * np.abs(whole).max()


@pytest.mark.parametrize("channels", [3, 5, 6, 7])
def test_streams_stay_independent_when_a_neighbour_holds_nan(aw, oracle, monkeypatch, channels):
    """Layouts whose frames are not whole float4s are read with 16-byte loads that run into the next frame — after a stream's last
    history frame that is the NEXT stream's history.  A NaN there must not reach this stream (the reference's streams share nothing):
    partitioned kernels (head windows read the history) and the long-window kernels."""
    taps, S, F = 20000, 3, 9000
    h = oracle.synth_hrir(14, taps, seed=5)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    x = oracle.synth_input(S, 2 * F, channels, seed=3)
    bad = x.copy()
    bad[1] = np.nan                                    # stream 1 is poisoned from the first call on
    for lw in ("0", "32"):
        monkeypatch.setenv("AW_LW", lw)
        sp = aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
        sp.process(bad[:, :F])
        y = sp.process(bad[:, F:]) if lw == "0" else sp.process(np.ascontiguousarray(np.concatenate([bad[:, F:]] * 8, axis=1)))[:, :F]
        for s in (0, 2):
            assert np.isfinite(y[s]).all(), (lw, s)
            assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)[F:]) < TOL, (lw, s)


def _write_float_wav(path, tracks, rate):
    """[track][frame] float32 -> RIFF/WAVE, IEEE float, interleaved."""
    import struct
    n, frames = tracks.shape
    data = np.ascontiguousarray(tracks.T).astype("<f4").tobytes()
    block = n * 4
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, n, rate, rate * block, block, 32) + b"data" + struct.pack("<I", len(data)) + data
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


@pytest.mark.parametrize("speakers", [2, 6, 8, ["FL", "FR", "FC", "BL", "BR", "SL", "SR"], ["FC", "LFE"]])
def test_seven_track_hrir_through_preset_activation(aw, oracle, tmp_path, speakers):
    """A11 end to end: a 7-track (left-ear-only) HeSuVi file takes hesuvi7Channel (HRIRManager.swift:355-360) — FL = (0, 1) but
    FR = (1, 0): the right speakers reuse the left-ear tracks mirrored; FC and LFE both read track 2 for BOTH ears; BL / BR = (3, 4) /
    (4, 3), SL / SR = (5, 6) / (6, 5) (VirtualSpeaker.swift:224-250).  Audio through aw_preset_activate against the oracle's own
    assembly of the same file and the float64 truth, with distinct seeded tracks so that any swapped or shared index shows."""
    h = oracle.synth_hrir(7, 1500, seed=77)
    path = str(tmp_path / "seven.wav")
    _write_float_wav(path, h, 48000)
    layout = aw.InputLayout.detect(speakers) if isinstance(speakers, int) else aw.InputLayout(speakers, "custom")
    names = list(layout.channels)
    mgr = aw.HRIRManager()
    sp = mgr.activatePreset(path, 48000.0, layout, n_streams=2)
    assert mgr.isReady
    w = oracle.wav_load(path)
    assert w.channel_count == 7 and np.array_equal(w.audio_data, h)
    tracks, lt, rt = oracle.assemble_tracks(w, names)
    want = {"FL": (0, 1), "FR": (1, 0), "FC": (2, 2), "LFE": (2, 2), "BL": (3, 4), "BR": (4, 3), "SL": (5, 6), "SR": (6, 5)}
    assert [(int(a), int(b)) for a, b in zip(lt, rt)] == [want[s] for s in names]
    x = oracle.synth_input(2, 20011, len(names), seed=5)
    y = sp.process(x)
    for s in range(2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], tracks, lt, rt)) < TOL
    assert oracle.peak_rel_error(y[1, :6000], oracle.spatialize_f32(x[1:2, :6000], tracks, lt, rt)[0]) < TOL
    if "FC" in names and "LFE" in names:
        # FC and LFE share track 2 for both ears: swapping the two input channels changes nothing, and both ears get the same signal from them
        i, j = names.index("FC"), names.index("LFE")
        xs = np.zeros_like(x)
        xs[:, :, i], xs[:, :, j] = x[:, :, j], x[:, :, i]
        sp.reset()
        a = sp.process(xs)
        xo = np.zeros_like(x)
        xo[:, :, i], xo[:, :, j] = x[:, :, i], x[:, :, j]
        sp.reset()
        b = sp.process(xo)
        assert np.allclose(a, b, rtol=0, atol=2e-6 * np.abs(b).max()) and np.allclose(b[..., 0], b[..., 1], rtol=0, atol=2e-6 * np.abs(b).max())
    if "FL" in names and "FR" in names:
        # FR is FL mirrored: the same signal on FR alone gives FL's output with the ears exchanged
        i, j = names.index("FL"), names.index("FR")
        xl, xr = np.zeros_like(x), np.zeros_like(x)
        xl[:, :, i] = x[:, :, 0]
        xr[:, :, j] = x[:, :, 0]
        sp.reset()
        yl = sp.process(xl)
        sp.reset()
        yr = sp.process(xr)
        assert np.allclose(yl[..., 0], yr[..., 1], rtol=0, atol=2e-6 * np.abs(yl).max()) and np.allclose(yl[..., 1], yr[..., 0], rtol=0, atol=2e-6 * np.abs(yl).max())
