"""The overlap-add form of the fused tile (airwave_amd/csrc/device/tile_ola.hpp) through the C ABI, against the float64 truth and the
float32 port of the reference algorithm.  Contexts created with AW_OLA_MIN_BLOCKS=0 run it on calls of every size (the policy keeps
short calls on the overlap-save tile); one test runs the shipped policy on a batch large enough to take it by itself."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


@pytest.fixture()
def forced(aw, monkeypatch):
    """A context whose path-0 spatializers run the overlap-add tile on every call (knobs are read once, at context creation)."""
    monkeypatch.setenv("AW_OLA_MIN_BLOCKS", "0")
    monkeypatch.setenv("AW_OLA", "1")
    return aw.Context(0)


def _maps(channels):
    return (np.arange(channels) % 14).astype(np.int32), ((np.arange(channels) * 3 + 7) % 14).astype(np.int32)


@pytest.mark.parametrize("channels", [4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_every_layout_of_the_overlap_add_tile_matches_truth_and_port(aw, oracle, golden_dir, forced, channels):
    """RoomSH1.0 (4320 taps: blocks of 7 x 512 frames), 5 streams of 30 001 frames: 9 blocks per stream, the last one ragged; 45 blocks on
    45 workgroups, so every run starts inside a stream or at its start and rebuilds its carry."""
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    lt, rt = _maps(channels)
    sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=forced), lt, rt, n_streams=5, ctx=forced)
    assert sp.info()["overlap_add_rows_policy"] == 7
    x = oracle.synth_input(5, 30001, channels, seed=channels)
    y = sp.process(x)
    assert sp.info()["overlap_add_rows"] == 7 and sp.info()["path"] == 0 and sp.info()["fft"] == 8192
    assert not np.isnan(y).any()
    yo = oracle.spatialize_f32(x[:, :6000], w.audio_data, lt, rt)
    for s in range(5):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], w.audio_data, lt, rt)) < TOL
        assert oracle.peak_rel_error(y[s, :6000], yo[s]) < TOL


@pytest.mark.parametrize("channels,taps,rows", [(8, 3969, 8), (14, 3969, 8), (7, 4097, 8), (8, 4098, 7), (14, 4609, 7), (8, 4610, 6), (14, 5121, 6),
                                                (4, 5122, 0), (16, 5633, 0), (12, 2000, 8), (9, 4320, 7), (13, 3969, 8), (15, 5121, 6), (5, 4098, 0), (3, 4320, 0), (2, 4320, 0)])
def test_block_length_follows_the_hrir_length(aw, oracle, forced, channels, taps, rows):
    """H = 8 rows of 512 frames up to 4097 taps, 7 up to 4609, 6 up to 5121; longer HRIRs and layouts without a kernel
    (mono, stereo, 3 and 5 channels: their 16384-frame tiles are faster) stay on the overlap-save tiles.  Each at the edge of its range."""
    h = oracle.synth_hrir(14, taps, seed=taps)
    lt, rt = _maps(channels)
    sp = aw.Spatializer(aw.HRIR(h, ctx=forced), lt, rt, n_streams=2, ctx=forced)
    x = oracle.synth_input(2, 20011, channels, seed=3)
    y = sp.process(x)
    assert sp.info()["overlap_add_rows"] == rows, sp.info()
    for s in range(2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], h, lt, rt)) < TOL


@pytest.mark.parametrize("channels", [8, 14])
def test_state_carries_across_ragged_calls_and_across_tiles(aw, oracle, golden_dir, monkeypatch, channels):
    """The state between calls is the input history whatever tile ran: a stream cut into calls of 9000 / 1 / 100 / 20 000 / 511 / 7000
    frames equals the one-call result — on a context that runs the overlap-add tile on every call, and on one that alternates: calls
    below 2 blocks per workgroup (of 8) take the overlap-save tile there, the others the overlap-add tile, on the same history buffer."""
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    lt, rt = _maps(channels)
    cuts = [9000, 1, 100, 20000, 511, 7000]
    x = oracle.synth_input(3, sum(cuts), channels, seed=17)
    ref = [oracle.spatialize_f64(x[s], w.audio_data, lt, rt) for s in range(3)]
    for min_blocks, wgs in (("0", "256"), ("2", "8")):
        monkeypatch.setenv("AW_OLA", "1")
        monkeypatch.setenv("AW_OLA_MIN_BLOCKS", min_blocks)
        monkeypatch.setenv("AW_PERSISTENT_WGS", wgs)           # 8 workgroups: 3 streams x 6 blocks (20 000 frames) >= 2 x 8, the shorter calls are not
        ctx = aw.Context(0)
        sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=3, ctx=ctx)
        used, pos, out = [], 0, []
        for n in cuts:
            out.append(sp.process(x[:, pos:pos + n]))
            used.append(sp.info()["overlap_add_rows"])
            pos += n
        y = np.concatenate(out, axis=1)
        assert used == ([7] * 6 if min_blocks == "0" else [0, 0, 0, 7, 0, 0]), used
        for s in range(3):
            assert oracle.peak_rel_error(y[s], ref[s]) < TOL
        sp.reset()                                                 # reset() = zero history: the first call again, bit for bit
        assert np.array_equal(sp.process(x[:, :9000]), out[0])


def test_result_does_not_depend_on_the_number_of_workgroups(aw, oracle, golden_dir, monkeypatch):
    """Runs rebuild their carry from the input alone (two warm-up blocks), so where the launch cuts the streams changes no bit:
    8, 100 and 256 persistent workgroups give identical output."""
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    lt, rt = _maps(14)
    x = oracle.synth_input(4, 50000, 14, seed=23)
    outs = []
    for wgs in ("8", "100", "256"):
        monkeypatch.setenv("AW_OLA", "1")
        monkeypatch.setenv("AW_OLA_MIN_BLOCKS", "0")
        monkeypatch.setenv("AW_PERSISTENT_WGS", wgs)
        ctx = aw.Context(0)
        sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=4, ctx=ctx)
        outs.append(sp.process(x))
        assert sp.info()["overlap_add_rows"] == 7
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert oracle.peak_rel_error(outs[0][3], oracle.spatialize_f64(x[3], w.audio_data, lt, rt)) < TOL


def test_streams_stay_independent_when_a_neighbour_holds_nan(aw, oracle, golden_dir, forced):
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    lt, rt = _maps(14)
    sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=forced), lt, rt, n_streams=3, ctx=forced)
    x = oracle.synth_input(3, 12000, 14, seed=4)
    x[1, 11000:] = np.nan                                          # also the history the second call's warm-up blocks read
    sp.process(x)
    x2 = oracle.synth_input(3, 9000, 14, seed=5)
    y2 = sp.process(x2)
    for s in (0, 2):
        full = np.concatenate([x[s], x2[s]])
        assert oracle.peak_rel_error(y2[s], oracle.spatialize_f64(full, w.audio_data, lt, rt)[12000:]) < TOL
    assert np.isnan(y2[1]).any()


def test_shipped_policy_takes_the_tile_on_a_full_batch(aw, oracle, golden_dir):
    """No knob: cfg 2 with 14-channel input (north_star's literal case) at full size — 128 streams x 10 s — runs the overlap-add tile
    (134 blocks per stream, 67 per workgroup), a 1-stream call of the same spatializer shape does not (too few blocks per workgroup)."""
    import torch
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    w = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    cmap = aw.HRIRChannelMap.parseHeSuViFormat(open(os.path.join(golden_dir, "hesuvi14_custom_map.txt")).read())
    layout = aw.InputLayout.detect(14)
    lt, rt = cmap.resolve(layout, 14)
    S, F, C = 128, 480000, 14
    sp = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, C, seed=oracle.SYNTH_SEED)
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    i = sp.info()
    assert i["overlap_add_rows"] == 7 and i["long_window_rows"] == 0, i
    assert torch.isfinite(y).all()
    for s in (0, 77, 127):
        xs = x[s, :9000].cpu().numpy()
        assert oracle.peak_rel_error(y[s, :9000].cpu().numpy(), oracle.spatialize_f64(xs, w.audio_data, lt, rt)) < TOL
        tail_in = x[s, F - 3000 - 4319:].cpu().numpy()
        assert oracle.peak_rel_error(y[s, F - 3000:].cpu().numpy(), oracle.spatialize_f64(tail_in, w.audio_data, lt, rt)[4319:]) < TOL
    # a second call continues the streams (history) — checked at its head
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    xin = np.concatenate([x[5, F - 4319:].cpu().numpy(), x[5, :4096].cpu().numpy()])
    assert oracle.peak_rel_error(y[5, :4096].cpu().numpy(), oracle.spatialize_f64(xin, w.audio_data, lt, rt)[4319:]) < TOL
    one = aw.Spatializer(aw.HRIR(w.audio_data, ctx=ctx), lt, rt, n_streams=1, ctx=ctx)
    one.process(oracle.synth_input(1, 48000, 14))
    assert one.info()["overlap_add_rows"] == 0 and one.info()["overlap_add_rows_policy"] == 7
