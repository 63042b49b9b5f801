"""The host entry of the batch spatializer (aw_spatializer_process_host): a multi-stream batch crosses PCIe in chunks of streams, double
buffered on three HIP streams.  The reference's callers own host buffers (AudioPipeline.swift:3-11); its semantics per stream are still
ConvolutionEngine.process summed over speakers (ConvolutionEngine.swift:232-367, RealtimeAudioProcessor.swift:141-172).  Chunking must be
invisible: the samples equal the device entry's bit for bit (same kernels per stream), state carries across calls, ragged last chunks and
both kernel families (fused tiles, long-window kernels) are covered, on page-locked and pageable buffers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_reference(aw, torch, ctx, h, lt, rt, x_host, splits):
    S, F, C = x_host.shape
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    outs, at = [], 0
    for n in splits:
        x = torch.from_numpy(np.ascontiguousarray(x_host[:, at:at + n])).cuda()
        y = torch.empty((S, n, 2), dtype=torch.float32, device="cuda")
        sp.process_device(x.data_ptr(), y.data_ptr(), n)
        torch.cuda.synchronize()
        outs.append(y.cpu().numpy())
        at += n
    return np.concatenate(outs, axis=1)


@pytest.mark.parametrize("taps,channels,streams,frames", [(4320, 8, 37, 200000), (32768, 7, 21, 300000), (600, 2, 64, 900000)])
@pytest.mark.parametrize("pinned", [True, False])
def test_chunked_host_entry_equals_the_device_entry(oracle, taps, channels, streams, frames, pinned, monkeypatch):
    import torch
    import airwave_amd as aw
    monkeypatch.setenv("AW_HOST_CHUNK_MB", "8")                        # (read once, at context creation) small chunks: many of them, a ragged last one
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    monkeypatch.delenv("AW_HOST_CHUNK_MB")
    h = oracle.synth_hrir(14, taps, seed=3)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    xd = torch.empty((streams, frames, channels), dtype=torch.float32, device="cuda")
    ctx.synth_fill(xd.data_ptr(), streams, frames, channels, seed=99)
    x_all = xd.cpu().numpy()
    del xd
    splits = [frames // 2, frames - frames // 2]                       # two calls: the convolution tail crosses the call boundary
    ref = _device_reference(aw, torch, ctx, h, lt, rt, x_all, splits)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=streams, ctx=ctx)
    sp.reserve_host(max(splits))
    allocs = sp.info()["device_allocs"]
    got, at = [], 0
    for n in splits:
        if pinned:
            x, y = ctx.pinned_empty((streams, n, channels)), ctx.pinned_empty((streams, n, 2))
            x[...] = x_all[:, at:at + n]
            allocs += 2
        else:
            x, y = np.ascontiguousarray(x_all[:, at:at + n]), np.full((streams, n, 2), np.nan, np.float32)
        sp.process_host_into(x, y)
        assert sp.info()["device_allocs"] == allocs                    # reserve_host sized the staging: the entry allocates nothing
        got.append(np.array(y))
        at += n
    got = np.concatenate(got, axis=1)
    chunk = sp.info()["host_chunk_streams"]
    assert 0 < chunk < streams, chunk                                  # really chunked (37 and 21 streams: with a ragged last chunk)
    assert np.array_equal(got, ref)                                    # chunking is invisible: the same kernels saw every stream
    for s in (0, streams - 1):
        assert oracle.peak_rel_error(got[s, :20000], oracle.spatialize_f64(x_all[s, :20000], h, lt, rt)) < 1e-5
        n0 = frames - 3000 - (taps - 1)
        assert oracle.peak_rel_error(got[s, -3000:], oracle.spatialize_f64(x_all[s, n0:], h, lt, rt)[-3000:]) < 1e-5


def test_unreserved_host_entry_and_small_batches_stay_in_one_piece(oracle):
    """Plug-in shaped calls (one stream, a few frames: aw_engine_* / aw_realtime_* sit on this entry) are not chunked; an unreserved
    multi-stream call sizes its staging itself and gives the same samples."""
    import torch
    import airwave_amd as aw
    ctx = aw.Context(0)
    h = oracle.synth_hrir(14, 4320, seed=4)
    lt, rt = np.array([0, 8], np.int32), np.array([1, 7], np.int32)
    x1 = oracle.synth_input(1, 5000, 2)
    sp1 = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=1, ctx=ctx)
    y1 = sp1.process(x1)
    assert sp1.info()["host_chunk_streams"] == 0
    assert oracle.peak_rel_error(y1[0], oracle.spatialize_f64(x1[0], h, lt, rt)) < 1e-5
    S, F = 48, 700000                                                   # 269 MB of input: three chunks of the default 96 MB
    xd = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    ctx.synth_fill(xd.data_ptr(), S, F, 2, seed=7)
    torch.cuda.synchronize()
    x = xd.cpu().numpy()
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    y = sp.process(x)                                                   # no reserve at all
    assert sp.info()["host_chunk_streams"] > 0
    for s in (0, 17, S - 1):
        assert oracle.peak_rel_error(y[s, :30000], oracle.spatialize_f64(x[s, :30000], h, lt, rt)) < 1e-5


def test_host_entry_at_cfg2_size_against_the_oracle(oracle, golden_dir):
    """BASELINE cfg 2 through the host entry at full size: 128 streams x 10 s of 7.1 -> RoomSH1.0, page-locked buffers in and out."""
    import os
    import torch
    import airwave_amd as aw
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    tracks, lt, rt = oracle.assemble_tracks(wav, oracle.layout_detect(8))
    S, F, C = 128, 480000, 8
    xd = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
    ctx.synth_fill(xd.data_ptr(), S, F, C, seed=0xA17AE)
    x, y = ctx.pinned_empty((S, F, C)), ctx.pinned_empty((S, F, 2))
    ctx.d2h(x, xd.data_ptr())
    yd = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    spd = aw.Spatializer(aw.HRIR(tracks, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    spd.process_device(xd.data_ptr(), yd.data_ptr(), F)
    torch.cuda.synchronize()
    sp = aw.Spatializer(aw.HRIR(tracks, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    sp.reserve_host(F)
    y[...] = np.nan
    sp.process_host_into(x, y)
    assert sp.info()["host_chunk_streams"] > 0
    assert np.array_equal(np.asarray(y), yd.cpu().numpy())
    L = tracks.shape[1]
    for s in (0, 63, 127):
        assert oracle.peak_rel_error(y[s, :8192], oracle.spatialize_f64(x[s, :8192], tracks, lt, rt)) < 1e-5
        assert oracle.peak_rel_error(y[s, -4096:], oracle.spatialize_f64(x[s, F - 4096 - (L - 1):], tracks, lt, rt)[-4096:]) < 1e-5
