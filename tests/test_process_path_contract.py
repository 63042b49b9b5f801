""""creation may block, process must not allocate" — and no environment lookups on a process path.
Source-level part (CPU): the bodies of the process entries contain no getenv.  GPU part: after reserve(), process calls of
up to that size leave the device's free memory unchanged (no hipMalloc / hipFree), on the partitioned and the fused path."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _body(src: str, name: str) -> str:
    m = re.search(r"^[\w\s\*]*\b" + re.escape(name) + r"\(", src, re.M)
    assert m, name
    j = src.index("{", m.end())
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def test_no_environment_lookups_on_process_paths():
    rt = open(os.path.join(ROOT, "airwave_amd", "csrc", "runtime.cpp")).read()
    for fn in ("sp_process_fused", "sp_process_partitioned", "sp_process_longwin", "sp_run_streams", "sp_begin_call", "aw_spatializer_process",
               "aw_spatializer_process_host", "aw_spatializer_process_planar", "aw_realtime_process", "aw_engine_process", "part_plan",
               "part_ensure_scratch", "host_chunk_streams", "host_stage_buffers", "aw_spatializer_reserve", "aw_spatializer_reserve_host"):
        assert "getenv" not in _body(rt, fn), fn
    eq = open(os.path.join(ROOT, "airwave_amd", "csrc", "eq_runtime.cpp")).read()
    assert "getenv" not in eq
    for f in ("kernels.hip", "march_kernels.hip", "eq_kernels.hip"):
        src = open(os.path.join(ROOT, "airwave_amd", "csrc", "device", f)).read()
        for m in re.finditer(r"getenv", src):       # only inside prepare_kernels (context creation)
            start = src.rfind("\nhipError_t ", 0, m.start())
            assert src[start:start + 60].lstrip().startswith("hipError_t prepare_kernels("), (f, src[start:start + 60])


def test_every_status_returning_entry_point_is_an_exception_barrier():
    """include/airwave_hip.h promises "no exceptions": every `aw_status aw_*` definition is a function-try-block whose handler maps what
    was thrown to a status (runtime.hpp: awr::caught), so std::bad_alloc out of a container never crosses the C ABI."""
    n = 0
    for rel in ("runtime.cpp", "eq_runtime.cpp", os.path.join("host", "host_api.cpp"), os.path.join("host", "eq.cpp"), os.path.join("host", "tables.cpp")):
        src = open(os.path.join(ROOT, "airwave_amd", "csrc", rel)).read()
        for m in re.finditer(r"^aw_status (aw_\w+)\(", src, re.M):
            head = src[m.start():src.index("{", m.end())]
            assert head.rstrip().endswith("try"), (rel, m.group(1))
            n += 1
        assert src.count("AW_NOEXCEPT_TAIL") == len(re.findall(r"^aw_status aw_\w+\(", src, re.M)), rel
    assert n >= 60


@pytest.mark.gpu
def test_an_exception_inside_an_entry_point_comes_back_as_a_status():
    """A track set whose size no std::vector can hold: vector::assign throws std::length_error before it reads a byte; the caller gets
    AW_ERR_OUT_OF_MEMORY and a message, the process lives, and the context still works afterwards."""
    import ctypes
    import airwave_amd as aw
    from airwave_amd import _capi
    lib = _capi.load()
    ctx = aw.Context()
    h = ctypes.c_void_p()
    one = np.zeros(4, np.float32)
    st = lib.aw_hrir_create(ctx._h, one.ctypes.data_as(_capi.c_float_p), 2 ** 31 - 1, 2 ** 31 - 1, 48000.0, ctypes.byref(h))
    assert lib.aw_status_string(st) == b"out of memory" and not h.value
    assert lib.aw_last_error_message()
    hr = aw.HRIR(np.ones((2, 8), np.float32), 48000.0, ctx=ctx)
    sp = aw.Spatializer(hr, [0, 1], [1, 0], 1, ctx=ctx)
    assert sp.process(np.ones((1, 64, 2), np.float32)).shape == (1, 64, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("taps,channels", [(20000, 7), (40000, 14), (4320, 8)])
def test_process_does_not_allocate_after_reserve(oracle, taps, channels):
    """reserve(max_frames) sizes every grow-only internal buffer; afterwards calls of any size up to it leave them alone
    (info()["scratch_bytes"] constant) and, once the HIP runtime has loaded each kernel, the device's free memory too."""
    import torch
    import airwave_amd as aw
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    h = oracle.synth_hrir(14, taps, seed=7)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    S, F = 3, 50000
    x = torch.empty((S, F, channels), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, channels)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sizes = (1000, F, 4097, F)
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    partitioned = sp.info()["path"] == 1
    sp.reserve(F)
    cap0 = sp.info()["scratch_bytes"]
    assert (cap0 > 0) == partitioned
    allocs0, copies0 = sp.info()["device_allocs"], sp.info()["sync_copies"]
    for n in sizes:
        sp.process_device(x.data_ptr(), y.data_ptr(), n)
        assert sp.info()["scratch_bytes"] == cap0, n
        # a reserved spatializer performs no hipMalloc and no blocking table upload on its process path (the library counts its own)
        assert (sp.info()["device_allocs"], sp.info()["sync_copies"]) == (allocs0, copies0), n
    torch.cuda.synchronize()
    flat = x.view(-1, channels)                     # a call of n frames reads streams packed with stride n: stream 1 = rows [n, 2n)
    ref = oracle.spatialize_f64(np.concatenate([flat[n:2 * n].cpu().numpy() for n in sizes]), h, lt, rt)
    assert oracle.peak_rel_error(y[1].cpu().numpy(), ref[-F:]) < 1e-5          # and the calls still continue one stream exactly
    # second round, every kernel already loaded by the runtime: the device's free memory does not move at all
    free0 = torch.cuda.mem_get_info()[0]
    for n in sizes:
        sp.process_device(x.data_ptr(), y.data_ptr(), n)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, n
    if partitioned:
        # the scratch is a pool of the CONTEXT: a second spatializer on the same context (a preset change) finds it there ...
        sp2 = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        a0 = sp2.info()["device_allocs"]
        sp2.reserve(F)
        assert sp2.info()["scratch_bytes"] == cap0 and sp2.info()["reserve_scratch_ms"] < 50.0
        sp2.process_device(x.data_ptr(), y.data_ptr(), F)
        torch.cuda.synchronize()
        assert oracle.peak_rel_error(y[1].cpu().numpy(), oracle.spatialize_f64(flat[F:2 * F].cpu().numpy(), h, lt, rt)) < 1e-5
        del a0
        # ... and on a fresh context, without reserve, the same buffers grow with the calls (what reserve() is for)
        ctx3 = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        sp3 = aw.Spatializer(aw.HRIR(h, ctx=ctx3), lt, rt, n_streams=S, ctx=ctx3)
        sp3.process_device(x.data_ptr(), y.data_ptr(), 1000)
        small = sp3.info()["scratch_bytes"]
        sp3.process_device(x.data_ptr(), y.data_ptr(), F)
        assert small < sp3.info()["scratch_bytes"] <= cap0


@pytest.mark.gpu
def test_mixed_rate_batch_uses_the_reference_map_chooser(oracle):
    """HRIRManager.swift:355-360: 7 tracks -> hesuvi7, anything else -> hesuvi14 (an 8-track HRIR then fails the bounds
    check of the 14-channel map instead of silently taking the 7-channel one)."""
    import airwave_amd as aw
    layout = aw.InputLayout.detect(8)
    h8 = oracle.synth_hrir(8, 300, seed=1)
    with pytest.raises(Exception) as ei:
        aw.MixedRateBatch(h8, 48000.0, layout, [48000.0])
    assert "out of range" in str(ei.value) or "Invalid channel mapping" in str(ei.value)
    h7 = oracle.synth_hrir(7, 300, seed=1)
    b = aw.MixedRateBatch(h7, 48000.0, aw.InputLayout.detect(2), [48000.0])
    assert list(b.left_track) == [0, 1] and list(b.right_track) == [1, 0]          # hesuvi7: FL(0,1) FR(1,0)


@pytest.mark.gpu
@pytest.mark.parametrize("wgs", ["3", "12"])
def test_tiny_persistent_grids_still_compute_every_tile(oracle, wgs, monkeypatch):
    """AW_PERSISTENT_WGS below 8 (a documented knob, or a device with fewer than 8 CUs): the persistent kernels deal tiles to 8 XCD
    groups by blockIdx % 8, so the grid is clamped to >= 8 workgroups — otherwise whole groups of tiles would never be computed.
    The knob is read when the CONTEXT is created."""
    import airwave_amd as aw
    monkeypatch.setenv("AW_PERSISTENT_WGS", wgs)
    ctx = aw.Context(0)
    monkeypatch.delenv("AW_PERSISTENT_WGS")
    for taps, channels in ((4320, 8), (4320, 7), (20000, 4), (900, 14)):
        h = oracle.synth_hrir(14, taps, seed=5)
        lt = (np.arange(channels) % 14).astype(np.int32)
        rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
        x = oracle.synth_input(2, 60000, channels, seed=taps)
        y = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=2, ctx=ctx).process(x)
        assert np.isfinite(y).all()
        assert oracle.peak_rel_error(y[1], oracle.spatialize_f64(x[1], h, lt, rt)) < 1e-5, (taps, channels)


@pytest.mark.gpu
@pytest.mark.parametrize("taps", [4320, 20000])
def test_realtime_adapter_allocates_everything_at_creation(oracle, taps):
    """RealtimeAudioProcessor.swift:30-62 allocates pending / block / FIFO buffers in init and nothing in process (:77-119); SURVEY 8b:
    "process must not allocate".  The adapter: host vectors and the device side (kernel scratch + the host entry's staging) are sized by
    aw_realtime_create for the longest device call a callback can cause; across callbacks of every size 1..4096 neither the host
    capacity, nor the device bytes, nor the library's count of device allocations moves — and the output still matches the oracle."""
    import airwave_amd as aw
    ctx = aw.Context(0)
    h = oracle.synth_hrir(14, taps, seed=11)
    hr = aw.HRIR(h, ctx=ctx)
    p = aw.RealtimeAudioProcessor(hr, [(0, 1), (8, 7)], 512, 4096)
    ref = oracle.RealtimeAudioProcessor([(h[0], h[1]), (h[8], h[7])], 512, 4096)
    held = p.info()
    assert held["host_bytes"] >= 4 * (2 * 512 + 2 * 2 * (4096 + 512) + 2 * (4096 + 512)) and (held["device_bytes"] > 0)
    rng = np.random.default_rng(5)
    sizes = [1, 127, 128, 511, 512, 513, 1024, 4095, 4096, 3, 700, 4096, 4096, 1, 2048]
    got, exp = [], []
    for n in sizes:
        l, r = (rng.random(n, dtype=np.float32) - 0.5), (rng.random(n, dtype=np.float32) - 0.5)
        ol, orr = p.process(l, r)
        assert p.info() == held, (n, p.info(), held)
        el, er = ref.process(l, r)
        got += [ol, orr]; exp += [el, er]
    got, exp = np.concatenate(got), np.concatenate(exp)
    assert np.max(np.abs(got - exp)) <= 1e-5 * np.max(np.abs(exp))
    p.reset()
    assert p.info() == held


@pytest.mark.gpu
def test_two_handles_of_one_context_driven_from_two_threads(oracle):
    """"A handle is single-threaded, independent handles are independent"  — also when two handles share a context, whose
    scratch pool they then share: each call's kernels are queued under the context's launch lock, so one call's split / rows / merge (or
    forward / march / inverse) sequence is never interleaved with the other's on the stream.  Two path-1 spatializers (both kernel families
    that use the pool), two threads, forty interleaved calls: every output equals the single-threaded one."""
    import threading
    import torch
    import airwave_amd as aw
    ctx = aw.Context(0)
    h = oracle.synth_hrir(14, 20000, seed=3)
    shapes = [(7, 3, 70000), (2, 5, 9000)]               # (channels, streams, frames): a long-window call and a partitioned one
    sps, xs, refs = [], [], []
    for C, S, F in shapes:
        lt = (np.arange(C) % 14).astype(np.int32)
        rt = ((np.arange(C) + 7) % 14).astype(np.int32)
        x = torch.empty((S, F, C), dtype=torch.float32, device="cuda")
        ctx.synth_fill(x.data_ptr(), S, F, C, seed=C)
        one = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
        outs = []
        for _ in range(20):
            one.process_device(x.data_ptr(), y.data_ptr(), F)
            ctx.synchronize()
            outs.append(y.cpu().numpy().copy())
        refs.append(outs)
        sps.append(aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx))
        xs.append(x)
    got = [[], []]
    errors = []

    def drive(i):
        try:
            C, S, F = shapes[i]
            ys = [torch.empty((S, F, 2), dtype=torch.float32, device="cuda") for _ in range(20)]
            for k in range(20):
                sps[i].process_device(xs[i].data_ptr(), ys[k].data_ptr(), F)        # asynchronous: the two threads' launches interleave
            ctx.synchronize()
            got[i] = [y.cpu().numpy() for y in ys]
        except Exception as e:          # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for k in range(20):
            assert np.array_equal(got[i][k], refs[i][k]), (i, k)


@pytest.mark.gpu
def test_context_scratch_pool_sized_at_start_up(oracle):
    """aw_context_reserve_scratch: a host pays the one large hipMalloc when it creates the context; the spatializers created on it later
    find the pool there (their reserve adds no scratch allocation), share it, and still give the right samples."""
    import torch
    import airwave_amd as aw
    ctx = aw.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    assert ctx.scratch_bytes == 0
    ctx.reserve_scratch(1 << 30)
    assert ctx.scratch_bytes >= 1 << 30
    h = oracle.synth_hrir(14, 20000, seed=9)
    lt, rt = np.array([0, 8, 6], np.int32), np.array([1, 7, 13], np.int32)
    S, F = 2, 120000
    x = torch.empty((S, F, 3), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, 3, seed=4)
    y = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
    pool = ctx.scratch_bytes
    sp.reserve(F)
    assert ctx.scratch_bytes == pool and sp.info()["reserve_scratch_ms"] < 20.0          # nothing to allocate: the pool was there
    ctx.reserve_scratch(1 << 20)                                                           # never shrinks
    assert ctx.scratch_bytes == pool
    sp.process_device(x.data_ptr(), y.data_ptr(), F)
    torch.cuda.synchronize()
    assert oracle.peak_rel_error(y[1].cpu().numpy(), oracle.spatialize_f64(x[1].cpu().numpy(), h, lt, rt)) < 1e-5
