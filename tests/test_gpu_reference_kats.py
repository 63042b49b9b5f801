"""The reference's own known-answer tests for the path, run through the HIP engine via the C ABI.
Mirrors AirwaveTests/ConvolutionEngineTests.swift and AirwaveTests/RealtimeAudioProcessorTests.swift
test for test (the CPU oracle runs the same cases in tests/test_oracle_reference_kats.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK = 8


@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


def make_engine(aw):
    return aw.ConvolutionEngine([1, 0, 0, 0, 0, 0, 0, 0], blockSize=BLOCK)      # ConvolutionEngineTests.swift:7-10


def test_impulse_preserves_sample_order(aw):
    e = make_engine(aw)
    x = np.array([0.25, -0.5, 1, 0.75, -1, 0.125, 0.5, -0.25], dtype=np.float32)
    y = e.process(x)
    assert np.all(np.abs(y - x) < 1e-4)                                          # :12-20


def test_reset_clears_overlap_and_frequency_history(aw):
    e = make_engine(aw)
    x = np.zeros(BLOCK, dtype=np.float32)
    x[BLOCK - 1] = 1
    e.process(x)
    e.reset()
    y = e.process(np.zeros(BLOCK, dtype=np.float32))
    assert np.all(np.abs(y) < 1e-4)                                              # :22-34


def test_multiple_blocks_remain_finite(aw):
    e = make_engine(aw)
    x = (np.arange(BLOCK, dtype=np.float32) / np.float32(7)).astype(np.float32)
    for _ in range(64):
        y = e.process(x)
        assert np.all(np.isfinite(y))                                            # :36-46
        x = (-x * np.float32(0.97) + np.float32(0.01)).astype(np.float32)


def test_identical_input_after_reset_produces_identical_output(aw):
    e = make_engine(aw)
    x = np.arange(-0.75, 0.75 + 1e-9, 0.2)[:BLOCK].astype(np.float32)
    first = e.process(x)
    e.reset()
    second = e.process(x)
    assert np.all(np.abs(first - second) < 1e-4)                                 # :48-59


def test_wrong_frame_count_is_ignored(aw):
    e = make_engine(aw)                                                          # ConvolutionEngine.swift:370-373
    assert e.process(np.zeros(BLOCK, dtype=np.float32), frameCount=BLOCK - 1) is None


def test_engine_matches_oracle_blockwise(aw, oracle):
    rng = np.random.default_rng(11)
    h = rng.standard_normal(1300).astype(np.float32)
    ge, oe = aw.ConvolutionEngine(h, 512), oracle.ConvolutionEngine(h, 512)
    acc_g, acc_o = np.full(512, 2.0, np.float32), np.full(512, 2.0, np.float32)
    for i in range(6):
        x = rng.uniform(-0.5, 0.5, 512).astype(np.float32)
        if i % 2:
            ge.processAndAccumulate(x, acc_g)                                    # ConvolutionEngine.swift:388-394
            oe.process_and_accumulate(x, acc_o)
            assert np.max(np.abs(acc_g - acc_o)) < 1e-5 * max(1.0, np.abs(acc_o).max())
        else:
            yg, yo = ge.process(x), oe.process(x)
            assert np.max(np.abs(yg - yo)) < 1e-5 * np.abs(yo).max()


# ---- RealtimeAudioProcessorTests.swift -------------------------------------------------------------
RT_BLOCK, RT_MAX = 512, 4096


def make_processor(aw, renderer_count=2):
    # one-tap HRIRs with gains 1 and 2 (RealtimeAudioProcessorTests.swift:8-28): tracks [1.0], [2.0]
    hrir = aw.HRIR(np.array([[1.0], [2.0]], dtype=np.float32))
    return aw.RealtimeAudioProcessor(hrir, [(i, i) for i in range(renderer_count)], blockSize=RT_BLOCK,
                                     maxFramesPerCallback=RT_MAX)


def run(p, size, left_value=1.0, right_value=2.0):
    return p.process(np.full(size, left_value, np.float32), np.full(size, right_value, np.float32))


def test_all_required_callback_sizes_write_finite_output(aw):
    for size in [1, 64, 128, 256, 511, 512, 513, 768, 1024, 4096]:              # :59-66
        l, r = run(make_processor(aw), size)
        assert np.all(np.isfinite(l)) and np.all(np.isfinite(r)), size


def test_mixed_callback_sequence_preserves_order_after_adapter_latency(aw, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "kat_reference.npz"))
    p = make_processor(aw, renderer_count=1)                                     # :68-78
    out = np.concatenate([run(p, int(size))[0] for size in g["adapter_sizes"]])
    assert out.size == 6913
    assert np.all(out[:384] == 0)
    assert np.all(np.abs(out[384:] - 1) < 1e-4)
    assert np.all(np.abs(out - g["adapter_expected"]) < 1e-4)


def test_two_renderers_sum_left_and_right_inputs(aw):
    l, r = run(make_processor(aw, 2), 1024)                                      # RealtimeAudioProcessor.swift:141-164
    assert np.all(np.abs(l[:512] - 5) < 1e-4) and np.all(np.abs(r[:512] - 5) < 1e-4)


def test_more_than_two_renderers_are_capped(aw):
    hrir = aw.HRIR(np.array([[1.0], [2.0], [100.0]], dtype=np.float32))
    p = aw.RealtimeAudioProcessor(hrir, [(0, 0), (1, 1), (2, 2)], RT_BLOCK, RT_MAX)   # min(renderers.count, 2) :145
    l, _ = p.process(np.full(512, 1, np.float32), np.full(512, 2, np.float32))
    assert np.all(np.abs(l - 5) < 1e-4)


def test_reset_clears_pending_input_and_queued_output(aw):
    p = make_processor(aw, 1)                                                    # :80-88
    run(p, 512)
    p.reset()
    l, r = run(p, 1)
    assert l.tolist() == [0] and r.tolist() == [0]


def test_underflow_silence_and_mono_duplication(aw):
    p = make_processor(aw, 1)                                                    # :90-97
    ul, ur = run(p, 3, 0.5, 0.5)
    assert ul.tolist() == [0, 0, 0] and ur.tolist() == ul.tolist()
    l, r = run(p, 512, 0.5, 0.5)
    # The reference asserts bitwise L == R here because its two ears are two identical engines.
    # The HIP engine computes both ears as the real/imaginary parts of ONE complex transform, so
    # equal-by-construction ears agree to rounding (1e-6 of peak), not bitwise (DESIGN.md "Deviations").
    assert np.max(np.abs(l - r)) <= 1e-6 * np.max(np.abs(l))
    assert np.all(np.abs(l - 0.5) < 1e-4)


def test_canaries_remain_unchanged(aw):
    p = make_processor(aw, 1)                                                    # :99-126
    size, canary = 4096, np.float32(12345)
    inp = np.zeros(size + 2, dtype=np.float32)
    out = np.full(size + 2, canary, dtype=np.float32)
    p.process(inp[1:size + 1], None, out[1:size + 1], out[1:size + 1])
    assert inp[0] == 0 and inp[size + 1] == 0 and out[0] == canary and out[size + 1] == canary


def test_frame_count_above_maximum_is_rejected(aw):
    p = make_processor(aw, 1)                                                    # precondition :85
    with pytest.raises(aw.AirwaveError):
        run(p, RT_MAX + 1)


def test_realtime_matches_oracle_on_real_hrir(aw, oracle, golden_dir):
    import os
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav"))
    hrir = aw.HRIR(wav.audio_data)
    gp = aw.RealtimeAudioProcessor(hrir, [(0, 1), (8, 7)], 512, 4096)
    op = oracle.RealtimeAudioProcessor([(wav.audio_data[0], wav.audio_data[1]), (wav.audio_data[8], wav.audio_data[7])], 512, 4096)
    rng = np.random.default_rng(2)
    gl, ol = [], []
    for size in [128, 700, 33, 4096, 512, 1]:
        l = rng.uniform(-0.5, 0.5, size).astype(np.float32)
        r = rng.uniform(-0.5, 0.5, size).astype(np.float32)
        a, b = gp.process(l, r), op.process(l, r)
        gl.append(np.stack(a, 1)); ol.append(np.stack(b, 1))
    g, o = np.concatenate(gl), np.concatenate(ol)
    assert oracle.peak_rel_error(g, o) < 1e-5


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_realtime_random_callback_sequences_match_oracle(aw, oracle, golden_dir, seed):
    """Random callback sizes (1..4096, with mono callbacks and a reset in the middle) through the pending/FIFO adapter:
    every output sample against the oracle's restatement of RealtimeAudioProcessor (:77-190)."""
    import os
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    hrir = aw.HRIR(wav.audio_data)
    gp = aw.RealtimeAudioProcessor(hrir, [(0, 1), (8, 7)], 512, 4096)
    op = oracle.RealtimeAudioProcessor([(wav.audio_data[0], wav.audio_data[1]), (wav.audio_data[8], wav.audio_data[7])], 512, 4096)
    rng = np.random.default_rng(seed)
    g, o = [], []
    for i in range(14):
        size = int(rng.choice([1, 7, 64, 511, 512, 513, 1000, 2048, 4095, 4096, int(rng.integers(1, 4097))]))
        l = rng.uniform(-0.5, 0.5, size).astype(np.float32)
        r = None if rng.random() < 0.2 else rng.uniform(-0.5, 0.5, size).astype(np.float32)
        if i == 8:
            gp.reset(); op.reset()
        a, b = gp.process(l, r), op.process(l, r)
        g.append(np.stack(a, 1)); o.append(np.stack(b, 1))
    g, o = np.concatenate(g), np.concatenate(o)
    assert oracle.peak_rel_error(g, o) < 1e-5


def test_ten_seconds_of_stereo_input_across_performance_callback_sizes(aw):
    """RealtimeAudioProcessorTests.swift:128-166 as written there: ten seconds of 0.25 on both inputs through one renderer, in 128-,
    512- and 1024-frame callbacks; every callback's output stays finite.  (Round 5: a callback costs ~25 us, so the reference's full
    ten seconds fit the suite — rounds 1-4 ran one second.)  Beyond the reference's assertions: with the one-tap gain-1 HRIR the
    steady-state output IS the input, so the last callback is 0.25 on the left ear to float32 rounding."""
    for size in (128, 512, 1024):
        p = make_processor(aw, renderer_count=1)
        x = np.full(size, 0.25, np.float32)
        frames = 0
        while frames < 48000 * 10:
            l, r = p.process(x, x)
            frames += size
            assert np.all(np.isfinite(l)) and np.all(np.isfinite(r))
        assert frames >= 48000 * 10
        assert np.max(np.abs(l - 0.25)) < 1e-6, size
