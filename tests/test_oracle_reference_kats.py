"""Pins the CPU oracle against every known-answer test the reference holds for the hot path.

Each test re-expresses one XCTest of the reference (file:line cited) against oracle/airwave_oracle.c.
The same cases run against the HIP path in tests/test_gpu_reference_kats.py.
"""
import numpy as np
import pytest

BLOCK = 8


def make_engine(oracle):
    # ConvolutionEngineTests.swift:7-10  impulse [1,0,0,0,0,0,0,0], blockSize 8
    return oracle.ConvolutionEngine([1, 0, 0, 0, 0, 0, 0, 0], block_size=BLOCK)


def test_impulse_preserves_sample_order(oracle):
    # ConvolutionEngineTests.swift:12-20
    e = make_engine(oracle)
    x = np.array([0.25, -0.5, 1, 0.75, -1, 0.125, 0.5, -0.25], dtype=np.float32)
    y = e.process(x)
    assert np.all(np.abs(y - x) < 1e-4)


def test_reset_clears_overlap_and_frequency_history(oracle):
    # ConvolutionEngineTests.swift:22-34
    e = make_engine(oracle)
    x = np.zeros(BLOCK, dtype=np.float32)
    x[BLOCK - 1] = 1
    e.process(x)
    e.reset()
    y = e.process(np.zeros(BLOCK, dtype=np.float32))
    assert np.all(np.abs(y) < 1e-4)


def test_multiple_blocks_remain_finite(oracle):
    # ConvolutionEngineTests.swift:36-46
    e = make_engine(oracle)
    x = (np.arange(BLOCK, dtype=np.float32) / np.float32(7)).astype(np.float32)
    for _ in range(64):
        y = e.process(x)
        assert np.all(np.isfinite(y))
        x = (-x * np.float32(0.97) + np.float32(0.01)).astype(np.float32)


def test_identical_input_after_reset_produces_identical_output(oracle):
    # ConvolutionEngineTests.swift:48-59
    e = make_engine(oracle)
    x = np.arange(-0.75, 0.75 + 1e-9, 0.2, dtype=np.float64)[:BLOCK].astype(np.float32)
    first = e.process(x)
    e.reset()
    second = e.process(x)
    assert np.all(np.abs(first - second) < 1e-4)


def test_engine_rejects_non_block_sized_input(oracle):
    # ConvolutionEngine.swift:370-373: wrapper silently ignores count != blockSize
    e = make_engine(oracle)
    with pytest.raises(AssertionError):
        e.process(np.zeros(BLOCK + 1, dtype=np.float32))


# ---- RealtimeAudioProcessorTests.swift -------------------------------------------------------
RT_BLOCK, RT_MAX = 512, 4096


def make_processor(oracle, renderer_count=2):
    # RealtimeAudioProcessorTests.swift:8-28: one-tap HRIRs with gains 1 and 2
    rend = [([float(i + 1)], [float(i + 1)]) for i in range(renderer_count)]
    return oracle.RealtimeAudioProcessor(rend, block_size=RT_BLOCK, max_frames_per_callback=RT_MAX)


def run(p, size, left_value=1.0, right_value=2.0):
    l = np.full(size, left_value, dtype=np.float32)
    r = np.full(size, right_value, dtype=np.float32)
    return p.process(l, r)


def test_all_required_callback_sizes_write_finite_output(oracle):
    # RealtimeAudioProcessorTests.swift:59-66
    for size in [1, 64, 128, 256, 511, 512, 513, 768, 1024, 4096]:
        l, r = run(make_processor(oracle), size)
        assert np.all(np.isfinite(l)) and np.all(np.isfinite(r)), size


def test_mixed_callback_sequence_preserves_order_after_adapter_latency(oracle):
    # RealtimeAudioProcessorTests.swift:68-78
    p = make_processor(oracle, renderer_count=1)
    out = np.concatenate([run(p, size)[0] for size in [128, 128, 128, 128, 513, 768, 1024, 4096]])
    assert out.size == 6913
    assert np.all(out[:384] == 0)
    assert np.all(np.abs(out[384:] - 1) < 1e-4)


def test_two_renderers_sum_left_and_right_inputs(oracle):
    # processPendingBlock (RealtimeAudioProcessor.swift:141-164): renderer 0 fed from L, 1 from R,
    # gains 1 and 2 -> both ears = 1*L + 2*R = 1 + 4 = 5 after the first block.
    p = make_processor(oracle, renderer_count=2)
    l, r = run(p, 1024)
    assert np.all(np.abs(l[:512] - 5) < 1e-4) and np.all(np.abs(r[:512] - 5) < 1e-4)


def test_reset_clears_pending_input_and_queued_output(oracle):
    # RealtimeAudioProcessorTests.swift:80-88
    p = make_processor(oracle, renderer_count=1)
    run(p, 512)
    p.reset()
    l, r = run(p, 1)
    assert l.tolist() == [0] and r.tolist() == [0]


def test_underflow_silence_and_mono_duplication(oracle):
    # RealtimeAudioProcessorTests.swift:90-97
    p = make_processor(oracle, renderer_count=1)
    ul, ur = run(p, 3, 0.5, 0.5)
    assert ul.tolist() == [0, 0, 0] and ur.tolist() == ul.tolist()
    l, r = run(p, 512, 0.5, 0.5)
    assert np.array_equal(l, r)


def test_mono_input_right_nil_duplicates_left(oracle):
    # RealtimeAudioProcessor.swift:95-107 (inputRight == nil), canary test :99-126 uses it
    p = make_processor(oracle, renderer_count=2)
    x = np.full(512, 0.25, dtype=np.float32)
    l, r = p.process(x, None)
    assert np.all(np.abs(l - 0.75) < 1e-4) and np.all(np.abs(r - 0.75) < 1e-4)


def test_more_than_max_frames_is_rejected(oracle):
    # precondition(frameCount <= maxFramesPerCallback)  RealtimeAudioProcessor.swift:85
    p = make_processor(oracle, renderer_count=1)
    with pytest.raises(ValueError):
        run(p, RT_MAX + 1)
