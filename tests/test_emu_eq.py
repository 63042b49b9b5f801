"""CPU thread emulation of the EQ cascade kernel (the code hipcc compiles, device/eq_cascade.hpp) against the
oracle's sequential recurrence: span boundaries, partial spans, tails, carried state, poles near the unit circle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

FILTERS = [(1, 105.0, -2.8, 0.7), (0, 65.3, 1.0, 1.68), (0, 1000.0, 6.0, 0.707), (2, 10000.0, -5.2, 0.7), (0, 20.0, 3.0, 4.0)]


def _oracle_run(oracle, fs, preamp, filters, calls):
    d = oracle.EqualizerDefinition(preamp, [oracle.EqualizerFilter(1, None, True, t, f, g, q) for t, f, g, q in filters])
    outs = [np.empty_like(c) for c in calls]
    for s in range(calls[0].shape[0]):
        st = oracle.eq_prepare(d, fs)
        for c, o in zip(calls, outs):
            o[s, :, 0], o[s, :, 1] = st.process(c[s, :, 0], c[s, :, 1])
    return outs


@pytest.mark.parametrize("ear_split", [False, True])
@pytest.mark.parametrize("frames", [5, 32, 63, 8192, 8192 + 32 * 3 + 7, 2 * 8192 + 32])
def test_cascade_matches_sequential_recurrence(oracle, frames, ear_split):
    rng = np.random.default_rng(frames)
    x = rng.uniform(-0.5, 0.5, (2, frames, 2)).astype(np.float32)
    x2 = rng.uniform(-0.5, 0.5, (2, 333, 2)).astype(np.float32)
    y, z = emu.eq_process(x, 48000.0, -2.56, FILTERS, ear_split=ear_split)
    y2, _ = emu.eq_process(x2, 48000.0, -2.56, FILTERS, z, ear_split=ear_split)   # the stream continues: state carried
    e, e2 = _oracle_run(oracle, 48000.0, -2.56, FILTERS, [x, x2])
    # Float64 reassociation only: at most 1 ulp of the Float32 output
    assert np.max(np.abs(y - e)) <= 6e-8 and np.max(np.abs(y2 - e2)) <= 6e-8
    if frames < 32:
        assert np.array_equal(y, e)                                 # the sequential kernel is the recurrence itself


def test_unity_and_many_filters(oracle):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (1, 1000, 2)).astype(np.float32)
    y, _ = emu.eq_process(x, 44100.0, 0.0, [])
    assert np.array_equal(y, x)
    many = [(i % 3, 100.0 + 300.0 * i, ((i % 5) - 2) * 1.5, 0.5 + 0.1 * i) for i in range(64)]
    y, _ = emu.eq_process(x, 96000.0, -6.0, many)
    (e,) = _oracle_run(oracle, 96000.0, -6.0, many, [x])
    assert np.max(np.abs(y - e)) <= 1e-6 * np.max(np.abs(e))
