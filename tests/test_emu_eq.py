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


def test_table_powers_are_exact_to_the_last_bit_or_two():
    """The kernel's tables are powers of the zero-input state matrix M up to M^(64 chunk).  For a low-frequency section M is
    nearly defective (a double pole next to z = 1) and powers formed in double lose n eps / angle^2 of their entries: 1e-7 at
    n = 2048, which reached the Float32 output at the wave boundaries (4.6 ulp of the peak on a 64-filter script of
    tools/fuzz_eq.py).  The host forms them in double-double; checked against exact rational arithmetic."""
    from fractions import Fraction
    filters = [(0, 20.0, 6.0, 6.0), (1, 23.0, 5.4, 2.49), (2, 31.0, -5.9, 4.7), (0, 45.0, 11.7, 1.7), (1, 400.0, -3.0, 0.7), (2, 9000.0, 4.0, 1.0)]
    tab, plane, chunk = emu.eq_tables(96000.0, 0.0, filters)
    for k in range(len(filters)):
        a1, a2 = Fraction(float(tab[k, 3])), Fraction(float(tab[k, 4]))
        M = ((-a1, Fraction(1)), (-a2, Fraction(0)))

        def mul(a, b):
            return ((a[0][0] * b[0][0] + a[0][1] * b[1][0], a[0][0] * b[0][1] + a[0][1] * b[1][1]),
                    (a[1][0] * b[0][0] + a[1][1] * b[1][0], a[1][0] * b[0][1] + a[1][1] * b[1][1]))

        def close(got, exact):
            ex = np.array([[float(v) for v in row] for row in exact]).reshape(-1)
            scale = max(np.max(np.abs(ex)), 1e-300)
            assert np.max(np.abs(np.asarray(got).reshape(-1) - ex)) <= 4e-16 * scale, (k, np.asarray(got), ex)

        Mj = ((Fraction(1), Fraction(0)), (Fraction(0), Fraction(1)))
        for j in range(chunk):
            ex = np.array([float(Mj[0][0]), float(Mj[0][1])])
            assert np.max(np.abs(tab[k, 5 + 2 * j:7 + 2 * j] - ex)) <= 4e-16 * max(np.max(np.abs(ex)), 1e-300)
            Mj = mul(M, Mj)
        P = Mj
        Pl = P
        for m in range(64):                      # plane[m] = P^(m+1)
            close(plane[k, m], Pl)
            Pl = mul(P, Pl)
        Ps = P
        for s_ in range(7):                      # P^(2^s)
            close(tab[k, 5 + 2 * chunk + 4 * s_:9 + 2 * chunk + 4 * s_], Ps)
            Ps = mul(Ps, Ps)


def test_low_frequency_sections_keep_full_precision(oracle):
    """Sections with a double pole next to z = 1 (20-45 Hz at 96 kHz): the kernel's tables are powers of the state matrix up
    to M^2048, which lose n eps / angle^2 when they are formed in double (4.6 ulp of the peak at the wave boundaries of a
    64-filter script of tools/fuzz_eq.py, before the host formed them in double-double).  Three spans, so that carried
    state crosses every wave boundary."""
    rng = np.random.default_rng(77)
    lows = [(int(i % 3), 20.0 + 3.1 * i, (11.5 if i % 2 else -10.5), 2.0 + 0.5 * (i % 8)) for i in range(8)]
    frames = 2 * 8192 + 1500
    x = rng.uniform(-0.5, 0.5, (1, frames, 2)).astype(np.float32)
    y, _ = emu.eq_process(x, 96000.0, -3.0, lows)
    (e,) = _oracle_run(oracle, 96000.0, -3.0, lows, [x])
    peak = float(np.max(np.abs(e)))
    assert np.max(np.abs(y - e)) <= 1.5 * 2.0 ** -23 * max(1.0, peak)
