"""The long-window path's device source (airwave_amd/csrc/device/tile_lw.hpp: split -> rows -> merge), compiled for the
host and run under thread emulation (tests/emu/), against the float64 truth.  Index math, twiddles, table layout and LDS
hazards are checked here on the CPU-only container; the parity tests proper are the -m gpu ones."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

TOL = 1e-5


def _case(oracle, channels, taps, frames, R, seed=5, hop=None, hist_frames=0):
    h = oracle.synth_hrir(14, taps, seed=seed)
    lt = np.array([(2 * c) % 14 for c in range(channels)], dtype=np.int32)
    rt = np.array([(2 * c + 1) % 14 for c in range(channels)], dtype=np.int32)
    x = oracle.synth_input(1, hist_frames + frames, channels)
    ref = oracle.spatialize_f64(x[0], h, lt, rt)
    return h, lt, rt, x, ref


@pytest.mark.parametrize("channels,rows_pb", [(7, 2), (8, 2), (1, 2), (2, 2), (5, 2), (7, 1), (6, 1), (1, 1), (14, 1), (9, 1), (13, 2), (16, 1), (12, 1), (11, 1),
                                              (7, 16), (8, 16), (1, 16), (14, 16), (13, 16)])
def test_emulated_long_window_matches_truth(oracle, channels, rows_pb):
    # one window of 32 x 4096 frames holds the whole call: history (zeros), input, zero fill past the end;
    # rows_pb = channel pairs per batch of the rows kernel (1: the two-workgroups-per-CU form; 16: the 16-points-per-thread
    # kernel of tile_lw16.hpp, 256-thread workgroups)
    taps, frames = 9000, 100000
    h, lt, rt, x, ref = _case(oracle, channels, taps, frames, 32)
    y = emu.longwin(x, h, lt, rt, R=32, rows_pb=rows_pb)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL


@pytest.mark.parametrize("rows_pb", [2, 16])
def test_emulated_long_window_two_windows_and_history(oracle, rows_pb):
    # two windows (hop < frames), a history buffer carried from a previous call, an unmapped channel
    taps, R = 40000, 32
    N = R * 4096
    hop = 70016                         # N - hop = 61056 >= taps - 1
    hist_len = N - hop
    frames = 120000
    h, lt, rt, x, ref = _case(oracle, 6, taps, frames, R, hist_frames=hist_len)
    lt[3] = -1
    ref = oracle.spatialize_f64(x[0], h, lt, rt)
    hist = x[:, :hist_len].copy()
    hist_out = np.full((1, hist_len, 6), np.nan, dtype=np.float32)
    y = emu.longwin(x[:, hist_len:], h, lt, rt, R=R, hop=hop, hist=hist, hist_out=hist_out, rows_pb=rows_pb)
    assert not np.isnan(y).any()
    assert np.array_equal(hist_out, x[:, -hist_len:])          # the split kernel carried the convolution tail
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[hist_len:, ear]) < TOL


@pytest.mark.parametrize("n", range(2, 17))
def test_in_register_dfts_of_every_size(n):
    # the split / merge kernels' DFT over the register index (tile_lw.hpp lw_fft / lw_odd_dft): powers of two, odd primes (direct),
    # mixed sizes (Cooley-Tukey with compile-time twiddles)
    rng = np.random.default_rng(n)
    v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    j = np.arange(n)
    for inv in (False, True):
        for odd in (False, True):
            h = 0.5 if odd else 0.0
            # forward: out[k] = sum_j v[j] w^{j (k + h)};  inverse: out[j] = sum_k v[k] conj(w)^{j (k + h)}
            M = np.exp(-2j * np.pi * (j[:, None] + h) * j[None, :] / n) if not inv else np.exp(2j * np.pi * j[:, None] * (j[None, :] + h) / n)
            ref = M @ v.astype(np.complex128)
            got = emu.lw_dft(v, inv, odd)
            assert np.abs(got - ref).max() < 2e-6 * np.abs(ref).max(), (n, inv, odd)


@pytest.mark.parametrize("R,channels,rows_pb", [(40, 3, 16), (48, 7, 16), (56, 2, 1), (72, 3, 16), (80, 5, 16), (96, 1, 16), (112, 9, 1), (120, 2, 16), (40, 13, 16)])
def test_emulated_window_lengths_that_are_not_powers_of_two(oracle, R, channels, rows_pb):
    # R = 8 RA, RA = 5, 6, 7, 9, 10, 12, 14, 15: the same three kernels, tables from the mixed-radix host transform
    taps, frames = 9000, 30000
    h, lt, rt, x, ref = _case(oracle, channels, taps, frames, R)
    y = emu.longwin(x, h, lt, rt, R=R, rows_pb=rows_pb)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL


@pytest.mark.parametrize("R,rows_pb", [(64, 2), (128, 2), (64, 16), (128, 16)])
def test_emulated_long_window_larger_radix(oracle, R, rows_pb):
    # R = 64 (in-thread radix 8) and R = 128 (radix 16): a short call in a large window — the kernels do the full work
    taps, frames = 33000, 20000
    h, lt, rt, x, ref = _case(oracle, 3, taps, frames, R)
    y = emu.longwin(x, h, lt, rt, R=R, rows_pb=rows_pb)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL


def test_emulated_split_kernel_carries_a_short_calls_tail(oracle):
    # a call shorter than the history: the new history is part old history, part input
    taps, R = 33000, 32
    N = R * 4096
    hist_len = 36864
    hop = N - hist_len
    frames = 5000
    h, lt, rt, x, ref = _case(oracle, 3, taps, frames, R, hist_frames=hist_len)
    hist = x[:, :hist_len].copy()
    hist_out = np.full((1, hist_len, 3), np.nan, dtype=np.float32)
    y = emu.longwin(x[:, hist_len:], h, lt, rt, R=R, hop=hop, hist=hist, hist_out=hist_out)
    assert np.array_equal(hist_out, x[:, -hist_len:])
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[hist_len:, ear]) < TOL


def test_emulated_wide_layout_carries_history(oracle):
    # 14 channels, two windows, history in and out through the wide split kernel (both channel halves of a frame in one wave);
    # the call's very last frame is read through the padded tail copy
    taps, R, C = 30000, 32, 14
    N = R * 4096
    hist_len = 32768
    hop = N - hist_len
    frames = 110000
    h, lt, rt, x, ref = _case(oracle, C, taps, frames, R, hist_frames=hist_len)
    hist = x[:, :hist_len].copy()
    hist_out = np.full((1, hist_len, C), np.nan, dtype=np.float32)
    y = emu.longwin(x[:, hist_len:], h, lt, rt, R=R, hop=hop, hist=hist, hist_out=hist_out, rows_pb=1)
    assert np.array_equal(hist_out, x[:, -hist_len:])
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[hist_len:, ear]) < TOL


@pytest.mark.parametrize("rows", [32, 40, 48, 56, 64, 72, 80, 96, 112, 120, 128])
@pytest.mark.parametrize("n_sw", [1, 3, 128, 1021])
def test_row_tiles_cover_every_row_pair_once_and_balance_the_xcd_groups(rows, n_sw):
    """The rows kernels pin row pairs to the 8 XCD groups (table slice in that XCD's L2).  Window lengths with R/2 = 4 mod 8 (40, 56, 72,
    120 rows) leave four row pairs over: their tiles are dealt to all eight groups in equal shares instead of whole pairs to four of them
    (round-4 advisor finding: 8 row pairs against 7 at R = 120).  Every (row pair, stream-window) exactly once; groups within one tile."""
    hits, per = emu.lw_row_map(rows // 2, n_sw)
    assert (hits == 1).all()
    assert per.max() - per.min() <= 1 or (rows // 2) % 8 == 0 and per.max() == per.min(), per
