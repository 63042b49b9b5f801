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


@pytest.mark.parametrize("channels", [7, 8, 1, 2, 5])
def test_emulated_long_window_matches_truth(oracle, channels):
    # one window of 32 x 4096 frames holds the whole call: history (zeros), input, zero fill past the end
    taps, frames = 9000, 100000
    h, lt, rt, x, ref = _case(oracle, channels, taps, frames, 32)
    y = emu.longwin(x, h, lt, rt, R=32)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL


def test_emulated_long_window_two_windows_and_history(oracle):
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
    y = emu.longwin(x[:, hist_len:], h, lt, rt, R=R, hop=hop, hist=hist)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[hist_len:, ear]) < TOL


@pytest.mark.parametrize("R", [64, 128])
def test_emulated_long_window_larger_radix(oracle, R):
    # R = 64 (in-thread radix 8) and R = 128 (radix 16): a short call in a large window — the kernels do the full work
    taps, frames = 33000, 20000
    h, lt, rt, x, ref = _case(oracle, 3, taps, frames, R)
    y = emu.longwin(x, h, lt, rt, R=R)
    assert not np.isnan(y).any()
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL
