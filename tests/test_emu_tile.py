"""The HIP tile kernel's source (airwave_amd/csrc/device/tile_ols.hpp), compiled for the host and
run under thread emulation (tests/emu/), against the float64 truth.  Catches index-math and
LDS-hazard bugs on the CPU-only container; the real parity tests are the -m gpu ones."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

TOL = 1e-5


def test_small_butterflies_match_numpy():
    rng = np.random.default_rng(0)
    for n in (4, 8, 16):
        v = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.fft(v.astype(np.complex128))
        assert np.abs(emu.fft_small(v) - ref).max() < 2e-6 * n
        iref = np.fft.ifft(v.astype(np.complex128)) * n
        assert np.abs(emu.fft_small(v, True) - iref).max() < 2e-6 * n


@pytest.mark.parametrize("channels", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, (10, 5), (14, 5)])
def test_emulated_tile_matches_truth(oracle, golden_dir, channels):
    # (channels, 5): the two-pass form of the wide layouts (AW_WIDE_TWO_PASS=1); the default is one pass over two eight-channel groups
    channels, variant = channels if isinstance(channels, tuple) else (channels, 1)
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    spk = oracle.layout_detect(8)[:channels] if channels <= 8 else oracle.layout_detect(8) + ["FL", "FR", "BL", "BR", "SL", "SR", "FC", "LFE"][: channels - 8]
    cmap = oracle.map_hesuvi14(oracle.layout_detect(8))
    tracks = wav.audio_data
    lt = np.array([cmap[s][0] for s in spk], dtype=np.int32)
    rt = np.array([cmap[s][1] for s in spk], dtype=np.int32)
    frames = 16500                   # boundary tiles (history, ragged end) and interior ones
    x = oracle.synth_input(1, frames, channels)
    y = emu.fused_ols(x, tracks, lt, rt, variant=variant)
    assert not np.isnan(y).any()
    ref = oracle.spatialize_f64(x[0], tracks, lt, rt)
    for ear in range(2):
        assert oracle.peak_rel_error(y[0, :, ear], ref[:, ear]) < TOL


def test_emulated_tile_history_carry(oracle):
    # second call continues the first: history = tail of the first call's input
    h = oracle.synth_hrir(2, 300, seed=3)
    x = oracle.synth_input(1, 1200, 2)
    hop = 8192 - 299
    ref = oracle.spatialize_f64(x[0], h, [0, 1], [1, 0])
    hist = np.zeros((1, 8192 - hop, 2), dtype=np.float32)
    hist[0] = x[0, 700 - (8192 - hop):700]      # the last hist_len frames before the second call
    y2 = emu.fused_ols(x[:, 700:], h, [0, 1], [1, 0], hop=hop, hist=hist)
    assert oracle.peak_rel_error(y2[0], ref[700:]) < TOL


def test_emulated_partitioned_long_hrir(oracle, golden_dir):
    # cfg 3 golden: synthetic 32768-tap HRIR (P = 8 partitions of 4096), 7 speakers
    g = np.load(os.path.join(golden_dir, "cfg3_longtap_7spk.npz"))
    h = oracle.synth_hrir(14, int(g["hrir_taps"]), seed=int(g["hrir_seed"]))
    x = oracle.synth_input(1, int(g["frames"]), 7, seed=int(g["seed"]))
    y = emu.partitioned(x, h, g["left_track"], g["right_track"])
    assert not np.isnan(y).any()
    assert oracle.peak_rel_error(y[0], g["expected"]) < TOL


@pytest.mark.parametrize("channels", [5, 7, 8])
def test_emulated_partitioned_interior_windows(oracle, channels):
    """Long enough for interior windows of the forward kernel (whole-frame vector loads; 7-channel frames run into
    the next frame) next to the boundary ones; 9000 taps = 3 partitions."""
    h = oracle.synth_hrir(14, 9000, seed=5)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    x = oracle.synth_input(1, 13001, channels, seed=channels)
    y = emu.partitioned(x, h, lt, rt)
    assert not np.isnan(y).any()
    assert oracle.peak_rel_error(y[0], oracle.spatialize_f64(x[0], h, lt, rt)) < TOL


@pytest.mark.parametrize("channels,taps,frames", [(1, 700, 21000), (2, 300, 5000), (2, 4320, 40001), (8, 4320, 30011), (7, 4319, 29000), (5, 1001, 20000), (5, 4320, 30001), (3, 6001, 29001), (4, 777, 17000)])
def test_emulated_16384_window_path(oracle, channels, taps, frames):
    """tile_ols2.hpp: the polyphase (half-rate, 2C pseudo-channels, two output spectra) form of a 16384-frame window,
    interior and boundary kernels, odd tap and frame counts."""
    h = oracle.synth_hrir(14, taps, seed=3)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    x = oracle.synth_input(1, frames, channels, seed=channels)
    y = emu.fused_ols(x, h, lt, rt, variant=2)
    assert not np.isnan(y).any()
    assert oracle.peak_rel_error(y[0], oracle.spatialize_f64(x[0], h, lt, rt)) < TOL


@pytest.mark.parametrize("cmac", ["march", "group"])
def test_emulated_partitioned_more_than_eight_partitions(oracle, cmac):
    """P = 10 partitions: both CMAC kernels take the partitions 8 at a time, so this exercises a second, partial pass
    (marched kernel: a PQ = 4 pass with two zero-table slots that accumulates into W; block-group kernel: groups that
    straddle the chunk)."""
    h = oracle.synth_hrir(14, 40000, seed=8)
    lt = np.array([0, 3, 5], np.int32)
    rt = np.array([1, 4, 13], np.int32)
    x = oracle.synth_input(1, 50001, 3, seed=2)
    y = emu.partitioned(x, h, lt, rt, cmac=cmac)
    assert not np.isnan(y).any()
    assert oracle.peak_rel_error(y[0], oracle.spatialize_f64(x[0], h, lt, rt)) < TOL


def test_marched_cmac_slots_cover_every_bin_once():
    """tile_march.hpp: slot -> (bin, partner bin) must hit each of the 8192 storage indices exactly once, each slot's two
    bins must be k and N - k, and the two self-paired bins (k = 0, N/2) get slots of their own."""
    N = 8192
    seen = np.zeros(N, int)
    for j in range(N // 2 + 1):
        if j < 7 * 512:
            row, col = 1 + (j >> 9), j & 511
            i, pi = row * 512 + col, (16 - row) * 512 + (511 - col)
        elif j < 7 * 512 + 256:
            col = j - 7 * 512
            i, pi = col, (512 - col) & 511
        elif j < 8 * 512:
            col = j - (7 * 512 + 256)
            i, pi = 8 * 512 + col, 8 * 512 + 511 - col
        else:
            i = pi = 256
        k, kp = (i >> 9) + 16 * (i & 511), (pi >> 9) + 16 * (pi & 511)      # storage index = k1 * 512 + k2, k = k1 + 16 k2
        assert (k + kp) % N == 0
        seen[i] += 1
        if pi != i:
            seen[pi] += 1
    assert (seen == 1).all()


def test_half_wave_row_transform_against_numpy():
    """sub_fft512h_fwd / _inv (tile_ols.hpp): a half-wave per 512-point row, 16 x 2 x 16 with one LDS transpose and a permlane radix-2
    stage.  Forward = numpy's FFT in the bin order hl_col(lane) + 32 kb; inverse = its unnormalised inverse (512 x the input)."""
    rng = np.random.default_rng(12)
    rows = (rng.standard_normal((2, 512)) + 1j * rng.standard_normal((2, 512))).astype(np.complex64)
    fwd, back = emu.sub_fft512h(rows)
    ref = np.fft.fft(rows.astype(np.complex128), axis=1)
    assert np.max(np.abs(fwd - ref)) <= 2e-6 * np.max(np.abs(ref))
    assert np.max(np.abs(back / 512 - rows)) <= 2e-6 * np.max(np.abs(rows))
    delta = np.zeros((2, 512), np.complex64); delta[0, 1] = 1; delta[1, 511] = 1j      # single bins: exact twiddle placement
    f2, _ = emu.sub_fft512h(delta)
    k = np.arange(512)
    assert np.max(np.abs(f2[0] - np.exp(-2j * np.pi * k / 512))) < 1e-6
    assert np.max(np.abs(f2[1] - 1j * np.exp(2j * np.pi * k / 512))) < 1e-6
