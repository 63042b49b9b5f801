"""The overlap-add tile's source (airwave_amd/csrc/device/tile_ola.hpp), compiled for the host and run under thread emulation
(tests/emu/), against the float64 truth: block / carry / warm-up index math and the buffer-load range checks, on the CPU-only
container.  The GPU parity tests of the same tile are tests/test_gpu_ola.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

TOL = 1e-5


def _room_map(oracle, golden_dir, channels):
    wav = oracle.wav_load(os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav"))
    spk = oracle.layout_detect(8)[:channels] if channels <= 8 else oracle.layout_detect(8) + ["FL", "FR", "BL", "BR", "SL", "SR", "FC", "LFE"][: channels - 8]
    cmap = oracle.map_hesuvi14(oracle.layout_detect(8))
    lt = np.array([cmap[s][0] for s in spk], dtype=np.int32)
    rt = np.array([cmap[s][1] for s in spk], dtype=np.int32)
    return wav.audio_data, lt, rt


@pytest.mark.parametrize("channels,workgroups", [(8, 3), (14, 2), (7, 4), (2, 1), (16, 2), (12, 5), (6, 3), (13, 3)])
def test_emulated_overlap_add_tile_matches_truth(oracle, golden_dir, channels, workgroups):
    """4320 taps -> blocks of 7 x 512 frames.  16 500 frames = 5 blocks per stream (the last one ragged); the runs cut streams in the
    middle (carry rebuilt from two warm-up blocks), start at stream starts (warm-up blocks in the zero history) and cross stream ends."""
    tracks, lt, rt = _room_map(oracle, golden_dir, channels)
    x = oracle.synth_input(2, 16500, channels)
    y = emu.fused_ola(x, tracks, lt, rt, workgroups=workgroups)
    assert not np.isnan(y).any()
    for s in range(2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], tracks, lt, rt)) < TOL


def test_emulated_overlap_add_result_does_not_depend_on_the_run_cut(oracle, golden_dir):
    """Runs rebuild their carry from the input alone: one workgroup for everything and seven workgroups give the same bits."""
    tracks, lt, rt = _room_map(oracle, golden_dir, 8)
    x = oracle.synth_input(2, 12000, 8, seed=5)
    a = emu.fused_ola(x, tracks, lt, rt, workgroups=1)
    b = emu.fused_ola(x, tracks, lt, rt, workgroups=7)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("channels,taps,H", [(8, 3969, 8), (14, 4000, 8), (8, 5500, 5), (14, 5000, 6), (5, 6100, 4), (7, 4609, 7), (2, 300, 8), (9, 3969, 8)])
def test_emulated_overlap_add_block_lengths(oracle, channels, taps, H):
    """Every block length the library carries (H = 5 .. 8 rows of 512 frames) and the shortest the tile supports (4), at the longest
    HRIR each one holds or near it; a short HRIR on the longest block."""
    h = oracle.synth_hrir(14, taps, seed=taps)
    lt = (np.arange(channels) % 14).astype(np.int32)
    rt = ((np.arange(channels) + 7) % 14).astype(np.int32)
    x = oracle.synth_input(1, 3 * 512 * H + 777, channels, seed=H)
    y = emu.fused_ola(x, h, lt, rt, H=H, workgroups=2)
    assert not np.isnan(y).any()
    assert oracle.peak_rel_error(y[0], oracle.spatialize_f64(x[0], h, lt, rt)) < TOL


def test_emulated_overlap_add_history_carry(oracle, golden_dir):
    """A second call continues the first: the warm-up blocks read the history rows (the last frames of the first call's input); rows
    before the kept history and frames past the end come back as zeros from the range-checked loads.  History longer than taps - 1
    (the runtime keeps N - the aligned overlap-save hop rows) changes nothing."""
    tracks, lt, rt = _room_map(oracle, golden_dir, 8)
    x = oracle.synth_input(1, 14000, 8, seed=9)
    ref = oracle.spatialize_f64(x[0], tracks, lt, rt)
    for cut, hist_len in ((5000, 4319), (9111, 4352), (700, 4352)):
        hist = np.zeros((1, hist_len, 8), dtype=np.float32)
        n = min(cut, hist_len)
        hist[0, hist_len - n:] = x[0, cut - n:cut]
        y2 = emu.fused_ola(x[:, cut:], tracks, lt, rt, hist=hist, workgroups=2)
        assert oracle.peak_rel_error(y2[0], ref[cut:]) < TOL, (cut, hist_len)


def test_emulated_overlap_add_streams_are_independent(oracle, golden_dir):
    """A NaN in one stream (input and history) never reaches its neighbours: blocks, carries and warm-ups are per stream."""
    tracks, lt, rt = _room_map(oracle, golden_dir, 14)
    x = oracle.synth_input(3, 9000, 14, seed=2)
    x[1, 4000:4100] = np.nan
    hist = np.zeros((3, 4319, 14), dtype=np.float32)
    hist[1] = np.nan
    y = emu.fused_ola(x, tracks, lt, rt, hist=hist, workgroups=4)
    for s in (0, 2):
        assert oracle.peak_rel_error(y[s], oracle.spatialize_f64(x[s], tracks, lt, rt)) < TOL
    assert np.isnan(y[1]).any()


def test_emulated_overlap_add_tuning_variants_stay_correct(golden_dir):
    """The measured-and-not-kept forms of the tile stay behind macros: frames requested behind the last table request, tables
    requested before the row transform, both pairs of a batch through the row transforms together.  They must stay CORRECT — someone
    will A/B them again.  Emulated in a process of its own (the harness is built per set of defines)."""
    import subprocess
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, os.path.join('tests', 'emu')); sys.path.insert(0, 'oracle')\n"
        "import emu, airwave_oracle as orc\n"
        "wav = orc.wav_load(os.path.join('tests', 'golden', 'hrtf', 'RoomSH1.0.wav'))\n"
        "cmap = orc.map_hesuvi14(orc.layout_detect(8)); spk = orc.layout_detect(8)[:6]\n"
        "lt = np.array([cmap[s][0] for s in spk], np.int32); rt = np.array([cmap[s][1] for s in spk], np.int32)\n"
        "x = orc.synth_input(2, 9000, 6, seed=3)\n"
        "y = emu.fused_ola(x, wav.audio_data, lt, rt, workgroups=3)\n"
        "print(max(orc.peak_rel_error(y[s], orc.spatialize_f64(x[s], wav.audio_data, lt, rt)) for s in range(2)))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # six channels = three pairs: the first batch has two pairs (together), the last batch one (early tables, prefetch behind its last table request)
    for defines in ("-DAW_OLA_PAIR2=1 -DAW_OLA_PAIR2_MAXNP=3 -DAW_OLA_PREFETCH_MID=1 -DAW_OLA_TAB_EARLY=1",):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, AW_EMU_DEFINES=defines), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert float(r.stdout.strip().splitlines()[-1]) < TOL, (defines, r.stdout)
