"""The float32 engine restatement vs the float64 truth goldens (tolerance: north_star's 1e-5,
peak-normalised per SURVEY.md §7), plus oracle self-consistency properties."""
import os

import numpy as np
import pytest

TOL = 1e-5  # BASELINE.json north_star: <= 1e-5 max rel error vs ConvolutionEngine


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def hrtf(oracle, golden_dir, name):
    return oracle.wav_load(os.path.join(golden_dir, "hrtf", name))


@pytest.mark.parametrize("gold,wavname,speakers", [
    ("cfg1_neutral_stereo.npz", "NeutralSH1.0.wav", ["FL", "FR"]),
    ("cfg2_room_71.npz", "RoomSH1.0.wav", ["FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR"]),
    ("cfg3_stage_7spk.npz", "StageSH1.0.wav", ["FL", "FR", "FC", "BL", "BR", "SL", "SR"]),
])
def test_f32_engine_network_matches_f64_truth(oracle, golden_dir, gold, wavname, speakers):
    g = load(golden_dir, gold)
    wav = hrtf(oracle, golden_dir, wavname)
    tracks, lt, rt = oracle.assemble_tracks(wav, speakers)
    assert np.array_equal(lt, g["left_track"]) and np.array_equal(rt, g["right_track"])
    x = oracle.synth_input(1, int(g["frames"]), len(speakers), seed=int(g["seed"]))[0]
    y = oracle.spatialize_f32(x, tracks, lt, rt)
    for ear in range(2):
        assert oracle.peak_rel_error(y[:, ear], g["expected"][:, ear]) < TOL


def test_long_tap_p64(oracle, golden_dir):
    g = load(golden_dir, "cfg3_longtap_7spk.npz")
    h = oracle.synth_hrir(14, int(g["hrir_taps"]), seed=int(g["hrir_seed"]))
    x = oracle.synth_input(1, int(g["frames"]), 7, seed=int(g["seed"]))[0]
    e = oracle.ConvolutionEngine(h[0], 512)
    assert e.partition_count == 64
    y = oracle.spatialize_f32(x, h, g["left_track"], g["right_track"])
    for ear in range(2):
        assert oracle.peak_rel_error(y[:, ear], g["expected"][:, ear]) < TOL


def test_custom_14ch_text_map(oracle, golden_dir):
    g = load(golden_dir, "cfg3_stage_14ch_custom.npz")
    wav = hrtf(oracle, golden_dir, "StageSH1.0.wav")
    cmap = oracle.parse_hesuvi_format(open(os.path.join(golden_dir, "hesuvi14_custom_map.txt")).read())
    tracks, lt, rt = oracle.assemble_tracks(wav, oracle.layout_detect(14), channel_map=cmap)
    assert np.array_equal(lt, g["left_track"]) and np.array_equal(rt, g["right_track"])
    x = oracle.synth_input(1, int(g["frames"]), 14, seed=int(g["seed"]))[0]
    y = oracle.spatialize_f32(x, tracks, lt, rt)
    assert oracle.peak_rel_error(y, g["expected"]) < TOL


def test_block_size_independence(oracle):
    # exact streaming linear convolution: results do not depend on the partition size
    rng = np.random.default_rng(7)
    h = rng.standard_normal(700).astype(np.float32)
    x = rng.uniform(-0.5, 0.5, 4096).astype(np.float32)
    ref = oracle.direct_conv_f64(x, h)
    for b in (8, 64, 512, 2048):
        e = oracle.ConvolutionEngine(h, b)
        y = np.concatenate([e.process(x[i:i + b]) for i in range(0, x.size, b)])
        assert oracle.peak_rel_error(y, ref) < TOL


def test_linearity_and_process_and_accumulate(oracle):
    rng = np.random.default_rng(3)
    h = rng.standard_normal(1500).astype(np.float32)
    a = rng.uniform(-0.5, 0.5, 512).astype(np.float32)
    e1 = oracle.ConvolutionEngine(h, 512)
    e2 = oracle.ConvolutionEngine(h, 512)
    acc = np.full(512, 3.0, dtype=np.float32)
    e1.process_and_accumulate(a, acc)           # ConvolutionEngine.swift:388-394
    assert np.allclose(acc - 3.0, e2.process(a), atol=1e-6)


def test_synth_matches_c_implementation(oracle):
    L = oracle.lib()
    x = oracle.synth_input(3, 17, 5, seed=oracle.SYNTH_SEED)
    for s, i in [(0, 0), (1, 11), (2, 84)]:
        assert x[s].reshape(-1)[i] == L.orc_synth_value(oracle.SYNTH_SEED, s, i)
    assert x.min() >= -0.5 and x.max() < 0.5
    assert abs(float(oracle.synth_input(1, 100000, 2)[0].mean())) < 5e-3


def test_wav_decode_digests(oracle, golden_dir):
    d = load(golden_dir, "wav_decode_digests.npz")
    for name in ["NeutralSH1.0", "RoomSH1.0", "StageSH1.0"]:
        w = hrtf(oracle, golden_dir, name + ".wav")
        k = name.replace(".", "_")
        assert [w.channel_count, w.frame_count, int(w.sample_rate)] == d[k + "_shape"].tolist() == [14, 4320, 48000]
        assert np.array_equal(w.audio_data[:, :8], d[k + "_first8"])
        assert np.array_equal(w.audio_data[:, -8:], d[k + "_last8"])
