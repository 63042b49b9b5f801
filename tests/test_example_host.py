"""examples/offline_batch.c — the C ABI used from plain C99 by an offline batch host (preset activation, reserve, page-locked host
buffers, the pipelined host entry).  CPU: it compiles as strict C99 against include/airwave_hip.h alone and fails loudly without a
device.  GPU: its output equals the float64 convolution of the same PCM with the preset's HRIR pairs (HRIRManager.swift:347-446 for the
assembly, ConvolutionEngine.swift:232-367 summed over speakers for the samples) to 1e-5 of peak."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "offline_batch.c")
EXE = os.path.join(ROOT, "examples", "offline_batch")


def build():
    lib_dir = os.path.join(ROOT, "airwave_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + lib_dir,
                    "-lairwave_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", EXE], check=True)


def test_example_is_strict_c99_and_needs_a_device():
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("a device is present: the no-device half runs in the CPU container")
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "hrtf", "RoomSH1.0.wav"), "2", "0.1"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_example_output_matches_the_oracle(oracle, golden_dir, tmp_path):
    build()
    wav = os.path.join(golden_dir, "hrtf", "RoomSH1.0.wav")
    S, seconds, C = 5, 0.25, 8
    F = int(seconds * 48000)
    out = str(tmp_path / "out.f32")
    r = subprocess.run([EXE, wav, str(S), str(seconds), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"streams {S} frames {F}" in r.stdout
    y = np.fromfile(out, dtype=np.float32).reshape(S, F, 2)
    # the example's PCM: a 32-bit linear congruential sequence over the whole [stream][frame][channel] array
    n = S * F * C
    state, vals = 12345, np.empty(n, np.uint32)
    for i in range(n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = state
    x = ((vals >> 8).astype(np.float32) / np.float32(16777216.0) - np.float32(0.5)).reshape(S, F, C)
    w = oracle.wav_load(wav)
    tracks, lt, rt = oracle.assemble_tracks(w, oracle.layout_detect(C), 48000.0)          # the oracle's restatement of activatePreset
    ref = np.stack([oracle.spatialize_f64(x[s], tracks, lt, rt) for s in range(S)])
    assert oracle.peak_rel_error(y, ref) < 1e-5


SRC_EQ = os.path.join(ROOT, "examples", "offline_batch_eq.c")
EXE_EQ = os.path.join(ROOT, "examples", "offline_batch_eq")


def build_eq():
    lib_dir = os.path.join(ROOT, "airwave_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"), SRC_EQ, "-L" + lib_dir,
                    "-lairwave_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", EXE_EQ], check=True)


def test_equalizer_example_is_strict_c99_folds_on_the_host_and_needs_a_device_only_then():
    """examples/offline_batch_eq.c: parser, WAV loader, channel map and aw_eq_fold_hrir are host code and run; the first device call
    fails loudly in the CPU container."""
    import torch
    build_eq()
    if torch.cuda.is_available():
        pytest.skip("a device is present: the no-device half runs in the CPU container")
    args = [EXE_EQ, os.path.join(ROOT, "tests", "golden", "hrtf", "StageSH1.0.wav"), os.path.join(ROOT, "tests", "golden", "eq", "CCA CRA ParametricEq.txt"), "2", "0.1"]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 1 and "aw_context_create" in r.stderr and "no HIP device" in r.stderr, (r.returncode, r.stderr)
    r = subprocess.run(args[:2] + [os.path.join(ROOT, "tests", "golden", "hesuvi14_custom_map.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "equalizer preset:" in r.stderr                      # not an Equalizer APO file: the parser's issues


@pytest.mark.gpu
def test_equalizer_example_output_matches_the_two_effect_order(oracle, golden_dir, tmp_path):
    """Spatializer + equalizer in one pass from plain C: equal to float64 convolution -> float32 -> the oracle's sequential Float64 cascade
    (AudioEffectGraph.swift:195-211 order) to 1e-5 of the peak."""
    build_eq()
    wav = os.path.join(golden_dir, "hrtf", "StageSH1.0.wav")
    preset = os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt")
    S, seconds, C = 3, 0.5, 8
    F = int(seconds * 48000)
    out = str(tmp_path / "out.f32")
    r = subprocess.run([EXE_EQ, wav, preset, str(S), str(seconds), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"streams {S} frames {F} hrir taps 4320 + equalizer response" in r.stdout
    y = np.fromfile(out, dtype=np.float32).reshape(S, F, 2)
    n = S * F * C
    state, vals = 12345, np.empty(n, np.uint32)
    for i in range(n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = state
    x = ((vals >> 8).astype(np.float32) / np.float32(16777216.0) - np.float32(0.5)).reshape(S, F, C)
    w = oracle.wav_load(wav)
    tracks, lt, rt = oracle.assemble_tracks(w, oracle.layout_detect(C), 48000.0)
    od = oracle.eq_parse(open(preset, "rb").read(), preset)
    for s in range(S):
        sp = oracle.spatialize_f64(x[s], tracks, lt, rt).astype(np.float32)
        el, er = oracle.eq_prepare(od, 48000.0).process(np.ascontiguousarray(sp[:, 0]), np.ascontiguousarray(sp[:, 1]))
        assert oracle.peak_rel_error(y[s], np.stack([el, er], axis=1)) < 1e-5, s
