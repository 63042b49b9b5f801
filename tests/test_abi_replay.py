"""The boundary as a compiled host would use it: tests/harness/abi_replay.cpp (plain g++, only
include/airwave_hip.h[pp]) replays create -> process x N -> reset -> process -> destroy."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "harness", "abi_replay.cpp")
EXE = os.path.join(ROOT, "tests", "harness", "abi_replay")


def build():
    lib_dir = os.path.join(ROOT, "airwave_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", EXE, SRC, "-L" + lib_dir, "-lairwave_hip",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_harness_compiles_against_the_c_abi_without_hip_headers():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_replay_matches_oracle(oracle, golden_dir, tmp_path):
    build()
    wav_path = os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav")
    out = str(tmp_path / "dump.bin")
    subprocess.run([EXE, wav_path, out], check=True)
    got = np.fromfile(out, dtype=np.float32)
    half = got.size // 2
    # determinism after reset: second pass equals the first (ConvolutionEngineTests.swift:48-59 in spirit)
    assert np.max(np.abs(got[:half] - got[half:])) <= 1e-6
    # same sequence through the CPU oracle
    w = oracle.wav_load(wav_path)
    proc = oracle.RealtimeAudioProcessor([(w.audio_data[0], w.audio_data[1]), (w.audio_data[8], w.audio_data[7])], 512, 4096)
    eng = oracle.ConvolutionEngine(w.audio_data[0], 512)
    state = 12345

    def rnd():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return np.float32(state >> 8) / np.float32(16777216.0) - np.float32(0.5)

    exp = []
    for n in [128, 512, 700, 4096, 1]:
        lr = np.array([rnd() for _ in range(2 * n)], dtype=np.float32)
        l, r = proc.process(lr[0::2], lr[1::2])
        exp += [l, r]
    blk = np.array([rnd() for _ in range(512)], dtype=np.float32)
    exp.append(eng.process(blk))
    exp = np.concatenate(exp)
    assert exp.size == half
    assert np.max(np.abs(got[:half] - exp)) <= 1e-5 * np.max(np.abs(exp))


@pytest.mark.gpu
def test_replay_eq_through_the_cpp_mirror(oracle, golden_dir, tmp_path):
    """The EQ half of the graph from a compiled host: aw::EqualizerDefinition::parse + aw::ParametricEqualizerProcessor."""
    build()
    wav_path = os.path.join(golden_dir, "hrtf", "NeutralSH1.0.wav")
    preset = os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt")
    out = str(tmp_path / "dump_eq.bin")
    subprocess.run([EXE, wav_path, out, preset], check=True)
    got = np.fromfile(out, dtype=np.float32)[-6 * 1024:]
    p = oracle.ParametricEqualizerProcessor(48000.0, 512)
    p.set_target(oracle.eq_parse(open(preset, "rb").read(), "f.txt"))
    state = 777

    def rnd():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return np.float32(state >> 8) / np.float32(16777216.0) - np.float32(0.5)

    exp = []
    for call in range(6):
        if call == 4:
            p.drain_retired_states()
            p.set_target(None)
        lr = np.array([rnd() for _ in range(1024)], dtype=np.float32)
        l, r = p.process(lr[0::2], lr[1::2])
        exp += [l, r]
    exp = np.concatenate(exp)
    assert np.max(np.abs(got - exp)) <= 2e-7
