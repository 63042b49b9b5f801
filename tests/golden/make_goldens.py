#!/usr/bin/env python3
"""Generates tests/golden/*.npz (run in the build container; outputs are committed).

No reference code runs here: the reference (Swift + Apple Accelerate) cannot be compiled or
imported on Linux.  Vectors are (a) the reference's own known-answer tests re-expressed as
arrays, (b) float64 linear-convolution truth for seeded inputs against the reference's bundled
HRIR assets (tests/golden/hrtf/*.wav, data files copied from assets/hrtf/), (c) table dumps of
the channel maps, (d) WAV-decode digests.  Inputs are regenerated from the seeded counter RNG
(oracle.synth_input), so only expected outputs are stored.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import airwave_oracle as orc  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes")


def hrtf(name):
    return orc.wav_load(os.path.join(HERE, "hrtf", name))


def case(wav_tracks, speakers, frames, seed, tracks_are_wav=True, cmap=None):
    wav = wav_tracks if tracks_are_wav else None
    if wav is not None:
        tracks, lt, rt = orc.assemble_tracks(wav, speakers, channel_map=cmap)
    else:
        tracks, lt, rt = wav_tracks
    x = orc.synth_input(1, frames, len(speakers), seed=seed)[0]
    y = orc.spatialize_f64(x, tracks, lt, rt)
    return dict(expected=y, left_track=lt, right_track=rt, frames=np.int64(frames), seed=np.int64(seed),
                channels=np.int64(len(speakers)))


def main():
    # (1) KATs: impulse HRIR identity (ConvolutionEngineTests.swift:12-20) and the 6913-frame adapter
    # sequence (RealtimeAudioProcessorTests.swift:68-78)
    kat_in = np.array([0.25, -0.5, 1, 0.75, -1, 0.125, 0.5, -0.25], dtype=np.float32)
    sizes = np.array([128, 128, 128, 128, 513, 768, 1024, 4096], dtype=np.int64)
    adapter = np.concatenate([np.zeros(384), np.ones(6913 - 384)]).astype(np.float32)
    save("kat_reference.npz", impulse_hrir=np.eye(1, 8, dtype=np.float32)[0], impulse_in=kat_in,
         impulse_out=kat_in.copy(), adapter_sizes=sizes, adapter_expected=adapter)

    # (2) cfg 1: NeutralSH1.0 stereo (tracks 0,1,8,7), 8 blocks of 512
    save("cfg1_neutral_stereo.npz", **case(hrtf("NeutralSH1.0.wav"), orc.layout_detect(2), 8 * 512, orc.SYNTH_SEED))
    # (3) cfg 2: RoomSH1.0 7.1, 4 blocks (+ a ragged tail so non-multiple-of-512 lengths are covered)
    save("cfg2_room_71.npz", **case(hrtf("RoomSH1.0.wav"), orc.layout_detect(8), 4 * 512 + 37, orc.SYNTH_SEED))
    # (4) 14-track / 7-speaker case (cfg 3 primary layout) on StageSH1.0
    save("cfg3_stage_7spk.npz", **case(hrtf("StageSH1.0.wav"), ["FL", "FR", "FC", "BL", "BR", "SL", "SR"], 3 * 512, orc.SYNTH_SEED))
    # 14 custom input channels through a parseHeSuViFormat text map (cfg 3 secondary)
    text = open(os.path.join(HERE, "hesuvi14_custom_map.txt")).read()
    cmap = orc.parse_hesuvi_format(text)
    save("cfg3_stage_14ch_custom.npz", **case(hrtf("StageSH1.0.wav"), orc.layout_detect(14), 2 * 512, orc.SYNTH_SEED, cmap=cmap))
    # (5) synthetic long-tap HRIR, P = 64 partitions at B = 512 (L = 32768), 7 speakers, short input
    long_h = orc.synth_hrir(14, 32768, seed=1234)
    lt = np.array([0, 8, 6, 4, 12, 2, 10], dtype=np.int32)
    rt = np.array([1, 7, 13, 5, 11, 3, 9], dtype=np.int32)
    x = orc.synth_input(1, 9000, 7, seed=orc.SYNTH_SEED)[0]
    save("cfg3_longtap_7spk.npz", expected=orc.spatialize_f64(x, long_h, lt, rt), left_track=lt, right_track=rt,
         frames=np.int64(9000), seed=np.int64(orc.SYNTH_SEED), channels=np.int64(7), hrir_seed=np.int64(1234),
         hrir_taps=np.int64(32768))

    # (6) track-index tables for every InputLayout x {hesuvi14, hesuvi7}
    rows = []
    for n in [1, 2, 6, 7, 8, 12, 14]:
        spk = orc.layout_detect(n)
        for kind, fn in (("hesuvi14", orc.map_hesuvi14), ("hesuvi7", orc.map_hesuvi7)):
            m = fn(spk)
            for i, s in enumerate(spk):
                l, r = m.get(s, (-1, -1))
                rows.append((n, 14 if kind == "hesuvi14" else 7, i, l, r))
    save("channel_map_tables.npz", rows=np.array(rows, dtype=np.int32))

    # (7) WAV decode digests of the bundled HRIRs
    dig = {}
    for name in ["NeutralSH1.0", "RoomSH1.0", "StageSH1.0"]:
        path = os.path.join(HERE, "hrtf", name + ".wav")
        w = orc.wav_load(path)
        key = name.replace(".", "_")
        dig[key + "_shape"] = np.array([w.channel_count, w.frame_count, int(w.sample_rate)], dtype=np.int64)
        dig[key + "_first8"] = w.audio_data[:, :8].copy()
        dig[key + "_last8"] = w.audio_data[:, -8:].copy()
        dig[key + "_sha256"] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), dtype=np.uint8)
        dig[key + "_sum"] = w.audio_data.astype(np.float64).sum(axis=1)
    save("wav_decode_digests.npz", **dig)


if __name__ == "__main__":
    main()
