"""CPU ORACLE — test infrastructure, NOT product code.

Python side of the oracle for the Airwave HRIR convolution hot path:

* ctypes bindings to ``oracle/airwave_oracle.c`` (float32 restatement of
  ``ConvolutionEngine`` / ``RealtimeAudioProcessor`` and the N-speaker downmix);
* the authoritative float64 truth (direct / float64-FFT linear convolution);
* numpy/pure-Python restatements of the host-side data model the hot path consumes:
  ``InputLayout`` / ``HRIRChannelMap`` (Airwave/VirtualSpeaker.swift:59-346), the WAV
  decode contract (Airwave/WAVLoader.swift:26-99), ``Resampler`` (Airwave/Resampler.swift:31-68)
  and the renderer assembly rules of ``HRIRManager.activatePreset``
  (Airwave/HRIRManager.swift:347-446).

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import
this module.  The product package ``airwave_amd`` never does.

Parity pin: the reference is Swift on Apple's closed Accelerate/AVFoundation frameworks and can be
neither compiled nor imported here.  The oracle is pinned against every known-answer test the
reference holds for the path (tests/test_oracle_reference_kats.py re-expresses
AirwaveTests/ConvolutionEngineTests.swift:12-59 and
AirwaveTests/RealtimeAudioProcessorTests.swift:59-126) and, where the reference holds no vector
(real HRIRs, channel maps, WAV decode, resampler: "parity unpinned" by the reference itself),
against the mathematical definition (float64 linear convolution).
"""
from __future__ import annotations

import ctypes
import os
import struct
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libairwave_oracle.so")
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile oracle/airwave_oracle.c (gcc) into oracle/_build/."""
    src = os.path.join(_HERE, "airwave_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "airwave_oracle.h")))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = ctypes.CDLL(_LIB_PATH)
    L.orc_engine_create.restype = ctypes.c_void_p
    L.orc_engine_create.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int]
    L.orc_engine_destroy.argtypes = [ctypes.c_void_p]
    L.orc_engine_process.argtypes = [ctypes.c_void_p, c_float_p, c_float_p]
    L.orc_engine_process_accumulate.argtypes = [ctypes.c_void_p, c_float_p, c_float_p]
    L.orc_engine_reset.argtypes = [ctypes.c_void_p]
    L.orc_engine_block_size.argtypes = [ctypes.c_void_p]
    L.orc_engine_partition_count.argtypes = [ctypes.c_void_p]
    L.orc_realtime_create.restype = ctypes.c_void_p
    L.orc_realtime_create.argtypes = [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.orc_realtime_destroy.argtypes = [ctypes.c_void_p]
    L.orc_realtime_process.argtypes = [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int]
    L.orc_realtime_reset.argtypes = [ctypes.c_void_p]
    L.orc_spatializer_create.restype = ctypes.c_void_p
    L.orc_spatializer_create.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p, ctypes.c_int]
    L.orc_spatializer_destroy.argtypes = [ctypes.c_void_p]
    L.orc_spatializer_process.argtypes = [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_int64]
    L.orc_spatializer_reset.argtypes = [ctypes.c_void_p]
    L.orc_spatializer_batch.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int32_p, c_int32_p,
                                        ctypes.c_int, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    L.orc_direct_conv_f64.argtypes = [c_float_p, ctypes.c_int64, ctypes.c_int64, c_float_p, ctypes.c_int, c_double_p, ctypes.c_int]
    L.orc_synth_value.restype = ctypes.c_float
    L.orc_synth_value.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    L.orc_synth_fill.argtypes = [c_float_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64]
    _lib = L
    return L


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


# ------------------------------------------------------------------------------------------------
# C-backed restatements
# ------------------------------------------------------------------------------------------------
class ConvolutionEngine:
    """Airwave/ConvolutionEngine.swift:14-408 (float32, packed-format UPOLS)."""

    def __init__(self, hrir_samples, block_size: int = 512):
        h = _f32(hrir_samples)
        self._h = lib().orc_engine_create(_fp(h), int(h.size), int(block_size))
        if not self._h:
            raise ValueError("ConvolutionEngine init failed (init? returned nil)")
        self.block_size = int(block_size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_engine_destroy(self._h)
            self._h = None

    @property
    def partition_count(self) -> int:
        return lib().orc_engine_partition_count(self._h)

    def process(self, block) -> np.ndarray:
        x = _f32(block)
        # process(input:[Float], output:, frameCount:) silently returns when count != blockSize
        # (ConvolutionEngine.swift:370-373); here that is an assertion because tests want to know.
        assert x.size == self.block_size
        y = np.zeros(self.block_size, dtype=np.float32)
        lib().orc_engine_process(self._h, _fp(x), _fp(y))
        return y

    def process_and_accumulate(self, block, acc: np.ndarray) -> None:
        x = _f32(block)
        assert x.size == self.block_size and acc.dtype == np.float32 and acc.size == self.block_size
        lib().orc_engine_process_accumulate(self._h, _fp(x), _fp(acc))

    def reset(self) -> None:
        lib().orc_engine_reset(self._h)


class RealtimeAudioProcessor:
    """Airwave/RealtimeAudioProcessor.swift:11-191.  `renderers` = [(left_ir, right_ir), ...]."""

    def __init__(self, renderers: Sequence[Tuple[Sequence[float], Sequence[float]]], block_size: int = 512,
                 max_frames_per_callback: int = 4096):
        n = len(renderers)
        ir_len = max([1] + [max(len(l), len(r)) for l, r in renderers])
        left = np.zeros((max(n, 1), ir_len), dtype=np.float32)
        right = np.zeros((max(n, 1), ir_len), dtype=np.float32)
        for i, (l, r) in enumerate(renderers):
            left[i, : len(l)] = l
            right[i, : len(r)] = r
        self._h = lib().orc_realtime_create(_fp(left), _fp(right), n, ir_len, block_size, max_frames_per_callback)
        if not self._h:
            raise ValueError("RealtimeAudioProcessor create failed")
        self.block_size = block_size
        self.max_frames_per_callback = max_frames_per_callback

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_realtime_destroy(self._h)
            self._h = None

    def process(self, input_left, input_right=None) -> Tuple[np.ndarray, np.ndarray]:
        l = _f32(input_left)
        r = None if input_right is None else _f32(input_right)
        n = int(l.size)
        out_l = np.full(n, np.nan, dtype=np.float32)
        out_r = np.full(n, np.nan, dtype=np.float32)
        rc = lib().orc_realtime_process(self._h, _fp(l), None if r is None else _fp(r), _fp(out_l), _fp(out_r), n)
        if rc != 0:
            raise ValueError("frameCount exceeds maxFramesPerCallback (reference precondition)")
        return out_l, out_r

    def reset(self) -> None:
        lib().orc_realtime_reset(self._h)


def spatialize_f32(x: np.ndarray, tracks: np.ndarray, left_track, right_track, block_size: int = 512,
                   threads: int = 1) -> np.ndarray:
    """float32 restatement of the per-block engine network.  x: [streams][frames][C] (or [frames][C])."""
    squeeze = x.ndim == 2
    xs = _f32(x[None] if squeeze else x)
    S, F, C = xs.shape
    tr = _f32(tracks)
    lt = np.ascontiguousarray(np.asarray(left_track, dtype=np.int32))
    rt = np.ascontiguousarray(np.asarray(right_track, dtype=np.int32))
    assert lt.size == C and rt.size == C
    pad = (-F) % block_size
    if pad:
        xs = np.concatenate([xs, np.zeros((S, pad, C), dtype=np.float32)], axis=1)
    out = np.zeros((S, F + pad, 2), dtype=np.float32)
    rc = lib().orc_spatializer_batch(_fp(tr), tr.shape[0], tr.shape[1], C, lt.ctypes.data_as(c_int32_p),
                                     rt.ctypes.data_as(c_int32_p), block_size, _fp(xs), _fp(out), S, F + pad, threads)
    if rc != 0:
        raise ValueError("oracle spatializer failed (invalid channel mapping or no renderers)")
    out = out[:, :F]
    return out[0] if squeeze else out


# ------------------------------------------------------------------------------------------------
# float64 truth
# ------------------------------------------------------------------------------------------------
def direct_conv_f64(x, h) -> np.ndarray:
    """y[n] = sum_k h[k] x[n-k] in float64, first len(x) samples (streaming linear convolution)."""
    x = np.asarray(x, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    if x.size * h.size <= 1 << 26:
        return np.convolve(x, h)[: x.size]
    from scipy.signal import oaconvolve  # float64 FFT: error ~1e-15, far below the 1e-5 budget
    return oaconvolve(x, h)[: x.size]


def spatialize_f64(x: np.ndarray, tracks: np.ndarray, left_track, right_track) -> np.ndarray:
    """Truth for the N-speaker downmix.  x: [frames][C] -> [frames][2] float64.
    out_L = sum_c x_c * h[left_track[c]], out_R likewise; unmapped (<0) channels skipped."""
    x = np.asarray(x)
    F, C = x.shape
    out = np.zeros((F, 2), dtype=np.float64)
    for c in range(C):
        l, r = int(left_track[c]), int(right_track[c])
        if l < 0 or r < 0:
            continue
        out[:, 0] += direct_conv_f64(x[:, c], tracks[l])
        out[:, 1] += direct_conv_f64(x[:, c], tracks[r])
    return out


def peak_rel_error(y, ref) -> float:
    """max|y - ref| / max|ref| per array (pointwise relative error is meaningless at
    zero crossings)."""
    ref = np.asarray(ref, dtype=np.float64)
    den = float(np.max(np.abs(ref)))
    num = float(np.max(np.abs(np.asarray(y, dtype=np.float64) - ref)))
    return num / den if den > 0 else num


# ------------------------------------------------------------------------------------------------
# synthetic input
# ------------------------------------------------------------------------------------------------
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


SYNTH_SEED = 0xA17AE


def synth_input(n_streams: int, frames: int, n_channels: int, seed: int = SYNTH_SEED, first_stream: int = 0) -> np.ndarray:
    """U(-0.5, 0.5) counter-based input, identical to orc_synth_fill and the HIP fill kernel."""
    per = frames * n_channels
    idx = np.arange(per, dtype=np.uint64)
    out = np.empty((n_streams, frames, n_channels), dtype=np.float32)
    with np.errstate(over="ignore"):
        for s in range(n_streams):
            key = (np.uint64(seed) + np.uint64(first_stream + s)) * _GOLDEN + idx
            top = (_splitmix64(key) >> np.uint64(40)).astype(np.float32)
            out[s] = (top * np.float32(1.0 / 16777216.0) - np.float32(0.5)).reshape(frames, n_channels)
    return out


def synth_hrir(n_tracks: int = 14, taps: int = 4320, seed: int = 1234, tau: Optional[float] = None) -> np.ndarray:
    """Seeded exponentially-decaying noise HRIR (SURVEY.md §8c/§8d cfg 3: tau = taps/6)."""
    rng = np.random.default_rng(seed)
    tau = taps / 6.0 if tau is None else tau
    env = np.exp(-np.arange(taps) / tau)
    h = rng.standard_normal((n_tracks, taps)) * env
    h /= np.sqrt((h ** 2).sum(axis=1, keepdims=True))
    return h.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# VirtualSpeaker / InputLayout / HRIRChannelMap   (Airwave/VirtualSpeaker.swift)
# ------------------------------------------------------------------------------------------------
NAMED_SPEAKERS = ["FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR", "TFL", "TFR", "TBL", "TBR", "FLC", "FRC", "BC"]  # :11-31

_LEFT_SIDE = {"FL", "BL", "SL", "TFL", "TBL", "FLC"}      # :143
_RIGHT_SIDE = {"FR", "BR", "SR", "TFR", "TBR", "FRC"}     # :148


def layout_detect(channel_count: int) -> List[str]:
    """InputLayout.detect(channelCount:)  VirtualSpeaker.swift:88-99.  Custom speakers are "Ch<i>"."""
    if channel_count == 2:
        return ["FL", "FR"]
    if channel_count == 6:
        return ["FL", "FR", "FC", "LFE", "BL", "BR"]
    if channel_count == 8:
        return ["FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR"]
    if channel_count == 12:
        return ["FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR", "TFL", "TFR", "TBL", "TBR"]
    return [f"Ch{i}" for i in range(channel_count)]


ChannelMap = Dict[str, Tuple[int, int]]


def map_hesuvi14(speakers: Sequence[str]) -> ChannelMap:
    """hesuvi14Channel  VirtualSpeaker.swift:270-297"""
    table = {"FL": (0, 1), "FR": (8, 7), "FC": (6, 13), "LFE": (6, 13), "BL": (4, 5), "BR": (12, 11),
             "SL": (2, 3), "SR": (10, 9)}
    return {s: table[s] for s in speakers if s in table}


def map_hesuvi7(speakers: Sequence[str]) -> ChannelMap:
    """hesuvi7Channel  VirtualSpeaker.swift:224-250"""
    table = {"FL": (0, 1), "FR": (1, 0), "FC": (2, 2), "LFE": (2, 2), "BL": (3, 4), "BR": (4, 3),
             "SL": (5, 6), "SR": (6, 5)}
    return {s: table[s] for s in speakers if s in table}


def map_interleaved_pairs(speakers: Sequence[str]) -> ChannelMap:
    """interleavedPairs  VirtualSpeaker.swift:126-159 (right-side speakers swap the pair)."""
    m: ChannelMap = {}
    for i, s in enumerate(speakers):
        b = 2 * i
        m[s] = (b + 1, b) if s in _RIGHT_SIDE else (b, b + 1)
    return m


def map_split_blocks(speakers: Sequence[str]) -> ChannelMap:
    """splitBlocks  VirtualSpeaker.swift:200-209"""
    n = len(speakers)
    return {s: (i, i + n) for i, s in enumerate(speakers)}


_ALIASES = {"FL": "FL", "L": "FL", "FR": "FR", "R": "FR", "FC": "FC", "C": "FC", "LFE": "LFE", "SUB": "LFE",
            "BL": "BL", "RL": "BL", "BR": "BR", "RR": "BR", "SL": "SL", "SR": "SR", "TFL": "TFL", "TFR": "TFR",
            "TBL": "TBL", "TBR": "TBR"}


def _swift_int(text: str) -> Optional[int]:
    """Swift's Int(String): optional sign then ASCII digits only, no whitespace."""
    t = text
    if t[:1] in "+-":
        t = t[1:]
    if not t or not all("0" <= ch <= "9" for ch in t):
        return None
    v = int(text)
    return v if -(1 << 63) <= v < (1 << 63) else None          # Int(String) is nil on Int64 overflow


def parse_hesuvi_format(text: str) -> ChannelMap:
    """parseHeSuViFormat  VirtualSpeaker.swift:301-346.  Unknown names become custom speakers keyed by
    the name as written; a later line for the same speaker overwrites an earlier one."""
    m: ChannelMap = {}
    # components(separatedBy: .newlines) splits at every newline scalar (U+000A-000D, U+0085, U+2028, U+2029; CR LF gives an empty
    # component, which is skipped); .whitespaces is general category Zs plus TAB; String.uppercased() is the full Unicode mapping
    # (a long s or an "fl" ligature therefore reaches the alias table: Python's str.upper() is the same mapping)
    ws = " \t\u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u202f\u205f\u3000"
    for line in _re.split("[\n\x0b\x0c\r\x85\u2028\u2029]", text):
        t = line.strip(ws)
        if not t or t.startswith("#") or t.startswith(";"):
            continue
        parts = t.split("=")
        if len(parts) != 2:
            continue
        name = parts[0].strip(ws)
        idx = [v for v in (_swift_int(p.strip(ws)) for p in parts[1].strip(ws).split(",")) if v is not None]
        if len(idx) != 2:
            continue
        speaker = _ALIASES.get(name.upper(), name)
        m[speaker] = (idx[0], idx[1])
    return m


# ------------------------------------------------------------------------------------------------
# WAVLoader contract (Airwave/WAVLoader.swift:26-99): any WAV -> planar float32 [channels][frames]
# ------------------------------------------------------------------------------------------------
@dataclass
class WAVData:
    sample_rate: float
    channel_count: int
    frame_count: int
    audio_data: np.ndarray  # [channels][frames] float32


def wav_load(path: str) -> WAVData:
    b = open(path, "rb").read()
    if len(b) < 12 or b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("fileReadError: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, sz = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        body = b[pos + 8:pos + 8 + sz]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
        pos += 8 + sz + (sz & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError("fileReadError: missing fmt/data chunk")
    tag, ch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:
        tag = struct.unpack("<H", fmt[24:26])[0]
    if ch <= 0:
        raise ValueError("invalidChannelCount")
    bps = bits // 8
    frames = len(data) // (bps * ch) if bps else 0
    if frames <= 0:
        raise ValueError("emptyFile")
    raw = data[: frames * bps * ch]
    if tag == 3 and bits == 32:
        a = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        a = np.frombuffer(raw, dtype="<f8").astype(np.float32)
    elif tag == 1 and bits == 16:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / np.float32(32768.0)            # :77
    elif tag == 1 and bits == 32:
        a = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)  # :86
    elif tag == 1 and bits == 24:
        u = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = u[:, 0] | (u[:, 1] << 8) | (u[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        a = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif tag == 1 and bits == 8:
        a = ((np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0).astype(np.float32)
    else:
        raise ValueError("unsupportedFormat")
    planar = np.ascontiguousarray(a.reshape(frames, ch).T)
    return WAVData(float(rate), int(ch), int(frames), planar)


# ------------------------------------------------------------------------------------------------
# Resampler (Airwave/Resampler.swift:31-68)
# ------------------------------------------------------------------------------------------------
def resample_output_count(count: int, from_rate: float, to_rate: float) -> int:
    stride = from_rate / to_rate                       # :37
    return int(float(count) / stride)                  # :38


def resample_intended(x, from_rate: float, to_rate: float) -> np.ndarray:
    """What Resampler.swift's comments (:49-50) intend: out[i] = lerp(input, i * fromRate/toRate).
    The control ramp is float32 (vDSP_vramp with Float start/step, :54-56); positions past the last
    sample hold it."""
    x = _f32(x)
    if abs(from_rate - to_rate) < 0.01:                # :33
        return x.copy()
    n_out = resample_output_count(x.size, from_rate, to_rate)
    if n_out <= 0:
        return np.zeros(0, dtype=np.float32)
    step = np.float32(from_rate / to_rate)
    pos = (np.arange(n_out, dtype=np.float32) * step).astype(np.float32)     # vDSP_vramp :56
    i0 = np.floor(pos).astype(np.int64)
    frac = (pos - i0.astype(np.float32)).astype(np.float32)
    i0c = np.minimum(i0, x.size - 1)
    i1c = np.minimum(i0 + 1, x.size - 1)
    return (x[i0c] + frac * (x[i1c] - x[i0c])).astype(np.float32)


def resample_vgenp_literal(x, from_rate: float, to_rate: float) -> np.ndarray:
    """The call Resampler.swift:56-65 literally makes: control = vDSP_vramp(0, Float(stride)) (outputCount entries), then
    vDSP_vgenp(A=input, B=control, C=output, N=outputCount, M=input.count).  vgenp as Apple documents it: the knots
    (B[m], A[m]) define a piecewise-linear function evaluated at the integers n = 0..N-1; C[n] = A[0] for n <= B[0] and
    A[M-1] for n > B[M-1].  Written as the knot search, not as a closed form.  (For stride > 1 the evaluation points never
    need a knot index >= outputCount, so the entries vgenp would read past the control array do not matter.)
    Unpinned by the reference's tests and by Apple's closed implementation; kept to show what the shipped code computes."""
    x = _f32(x)
    if abs(from_rate - to_rate) < 0.01:
        return x.copy()
    n_out = resample_output_count(x.size, from_rate, to_rate)
    if n_out <= 0:
        return np.zeros(0, dtype=np.float32)
    step = np.float32(from_rate / to_rate)
    M = x.size
    knots = (np.arange(M, dtype=np.float32) * step).astype(np.float32)          # B[m] (the reference allocates only n_out of them)
    out = np.empty(n_out, dtype=np.float32)
    m = 0
    for n in range(n_out):
        xn = np.float32(n)
        if xn <= knots[0]:
            out[n] = x[0]
        elif xn > knots[M - 1]:
            out[n] = x[M - 1]
        else:
            while m + 1 < M - 1 and knots[m + 1] < xn:
                m += 1
            b0, b1 = knots[m], knots[m + 1]
            out[n] = np.float32(x[m] + (x[m + 1] - x[m]) * np.float32((xn - b0) / (b1 - b0)))
    return out


# ------------------------------------------------------------------------------------------------
# Renderer assembly (HRIRManager.activatePreset, Airwave/HRIRManager.swift:347-446)
# ------------------------------------------------------------------------------------------------
class InvalidChannelMapping(ValueError):
    pass


class ConvolutionSetupFailed(ValueError):
    pass


def assemble_tracks(wav: WAVData, speakers: Sequence[str], target_rate: Optional[float] = None,
                    channel_map: Optional[ChannelMap] = None):
    """Returns (tracks[n_tracks][taps], left_track[C], right_track[C]) with -1 for skipped speakers.
    Map choice :355-360 (7 tracks -> hesuvi7 else hesuvi14), skip :370-372, bounds :375-379,
    resample when |wavRate - targetRate| > 0.01 :389-403, empty -> convolutionSetupFailed :420-422."""
    cmap = channel_map if channel_map is not None else (
        map_hesuvi7(speakers) if wav.channel_count == 7 else map_hesuvi14(speakers))
    left = np.full(len(speakers), -1, dtype=np.int32)
    right = np.full(len(speakers), -1, dtype=np.int32)
    any_mapped = False
    for i, s in enumerate(speakers):
        if s not in cmap:
            continue
        l, r = cmap[s]
        if not (l < wav.channel_count and r < wav.channel_count) or l < 0 or r < 0:
            raise InvalidChannelMapping(f"HRIR indices ({l}, {r}) out of range for {wav.channel_count} channels")
        left[i], right[i] = l, r
        any_mapped = True
    if not any_mapped:
        raise ConvolutionSetupFailed("No valid renderers created")
    tracks = wav.audio_data
    if target_rate is not None and abs(wav.sample_rate - target_rate) > 0.01:
        tracks = np.stack([resample_intended(t, wav.sample_rate, target_rate) for t in wav.audio_data])
    return np.ascontiguousarray(tracks, dtype=np.float32), left, right


# ================================================================================================
# Parametric EQ ("next" row) — restatement of BiquadCoefficientBuilder.swift,
# EqualizerAPOParser.swift, ParametricEqualizerProcessor.swift.  Pinned by the reference's own golden
# numbers (tests/test_oracle_eq_kats.py re-expresses ParametricEqualizerProcessorTests.swift and
# EqualizerAPOParserTests.swift).
# ================================================================================================
import re as _re

PEAKING, LOW_SHELF, HIGH_SHELF = 0, 1, 2
_BIQUAD_ERRORS = {1: "invalidSampleRate", 2: "invalidFrequency", 3: "invalidQ", 4: "nonFiniteInput", 5: "nonFiniteCoefficients"}


class _Biquad(ctypes.Structure):
    _fields_ = [("b0", ctypes.c_double), ("b1", ctypes.c_double), ("b2", ctypes.c_double), ("a1", ctypes.c_double), ("a2", ctypes.c_double)]


def _eq_lib():
    L = lib()
    if not hasattr(L, "_eq_typed"):
        L.orc_biquad_make.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(_Biquad)]
        L.orc_eq_state_create.restype = ctypes.c_void_p
        L.orc_eq_state_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.POINTER(_Biquad), ctypes.c_int]
        L.orc_eq_state_destroy.argtypes = [ctypes.c_void_p]
        L.orc_eq_state_reset.argtypes = [ctypes.c_void_p]
        L.orc_eq_state_process.argtypes = [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int]
        L._eq_typed = True
    return L


class BiquadCoefficientError(ValueError):
    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


def biquad_make(ftype: int, gain_db: float, frequency_hz: float, q: float, sample_rate: float):
    """BiquadCoefficientBuilder.make  BiquadCoefficientBuilder.swift:29-107 -> (b0, b1, b2, a1, a2)"""
    out = _Biquad()
    rc = _eq_lib().orc_biquad_make(ftype, gain_db, frequency_hz, q, sample_rate, ctypes.byref(out))
    if rc != 0:
        raise BiquadCoefficientError(_BIQUAD_ERRORS[rc])
    return (out.b0, out.b1, out.b2, out.a1, out.a2)


@dataclass
class EqualizerFilter:            # EqualizerPreset.swift:9-17
    source_line: int
    source_number: Optional[int]
    is_enabled: bool
    type: int
    frequency_hz: float
    gain_db: float
    q: float


@dataclass
class EqualizerDefinition:        # EqualizerPreset.swift:19-27
    preamp_db: float = 0.0
    filters: List[EqualizerFilter] = field(default_factory=list)


class EqualizerParseError(ValueError):
    def __init__(self, filename, issues):
        self.filename, self.issues = filename, issues     # issues: [(line or None, reason)]
        det = "; ".join((f"line {l}: {r}" if l is not None else r) for l, r in issues)
        super().__init__(f"Could not read {filename}: {det}")


# NSRegularExpression is ICU: \s is [\t\n\f\r\p{Z}] (NOT Python's str.isspace(), which adds U+001C-001F, VT and NEL), and
# .caseInsensitive compares simple case foldings: besides the ASCII pairs, U+212A KELVIN SIGN folds to k and U+017F LONG S to s
# (Python's re.IGNORECASE would also pair i with U+0130 / U+0131, which ICU does not) — so both are spelled out here.
_S = "[\\t\\n\\f\\r \u00a0\u1680\u2000-\u200a\u2028\u2029\u202f\u205f\u3000]"
_NS = "[^" + _S[1:]
_FOLD = {"k": "Kk\u212a", "s": "Ss\u017f"}


def _icu(pattern: str) -> "_re.Pattern":
    out, i = [], 0
    while i < len(pattern):
        c = pattern[i]
        if c == "\\":
            out.append({"s": _S, "S": _NS}.get(pattern[i + 1], pattern[i:i + 2])); i += 2; continue
        if c == "[":                           # [0-9]: as written
            j = pattern.index("]", i); out.append(pattern[i:j + 1]); i = j + 1; continue
        out.append("[" + _FOLD.get(c.lower(), c.upper() + c.lower()) + "]" if c.isalpha() else c)
        i += 1
    return _re.compile("".join(out))


_PREAMP_RE = _icu(r"^Preamp\s*:\s*(\S+)\s+dB$")
_FILTER_RE = _icu(r"^Filter(?:\s+([0-9]+))?\s*:\s+(ON|OFF)\s+(PK|LSC|HSC)\s+Fc\s+(\S+)\s+Hz\s+Gain\s+(\S+)\s+dB\s+Q\s+(\S+)$")
# CharacterSet.whitespacesAndNewlines: general category Z*, TAB, U+000A-000D, U+0085
_TRIM = "\t\n\x0b\x0c\r\x85 \u00a0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000"


def _finite_double(text: str) -> Optional[float]:
    """Swift Double(String): decimal / exponent / hex forms, 'inf'/'nan' spellings parse but are rejected as
    non-finite; no surrounding whitespace, no underscores."""
    if not _re.fullmatch(r"[+-]?(([0-9]+\.?[0-9]*([eE][+-]?[0-9]+)?)|(\.[0-9]+([eE][+-]?[0-9]+)?)|(0[xX][0-9a-fA-F]+\.?[0-9a-fA-F]*([pP][+-]?[0-9]+)?)|inf|infinity|nan)", text, _re.IGNORECASE):
        return None
    try:
        v = float.fromhex(text) if _re.match(r"[+-]?0[xX]", text) else float(text)
    except ValueError:
        return None
    return v if np.isfinite(v) else None


def eq_parse(data: bytes, filename: str = "preset.txt") -> EqualizerDefinition:
    """EqualizerAPOParser.parse  EqualizerAPOParser.swift:36-151"""
    if len(data) > 1_048_576:
        raise EqualizerParseError(filename, [(None, "file exceeds the 1 MiB limit")])
    try:
        source = data.decode("utf-8")
    except UnicodeDecodeError:
        raise EqualizerParseError(filename, [(None, "file is not valid UTF-8")])
    if source[:1] == "﻿":
        source = source[1:]
    preamp, has_preamp, decl = 0.0, False, 0
    filters: List[EqualizerFilter] = []
    issues = []
    # components(separatedBy: .newlines) splits at EVERY newline scalar (CR LF therefore yields an empty
    # component and advances the line number twice) — unpinned by the reference tests, kept literal.
    for index, raw in enumerate(_re.split("[\n\r\x0b\x0c\x85\u2028\u2029]", source)):
        n = index + 1
        line = raw.strip(_TRIM)                      # :60
        if not line or line.startswith("#"):
            continue
        m = _PREAMP_RE.fullmatch(line)
        if m:
            if has_preamp:
                issues.append((n, "duplicate Preamp directive")); continue
            v = _finite_double(m.group(1))
            if v is None:
                issues.append((n, "Preamp must be a finite number")); continue
            preamp, has_preamp = v, True
            continue
        if line.lower().startswith("filter"):
            decl += 1
            if decl > 64:
                issues.append((n, "more than 64 filter declarations are not allowed")); continue
            m = _FILTER_RE.fullmatch(line)
            if not m:
                issues.append((n, "malformed Filter directive")); continue
            number = int(m.group(1)) if m.group(1) else None
            enabled = m.group(2).upper() == "ON"
            ftype = {"PK": PEAKING, "LSC": LOW_SHELF, "HSC": HIGH_SHELF}.get(m.group(3).upper())     # :91-99 ("P" + KELVIN SIGN matches, and is no type)
            if ftype is None:
                issues.append((n, "unsupported filter type")); continue
            f, g, q = _finite_double(m.group(4)), _finite_double(m.group(5)), _finite_double(m.group(6))
            num = []
            if f is not None:
                if f <= 0: num.append("frequency must be positive")
            else:
                num.append("frequency must be a finite number")
            if g is None: num.append("gain must be a finite number")
            if q is not None:
                if q <= 0: num.append("Q must be positive")
            else:
                num.append("Q must be a finite number")
            if num:
                issues.extend((n, r) for r in num); continue
            filters.append(EqualizerFilter(n, number, enabled, ftype, f, g, q))
            continue
        issues.append((n, "malformed Preamp directive" if line.lower().startswith("preamp") else "unsupported directive"))
    if not issues and preamp == 0 and not any(f.is_enabled for f in filters):
        issues.append((None, "effective configuration must contain a non-zero preamp or an enabled supported filter"))
    if issues:
        raise EqualizerParseError(filename, issues)
    return EqualizerDefinition(preamp, filters)


class EqualizerPreparationError(ValueError):
    def __init__(self, kind, detail=""):
        super().__init__(f"{kind} {detail}".strip())
        self.kind = kind


class ParametricEqualizerState:
    """ParametricEqualizerState  ParametricEqualizerProcessor.swift:16-98 (C-backed)."""

    def __init__(self, sample_rate: float, preamp_db: float, coefficients):
        arr = (_Biquad * max(len(coefficients), 1))(*[_Biquad(*c) for c in coefficients])
        self._h = _eq_lib().orc_eq_state_create(sample_rate, preamp_db, arr, len(coefficients))
        self.sample_rate, self.filter_count = sample_rate, len(coefficients)
        self.preamp_linear = 10.0 ** (preamp_db / 20.0)

    def __del__(self):
        if getattr(self, "_h", None):
            _eq_lib().orc_eq_state_destroy(self._h)
            self._h = None

    def reset(self):
        _eq_lib().orc_eq_state_reset(self._h)

    def process(self, left, right=None):
        l = _f32(left)
        r = None if right is None else _f32(right)
        ol = np.full(l.size, np.nan, np.float32)
        orr = np.full(l.size, np.nan, np.float32)
        _eq_lib().orc_eq_state_process(self._h, _fp(l), None if r is None else _fp(r), _fp(ol), _fp(orr), l.size)
        return ol, orr


def eq_prepare(definition: Optional[EqualizerDefinition], sample_rate: float) -> ParametricEqualizerState:
    """ParametricEqualizerProcessor.prepare  :168-212"""
    if not np.isfinite(sample_rate) or sample_rate <= 0:
        raise EqualizerPreparationError("invalidSampleRate")
    preamp = definition.preamp_db if definition else 0.0
    # pow(10, preampDB / 20) :176-179 — Python raises OverflowError where C returns +inf; a hugely NEGATIVE preamp underflows to 0, which is
    # finite and accepted by the reference (found by tools/fuzz_host.py in round 6: the guard used to treat |preamp| >= 1e4 as infinite)
    if not np.isfinite(preamp) or preamp / 20.0 > 308.0:
        raise EqualizerPreparationError("nonFinitePreamp")
    enabled = [f for f in (definition.filters if definition else []) if f.is_enabled]
    if len(enabled) > 64:
        raise EqualizerPreparationError("tooManyFilters", str(len(enabled)))
    coeffs = []
    for i, f in enumerate(enabled):
        try:
            coeffs.append(biquad_make(f.type, f.gain_db, f.frequency_hz, f.q, sample_rate))
        except BiquadCoefficientError as e:
            raise EqualizerPreparationError("invalidFilter", f"{i}: {e.kind}")
    return ParametricEqualizerState(sample_rate, preamp, coeffs)


class ParametricEqualizerProcessor:
    """ParametricEqualizerProcessor  :116-408: non-blocking target publication, 20 ms linear crossfades,
    newest-wins queueing, one-slot retirement that the control side drains."""

    CROSSFADE_SECONDS = 0.020
    MAX_CALLBACK_FRAMES = 4096

    def __init__(self, sample_rate: float, max_frames_per_callback: int = 4096):
        if not np.isfinite(sample_rate) or sample_rate <= 0:
            raise EqualizerPreparationError("invalidSampleRate")
        if not (0 < max_frames_per_callback <= self.MAX_CALLBACK_FRAMES):
            raise EqualizerPreparationError("tooManyFilters", str(max_frames_per_callback))
        self.sample_rate, self.max_frames = sample_rate, max_frames_per_callback
        self.unity = eq_prepare(None, sample_rate)
        self.active = self.unity
        self.transition_length = max(1, int(np.floor(sample_rate * self.CROSSFADE_SECONDS + 0.5)))   # .rounded(): half away from zero
        self.target = None            # published target (control side)
        self.audio_target = None
        self.observed = None
        self.t_from = self.t_to = None
        self.pending_target = None
        self.pending_retirement = None
        self.retired = None           # the one-slot retirement box
        self.t_frame = 0
        self.reset_requested = False

    def set_target(self, definition):                       # :226-228
        self.target = eq_prepare(definition, self.sample_rate)

    def reset(self):                                        # :230-234
        self.reset_requested = True

    def drain_retired_states(self):                         # :237-241
        self.retired = None

    def process(self, left, right=None):                    # :244-309
        l = _f32(left)
        r = None if right is None else _f32(right)
        n = l.size
        if n == 0:
            return l.copy(), l.copy()
        if n > self.max_frames:
            raise ValueError("frameCount exceeds maxFramesPerCallback")
        self._observe()
        self._flush_pending_retirement()
        if self.reset_requested:
            self.reset_requested = False
            for s in (self.active, self.t_from, self.t_to):
                if s is not None:
                    s.reset()
        out_l = np.empty(n, np.float32)
        out_r = np.empty(n, np.float32)
        off = 0
        while off < n:
            if self.t_from is None or self.t_to is None:
                a, b = self.active.process(l[off:], None if r is None else r[off:])
                out_l[off:], out_r[off:] = a, b
                return out_l, out_r
            seg = min(self.transition_length - self.t_frame, n - off)
            ol, orr = self.t_from.process(l[off:off + seg], None if r is None else r[off:off + seg])
            nl, nr = self.t_to.process(l[off:off + seg], None if r is None else r[off:off + seg])
            prog = (self.t_frame + np.arange(seg) + 1).astype(np.float64) / float(self.transition_length)
            inv = 1.0 - prog
            out_l[off:off + seg] = (ol.astype(np.float64) * inv + nl.astype(np.float64) * prog).astype(np.float32)
            out_r[off:off + seg] = (orr.astype(np.float64) * inv + nr.astype(np.float64) * prog).astype(np.float32)
            self.t_frame += seg
            off += seg
            if self.t_frame == self.transition_length:
                self._finish()
        return out_l, out_r

    def _observe(self):                                     # :311-333
        if self.target is not None:
            self.audio_target = self.target
        t = self.audio_target
        if t is None or t is self.observed:
            return
        self.observed = t
        if self.t_to is not None:
            if t is not self.t_to:
                self.pending_target = t
        elif self.pending_retirement is not None:
            self.pending_target = t
        elif t is not self.active:
            self._begin(t)

    def _begin(self, t):                                    # :349-354
        if t is self.active:
            return
        self.t_from, self.t_to, self.t_frame = self.active, t, 0

    def _finish(self):                                      # :356-371
        frm = self.t_from
        self.active, self.t_from, self.t_to, self.t_frame = self.t_to, None, None, 0
        if not self._retire(frm):
            return
        if self.pending_target is not None:
            p, self.pending_target = self.pending_target, None
            if p is not self.active:
                self._begin(p)

    def _retire(self, state):                               # :373-386
        if self.pending_retirement is not None:
            return False
        if self.retired is None:
            self.retired = state
            return True
        self.pending_retirement = state
        return False

    def _flush_pending_retirement(self):                    # :388-406
        if self.pending_retirement is None:
            return
        if self.retired is not None:
            return
        self.retired, self.pending_retirement = self.pending_retirement, None
        if self.pending_target is not None:
            p, self.pending_target = self.pending_target, None
            if p is not self.active:
                self._begin(p)
