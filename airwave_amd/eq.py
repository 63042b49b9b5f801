"""Python host mirror of the reference's parametric-EQ surface over the C ABI (include/airwave_hip.h,
"Parametric EQ" section).  Names follow the Swift sources: BiquadCoefficientBuilder,
EqualizerAPOParser, EqualizerDefinition/EqualizerFilter, ParametricEqualizerState,
ParametricEqualizerProcessor, EqualizerRuntimeEffect.  All arithmetic runs in the HIP library."""
from __future__ import annotations

import ctypes
import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _capi
from .api import AW_OK, AirwaveError, Context, _check, _f32, _finalizing, _fp, default_context

PEAKING, LOW_SHELF, HIGH_SHELF = 0, 1, 2
_BIQUAD_KINDS = {1: "invalidSampleRate", 2: "invalidFrequency", 3: "invalidQ", 4: "nonFiniteInput", 5: "nonFiniteCoefficients"}


class BiquadCoefficientError(AirwaveError):       # BiquadCoefficientBuilder.swift:11-27
    def __init__(self, status, message, kind):
        super().__init__(status, message)
        self.kind = _BIQUAD_KINDS.get(kind, str(kind))


class EqualizerParseError(AirwaveError):          # EqualizerAPOParser.swift:8-21
    def __init__(self, status, message, filename):
        super().__init__(status, message)
        self.filename = filename
        self.issues: List[Tuple[Optional[int], str]] = []
        for part in message.split("; "):
            m = re.match(r"line (\d+): (.*)", part)
            self.issues.append((int(m.group(1)), m.group(2)) if m else (None, part))
        self.errorDescription = f"Could not read {filename}: {message}"


class ParametricEqualizerPreparationError(AirwaveError):   # ParametricEqualizerProcessor.swift:100-115
    KINDS = {13: "invalidSampleRate", 14: "nonFinitePreamp", 15: "tooManyFilters", 16: "invalidFilter"}

    def __init__(self, status, message):
        super().__init__(status, message)
        self.kind = self.KINDS.get(status, str(status))
        m = re.search(r"\[kind (\d+), line (\d+)\]", message)
        self.filter_error = _BIQUAD_KINDS.get(int(m.group(1))) if m else None
        self.source_line = (int(m.group(2)) or None) if m else None


def _check_eq(status: int) -> None:
    if status in (13, 14, 15, 16):
        raise ParametricEqualizerPreparationError(status, (_capi.load().aw_last_error_message() or b"").decode("utf-8", "replace"))
    _check(status)


class BiquadCoefficientBuilder:
    @staticmethod
    def make(type: int, gainDB: float, frequencyHz: float, q: float, sampleRate: float) -> Tuple[float, float, float, float, float]:
        out = (ctypes.c_double * 5)()
        kind = ctypes.c_int32()
        lib = _capi.load()
        st = lib.aw_biquad_make(type, gainDB, frequencyHz, q, sampleRate, out, ctypes.byref(kind))
        if st != AW_OK:
            raise BiquadCoefficientError(st, (lib.aw_last_error_message() or b"").decode(), kind.value)
        return tuple(out)


@dataclass
class EqualizerFilter:                            # EqualizerPreset.swift:9-17
    sourceLine: int
    sourceNumber: Optional[int]
    isEnabled: bool
    type: int
    frequencyHz: float
    gainDB: float
    q: float


@dataclass
class EqualizerDefinition:                        # EqualizerPreset.swift:19-27
    preampDB: float = 0.0
    filters: List[EqualizerFilter] = field(default_factory=list)

    def _handle(self):
        lib = _capi.load()
        h = ctypes.c_void_p()
        _check(lib.aw_eq_definition_create(self.preampDB, ctypes.byref(h)))
        for i, f in enumerate(self.filters):
            _check(lib.aw_eq_definition_add_filter(h, int(f.isEnabled), f.type, f.frequencyHz, f.gainDB, f.q))
            _check(lib.aw_eq_definition_set_source(h, i, f.sourceLine, -1 if f.sourceNumber is None else f.sourceNumber))
        return h


class _DefHandle:
    def __init__(self, definition: Optional[EqualizerDefinition]):
        self.h = definition._handle() if definition is not None else None

    def __enter__(self):
        return self.h

    def __exit__(self, *a):
        if self.h is not None:
            _capi.load().aw_eq_definition_destroy(self.h)


class EqualizerAPOParser:
    maximumDataSize = 1_048_576
    maximumFilterCount = 64

    @staticmethod
    def parse(data: bytes, filename: str = "preset.txt") -> EqualizerDefinition:
        lib = _capi.load()
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(16384)
        st = lib.aw_eq_parse(data, len(data), ctypes.byref(h), buf, len(buf))
        if st == 12:
            raise EqualizerParseError(st, buf.value.decode("utf-8", "replace"), filename)
        _check(st)
        try:
            d = EqualizerDefinition(lib.aw_eq_definition_preamp_db(h), [])
            for i in range(lib.aw_eq_definition_filter_count(h)):
                line, num, en, ty = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
                f, g, q = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
                _check(lib.aw_eq_definition_filter(h, i, ctypes.byref(line), ctypes.byref(num), ctypes.byref(en), ctypes.byref(ty),
                                                   ctypes.byref(f), ctypes.byref(g), ctypes.byref(q)))
                d.filters.append(EqualizerFilter(line.value, None if num.value < 0 else num.value, bool(en.value), ty.value,
                                                 f.value, g.value, q.value))
            return d
        finally:
            lib.aw_eq_definition_destroy(h)


def _device_roundtrip(ctx: Context, fn, x: np.ndarray) -> np.ndarray:
    """x: [streams][frames][2] host array -> same shape, through device buffers."""
    out = np.full(x.shape, np.nan, np.float32)
    if x.size == 0:
        return out
    d = ctx.alloc(x.nbytes)
    try:
        ctx.h2d(d, x)
        fn(d, d, x.shape[1])
        ctx.d2h(out, d)
    finally:
        ctx.free(d)
    return out


def _as_batch(x, n_streams: int) -> Tuple[np.ndarray, bool]:
    a = _f32(x)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[None]
    assert a.ndim == 3 and a.shape[2] == 2 and a.shape[0] == n_streams, a.shape
    return np.ascontiguousarray(a), squeeze


class ParametricEqualizerState:
    """ParametricEqualizerProcessor.prepare(definition:sampleRate:) -> ParametricEqualizerState, for a batch."""

    maximumFilterCount = 64

    def __init__(self, definition: Optional[EqualizerDefinition], sampleRate: float, n_streams: int = 1, ctx: Optional[Context] = None):
        self._lib = _capi.load()
        self.ctx = ctx or default_context()
        self.n_streams = n_streams
        h = ctypes.c_void_p()
        with _DefHandle(definition) as dh:
            _check_eq(self._lib.aw_eq_state_create(self.ctx._h, dh, sampleRate, n_streams, ctypes.byref(h)))
        self._h = h
        self.sampleRate = sampleRate
        self.filterCount = self._lib.aw_eq_state_filter_count(h)
        self.preampLinear = self._lib.aw_eq_state_preamp_linear(h)

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_eq_state_destroy(self._h)
            self._h = None

    def reset(self) -> None:
        _check(self._lib.aw_eq_state_reset(self._h))

    def process_device(self, in_ptr: int, out_ptr: int, frames: int) -> None:
        _check(self._lib.aw_eq_state_process(self._h, ctypes.c_void_p(in_ptr), ctypes.c_void_p(out_ptr), frames))

    def process_batch(self, x) -> np.ndarray:
        a, squeeze = _as_batch(x, self.n_streams)
        y = _device_roundtrip(self.ctx, self.process_device, a)
        return y[0] if squeeze else y

    def process(self, inputLeft, inputRight=None) -> Tuple[np.ndarray, np.ndarray]:
        l = _f32(inputLeft)
        r = l if inputRight is None else _f32(inputRight)
        y = self.process_batch(np.stack([l, r], axis=1))
        return np.ascontiguousarray(y[:, 0]), np.ascontiguousarray(y[:, 1])


class EqualizerNotFoldable(AirwaveError):
    """aw_eq_fold_hrir: the equalizer's impulse response does not decay to the tolerance within the allowed length (AW_ERR_EQ_NOT_FOLDABLE)."""


@dataclass
class FoldedHRIR:
    """HRIR tracks with an equalizer folded in (aw_eq_fold_hrir): tracks[i] = (h_i * g)[0 : taps + responseTaps - 1]."""
    tracks: np.ndarray          # [n_tracks][taps + responseTaps - 1] float32
    responseTaps: int           # samples of the equalizer's impulse response that were kept
    tailBound: float            # (what was cut of it) / (its peak): the bound on the output difference relative to the spatializer output's peak


def fold_equalizer(tracks, definition: Optional[EqualizerDefinition], sampleRate: float, tailTolerance: float = 1e-7, maxTaps: int = 65536) -> FoldedHRIR:
    """The equalizer that follows the spatializer in the reference's graph (AudioEffectGraph.swift:195-211), folded into the HRIR tracks:
    EQ(x * h) = x * (h * g).  Host only.  Raises EqualizerNotFoldable when the response is too long (the caller then runs
    ParametricEqualizerState after the spatializer) and ParametricEqualizerPreparationError exactly as ParametricEqualizerState does."""
    lib = _capi.load()
    tr = np.ascontiguousarray(tracks, dtype=np.float32)
    assert tr.ndim == 2
    n, taps = tr.shape
    out_taps, resp, bound = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()

    def call(dh, out):
        st = lib.aw_eq_fold_hrir(dh, sampleRate, _fp(tr), n, taps, tailTolerance, maxTaps, out, ctypes.byref(out_taps), ctypes.byref(resp), ctypes.byref(bound))
        if st == 17:
            raise EqualizerNotFoldable(st, (lib.aw_last_error_message() or b"").decode("utf-8", "replace"))
        _check_eq(st)
    with _DefHandle(definition) as dh:
        call(dh, None)                                        # the length first, then the tracks
        folded = np.empty((n, out_taps.value), dtype=np.float32)
        call(dh, _fp(folded))
    return FoldedHRIR(folded, resp.value, bound.value)


class ParametricEqualizerProcessor:
    crossfadeDurationSeconds = 0.020
    maximumCallbackFrames = 4096

    def __init__(self, sampleRate: float, maxFramesPerCallback: int = 4096, n_streams: int = 1, ctx: Optional[Context] = None):
        """maxFramesPerCallback=0 lifts the per-call cap (batch use)."""
        self._lib = _capi.load()
        self.ctx = ctx or default_context()
        self.n_streams = n_streams
        h = ctypes.c_void_p()
        _check_eq(self._lib.aw_eq_create(self.ctx._h, sampleRate, n_streams, maxFramesPerCallback, ctypes.byref(h)))
        self._h = h
        self.sampleRate, self.maxFramesPerCallback = sampleRate, maxFramesPerCallback
        self.transitionLength = self._lib.aw_eq_transition_length(h)

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_eq_destroy(self._h)
            self._h = None

    @staticmethod
    def prepare(definition: Optional[EqualizerDefinition], sampleRate: float, n_streams: int = 1, ctx: Optional[Context] = None):
        return ParametricEqualizerState(definition, sampleRate, n_streams, ctx)

    def setTarget(self, definition: Optional[EqualizerDefinition]) -> None:
        with _DefHandle(definition) as dh:
            _check_eq(self._lib.aw_eq_set_target(self._h, dh))

    def reset(self) -> None:
        _check(self._lib.aw_eq_reset(self._h))

    def drainRetiredStates(self) -> None:
        _check(self._lib.aw_eq_drain_retired(self._h))

    def withPublicationLockForTesting(self, body) -> None:      # ParametricEqualizerProcessor.swift:229-233 (DEBUG)
        _check(self._lib.aw_eq_debug_hold_publication_lock(self._h, 1))
        try:
            body()
        finally:
            _check(self._lib.aw_eq_debug_hold_publication_lock(self._h, 0))

    @property
    def isTransitioning(self) -> bool:
        return bool(self._lib.aw_eq_is_transitioning(self._h))

    def process_device(self, in_ptr: int, out_ptr: int, frames: int) -> None:
        _check(self._lib.aw_eq_process(self._h, ctypes.c_void_p(in_ptr), ctypes.c_void_p(out_ptr), frames))

    def process_batch(self, x) -> np.ndarray:
        a, squeeze = _as_batch(x, self.n_streams)
        y = _device_roundtrip(self.ctx, self.process_device, a)
        return y[0] if squeeze else y

    def process(self, inputLeft, inputRight=None) -> Tuple[np.ndarray, np.ndarray]:
        """The StereoAudioProcessing entry (planar host buffers, one stream)."""
        l = _f32(inputLeft)
        r = None if inputRight is None else _f32(inputRight)
        ol = np.full(l.size, np.nan, np.float32)
        orr = np.full(l.size, np.nan, np.float32)
        _check(self._lib.aw_eq_process_planar(self._h, _fp(l), None if r is None else _fp(r), _fp(ol), _fp(orr), l.size))
        return ol, orr


class EqualizerAudioEffectError(RuntimeError):    # AudioEffectGraph.swift:28-45
    def __init__(self, kind: str, reason: str = "", line: Optional[int] = None):
        super().__init__(reason or kind)
        self.kind, self.reason, self.filterLine = kind, reason, line


class EqualizerRuntimeEffect:
    """EqualizerRuntimeEffect.swift:5-107 — the AudioEqualizerEffect the graph talks to."""

    def __init__(self, ctx: Optional[Context] = None):
        self.ctx = ctx
        self._processor: Optional[ParametricEqualizerProcessor] = None

    def _publish(self, definition):
        p = self._processor
        try:
            p.setTarget(definition)
            p.drainRetiredStates()
        except ParametricEqualizerPreparationError as e:      # :27-33, 41-47
            p.setTarget(None)
            p.drainRetiredStates()
            if e.kind == "invalidSampleRate":
                raise EqualizerAudioEffectError("invalidSampleRate", "Output sample rate is invalid.")
            if e.kind == "invalidFilter":
                reason = re.sub(r"^Filter \d+ is invalid: | \[kind.*$", "", str(e).split(": ", 1)[1])
                raise EqualizerAudioEffectError("invalidFilter", reason, e.source_line)
            if e.kind == "nonFinitePreamp":
                raise EqualizerAudioEffectError("invalidFilter", "Preamp produces a non-finite gain.")
            raise EqualizerAudioEffectError("invalidFilter", str(e).split(": ", 1)[1])

    def prepare(self, definition: Optional[EqualizerDefinition], sampleRate: float) -> None:     # :10-34
        if not (np.isfinite(sampleRate) and sampleRate > 0):
            raise EqualizerAudioEffectError("invalidSampleRate", "Output sample rate is invalid.")
        if self._processor is None or self._processor.sampleRate != sampleRate:
            self._processor = ParametricEqualizerProcessor(sampleRate, ctx=self.ctx)
        self._publish(definition)

    def setTarget(self, definition: Optional[EqualizerDefinition]) -> None:                      # :36-48
        if self._processor is None:
            raise EqualizerAudioEffectError("unavailable", "Equalizer has not been prepared for an output.")
        self._publish(definition)

    def process(self, inputLeft, inputRight=None) -> Tuple[np.ndarray, np.ndarray]:              # :50-78
        if self._processor is None:
            l = _f32(inputLeft)
            return l.copy(), (l.copy() if inputRight is None else _f32(inputRight).copy())
        return self._processor.process(inputLeft, inputRight)
