"""Host-side mirror of the reference's operator surface for the HRIR convolution path, over the C ABI.

Same names, argument meaning and error behaviour as the Swift types (cited per class), so parity
tests read like the reference's XCTests.  All arithmetic happens in libairwave_hip.so on the GPU;
nothing here computes audio.
"""
from __future__ import annotations

import ctypes
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import c_float_p, c_int32_p

AW_OK = 0
STATUS_NAMES = {
    1: "INVALID_ARGUMENT", 2: "OUT_OF_MEMORY", 3: "HIP", 4: "NO_DEVICE", 5: "INVALID_CHANNEL_MAPPING",
    6: "CONVOLUTION_SETUP_FAILED", 7: "INVALID_CHANNEL_COUNT", 8: "WAV_FILE_READ", 9: "WAV_EMPTY_FILE",
    10: "WAV_UNSUPPORTED_FORMAT", 11: "BLOCK_SIZE_MISMATCH", 12: "EQ_PARSE", 13: "EQ_INVALID_SAMPLE_RATE",
    14: "EQ_NON_FINITE_PREAMP", 15: "EQ_TOO_MANY_FILTERS", 16: "EQ_INVALID_FILTER",
}


class AirwaveError(RuntimeError):
    """Any non-zero aw_status.  `.status` is the code, `.name` its symbolic name."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"AW_ERR_{self.name}: {message}")


class HRIRError(AirwaveError):          # HRIRManager.swift:737-759
    pass


class WAVError(AirwaveError):           # WAVLoader.swift:127-147
    pass


def _finalizing() -> bool:
    """True while the interpreter shuts down.  Handles that own device memory are not destroyed then: module globals and reference
    cycles are collected in no particular order (a context can go before the objects created on it), and the HIP runtime's own exit
    handlers may already have run — the process is about to return everything to the driver anyway.  (Found by tools/fuzz_eq.py, whose
    last processor died at exit: std::bad_variant_access out of the HIP runtime, exit code 134 after a clean run.)"""
    return sys.is_finalizing()


def _check(status: int) -> None:
    if status == AW_OK:
        return
    msg = (_capi.load().aw_last_error_message() or b"").decode("utf-8", "replace")
    if status in (5, 6, 7):
        raise HRIRError(status, msg)
    if status in (8, 9, 10):
        raise WAVError(status, msg)
    raise AirwaveError(status, msg)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(c_int32_p)


class Context:
    """Device + stream + shared twiddle tables (FFTSetupManager analogue, FFTSetupManager.swift:41-60)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _capi.load()
        h = ctypes.c_void_p()
        if stream is None:
            _check(self._lib.aw_context_create(device, ctypes.byref(h)))
        else:
            _check(self._lib.aw_context_create_on_stream(device, ctypes.c_void_p(stream), ctypes.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.aw_context_destroy(self._h)
            self._h = None

    def __del__(self):
        if not _finalizing():
            self.close()

    def set_resampler(self, literal_vgenp: bool = False) -> None:
        """Which resampler activatePreset uses for this context: the intended interpolation (default) or the literal
        vDSP_vgenp call of the shipped reference."""
        _check(self._lib.aw_context_set_resampler(self._h, int(literal_vgenp)))

    def synchronize(self):
        _check(self._lib.aw_context_synchronize(self._h))

    @property
    def stream(self) -> int:
        return self._lib.aw_context_stream(self._h) or 0

    def timer_start(self):
        _check(self._lib.aw_context_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        _check(self._lib.aw_context_timer_stop(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def reserve_scratch(self, nbytes: int) -> None:
        """Size the context's scratch pool (shared by its spatializers) ahead of time: the one large hipMalloc at start-up."""
        _check(self._lib.aw_context_reserve_scratch(self._h, int(nbytes)))

    @property
    def scratch_bytes(self) -> int:
        return int(self._lib.aw_context_scratch_bytes(self._h))

    def bandwidth_probe(self, nbytes: int = 4 << 30, repetitions: int = 3) -> Dict[str, float]:
        """Measured read-only / write-only / copy rates of this device in GB/s (copy = bytes read + bytes written per second):
        the ceiling SURVEY.md 8d asks to quote next to the vendor peak."""
        r, w, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(self._lib.aw_context_bandwidth_probe(self._h, nbytes, repetitions, ctypes.byref(r), ctypes.byref(w), ctypes.byref(c)))
        return {"read": r.value, "write": w.value, "copy": c.value}

    def pcie_probe(self, nbytes: int = 1 << 30, repetitions: int = 2) -> Dict[str, float]:
        """Page-locked hipMemcpyAsync rates in GB/s: host to device, device to host, and both at once (bytes both ways per second)."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(self._lib.aw_context_pcie_probe(self._h, nbytes, repetitions, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"h2d": a.value, "d2h": b.value, "duplex": c.value}

    def pinned_empty(self, shape, dtype=np.float32) -> np.ndarray:
        """A numpy array over page-locked host memory (aw_host_alloc_pinned); freed when the array's base object dies."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _check(self._lib.aw_host_alloc_pinned(self._h, n, ctypes.byref(p)))
        owner = _PinnedOwner(self, p.value, max(n, 1))
        return np.asarray(owner)[:n].view(dtype).reshape(shape)

    # raw device memory for hosts without torch
    def alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        _check(self._lib.aw_device_alloc(self._h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, dptr: int):
        _check(self._lib.aw_device_free(self._h, ctypes.c_void_p(dptr)))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(self._lib.aw_memcpy_h2d(self._h, ctypes.c_void_p(dptr), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags["C_CONTIGUOUS"]
        _check(self._lib.aw_memcpy_d2h(self._h, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dptr), arr.nbytes))

    def synth_fill(self, dptr: int, n_streams: int, frames: int, n_channels: int, seed: int = 0xA17AE, first_stream: int = 0):
        _check(self._lib.aw_synth_fill(self._h, ctypes.c_void_p(dptr), n_streams, frames, n_channels, seed, first_stream))


class _PinnedOwner:
    """Base object of the numpy views over one page-locked allocation (array interface): alive as long as any view is, then frees it."""

    def __init__(self, ctx: "Context", ptr: int, nbytes: int):
        self._ctx, self._ptr = ctx, ptr
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            if self._ptr and getattr(self._ctx, "_h", None) and not _finalizing():
                self._ctx._lib.aw_host_free_pinned(self._ctx._h, ctypes.c_void_p(self._ptr))
        finally:
            self._ptr = 0


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- WAVLoader -------------------------------------------------------------------------------------
class WAVData:
    """WAVData (WAVLoader.swift:12-17): sampleRate, channelCount, frameCount, audioData [[Float]]."""

    def __init__(self, sample_rate: float, channel_count: int, frame_count: int, audio_data: np.ndarray):
        self.sample_rate = sample_rate
        self.channel_count = channel_count
        self.frame_count = frame_count
        self.audio_data = audio_data


class WAVLoader:
    @staticmethod
    def load(path: str) -> WAVData:
        """WAVLoader.load(from:)  WAVLoader.swift:26-99"""
        lib = _capi.load()
        h = ctypes.c_void_p()
        _check(lib.aw_wav_load(path.encode(), ctypes.byref(h)))
        try:
            ch, fr = lib.aw_wav_channel_count(h), lib.aw_wav_frame_count(h)
            data = np.ctypeslib.as_array(lib.aw_wav_planar(h), shape=(ch, fr)).copy()
            return WAVData(lib.aw_wav_sample_rate(h), ch, fr, data)
        finally:
            lib.aw_wav_destroy(h)


# ---- InputLayout / HRIRChannelMap --------------------------------------------------------------------
class InputLayout:
    """InputLayout (VirtualSpeaker.swift:59-100)."""

    def __init__(self, channels: Sequence[str], name: str = ""):
        lib = _capi.load()
        arr = (ctypes.c_char_p * max(len(channels), 1))(*[c.encode() for c in channels])
        h = ctypes.c_void_p()
        _check(lib.aw_layout_create(arr, len(channels), name.encode(), ctypes.byref(h)))
        self._h, self._lib = h, lib

    @classmethod
    def _wrap(cls, h):
        obj = cls.__new__(cls)
        obj._h, obj._lib = h, _capi.load()
        return obj

    @classmethod
    def detect(cls, channel_count: int) -> "InputLayout":
        h = ctypes.c_void_p()
        _check(_capi.load().aw_layout_detect(channel_count, ctypes.byref(h)))
        return cls._wrap(h)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.aw_layout_destroy(self._h)
            self._h = None

    @property
    def channels(self) -> List[str]:
        return [self._lib.aw_layout_speaker(self._h, i).decode() for i in range(self._lib.aw_layout_count(self._h))]

    @property
    def name(self) -> str:
        return self._lib.aw_layout_name(self._h).decode()


InputLayout.stereo = staticmethod(lambda: InputLayout.detect(2))          # type: ignore[attr-defined]
InputLayout.surround51 = staticmethod(lambda: InputLayout.detect(6))      # type: ignore[attr-defined]
InputLayout.surround71 = staticmethod(lambda: InputLayout.detect(8))      # type: ignore[attr-defined]
InputLayout.atmos714 = staticmethod(lambda: InputLayout.detect(12))       # type: ignore[attr-defined]


class HRIRChannelMap:
    """HRIRChannelMap (VirtualSpeaker.swift:103-347)."""

    def __init__(self, h):
        self._h, self._lib = h, _capi.load()

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.aw_map_destroy(self._h)
            self._h = None

    @classmethod
    def _make(cls, fn_name: str, arg) -> "HRIRChannelMap":
        h = ctypes.c_void_p()
        _check(getattr(_capi.load(), fn_name)(arg, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def hesuvi14Channel(cls, speakers: InputLayout):
        return cls._make("aw_map_hesuvi14", speakers._h)

    @classmethod
    def hesuvi7Channel(cls, speakers: InputLayout):
        return cls._make("aw_map_hesuvi7", speakers._h)

    @classmethod
    def interleavedPairs(cls, speakers: InputLayout):
        return cls._make("aw_map_interleaved_pairs", speakers._h)

    @classmethod
    def splitBlocks(cls, speakers: InputLayout):
        return cls._make("aw_map_split_blocks", speakers._h)

    @classmethod
    def parseHeSuViFormat(cls, text: str):
        return cls._make("aw_map_parse_text", text.encode())

    def getIndices(self, speaker: str) -> Optional[Tuple[int, int]]:
        l, r = ctypes.c_int32(), ctypes.c_int32()
        if self._lib.aw_map_get(self._h, speaker.encode(), ctypes.byref(l), ctypes.byref(r)):
            return (l.value, r.value)
        return None

    def hasMappingFor(self, speaker: str) -> bool:
        return self.getIndices(speaker) is not None

    def __len__(self):
        return self._lib.aw_map_count(self._h)

    def resolve(self, layout: InputLayout, n_tracks: int) -> Tuple[np.ndarray, np.ndarray]:
        n = len(layout.channels)
        lt = np.zeros(max(n, 1), dtype=np.int32)
        rt = np.zeros(max(n, 1), dtype=np.int32)
        _check(self._lib.aw_map_resolve(self._h, layout._h, n_tracks, _i32p(lt), _i32p(rt)))
        return lt[:n], rt[:n]


class Resampler:
    """Resampler (Resampler.swift:12-69)."""

    @staticmethod
    def resampleHighQuality(input, fromRate: float, toRate: float, literal_vgenp: bool = False) -> np.ndarray:
        """The intended interpolation out[i] = lerp(input, i * fromRate / toRate) by default; literal_vgenp=True is the
        call the reference actually makes (vDSP_vramp + vDSP_vgenp, Resampler.swift:56-65) as Apple documents vgenp."""
        lib = _capi.load()
        x = _f32(input)
        n = max(lib.aw_resample_output_count(x.size, fromRate, toRate), 0)
        out = np.zeros(max(n, 1), dtype=np.float32)
        cnt = ctypes.c_int32()
        fn = lib.aw_resample_vgenp if literal_vgenp else lib.aw_resample
        _check(fn(_fp(x), x.size, fromRate, toRate, _fp(out), out.size, ctypes.byref(cnt)))
        return out[: cnt.value]

    resample = resampleHighQuality


# ---- device objects -----------------------------------------------------------------------------------
class HRIR:
    def __init__(self, tracks, sample_rate: float = 48000.0, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        t = _f32(tracks)
        if t.ndim == 1:
            t = t[None]
        self._lib = _capi.load()
        h = ctypes.c_void_p()
        _check(self._lib.aw_hrir_create(self.ctx._h, _fp(t), t.shape[0], t.shape[1], sample_rate, ctypes.byref(h)))
        self._h = h
        self.n_tracks, self.taps = t.shape

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_hrir_destroy(self._h)
            self._h = None


class Spatializer:
    """Batch engine network: S streams x (one convolution per (input channel, ear)) + stereo downmix."""

    def __init__(self, hrir: HRIR, left_track, right_track, n_streams: int = 1, ctx: Optional[Context] = None, _h=None):
        self._lib = _capi.load()
        self.ctx = ctx or (hrir.ctx if hrir is not None else default_context())
        if _h is not None:
            self._h = _h
        else:
            lt = np.ascontiguousarray(left_track, dtype=np.int32)
            rt = np.ascontiguousarray(right_track, dtype=np.int32)
            assert lt.size == rt.size
            h = ctypes.c_void_p()
            _check(self._lib.aw_spatializer_create(self.ctx._h, hrir._h, lt.size, _i32p(lt), _i32p(rt), n_streams, 0,
                                                   ctypes.byref(h)))
            self._h = h
        self.n_streams = self._lib.aw_spatializer_stream_count(self._h)
        self.n_channels = self._lib.aw_spatializer_channel_count(self._h)

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_spatializer_destroy(self._h)
            self._h = None

    def info(self) -> Dict[str, int]:
        g = lambda i: int(self._lib.aw_spatializer_info(self._h, i))
        return {"fft": g(0), "hop": g(1), "partitions": g(2), "path": g(3), "history": g(4), "dominant_frames": g(5), "scratch_bytes": g(6),
                "long_window_rows": g(7),       # long-window kernels ran in the last call on windows of rows x 4096 frames (0: they did not)
                "long_window_rows_rest": g(8),  # ... and a last, shorter window of this many rows for the remainder (0: one window length)
                "long_window_table_sets": g(9), # table sets built so far (one per window length)
                # the last reserve(): float64 table build on host threads / table upload / scratch pool growth, milliseconds
                "reserve_tables_ms": g(10) / 1e3, "reserve_upload_ms": g(11) / 1e3, "reserve_scratch_ms": g(12) / 1e3,
                "device_allocs": g(13),         # device / page-locked allocations made so far on behalf of the context's handles
                "sync_copies": g(14),           # blocking table uploads likewise
                "host_chunk_streams": g(15),    # streams per staged chunk of the last host-entry call (0: one piece)
                "overlap_add_rows": g(16),      # the last call ran the overlap-add tile on blocks of 512 x this many frames (0: it did not)
                "overlap_add_rows_policy": g(17)}   # ... which this spatializer's calls do when they have enough blocks (0: never)

    def process_device(self, in_ptr: int, out_ptr: int, frames: int) -> None:
        _check(self._lib.aw_spatializer_process(self._h, ctypes.c_void_p(in_ptr), ctypes.c_void_p(out_ptr), frames))

    def process(self, x) -> np.ndarray:
        """x: [streams][frames][channels] (or [frames][channels] for one stream) host array."""
        a = _f32(x)
        squeeze = a.ndim == 2
        if squeeze:
            a = a[None]
        S, F, C = a.shape
        assert S == self.n_streams and C == self.n_channels
        out = np.full((S, F, 2), np.nan, dtype=np.float32)
        _check(self._lib.aw_spatializer_process_host(self._h, _fp(a), _fp(out), F))
        return out[0] if squeeze else out

    def process_planar(self, input_left, input_right=None) -> Tuple[np.ndarray, np.ndarray]:
        l = _f32(input_left)
        r = None if input_right is None else _f32(input_right)
        ol = np.full(l.size, np.nan, dtype=np.float32)
        orr = np.full(l.size, np.nan, dtype=np.float32)
        _check(self._lib.aw_spatializer_process_planar(self._h, _fp(l), None if r is None else _fp(r), _fp(ol), _fp(orr), l.size))
        return ol, orr

    def reserve(self, max_frames: int) -> None:
        """Size every internal device buffer for calls of up to max_frames frames: process never allocates afterwards."""
        _check(self._lib.aw_spatializer_reserve(self._h, int(max_frames)))

    def reserve_host(self, max_frames: int) -> None:
        """reserve() plus the device-side staging of the host entry (process / process_host_into)."""
        _check(self._lib.aw_spatializer_reserve_host(self._h, int(max_frames)))

    def process_host_into(self, x: np.ndarray, out: np.ndarray) -> None:
        """Host entry on caller-owned arrays (e.g. Context.pinned_empty): x [streams][frames][channels], out [streams][frames][2]."""
        assert x.dtype == np.float32 and out.dtype == np.float32 and x.flags["C_CONTIGUOUS"] and out.flags["C_CONTIGUOUS"]
        S, F, C = x.shape
        assert S == self.n_streams and C == self.n_channels and out.shape == (S, F, 2)
        _check(self._lib.aw_spatializer_process_host(self._h, _fp(x), _fp(out), F))

    def reset(self) -> None:
        _check(self._lib.aw_spatializer_reset(self._h))

    def set_profiling(self, on: bool) -> None:
        _check(self._lib.aw_spatializer_set_profiling(self._h, int(on)))

    def kernel_time(self) -> Tuple[int, float, str]:
        ms = ctypes.c_double()
        name = ctypes.c_char_p()
        n = self._lib.aw_spatializer_kernel_time(self._h, ctypes.byref(ms), ctypes.byref(name))
        return n, float(ms.value), (name.value or b"").decode()

    def stage_times(self) -> List[Tuple[str, float, int]]:
        """(kernel name, total ms, launches) of every separately timed launch since profiling was switched on."""
        out = []
        i = 0
        while True:
            name, ms, n = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int32()
            if not self._lib.aw_spatializer_stage_time(self._h, i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(n)):
                return out
            out.append(((name.value or b"").decode(), float(ms.value), int(n.value)))
            i += 1


class ConvolutionEngine:
    """ConvolutionEngine (ConvolutionEngine.swift:14-408) on the GPU."""

    def __init__(self, hrirSamples, blockSize: int = 512, ctx: Optional[Context] = None):
        self._lib = _capi.load()
        self.ctx = ctx or default_context()
        h = _f32(hrirSamples)
        e = ctypes.c_void_p()
        _check(self._lib.aw_engine_create(self.ctx._h, _fp(h), h.size, blockSize, ctypes.byref(e)))
        self._h = e
        self.blockSize = blockSize

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_engine_destroy(self._h)
            self._h = None

    def process(self, input, frameCount: Optional[int] = None) -> Optional[np.ndarray]:
        """process(input:output:frameCount:) — returns None (output untouched) when count != blockSize,
        like the Swift wrapper's silent return (ConvolutionEngine.swift:370-373)."""
        x = _f32(input)
        count = self.blockSize if frameCount is None else frameCount
        out = np.zeros(self.blockSize, dtype=np.float32)
        st = self._lib.aw_engine_process_n(self._h, _fp(x), _fp(out), count)
        if st == 11:
            return None
        _check(st)
        return out

    def processAndAccumulate(self, input, outputAccumulator: np.ndarray) -> None:
        x = _f32(input)
        assert outputAccumulator.dtype == np.float32 and outputAccumulator.size == self.blockSize
        _check(self._lib.aw_engine_process_accumulate(self._h, _fp(x), _fp(outputAccumulator)))

    def reset(self) -> None:
        _check(self._lib.aw_engine_reset(self._h))


class RealtimeAudioProcessor:
    """RealtimeAudioProcessor (RealtimeAudioProcessor.swift:11-191).  renderers: [(leftTrack, rightTrack)]
    index pairs into `hrir` (the VirtualSpeakerRenderer's two engines)."""

    def __init__(self, hrir: HRIR, renderers: Sequence[Tuple[int, int]], blockSize: int = 512,
                 maxFramesPerCallback: int = 4096):
        self._lib = _capi.load()
        self.ctx = hrir.ctx
        lt = np.ascontiguousarray([r[0] for r in renderers] or [0], dtype=np.int32)
        rt = np.ascontiguousarray([r[1] for r in renderers] or [0], dtype=np.int32)
        h = ctypes.c_void_p()
        _check(self._lib.aw_realtime_create(self.ctx._h, hrir._h, len(renderers), _i32p(lt), _i32p(rt), blockSize,
                                            maxFramesPerCallback, ctypes.byref(h)))
        self._h = h
        self.blockSize, self.maxFramesPerCallback = blockSize, maxFramesPerCallback

    def __del__(self):
        if getattr(self, "_h", None) and not _finalizing():
            self._lib.aw_realtime_destroy(self._h)
            self._h = None

    def process(self, inputLeft, inputRight=None, outputLeft: Optional[np.ndarray] = None,
                outputRight: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        l = _f32(inputLeft)
        r = None if inputRight is None else _f32(inputRight)
        n = l.size
        ol = outputLeft if outputLeft is not None else np.full(n, np.nan, dtype=np.float32)
        orr = outputRight if outputRight is not None else np.full(n, np.nan, dtype=np.float32)
        _check(self._lib.aw_realtime_process(self._h, _fp(l), None if r is None else _fp(r), _fp(ol), _fp(orr), n))
        return ol, orr

    def reset(self) -> None:
        _check(self._lib.aw_realtime_reset(self._h))

    def info(self) -> Dict[str, int]:
        """What the processor holds (all of it allocated in __init__, like RealtimeAudioProcessor.init, :30-62)."""
        g = lambda i: int(self._lib.aw_realtime_info(self._h, i))
        return {"host_bytes": g(0), "device_bytes": g(1), "device_allocs": g(2)}


class HRIRManager:
    """The activation + processing part of HRIRManager (HRIRManager.swift:316-449, 531-568):
    activatePreset builds the renderer network; process is the StereoAudioProcessing entry
    (AudioPipeline.swift:3-28) with passthrough when no preset is active."""

    def __init__(self, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self._lib = _capi.load()
        self.spatializer: Optional[Spatializer] = None
        self.errorMessage: Optional[str] = None

    @property
    def isReady(self) -> bool:                                   # AudioSpatialEffect.isReady
        return self.spatializer is not None

    isConvolutionActive = isReady

    def activatePreset(self, fileURL: str, targetSampleRate: float, inputLayout: InputLayout,
                       hrirMap: Optional[HRIRChannelMap] = None, n_streams: int = 1) -> Spatializer:
        sp = ctypes.c_void_p()
        st = self._lib.aw_preset_activate(self.ctx._h, fileURL.encode(), targetSampleRate, inputLayout._h,
                                          hrirMap._h if hrirMap is not None else None, n_streams, ctypes.byref(sp), None)
        if st != AW_OK:
            msg = (self._lib.aw_last_error_message() or b"").decode()
            self.errorMessage = "Failed to activate preset: " + msg       # HRIRManager.swift:441
            _check(st)
        self.errorMessage = None
        self.spatializer = Spatializer(None, None, None, ctx=self.ctx, _h=sp)
        return self.spatializer

    def deactivatePreset(self) -> None:
        self.spatializer = None

    def process(self, inputLeft, inputRight=None) -> Tuple[np.ndarray, np.ndarray]:
        if self.spatializer is None:                               # passthrough, HRIRManager.swift:550-559
            l = _f32(inputLeft)
            return l.copy(), (l.copy() if inputRight is None else _f32(inputRight).copy())
        return self.spatializer.process_planar(inputLeft, inputRight)

    def resetConvolutionState(self) -> None:
        if self.spatializer is not None:
            self.spatializer.reset()
