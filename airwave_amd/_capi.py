"""ctypes binding of include/airwave_hip.h — the same symbols a Swift module map would import.

Fails loudly when the shared library is missing: there is no Python/CPU fallback path.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIRWAVE_HIP_LIBRARY") or os.path.join(HERE, "libairwave_hip.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# every symbol include/airwave_hip.h declares: (restype, argtypes)
_V, _I32, _I64, _U64, _D, _SZ, _S = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double, ctypes.c_size_t, ctypes.c_char_p
SIGNATURES = {
    "aw_status_string": (_S, [_I32]),
    "aw_last_error_message": (_S, []),
    "aw_last_eq_filter_error": (_I32, [c_int32_p, c_int32_p, c_int32_p]),
    "aw_version": (_S, []),
    "aw_context_create": (_I32, [_I32, c_void_pp]),
    "aw_context_create_on_stream": (_I32, [_I32, _V, c_void_pp]),
    "aw_context_destroy": (None, [_V]),
    "aw_context_synchronize": (_I32, [_V]),
    "aw_context_stream": (_V, [_V]),
    "aw_context_timer_start": (_I32, [_V]),
    "aw_context_timer_stop": (_I32, [_V, c_float_p]),
    "aw_context_reserve_scratch": (_I32, [_V, _SZ]),
    "aw_context_scratch_bytes": (_SZ, [_V]),
    "aw_context_bandwidth_probe": (_I32, [_V, _SZ, _I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "aw_context_pcie_probe": (_I32, [_V, _SZ, _I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "aw_host_alloc_pinned": (_I32, [_V, _SZ, c_void_pp]),
    "aw_host_free_pinned": (_I32, [_V, _V]),
    "aw_device_alloc": (_I32, [_V, _SZ, c_void_pp]),
    "aw_device_free": (_I32, [_V, _V]),
    "aw_memcpy_h2d": (_I32, [_V, _V, _V, _SZ]),
    "aw_memcpy_d2h": (_I32, [_V, _V, _V, _SZ]),
    "aw_hrir_create": (_I32, [_V, c_float_p, _I32, _I32, _D, c_void_pp]),
    "aw_hrir_destroy": (None, [_V]),
    "aw_hrir_track_count": (_I32, [_V]),
    "aw_hrir_taps": (_I32, [_V]),
    "aw_hrir_sample_rate": (_D, [_V]),
    "aw_spatializer_create": (_I32, [_V, _V, _I32, c_int32_p, c_int32_p, _I32, _I32, c_void_pp]),
    "aw_spatializer_destroy": (None, [_V]),
    "aw_spatializer_process": (_I32, [_V, _V, _V, _I64]),
    "aw_spatializer_process_host": (_I32, [_V, c_float_p, c_float_p, _I64]),
    "aw_spatializer_process_planar": (_I32, [_V, c_float_p, c_float_p, c_float_p, c_float_p, _I32]),
    "aw_spatializer_reserve": (_I32, [_V, _I64]),
    "aw_spatializer_reserve_host": (_I32, [_V, _I64]),
    "aw_spatializer_reset": (_I32, [_V]),
    "aw_spatializer_stream_count": (_I32, [_V]),
    "aw_spatializer_channel_count": (_I32, [_V]),
    "aw_spatializer_info": (_I64, [_V, _I32]),
    "aw_spatializer_set_profiling": (_I32, [_V, _I32]),
    "aw_spatializer_kernel_time": (_I32, [_V, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_char_p)]),
    "aw_spatializer_stage_time": (_I32, [_V, _I32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    "aw_spatializer_debug_stamps": (_I32, [_V, ctypes.POINTER(ctypes.c_uint64), _I64, ctypes.POINTER(ctypes.c_int64)]),
    "aw_engine_create": (_I32, [_V, c_float_p, _I32, _I32, c_void_pp]),
    "aw_engine_destroy": (None, [_V]),
    "aw_engine_process": (_I32, [_V, c_float_p, c_float_p]),
    "aw_engine_process_n": (_I32, [_V, c_float_p, c_float_p, _I32]),
    "aw_engine_process_accumulate": (_I32, [_V, c_float_p, c_float_p]),
    "aw_engine_reset": (_I32, [_V]),
    "aw_engine_block_size": (_I32, [_V]),
    "aw_realtime_create": (_I32, [_V, _V, _I32, c_int32_p, c_int32_p, _I32, _I32, c_void_pp]),
    "aw_realtime_destroy": (None, [_V]),
    "aw_realtime_process": (_I32, [_V, c_float_p, c_float_p, c_float_p, c_float_p, _I32]),
    "aw_realtime_reset": (_I32, [_V]),
    "aw_realtime_info": (_I64, [_V, _I32]),
    "aw_wav_load": (_I32, [_S, c_void_pp]),
    "aw_wav_destroy": (None, [_V]),
    "aw_wav_sample_rate": (_D, [_V]),
    "aw_wav_channel_count": (_I32, [_V]),
    "aw_wav_frame_count": (_I32, [_V]),
    "aw_wav_channel": (c_float_p, [_V, _I32]),
    "aw_wav_planar": (c_float_p, [_V]),
    "aw_layout_detect": (_I32, [_I32, c_void_pp]),
    "aw_layout_create": (_I32, [ctypes.POINTER(ctypes.c_char_p), _I32, _S, c_void_pp]),
    "aw_layout_destroy": (None, [_V]),
    "aw_layout_count": (_I32, [_V]),
    "aw_layout_speaker": (_S, [_V, _I32]),
    "aw_layout_name": (_S, [_V]),
    "aw_map_hesuvi14": (_I32, [_V, c_void_pp]),
    "aw_map_hesuvi7": (_I32, [_V, c_void_pp]),
    "aw_map_interleaved_pairs": (_I32, [_V, c_void_pp]),
    "aw_map_split_blocks": (_I32, [_V, c_void_pp]),
    "aw_map_parse_text": (_I32, [_S, c_void_pp]),
    "aw_map_destroy": (None, [_V]),
    "aw_map_count": (_I32, [_V]),
    "aw_map_get": (_I32, [_V, _S, c_int32_p, c_int32_p]),
    "aw_map_resolve": (_I32, [_V, _V, _I32, c_int32_p, c_int32_p]),
    "aw_resample_output_count": (_I32, [_I32, _D, _D]),
    "aw_resample": (_I32, [c_float_p, _I32, _D, _D, c_float_p, _I32, c_int32_p]),
    "aw_resample_vgenp": (_I32, [c_float_p, _I32, _D, _D, c_float_p, _I32, c_int32_p]),
    "aw_context_set_resampler": (_I32, [_V, _I32]),
    "aw_preset_activate": (_I32, [_V, _S, _D, _V, _V, _I32, c_void_pp, c_void_pp]),
    "aw_synth_fill": (_I32, [_V, _V, _I32, _I64, _I32, _U64, _U64]),
    # parametric EQ row
    "aw_biquad_make": (_I32, [_I32, _D, _D, _D, _D, ctypes.POINTER(ctypes.c_double), c_int32_p]),
    "aw_eq_definition_create": (_I32, [_D, c_void_pp]),
    "aw_eq_definition_add_filter": (_I32, [_V, _I32, _I32, _D, _D, _D]),
    "aw_eq_definition_set_source": (_I32, [_V, _I32, _I32, _I64]),
    "aw_eq_definition_destroy": (None, [_V]),
    "aw_eq_definition_preamp_db": (_D, [_V]),
    "aw_eq_definition_filter_count": (_I32, [_V]),
    "aw_eq_definition_filter": (_I32, [_V, _I32, c_int32_p, ctypes.POINTER(ctypes.c_int64), c_int32_p, c_int32_p,
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "aw_eq_parse": (_I32, [_V, _SZ, c_void_pp, ctypes.c_char_p, _SZ]),
    "aw_eq_state_create": (_I32, [_V, _V, _D, _I32, c_void_pp]),
    "aw_eq_state_destroy": (None, [_V]),
    "aw_eq_state_reset": (_I32, [_V]),
    "aw_eq_state_process": (_I32, [_V, _V, _V, _I64]),
    "aw_eq_state_filter_count": (_I32, [_V]),
    "aw_eq_state_preamp_linear": (_D, [_V]),
    "aw_eq_fold_hrir": (_I32, [_V, _D, c_float_p, _I32, _I32, _D, _I32, c_float_p, c_int32_p, c_int32_p, ctypes.POINTER(ctypes.c_double)]),
    "aw_eq_create": (_I32, [_V, _D, _I32, _I32, c_void_pp]),
    "aw_eq_destroy": (None, [_V]),
    "aw_eq_set_target": (_I32, [_V, _V]),
    "aw_eq_reset": (_I32, [_V]),
    "aw_eq_drain_retired": (_I32, [_V]),
    "aw_eq_process": (_I32, [_V, _V, _V, _I64]),
    "aw_eq_process_planar": (_I32, [_V, c_float_p, c_float_p, c_float_p, c_float_p, _I32]),
    "aw_eq_debug_hold_publication_lock": (_I32, [_V, _I32]),
    "aw_eq_transition_length": (_I32, [_V]),
    "aw_eq_is_transitioning": (_I32, [_V]),
}

_lib = None


def load() -> ctypes.CDLL:
    """dlopen libairwave_hip.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python airwave_amd/build.py` (hipcc, gfx950). "
            "airwave_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
