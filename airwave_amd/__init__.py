"""airwave_amd — MI355X-native batch HRIR spatializer (drop-in for Airwave's ConvolutionEngine path).

The product is libairwave_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/airwave_hip.h).  This package is the Python host mirror of the reference's Swift operator
surface over that ABI.  Importing it loads the library; a missing build is an ImportError
(there is no CPU fallback).
"""
from . import _capi

_capi.load()

from .api import (  # noqa: E402
    AirwaveError, Context, ConvolutionEngine, HRIR, HRIRChannelMap, HRIRError, HRIRManager, InputLayout,
    RealtimeAudioProcessor, Resampler, Spatializer, WAVData, WAVError, WAVLoader, default_context,
)

from .batching import MixedRateBatch, RateBucket, bucket_by_rate, resample_tracks  # noqa: E402
from .graph import AudioEffectGraph, AudioEffectPreparationResult, AudioEffectWarning  # noqa: E402
from .eq import (  # noqa: E402
    BiquadCoefficientBuilder, BiquadCoefficientError, EqualizerAPOParser, EqualizerAudioEffectError, EqualizerDefinition,
    EqualizerFilter, EqualizerParseError, EqualizerRuntimeEffect, ParametricEqualizerPreparationError,
    ParametricEqualizerProcessor, ParametricEqualizerState, EqualizerNotFoldable, FoldedHRIR, fold_equalizer,
)

__all__ = [
    "AudioEffectGraph", "AudioEffectPreparationResult", "AudioEffectWarning",
    "MixedRateBatch", "RateBucket", "bucket_by_rate", "resample_tracks",
    "BiquadCoefficientBuilder", "BiquadCoefficientError", "EqualizerAPOParser", "EqualizerAudioEffectError",
    "EqualizerDefinition", "EqualizerFilter", "EqualizerParseError", "EqualizerRuntimeEffect",
    "ParametricEqualizerPreparationError", "ParametricEqualizerProcessor", "ParametricEqualizerState", "EqualizerNotFoldable", "FoldedHRIR", "fold_equalizer",
    "AirwaveError", "Context", "ConvolutionEngine", "HRIR", "HRIRChannelMap", "HRIRError", "HRIRManager",
    "InputLayout", "RealtimeAudioProcessor", "Resampler", "Spatializer", "WAVData", "WAVError", "WAVLoader",
    "default_context",
]
