"""AudioEffectGraph (Airwave/AudioEffectGraph.swift:64-246): composes the spatial effect and the equalizer in
the reference's order — spatial first, then EQ — with its passthrough rules.  Host-side glue over the two
effects of this package (`HRIRManager`, `EqualizerRuntimeEffect`); any objects with the same surface work,
which is how the reference's own graph tests are re-expressed (tests/test_effect_graph.py)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Set, Tuple

import numpy as np

from .api import _f32
from .eq import EqualizerAudioEffectError, EqualizerDefinition

SPATIAL, EQUALIZER = "spatial", "equalizer"          # AudioEffectKind


@dataclass
class AudioEffectWarning:                             # AudioEffectGraph.swift:14-17
    filterLine: Optional[int]
    reason: str


@dataclass
class AudioEffectPreparationResult:                   # :19-26
    runnableEffects: Set[str] = field(default_factory=set)
    equalizerWarning: Optional[AudioEffectWarning] = None

    @property
    def noEffectCanRun(self) -> bool:
        return not self.runnableEffects


class AudioEffectGraph:
    maximumCallbackFrames = 4096

    def __init__(self, spatial, equalizer, maxFramesPerCallback: int = 4096):
        if not (0 < maxFramesPerCallback <= self.maximumCallbackFrames):   # precondition :82
            raise ValueError("maxFramesPerCallback must be in 1...4096")
        self.spatial, self.equalizer, self.maxFramesPerCallback = spatial, equalizer, maxFramesPerCallback
        self._equalizerActive = False

    def _runnable(self) -> Set[str]:
        return {SPATIAL} if self.spatial.isReady else set()

    def prepare(self, sampleRate: float, equalizerDefinition: Optional[EqualizerDefinition]) -> AudioEffectPreparationResult:
        """prepare(for:equalizerDefinition:) :95-141 — `sampleRate` is output.nominalSampleRate."""
        runnable = self._runnable()
        try:
            self.equalizer.prepare(equalizerDefinition, sampleRate)
            self._equalizerActive = equalizerDefinition is not None
            if equalizerDefinition is not None:
                runnable.add(EQUALIZER)
            return AudioEffectPreparationResult(runnable, None)
        except EqualizerAudioEffectError as e:
            self._equalizerActive = False
            return AudioEffectPreparationResult(runnable, AudioEffectWarning(e.filterLine, e.reason or "Equalizer preparation failed."))
        except Exception as e:  # noqa: BLE001  (:129-140: any other error becomes a line-less warning)
            self._equalizerActive = False
            return AudioEffectPreparationResult(runnable, AudioEffectWarning(None, str(e)))

    def updateEqualizer(self, definition: Optional[EqualizerDefinition]) -> AudioEffectPreparationResult:
        """:143-177 — the processor stays in the callback path so that removing the EQ ramps to unity."""
        runnable = self._runnable()
        try:
            self.equalizer.setTarget(definition)
            self._equalizerActive = True
            if definition is not None:
                runnable.add(EQUALIZER)
            return AudioEffectPreparationResult(runnable, None)
        except EqualizerAudioEffectError as e:
            self._equalizerActive = True
            return AudioEffectPreparationResult(runnable, AudioEffectWarning(e.filterLine, e.reason or "Equalizer update failed."))
        except Exception as e:  # noqa: BLE001
            self._equalizerActive = True
            return AudioEffectPreparationResult(runnable, AudioEffectWarning(None, str(e)))

    def process(self, inputLeft, inputRight=None) -> Tuple[np.ndarray, np.ndarray]:
        """:180-241"""
        l = _f32(inputLeft)
        n = l.size
        if n == 0:
            return l.copy(), l.copy()
        if n > self.maxFramesPerCallback:                                   # precondition :188
            raise ValueError("frameCount exceeds maxFramesPerCallback")
        if self.spatial.isReady:
            sl, sr = self.spatial.process(l, inputRight)
            return self.equalizer.process(sl, sr) if self._equalizerActive else (sl, sr)
        r = l if inputRight is None else _f32(inputRight)                   # passthrough, mono duplicated :220-226,234-239
        if self._equalizerActive:
            return self.equalizer.process(l.copy(), r.copy())
        return l.copy(), r.copy()
