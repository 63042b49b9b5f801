"""Mixed-rate batches (BASELINE cfg 5): streams are bucketed by sample rate and every
bucket gets its own renderer network built exactly as HRIRManager.activatePreset builds one for an output
device of that rate — the HRIR (not the audio) is resampled to the stream rate with Resampler
(HRIRManager.swift:389-403, Resampler.swift:31-68) — so one preset serves 44.1/48/96 kHz streams side by
side.  Host-side orchestration over the kernels of the convolution path; nothing here touches samples."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from .api import Context, HRIR, HRIRChannelMap, InputLayout, Resampler, Spatializer, default_context


@dataclass
class RateBucket:
    sample_rate: float
    stream_ids: List[int]            # positions in the caller's stream list, in order
    hrir_taps: int                   # taps of the tracks the spatializer convolves with (HRIR resampled to the rate, equalizer folded in if it was)
    spatializer: Spatializer
    equalizer: Optional[object] = None      # ParametricEqualizerState that runs AFTER the spatializer (None: no equalizer, or folded into the HRIR)
    eq_response_taps: int = 0        # equalizer folded into the HRIR: samples of its impulse response that were kept (0: not folded)
    eq_tail_bound: float = 0.0       # ... and the bound on what the cut can change, relative to the spatializer output's peak


# An equalizer is folded into the HRIR (fold=None, the default) when that is the cheaper form: the folded HRIR is longer by the
# equalizer's response (thousands of taps for a low shelf at 100 Hz), which only lengthens the window overlap of the long-window kernels
# (cfg 4: 8640 -> 22 464 taps, +7 % of three memory-bound launches, against a Float64 cascade pass of 40 % of the step) but would push a
# short HRIR off the on-chip tile.  So: fold when the HRIR alone is already beyond the fused tiles, or when the folded one still fits them.
FUSED_TILE_TAPS = 5121


def _fold_pays(hrir_taps: int, folded_taps: int) -> bool:
    return hrir_taps > FUSED_TILE_TAPS or folded_taps <= FUSED_TILE_TAPS


def bucket_by_rate(stream_rates: Sequence[float]) -> Dict[float, List[int]]:
    """Stable bucketing: rates in first-appearance order, stream ids ascending inside a bucket."""
    buckets: Dict[float, List[int]] = {}
    for i, r in enumerate(stream_rates):
        buckets.setdefault(float(r), []).append(i)
    return buckets


def resample_tracks(tracks: np.ndarray, from_rate: float, to_rate: float, literal_vgenp: bool = False) -> np.ndarray:
    """Every HRIR track through Resampler.resampleHighQuality (identity when the rates agree, :33-35)."""
    if from_rate == to_rate:
        return np.ascontiguousarray(tracks, dtype=np.float32)
    return np.stack([Resampler.resampleHighQuality(t, from_rate, to_rate, literal_vgenp=literal_vgenp) for t in tracks])


class MixedRateBatch:
    """One preset, streams at several sample rates: `buckets[rate].spatializer` convolves that rate's streams."""

    def __init__(self, tracks, hrir_rate: float, layout: InputLayout, stream_rates: Sequence[float],
                 hrirMap: Optional[HRIRChannelMap] = None, ctx: Optional[Context] = None, literal_vgenp: bool = False,
                 equalizer=None, fold_equalizer: Optional[bool] = None, eq_tail_tolerance: float = 1e-7, eq_max_taps: int = 65536):
        """equalizer: an EqualizerDefinition applied after the spatializer, as AudioEffectGraph orders the two effects
        (AudioEffectGraph.swift:195-211).  fold_equalizer: True = folded into the HRIR (aw_eq_fold_hrir; raises EqualizerNotFoldable
        if its response is too long), False = the cascade kernel after the spatializer, None = whichever is cheaper (see _fold_pays),
        the cascade if it cannot be folded."""
        from .eq import EqualizerNotFoldable, ParametricEqualizerState, fold_equalizer as fold
        self.ctx = ctx or default_context()
        tracks = np.ascontiguousarray(tracks, dtype=np.float32)
        # the reference's chooser: 7-track files take the 7-channel map, everything else the 14-channel one
        # (HRIRManager.swift:355-360; aw_preset_activate does the same) — an 8..13-track HRIR then fails the bounds check
        cmap = hrirMap or (HRIRChannelMap.hesuvi7Channel(layout) if tracks.shape[0] == 7 else HRIRChannelMap.hesuvi14Channel(layout))
        lt, rt = cmap.resolve(layout, tracks.shape[0])
        self.left_track, self.right_track = lt, rt
        self.buckets: Dict[float, RateBucket] = {}
        for rate, ids in bucket_by_rate(stream_rates).items():
            tr = resample_tracks(tracks, hrir_rate, rate, literal_vgenp=literal_vgenp)     # literal_vgenp: what the shipped reference computes
            eq_state, folded = None, None
            if equalizer is not None and fold_equalizer is not False:
                try:
                    f = fold(tr, equalizer, rate, tailTolerance=eq_tail_tolerance, maxTaps=eq_max_taps)
                    if fold_equalizer or _fold_pays(int(tr.shape[1]), int(f.tracks.shape[1])):
                        folded, tr = f, f.tracks
                except EqualizerNotFoldable:
                    if fold_equalizer:
                        raise
            if equalizer is not None and folded is None:
                eq_state = ParametricEqualizerState(equalizer, float(rate), n_streams=len(ids), ctx=self.ctx)
            sp = Spatializer(HRIR(tr, rate, ctx=self.ctx), lt, rt, n_streams=len(ids), ctx=self.ctx)
            self.buckets[rate] = RateBucket(rate, ids, int(tr.shape[1]), sp, eq_state, folded.responseTaps if folded else 0, folded.tailBound if folded else 0.0)

    def process_device(self, in_ptrs: Dict[float, int], out_ptrs: Dict[float, int], frames: Dict[float, int]) -> None:
        """Per-rate device buffers ([bucket streams][frames][C] -> [..][frames][2]); all launches are queued on
        the context stream, buckets back to back."""
        for rate, b in self.buckets.items():
            b.spatializer.process_device(in_ptrs[rate], out_ptrs[rate], frames[rate])
            if b.equalizer is not None:
                b.equalizer.process_device(out_ptrs[rate], out_ptrs[rate], frames[rate])      # in place, same stream

    def process(self, streams: Sequence[np.ndarray], stream_rates: Sequence[float]) -> List[np.ndarray]:
        """Host convenience for tests: streams[i] is [frames_i][C] at stream_rates[i]; streams of one bucket must
        have equal length.  Returns the stereo outputs in the caller's order."""
        out: List[Optional[np.ndarray]] = [None] * len(streams)
        for rate, b in self.buckets.items():
            x = np.stack([np.asarray(streams[i], dtype=np.float32) for i in b.stream_ids])
            y = b.spatializer.process(x)
            if b.equalizer is not None:
                y = b.equalizer.process_batch(y)
            for k, i in enumerate(b.stream_ids):
                out[i] = y[k]
        return out  # type: ignore[return-value]

    def reset(self) -> None:
        for b in self.buckets.values():
            b.spatializer.reset()
            if b.equalizer is not None:
                b.equalizer.reset()
