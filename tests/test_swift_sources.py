"""The Swift drop-in cannot be compiled in this image (no swiftc); these checks pin the parts of its contract that a
reviewer would otherwise have to re-read every round: the render side never blocks on the publication lock and never
destroys a handle (HRIRManager.swift:133-147, 539-548)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "swift", "AirwaveHIP", "Sources", "AirwaveHIP", "HIPSpatialEffect.swift")


def _body(src: str, signature: str) -> str:
    i = src.index(signature)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += {"{": 1, "}": -1}.get(src[k], 0)
        if depth == 0:
            return src[j:k + 1]
        k += 1


def test_render_side_uses_try_lock_and_never_destroys():
    src = open(SRC).read()
    proc = _body(src, "public func process(")
    assert "stateLock.withLockIfAvailable" in proc and "stateLock.withLock " not in proc and "stateLock.withLock{" not in proc
    assert "aw_spatializer_destroy" not in proc
    assert "withLockIfAvailable" in _body(src, "private func flushAwaitingRetirement(")
    for fn in ("private func retire(", "private func flushAwaitingRetirement("):
        b = _body(src, fn)
        assert not re.search(r"retiredLock\.withLock\s*\{", b) and "aw_spatializer_destroy" not in b


def test_publication_defers_destruction():
    src = open(SRC).read()
    act = _body(src, "public func activatePreset(")
    assert "stateLock.withLock" in act and "aw_spatializer_destroy" not in act and "aw_spatializer_reserve" in act
    assert src.count("aw_spatializer_destroy(") == 1           # the only call: SpatializerBox.deinit
    assert "deinit { aw_spatializer_destroy(handle) }" in src


EQ = os.path.join(ROOT, "swift", "AirwaveHIP", "Sources", "AirwaveHIP", "HIPEqualizerEffect.swift")
LOCK = os.path.join(ROOT, "swift", "AirwaveHIP", "Sources", "AirwaveHIP", "TryLock.swift")


def test_equalizer_effect_publishes_behind_a_try_lock_and_retires_processors():
    """EqualizerRuntimeEffect.swift:6-8,57-61: `prepare` publishes a new processor under the lock and never destroys the one the
    render thread may be using; `process` takes a try-lock snapshot, never blocks, never destroys."""
    src = open(EQ).read()
    proc = _body(src, "public func process(")
    assert "processorLock.withLockIfAvailable" in proc and not re.search(r"processorLock\.withLock\s*\{", proc)
    assert "aw_eq_destroy" not in proc and "audioThreadProcessor" in proc
    prep = _body(src, "public func prepare(")
    assert "processorLock.withLock" in prep and "aw_eq_destroy" not in prep
    assert src.count("aw_eq_destroy(") == 1 and "deinit { aw_eq_destroy(handle) }" in src       # the only call: EqualizerBox.deinit
    for fn in ("private func retire(", "private func flushAwaitingRetirement("):
        b = _body(src, fn)
        assert not re.search(r"retiredLock\.withLock\s*\{", b) and "aw_eq_destroy" not in b
    assert "withLockIfAvailable" in _body(src, "private func flushAwaitingRetirement(")


def test_boxes_retain_their_context_and_the_package_builds_where_the_library_exists():
    """A handle's destroy call dereferences its context, so the boxes that own handles keep the context alive (teardown order);
    the try-lock is pthread based — `import os` / OSAllocatedUnfairLock exist on Darwin only, the ROCm library on Linux only."""
    sp, eq, lock = open(SRC).read(), open(EQ).read(), open(LOCK).read()
    assert "let context: HIPContext" in _body(sp, "final class SpatializerBox") and "let context: HIPContext" in _body(eq, "final class EqualizerBox")
    for src in (sp, eq):
        assert "import os" not in src and "OSAllocatedUnfairLock" not in src and "nonisolated(unsafe)" not in src
        assert "removeAll()" not in src                        # keepingCapacity: the render thread's later append must not allocate
    assert "pthread_mutex_trylock" in lock and "canImport(Glibc)" in lock
    act = _body(sp, "public func activatePreset(")
    assert re.search(r"guard aw_spatializer_reserve\(.*== AW_OK else", act)       # a failed reserve is an activation failure


def test_equalizer_effect_has_the_protocols_shape_and_maps_errors_like_the_reference():
    """AudioEqualizerEffect (AudioEffectGraph.swift:51-54): prepare(definition:sampleRate:) / setTarget(definition:) over a definition
    that mirrors EqualizerPreset.swift:9-27 field for field (sourceLine / sourceNumber included), and the error mapping of
    EqualizerRuntimeEffect.swift:80-100: .invalidFilter(line:reason:) with the sourceLine of the enabled filter the library names."""
    src = open(EQ).read()
    assert re.search(r"public func prepare\(definition: HIPEqualizerDefinition\?, sampleRate: Double\) throws", src)
    assert re.search(r"public func setTarget\(definition: HIPEqualizerDefinition\?\) throws", src)
    assert "preampDB: Double?, filters:" not in src                                  # round 5's shape is gone
    flt = _body(src, "public struct HIPEqualizerFilter")
    for field in ("let sourceLine: Int", "let sourceNumber: Int?", "let isEnabled: Bool", "let type: HIPEqualizerFilterType",
                  "let frequencyHz: Double", "let gainDB: Double", "let q: Double"):
        assert field in flt, field
    d = _body(src, "public struct HIPEqualizerDefinition")
    assert "let preampDB: Double" in d and "let filters: [HIPEqualizerFilter]" in d and "preampDB: Double = 0, filters: [HIPEqualizerFilter] = []" in d
    err = _body(src, "public enum HIPEqualizerError")
    assert "case invalidFilter(line: Int?, reason: String)" in err and "case invalidSampleRate" in err and "case unavailable(String)" in err
    assert "var filterLine: Int?" in err
    # the definition handed to the library carries the source lines; the mapping reads the structured error, not the message text
    mk = _body(src, "static func makeDefinitionHandle(")
    assert "aw_eq_definition_set_source" in mk and "f.sourceLine" in mk and "f.sourceNumber ?? -1" in mk
    m = _body(src, "static func map(")
    assert "aw_last_eq_filter_error(&index, &kind, &line)" in m and "filter(\\.isEnabled)" in m and "enabled[Int(index)].sourceLine" in m
    assert 'case AW_ERR_EQ_NON_FINITE_PREAMP:' in m and '"Preamp produces a non-finite gain."' in m
    assert "Equalizer supports at most 64 filters; received" in m and "case AW_ERR_EQ_INVALID_SAMPLE_RATE:" in m
    pub = _body(src, "private func publish(")
    assert pub.index("Self.map(st") < pub.index("aw_eq_set_target(box.handle, nil)")    # the error state is read before the unity fallback replaces it
    desc = _body(src, "static func biquadErrorDescription(")
    for text in ("Sample rate must be finite and positive.", "Frequency must be finite, positive, and below Nyquist.", "Q must be finite and positive.",
                 "Filter parameters must be finite.", "Filter coefficients must be finite."):
        assert text in desc                                                          # BiquadCoefficientBuilder.swift:18-26
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "final class HIPEqualizerEffectAdapter: AudioEqualizerEffect" in integ and "func prepare(definition: EqualizerDefinition?, sampleRate: Double) throws" in integ
    assert "EqualizerAudioEffectError.invalidFilter(line: line, reason: reason)" in integ


def test_batch_spatializer_folds_a_known_equalizer():
    src = open(os.path.join(ROOT, "swift", "AirwaveHIP", "Sources", "AirwaveHIP", "BatchSpatializer.swift")).read()
    init = _body(src, "public init(context: HIPContext, tracks:")
    assert "equalizer: HIPEqualizerDefinition? = nil" in src and init.count("aw_eq_fold_hrir(") == 2       # the length, then the tracks
    assert "aw_eq_definition_destroy(def)" in init and "foldedEqualizer = (Int(response), bound)" in init
