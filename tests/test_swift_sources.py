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
    assert src.count("aw_spatializer_destroy") == 1            # only in SpatializerBox.deinit
    assert "deinit { aw_spatializer_destroy(handle) }" in src
