"""Which sources a measurement was made on.  bench.py and the profile tools run on GPU boxes that have no .git, so the identity of
the device code is a digest of the files themselves; the git HEAD recorded by the last build that ran inside a checkout travels
along as information only (airwave_amd/.build_head, written by build.py)."""
from __future__ import annotations

import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEVICE_DIR = os.path.join(HERE, "csrc", "device")


def device_source_digest() -> str:
    """sha256 over every file under csrc/device (relative path + contents), first 16 hex digits."""
    h = hashlib.sha256()
    for d, _, fs in sorted(os.walk(DEVICE_DIR)):
        for f in sorted(fs):
            p = os.path.join(d, f)
            h.update(os.path.relpath(p, DEVICE_DIR).encode())
            h.update(b"\0")
            h.update(open(p, "rb").read())
            h.update(b"\0")
    return h.hexdigest()[:16]


def build_head() -> str:
    """git HEAD (+ '+dirty') as of the last build inside a checkout; 'unknown' if no build recorded one."""
    try:
        return open(os.path.join(HERE, ".build_head")).read().strip() or "unknown"
    except OSError:
        return "unknown"



CSRC_DIR = os.path.join(HERE, "csrc")


def host_source_digest() -> str:
    """sha256 over the host side of the library (runtime.cpp, eq_runtime.cpp, host/*): the launch policy, chunking and table builders
    that decide which kernels run and how many bytes a step moves — a traffic profile is only current if these agree too."""
    h = hashlib.sha256()
    files = [os.path.join(CSRC_DIR, f) for f in sorted(os.listdir(CSRC_DIR)) if f.endswith((".cpp", ".hpp", ".h"))]
    host = os.path.join(CSRC_DIR, "host")
    files += [os.path.join(host, f) for f in sorted(os.listdir(host))]
    for p in files:
        h.update(os.path.relpath(p, CSRC_DIR).encode())
        h.update(b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def build_flags() -> dict:
    """The flag manifest of the library that was last linked (airwave_amd/.build_flags.json, written by build.py): per-source hipcc
    flags, the environment knobs that shaped them and their digest.  {} if no build recorded one."""
    import json
    try:
        return json.load(open(os.path.join(HERE, ".build_flags.json")))
    except (OSError, ValueError):
        return {}


def build_flags_digest() -> str:
    return build_flags().get("build_flags_sha16", "unrecorded")
