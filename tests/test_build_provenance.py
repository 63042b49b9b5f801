"""Build provenance covers the compiler flags, not only the sources (round-5 review): the flag manifest of the linked library is
recorded and digested, a stray environment knob changes the digest, and timing-ablation macros (wrong results) cannot get into
libairwave_hip.so — build.py refuses them without a variant suffix and the device headers #error on them."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "airwave_amd")


def _build_module():
    spec = importlib.util.spec_from_file_location("_aw_build_under_test", os.path.join(PKG, "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_recorded_manifest_is_the_default_build():
    """The library in the tree was linked with exactly the flags build.py gives in a clean environment; provenance reads that digest."""
    from airwave_amd import provenance
    env = {k: os.environ.pop(k) for k in ("AW_MARCH_SLP", "AW_KERNELS_SLP", "AW_OLS2_EVEN_SLP", "AW_EXTRA_HIPCC_FLAGS", "HIPCC") if k in os.environ}
    try:
        want = _build_module().flag_manifest()
    finally:
        os.environ.update(env)
    rec = provenance.build_flags()
    assert rec["build_flags_sha16"] == want["build_flags_sha16"] == provenance.build_flags_digest() and rec["env"] == {}
    assert rec["flags_by_source"] == want["flags_by_source"] and "-fno-slp-vectorize" in rec["flags_by_source"]["device/kernels.hip"]
    assert len(provenance.host_source_digest()) == 16 and provenance.host_source_digest() != provenance.device_source_digest()


def test_an_environment_knob_changes_the_digest(monkeypatch):
    base = _build_module().flag_manifest()["build_flags_sha16"]
    for var, val in (("AW_KERNELS_SLP", "1"), ("AW_EXTRA_HIPCC_FLAGS", "-DAW_STAGGER_SLOTS=2"), ("AW_MARCH_SLP", "1")):
        monkeypatch.setenv(var, val)
        m = _build_module().flag_manifest()
        assert m["build_flags_sha16"] != base and m["env"] == {var: val}
        monkeypatch.delenv(var)
    assert _build_module().flag_manifest(defines=["AW_EQ_CHUNK=16"])["build_flags_sha16"] != base


def test_build_refuses_an_ablation_macro_in_the_product_library():
    lib = os.path.join(PKG, "libairwave_hip.so")
    before = os.path.getmtime(lib)
    env = dict(os.environ, AW_EXTRA_HIPCC_FLAGS="-DAW_ABL_NOFFT")
    p = subprocess.run([sys.executable, os.path.join(PKG, "build.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "ablation" in p.stderr and "wrong results" in p.stderr
    assert os.path.getmtime(lib) == before                                     # nothing was compiled or linked
    assert json.load(open(os.path.join(PKG, ".build_flags.json")))["env"] == {}
    b = _build_module()
    assert b.has_ablation(["-O3", "-DAW_LW_ABL_ROWS_NOTAB"]) and b.has_ablation(["-DAW_EQ_ABL=3"]) and not b.has_ablation(["-DAW_EQ_ABL=0", "-DAW_XA_REG=1"])


def test_device_headers_error_on_ablation_macros_without_the_marker():
    """Even past build.py (a hand-written hipcc line): every device translation unit includes cplx.hpp, which refuses the macros unless the
    build says it is an ablation build.  Checked with the host compiler on the header the CPU emulation shares."""
    hdr = os.path.join(PKG, "csrc", "device", "cplx.hpp")
    for macro in ("AW_ABL_NOFFT", "AW_LW_ABL_ROWS_NOSTORE", "AW_ABL_NOBARRIER", "AW_EQ_ABL=2", "AW_ABL2=1"):
        bad = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", f"-D{macro}", hdr], capture_output=True, text=True)
        assert bad.returncode != 0 and "timing-ablation macro in a product build" in bad.stderr, macro
        ok = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", f"-D{macro}", "-DAW_ABLATION_BUILD=1", hdr], capture_output=True, text=True)
        assert ok.returncode == 0, ok.stderr
    assert subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-DAW_EQ_ABL=0", hdr], capture_output=True).returncode == 0
