"""TEST-ONLY: ctypes access to the thread-emulated tile kernel (tests/emu/emu_harness.cpp)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_DEFS = os.environ.get("AW_EMU_DEFINES", "").split()          # e.g. "-DAW_SUBFFT_SKEW=1": emulate a tuning variant
_LIB = os.path.join(_HERE, "libemu.so" if not _DEFS else "libemu_variant.so")
_SRCS = [os.path.join(_HERE, "emu_harness.cpp"), os.path.join(_HERE, "emu_lw.cpp"), os.path.join(_ROOT, "airwave_amd/csrc/host/tables.cpp"),
         os.path.join(_ROOT, "airwave_amd/csrc/host/eq.cpp")]
_DEPS = _SRCS + [os.path.join(_HERE, "emu_ctx.hpp")] + [os.path.join(_ROOT, "airwave_amd/csrc/device", f) for f in ("tile_ols.hpp", "tile_ola.hpp", "tile_ols2.hpp", "tile_march.hpp", "tile_lw.hpp", "tile_lw16.hpp", "cplx.hpp", "eq_cascade.hpp")] + [
    os.path.join(_ROOT, "airwave_amd/csrc/host/tables.hpp"), os.path.join(_ROOT, "airwave_amd/csrc/host/eq.hpp")]
_lib = None

fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int32)


def lib():
    global _lib
    if _lib is None:
        if _DEFS or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(d) for d in _DEPS):
            # the translation units compile side by side (the long-window kernels alone are 176 template instantiations)
            units = [(src, []) for src in _SRCS] + [(os.path.join(_HERE, "emu_lw.cpp"), [f"-DEMU_LW_PART={k}"]) for k in (1, 2, 3)]
            objs = [f"{_LIB}.{i}.o" for i in range(len(units))]
            procs = [subprocess.Popen(["g++", "-std=c++20", "-O2", "-pthread", "-fPIC"] + _DEFS + extra + ["-c", src, "-o", o]) for (src, extra), o in zip(units, objs)]
            if any(p.wait() != 0 for p in procs):
                raise RuntimeError("emulation harness failed to compile")
            subprocess.run(["g++", "-shared", "-pthread", "-o", _LIB] + objs, check=True)
            for o in objs:
                os.remove(o)
        _lib = ctypes.CDLL(_LIB)
        _lib.emu_fused_ols.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip,
                                       ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.emu_fused_ola.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip,
                                       ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.emu_partitioned.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip,
                                         ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
        _lib.emu_longwin.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip,
                                     ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        _lib.emu_lw_row_map.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ip, ctypes.POINTER(ctypes.c_longlong)]
        _lib.emu_lw_row_map.restype = ctypes.c_longlong
        _lib.emu_fft_small.argtypes = [fp, ctypes.c_int, ctypes.c_int]
        _lib.emu_sub_fft512h.argtypes = [fp, fp, fp]
        _lib.emu_lw_dft.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.emu_eq_process.argtypes = [fp, fp, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_longlong, ctypes.c_double,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int]
    return _lib


def fused_ols(x, tracks, left_track, right_track, hop=None, hist=None, variant=1):
    """x: [streams][frames][C] float32 -> [streams][frames][2]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, F, C = x.shape
    tr = np.ascontiguousarray(tracks, dtype=np.float32)
    lt = np.ascontiguousarray(left_track, dtype=np.int32)
    rt = np.ascontiguousarray(right_track, dtype=np.int32)
    if hop is None:
        hop = 8192 - (tr.shape[1] - 1)
    out = np.full((S, F, 2), np.nan, dtype=np.float32)
    h = None
    if hist is not None:
        h = np.ascontiguousarray(hist, dtype=np.float32)
        assert h.shape == ((S, 8192 - hop, C) if variant != 2 else (S, 2 * (tr.shape[1] // 2), C))
    rc = lib().emu_fused_ols(x.ctypes.data_as(fp), out.ctypes.data_as(fp), None if h is None else h.ctypes.data_as(fp),
                             tr.ctypes.data_as(fp), tr.shape[0], tr.shape[1], C, lt.ctypes.data_as(ip),
                             rt.ctypes.data_as(ip), F, S, hop, variant)
    assert rc == 0
    return out


def fused_ola(x, tracks, left_track, right_track, H=None, hist=None, workgroups=3):
    """The overlap-add tile (tile_ola.hpp): x [streams][frames][C] -> [streams][frames][2]; hist: [streams][hist_len][C] (hist_len >= taps - 1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, F, C = x.shape
    tr = np.ascontiguousarray(tracks, dtype=np.float32)
    lt = np.ascontiguousarray(left_track, dtype=np.int32)
    rt = np.ascontiguousarray(right_track, dtype=np.int32)
    if H is None:
        H = min(8, (8192 - (tr.shape[1] - 1)) // 512)
    out = np.full((S, F, 2), np.nan, dtype=np.float32)
    h = None
    hist_len = tr.shape[1] - 1
    if hist is not None:
        h = np.ascontiguousarray(hist, dtype=np.float32)
        hist_len = h.shape[1]
        assert h.shape == (S, hist_len, C)
    rc = lib().emu_fused_ola(x.ctypes.data_as(fp), out.ctypes.data_as(fp), None if h is None else h.ctypes.data_as(fp),
                             tr.ctypes.data_as(fp), tr.shape[0], tr.shape[1], C, lt.ctypes.data_as(ip), rt.ctypes.data_as(ip), F, S, hist_len, H, workgroups)
    assert rc == 0, rc
    return out


def partitioned(x, tracks, left_track, right_track, hist=None, cmac="march"):
    """Long-HRIR path (hop 4096, P = ceil(taps/4096) partitions); cmac = "march" (tile_march.hpp, the default
    kernel) or "group" (the block-group kernel kept for more than 8 channel pairs)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, F, C = x.shape
    tr = np.ascontiguousarray(tracks, dtype=np.float32)
    lt = np.ascontiguousarray(left_track, dtype=np.int32)
    rt = np.ascontiguousarray(right_track, dtype=np.int32)
    out = np.full((S, F, 2), np.nan, dtype=np.float32)
    h = None if hist is None else np.ascontiguousarray(hist, dtype=np.float32)
    rc = lib().emu_partitioned(x.ctypes.data_as(fp), out.ctypes.data_as(fp), None if h is None else h.ctypes.data_as(fp),
                               tr.ctypes.data_as(fp), tr.shape[0], tr.shape[1], C, lt.ctypes.data_as(ip),
                               rt.ctypes.data_as(ip), F, S, 1 if cmac == "group" else 0)
    assert rc == 0
    return out


def longwin(x, tracks, left_track, right_track, R=32, hop=None, hist=None, rows_pb=2, hist_out=None):
    """Long-window path (tile_lw.hpp): windows of R x 4096 frames; x: [streams][frames][C] -> [streams][frames][2]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, F, C = x.shape
    tr = np.ascontiguousarray(tracks, dtype=np.float32)
    lt = np.ascontiguousarray(left_track, dtype=np.int32)
    rt = np.ascontiguousarray(right_track, dtype=np.int32)
    N = R * 4096
    if hop is None:
        hop = N - (tr.shape[1] - 1)
    out = np.full((S, F, 2), np.nan, dtype=np.float32)
    h = None
    if hist is not None:
        h = np.ascontiguousarray(hist, dtype=np.float32)
        assert h.shape == (S, N - hop, C)
    rc = lib().emu_longwin(x.ctypes.data_as(fp), out.ctypes.data_as(fp), None if h is None else h.ctypes.data_as(fp),
                           tr.ctypes.data_as(fp), tr.shape[0], tr.shape[1], C, lt.ctypes.data_as(ip), rt.ctypes.data_as(ip),
                           F, S, R, hop, rows_pb, None if hist_out is None else hist_out.ctypes.data_as(fp))
    assert rc == 0, rc
    return out


def lw_dft(v, inverse=False, odd=False):
    """The split / merge kernels' in-register DFT of len(v) points (n = 2^a q); odd: sampled at k + 1/2."""
    a = np.ascontiguousarray(np.asarray(v, dtype=np.complex64)).view(np.float32).copy()
    assert lib().emu_lw_dft(a.ctypes.data_as(fp), a.size // 2, int(inverse), int(odd)) == 0
    return a.view(np.complex64)


def sub_fft512h(rows):
    """(spectra in natural order, unnormalised inverse of those spectra) of two 512-point rows through the half-wave row transform."""
    a = np.ascontiguousarray(np.asarray(rows, dtype=np.complex64).reshape(2, 512)).view(np.float32)
    f = np.zeros_like(a); b = np.zeros_like(a)
    assert lib().emu_sub_fft512h(a.ctypes.data_as(fp), f.ctypes.data_as(fp), b.ctypes.data_as(fp)) == 0
    return f.view(np.complex64).reshape(2, 512), b.view(np.complex64).reshape(2, 512)


def fft_small(v, inverse=False):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.complex64)).view(np.float32).copy()
    assert lib().emu_fft_small(a.ctypes.data_as(fp), a.size // 2, int(inverse)) == 0
    return a.view(np.complex64)


def eq_process(x, sample_rate, preamp_db, filters, z=None, ear_split=False):
    """x: [streams][frames][2] float32; filters: [(type, fc, gain_db, q)].  Returns (y, z) with z the carried state."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    S, F, _ = x.shape
    fl = np.ascontiguousarray(np.asarray(filters, dtype=np.float64).reshape(-1, 4))
    K = fl.shape[0]
    z = np.zeros((S, max(K, 1), 4), np.float64) if z is None else np.ascontiguousarray(z, dtype=np.float64)
    out = np.full((S, F, 2), np.nan, dtype=np.float32)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib().emu_eq_process(x.ctypes.data_as(fp), out.ctypes.data_as(fp), z.ctypes.data_as(dp), S, F, sample_rate, preamp_db,
                              fl.ctypes.data_as(dp), K, int(ear_split))
    assert rc == K, rc
    return out, z


def eq_tables(sample_rate, preamp_db, filters):
    """Host-built tables of the EQ kernel: (tab [K][tab_doubles], plane [K][64][4], chunk)."""
    fl = np.ascontiguousarray(np.asarray(filters, dtype=np.float64).reshape(-1, 4))
    K = fl.shape[0]
    dp = ctypes.POINTER(ctypes.c_double)
    td, ch = ctypes.c_int(), ctypes.c_int()
    rc = lib().emu_eq_tables(ctypes.c_double(sample_rate), ctypes.c_double(preamp_db), fl.ctypes.data_as(dp), K, None, None, ctypes.byref(td), ctypes.byref(ch))
    assert rc == K, rc
    tab = np.empty((K, td.value), np.float64)
    plane = np.empty((K, 64, 4), np.float64)
    rc = lib().emu_eq_tables(ctypes.c_double(sample_rate), ctypes.c_double(preamp_db), fl.ctypes.data_as(dp), K, tab.ctypes.data_as(dp), plane.ctypes.data_as(dp),
                             ctypes.byref(td), ctypes.byref(ch))
    assert rc == K, rc
    return tab, plane, ch.value


def lw_row_map(n_rp: int, n_sw: int, groups: int = 8):
    """(visits per (row pair, stream-window), tiles per XCD group) of the rows kernels' tile map (tile_lw.hpp: lw_row_map / lw_row_tile)."""
    hits = np.zeros((n_rp, n_sw), dtype=np.int32)
    per = np.zeros(groups, dtype=np.int64)
    total = lib().emu_lw_row_map(n_rp, n_sw, groups, hits.ctypes.data_as(ip), per.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    assert total == n_rp * n_sw, total
    return hits, per
