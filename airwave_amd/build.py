"""Builds airwave_amd/libairwave_hip.so for gfx950 with hipcc (no torch, no cmake).

    python airwave_amd/build.py [--force]      (run as a script: importing the package needs the built library)

hipcc cross-compiles without a GPU; the .so stays in-tree (git-ignored) so it travels with the
repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libairwave_hip.so")
OBJ = os.path.join(HERE, "_build")
ARCH = "gfx950"

SOURCES = [
    "device/kernels.hip",
    "device/march_kernels.hip",
    "device/lw_kernels.hip",
    "device/lw_split_a.hip",
    "device/lw_split_b.hip",
    "device/lw_split_c.hip",
    "device/lw_split_d.hip",
    "device/lw_split_e.hip",
    "device/ols2_even_kernels.hip",
    "device/ola_kernels.hip",
    "device/ola_kernels_a.hip",
    "device/ola_kernels_b.hip",
    "device/ola_kernels_c.hip",
    "device/ola_kernels_d.hip",
    "device/ola_kernels_e.hip",
    "device/ola_kernels_f.hip",
    "device/eq_kernels.hip",
    "device/probe_kernels.hip",
    "device/prep_kernels.hip",
    "runtime.cpp",
    "eq_runtime.cpp",
    "host/eq.cpp",
    "host/tables.cpp",
    "host/host_api.cpp",
]
# per-source flags (see the header comment of the source)
EXTRA_FLAGS = {"device/march_kernels.hip": [] if os.environ.get("AW_MARCH_SLP") else ["-fno-slp-vectorize"],
               # the tile kernels too: SLP packs butterflies into v_pk_add/mul/fma_f32, which issue at HALF the rate of the scalar
               # forms on gfx950 (tools/ubench/valu_rate: 4.7 against 2.45 cycles) and need v_mov pairs on top — the FFT core alone
               # runs 3.03 instead of 3.78 us per transform without it (tools/ubench/fft_core), cfg 4 / 5 gain 10-13 %, cfg 3 4 %
               # long-window kernels: at their 16 waves per CU the first sub-FFT exchange in registers (permlane swaps) and plain
               # single ds_read_b64 (no read2 fusion) measure 4.88 -> 4.72-4.75 ms on the cfg 3 rows kernel (the 8-wave tile kernels: +-1 %)
               "device/lw_kernels.hip": ["-fno-slp-vectorize", "-DAW_XA_REG=1", "-DAW_LDS_ATOMIC_READS=1"],
               "device/lw_split_a.hip": ["-fno-slp-vectorize"], "device/lw_split_b.hip": ["-fno-slp-vectorize"], "device/lw_split_c.hip": ["-fno-slp-vectorize"],
               "device/lw_split_d.hip": ["-fno-slp-vectorize"], "device/lw_split_e.hip": ["-fno-slp-vectorize"],
               # the overlap-add tile kernels: the same tile code, the same reason
               "device/ola_kernels.hip": ["-fno-slp-vectorize"], "device/ola_kernels_a.hip": ["-fno-slp-vectorize"], "device/ola_kernels_b.hip": ["-fno-slp-vectorize"],
               "device/ola_kernels_c.hip": ["-fno-slp-vectorize"], "device/ola_kernels_d.hip": ["-fno-slp-vectorize"], "device/ola_kernels_e.hip": ["-fno-slp-vectorize"], "device/ola_kernels_f.hip": ["-fno-slp-vectorize"],
               "device/kernels.hip": [] if os.environ.get("AW_KERNELS_SLP") else ["-fno-slp-vectorize"],
               # the even 16384-frame layouts kept SLP through round 3 (their 8 x 8 x 8 form measured faster with it); on the half-wave row
               # transform they do not: 4 / 6 / 8 channels 66.8 / 46.5 / 33.4 -> 68.8 / 52.9 / 35.4 G frames/s (tools/ols2_ab.py)
               "device/ols2_even_kernels.hip": [] if os.environ.get("AW_OLS2_EVEN_SLP") else ["-fno-slp-vectorize"]}
HEADERS = sorted(os.path.relpath(os.path.join(d, f), CSRC) for d, _, fs in os.walk(CSRC) for f in fs if f.endswith((".hpp", ".h"))) + [
    "../../include/airwave_hip.h",
]


# environment variables that change what gets compiled: recorded (and digested) with the flags, so that two libraries built from the same
# sources with different tuning cannot share a provenance (round-5 review: the digest covered sources, not flags)
FLAG_ENV = ("AW_MARCH_SLP", "AW_KERNELS_SLP", "AW_OLS2_EVEN_SLP", "AW_EXTRA_HIPCC_FLAGS", "HIPCC")
ABLATION_PREFIXES = ("AW_ABL", "AW_LW_ABL", "AW_EQ_ABL")


def common_flags(defines=()):
    return ["-O3", "-std=c++17", "-fPIC", "-pthread", "-fvisibility=hidden", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-result", "-ffp-contract=fast"] + [f"-D{d}" for d in defines] + os.environ.get("AW_EXTRA_HIPCC_FLAGS", "").split()


def flag_manifest(defines=()) -> dict:
    """The full per-source flag list of a build plus the environment knobs behind it, and its digest (first 16 hex of sha256 over the
    canonical JSON).  airwave_amd/.build_flags.json holds the manifest of the library that was last linked; provenance.py reads it."""
    import hashlib
    import json
    by_source = {src: common_flags(defines) + EXTRA_FLAGS.get(src, []) for src in SOURCES}
    env = {k: os.environ.get(k, "") for k in FLAG_ENV if os.environ.get(k)}
    body = {"arch": ARCH, "flags_by_source": by_source, "env": env}
    body["build_flags_sha16"] = hashlib.sha256(json.dumps(body, sort_keys=True).encode()).hexdigest()[:16]
    return body


def has_ablation(flags) -> bool:
    return any(f.startswith("-D") and f[2:].startswith(ABLATION_PREFIXES) and not f.endswith("=0") for f in flags)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _record_head() -> None:
    """airwave_amd/.build_head = git HEAD (+dirty) when the build runs inside a checkout (provenance.py; the GPU boxes have no .git)."""
    root = os.path.dirname(HERE)
    if not os.path.isdir(os.path.join(root, ".git")):
        return
    try:
        head = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "airwave_amd/csrc"], capture_output=True, text=True, check=True).stdout.strip()
        with open(os.path.join(HERE, ".build_head"), "w") as f:
            f.write(head + ("+dirty" if dirty else "") + "\n")
    except Exception:
        pass


def build(force: bool = False, verbose: bool = False, stamps: bool = False, defines=(), suffix: str = "", only=()) -> str:
    """stamps=True builds the DIAGNOSTIC variant (phase time stamps, -DAW_STAMPS=1) as
    libairwave_hip_stamps.so; `defines` + `suffix` build tuning variants (libairwave_hip_<suffix>.so)
    for A/B runs.  Variants are never loaded unless AIRWAVE_HIP_LIBRARY points at them."""
    global OUT, OBJ
    defines = list(defines)
    # a timing ablation (wrong results) is only ever built as a suffixed variant that says what it is; cplx.hpp #errors on the macros otherwise
    if has_ablation(common_flags(defines)):
        if not suffix:
            raise RuntimeError("timing-ablation macros (AW_ABL_* / AW_LW_ABL_* / AW_EQ_ABL) give wrong results: build them as a variant "
                               "(suffix=..., libairwave_hip_<suffix>.so), never as libairwave_hip.so")
        defines.append("AW_ABLATION_BUILD=1")
    if stamps:
        suffix = suffix or "stamps"
        defines.append("AW_STAMPS=1")
    main_obj = OBJ
    if suffix:
        OUT = os.path.join(HERE, f"libairwave_hip_{suffix}.so")
        OBJ = os.path.join(HERE, f"_build_{suffix}")
    os.makedirs(OBJ, exist_ok=True)
    if suffix and only:          # a variant whose defines touch only some sources: the other objects are the main build's
        import shutil
        for src in SOURCES:
            o = src.replace("/", "_") + ".o"
            if src not in only and os.path.exists(os.path.join(main_obj, o)):
                shutil.copy2(os.path.join(main_obj, o), os.path.join(OBJ, o))
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    import json
    manifest = flag_manifest(defines)
    objs, jobs = [], []
    for src in SOURCES:
        spath = os.path.join(CSRC, src)
        opath = os.path.join(OBJ, src.replace("/", "_") + ".o")
        objs.append(opath)
        flags = manifest["flags_by_source"][src]
        # an object is stale when a source or header is newer — or when it was compiled with other flags (its .flags stamp)
        try:
            same_flags = open(opath + ".flags").read() == " ".join(flags) or (suffix and only and src not in only)
        except OSError:
            same_flags = bool(suffix and only and src not in only)
        if (force and (not only or src in only)) or (src in only) or not same_flags or _stale(opath, [spath] + hdrs):
            jobs.append((opath, flags, [hipcc()] + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", spath, "-o", opath]))
    if jobs:          # the translation units are independent: a few hipcc processes side by side (each peaks at ~2 GB)
        from concurrent.futures import ThreadPoolExecutor
        def run(job):
            opath, flags, cmd = job
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            with open(opath + ".flags", "w") as f:
                f.write(" ".join(flags))
        with ThreadPoolExecutor(max_workers=int(os.environ.get("AW_BUILD_JOBS", "8"))) as ex:
            list(ex.map(run, jobs))
    _record_head()
    flags_file = os.path.join(HERE, ".build_flags.json" if not suffix else f".build_flags_{suffix}.json")
    try:
        recorded = json.load(open(flags_file)).get("build_flags_sha16")
    except (OSError, ValueError):
        recorded = None
    if force or jobs or _stale(OUT, objs) or recorded != manifest["build_flags_sha16"]:
        cmd = [hipcc(), "-shared", "-fPIC", "-pthread", f"--offload-arch={ARCH}", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        with open(flags_file, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    return OUT


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    sfx = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--suffix=")), "")
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, stamps="--stamps" in sys.argv, defines=defs, suffix=sfx, only=only))
