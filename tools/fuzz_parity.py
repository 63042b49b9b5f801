"""Randomised GPU parity sweep (run on the GPU box; not part of the test suite): random channel counts, HRIR lengths,
stream counts, call splits and kernel choices against the float64 oracle.  Prints every failure and a summary."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import airwave_amd as aw
import airwave_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
rng = np.random.default_rng(seed)
TOL = 1e-5
fails, n, ola_used = [], 0, 0
t_end = time.time() + budget
while time.time() < t_end:
    C = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]))
    taps = int(rng.choice([1, 2, 3, 17, 255, 256, 1000, 2799, 2800, 4096, 4320, 5399, 5400, 5900, 6145, 6146, 8640, 12289, 12290, 16000]))
    if rng.random() < 0.3:
        taps = int(rng.integers(1, 14000))
    if rng.random() < 0.2:                       # long HRIRs: the partitioned path (marched CMAC; several passes above 8 partitions)
        taps = int(rng.choice([12290, 16384, 16385, 20000, 32768, 40000, 70000, int(rng.integers(12290, 80000))]))
    S = int(rng.choice([1, 2, 3, 5, 4, 7]))
    total = int(rng.integers(1, 60000)) if rng.random() < 0.8 else int(rng.choice([1, 2, 3839, 3840, 3841, 8191, 8192, 8193, 16384]))
    env = {}
    r = rng.random()
    if r < 0.25: env["AW_WINDOW"] = "8192"
    elif r < 0.5: env["AW_WINDOW"] = "16384"
    elif r < 0.6: env["AW_WINDOW"] = "4096"     # the partitioned path whatever the HRIR length
    r2 = rng.random()
    if r2 < 0.15: env["AW_PART_FWD"] = "1"
    elif r2 < 0.3: env["AW_PART_FWD"] = "2"
    if rng.random() < 0.1: env["AW_PART_CMAC"] = "group"
    if rng.random() < 0.1: env["AW_PART_HERM"] = "0"
    if rng.random() < 0.15: env["AW_SPEC_SCRATCH_MB"] = str(int(rng.choice([1, 3, 16])))      # several stream chunks
    r3 = rng.random()                            # long-window kernels (tile_lw.hpp): forced window length, forced off, or the policy
    if r3 < 0.3:
        env["AW_LW"] = str(int(rng.choice([32, 40, 48, 56, 64, 72, 80, 96, 112, 120, 128])))
        r4 = rng.random()                        # rows kernel: the 16-point form (default) or the 8-point forms of round 3
        if r4 < 0.2: env["AW_LW_ROWS_FORM"], env["AW_LW_ROWS_PB"] = "8", "2"
        elif r4 < 0.4: env["AW_LW_ROWS_FORM"] = "8"
        if rng.random() < 0.6: total = int(rng.integers(30000, 330000))          # calls that fill a window or more
    elif r3 < 0.4: env["AW_LW"] = "0"
    elif r3 < 0.6 and taps >= 8000:              # the policy's own choice on calls long enough for one or two window lengths
        total = int(rng.integers(150000, 700000)); S = int(rng.choice([1, 2, 6]))
    # round 5: the host entry in chunks of streams (1-MB chunks: a few streams each, ragged last chunk) and the host table builder
    if rng.random() < 0.3: env["AW_HOST_CHUNK_MB"] = "1"
    if rng.random() < 0.15: env["AW_LW_TABLES"] = "host"
    # round 6: the overlap-add tile (tile_ola.hpp) on calls of every size (AW_OLA_MIN_BLOCKS=0), wherever a kernel exists (AW_OLA=1) or
    # by the policy, on few or many persistent workgroups (where the launch cuts the streams into runs); HRIR lengths around its block steps
    r5 = rng.random()
    if r5 < 0.45:
        env["AW_OLA_MIN_BLOCKS"] = "0"
        if rng.random() < 0.7: env["AW_OLA"] = "1"
        if rng.random() < 0.5: env["AW_PERSISTENT_WGS"] = str(int(rng.choice([8, 9, 31, 100, 256])))
        if rng.random() < 0.7:
            C = int(rng.choice([4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]))
            taps = int(rng.choice([3585, 3969, 4097, 4098, 4320, 4609, 4610, 5000, 5121, 5122, int(rng.integers(2, 5200))]))
            env.pop("AW_WINDOW", None)
            if env.get("AW_LW") not in (None, "0") and rng.random() < 0.7: env.pop("AW_LW")
    elif r5 < 0.55: env["AW_OLA"] = "0"
    for k in ("AW_WINDOW", "AW_PART_FWD", "AW_PART_CMAC", "AW_PART_HERM", "AW_SPEC_SCRATCH_MB", "AW_LW", "AW_LW_ROWS_PB", "AW_LW_ROWS_FORM", "AW_HOST_CHUNK_MB", "AW_LW_TABLES", "AW_OLA", "AW_OLA_MIN_BLOCKS", "AW_PERSISTENT_WGS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    n_tracks = int(rng.choice([2, 7, 14]))
    h = orc.synth_hrir(n_tracks, taps, seed=int(rng.integers(1 << 30)))
    lt = rng.integers(-1 if rng.random() < 0.2 else 0, n_tracks, size=C).astype(np.int32)
    rt = rng.integers(0, n_tracks, size=C).astype(np.int32)
    rt[lt < 0] = -1                                   # an unmapped channel is skipped (both ears)
    if (lt < 0).all():
        lt[0], rt[0] = 0, n_tracks - 1
    x = orc.synth_input(S, total, C, seed=int(rng.integers(1 << 30)))
    cuts = sorted(set(int(c) for c in rng.integers(0, total + 1, size=int(rng.integers(0, 4))))) if total > 1 else []
    bounds = [0] + [c for c in cuts if 0 < c < total] + [total]
    if os.environ.get("AW_FUZZ_TRACE"):
        print("CASE", C, taps, S, total, bounds, env, n_tracks, flush=True)
    try:
        ctx = aw.Context(0) if any(k in env for k in ("AW_LW_ROWS_PB", "AW_LW_ROWS_FORM", "AW_HOST_CHUNK_MB", "AW_LW_TABLES", "AW_OLA_MIN_BLOCKS", "AW_PERSISTENT_WGS")) else None        # (those knobs are read when a context is created)
        sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx) if ctx else aw.Spatializer(aw.HRIR(h), lt, rt, n_streams=S)
        y = np.concatenate([sp.process(x[:, a:b]) for a, b in zip(bounds[:-1], bounds[1:])], axis=1)
        info = sp.info()
        ola_used += 1 if info.get("overlap_add_rows") else 0
        err = 0.0
        # the tolerance is relative to the output peak; a call of a few frames can have a peak far below the scale of the terms
        # that were summed (one frame: a single random dot product), so the peak is floored at a quarter of the typical output
        # level max|x| * sqrt(sum of the mapped tracks' energies) — seed 31 found a 1-frame call at 1.06e-5 of its own peak
        used = [t for t in list(lt) + list(rt) if t >= 0]
        floor = 0.25 * float(np.max(np.abs(x))) * float(np.sqrt(sum(float(np.sum(h[t].astype(np.float64) ** 2)) for t in used) / 2.0)) if used else 0.0
        for s in range(S):
            ref = orc.spatialize_f64(x[s], h, lt, rt)
            peak = max(float(np.max(np.abs(ref))), floor, 1e-30)
            err = max(err, float(np.max(np.abs(y[s].astype(np.float64) - ref))) / peak)
        ok = np.isfinite(y).all() and err < TOL
    except Exception as e:                              # noqa: BLE001
        ok, err, info = False, repr(e), None
    n += 1
    if not ok:
        fails.append((C, taps, S, total, bounds, env, n_tracks, lt.tolist(), rt.tolist(), info, err))
        print("FAIL", fails[-1], flush=True)
print(f"seed {seed}: {n} cases ({ola_used} whose last call ran the overlap-add tile), {len(fails)} failures")
