"""Host entry on pageable buffers, A/B of the output copy (AW_HOST_OUT_ASYNC=1: output copy threads beside the next chunk's copy-in; 0: on the
driver thread): G stereo frames/s, pinned for reference.   python tools/host_entry_ab.py [channels taps streams]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import airwave_amd as aw
C, taps, S = (int(a) for a in (sys.argv[1:4] + ["8", "4320", "128"][len(sys.argv) - 1:]))
F = 480000
rng = np.random.default_rng(1)
h = (rng.standard_normal((14, taps)) * np.exp(-np.arange(taps) / (taps / 6.0))).astype(np.float32)
lt = (np.arange(C) % 14).astype(np.int32); rt = ((np.arange(C) + 7) % 14).astype(np.int32)
x = rng.standard_normal((S, F, C), dtype=np.float32)
y = np.empty((S, F, 2), np.float32)
for rep in range(2):
    for mode in ("1", "0"):
        os.environ["AW_HOST_OUT_ASYNC"] = mode
        ctx = aw.Context(0)
        sp = aw.Spatializer(aw.HRIR(h, ctx=ctx), lt, rt, n_streams=S, ctx=ctx)
        sp.reserve_host(F)
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); sp.process_host_into(x, y); best = min(best, time.perf_counter() - t)
        xp, yp = ctx.pinned_empty((S, F, C)), ctx.pinned_empty((S, F, 2))
        xp[...] = x
        bp = 1e9
        for _ in range(3):
            t = time.perf_counter(); sp.process_host_into(xp, yp); bp = min(bp, time.perf_counter() - t)
        print(f"C={C} taps={taps} S={S} out_async={mode}: pageable {S * F / best / 1e9:.3f} G frames/s, pinned {S * F / bp / 1e9:.3f}, ratio {bp / best:.3f}, chunk {sp.info()['host_chunk_streams']}", flush=True)
        del sp, ctx, xp, yp
