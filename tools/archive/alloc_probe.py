"""hipMalloc / hipFree wall time by size on this box (why a scratch pool that must GROW costs what it costs): python tools/archive/alloc_probe.py"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def t(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
print("GiB   malloc ms   memset ms   free ms")
for gib in (1, 4, 8, 16, 19, 24, 32, 34, 41, 64):
    p = ctypes.c_void_p()
    n = ctypes.c_size_t(gib << 30)
    tm, rc = t(lambda: hip.hipMalloc(ctypes.byref(p), n))
    if rc != 0:
        print(gib, "hipMalloc failed", rc); continue
    ts, _ = t(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
    tf, _ = t(lambda: hip.hipFree(p))
    print(f"{gib:3d}   {tm:9.2f}   {ts:9.2f}   {tf:7.2f}", flush=True)
# a second allocation right after freeing a smaller one (the pool growing from 19 to 34 GiB)
p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(19 << 30)); hip.hipMemset(p, 0, ctypes.c_size_t(19 << 30)); hip.hipDeviceSynchronize()
tf, _ = t(lambda: hip.hipFree(p))
q = ctypes.c_void_p(); tm, _ = t(lambda: hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(34 << 30)))
print(f"grow 19 -> 34 GiB: free {tf:.2f} ms, malloc {tm:.2f} ms")
