"""Measured HBM copy ceiling of the box (SURVEY 8d asks for it beside the 8 TB/s vendor peak): device-to-device copy of a
4 GiB buffer, bytes read + written per second."""
import time
import torch
n = 1 << 30                      # float32 elements = 4 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"copy 4 GiB: {dt * 1e3:.3f} ms -> {2 * 4 * n / dt / 1e12:.2f} TB/s read+write")
