"""Measured HBM bandwidth of this box next to the 8 TB/s spec peak: device-to-device copy and a
read-only reduction over buffers much larger than the caches."""
import time

import torch

n = 1 << 28                                    # 2 GiB of f64
a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * n * 8),
                              ("sum  (read only)   ", lambda: a.sum(), n * 8)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {bytes_moved / dt / 1e12:.2f} TB/s")
