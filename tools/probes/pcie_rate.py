"""Host <-> device copy rates of this box: pageable and pinned host memory, both directions (what the host entry points pay)."""
import time

import numpy as np
import torch

n = 208 * 2 ** 20 // 8
dev = torch.device("cuda", 0)
d = torch.zeros(n, dtype=torch.float64, device=dev)
page = torch.from_numpy(np.zeros(n))
pin = torch.zeros(n, dtype=torch.float64).pin_memory()
torch.cuda.synchronize()
for name, host in (("pageable", page), ("pinned", pin)):
    for direction in ("h2d", "d2h"):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if direction == "h2d":
                d.copy_(host)
            else:
                host.copy_(d)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        print(f"{name:9s} {direction}: {n * 8 / best / 1e9:6.1f} GB/s ({best * 1e3:.1f} ms for 208 MiB)")
a = np.zeros(n)
b = np.ones(n)
t = time.perf_counter()
a[:] = b
print(f"host memcpy (one thread): {n * 8 / (time.perf_counter() - t) / 1e9:.1f} GB/s")
