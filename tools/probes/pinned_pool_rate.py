"""Copy rates of arrays from the mirror's page-locked pool (sasktran2_amd.hostmem) against ordinary NumPy arrays, through hipMemcpy."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sasktran2_amd.hostmem import pool  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 65 * 2 ** 20 // 8
d = torch.zeros(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, a in (("pool", pool.zeros((16, 51, n // 816), order="F")), ("numpy", np.zeros((16, 51, n // 816), order="F"))):
    nbytes = a.size * 8
    for kind, label in ((1, "h2d"), (4, "h2d default"), (2, "d2h")):
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            if kind == 2:
                hip.hipMemcpy(a.ctypes.data, d.data_ptr(), nbytes, 2)
            else:
                hip.hipMemcpy(d.data_ptr(), a.ctypes.data, nbytes, kind)
            best = min(best, time.perf_counter() - t)
        print(f"{name:6s} {label:12s} {nbytes / best / 1e9:6.1f} GB/s ({best * 1e3:.2f} ms for {nbytes / 2 ** 20:.0f} MiB)", flush=True)
print("pool pinned:", not pool._failed)
