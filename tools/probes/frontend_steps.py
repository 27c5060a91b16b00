"""Per-step host time of `Atmosphere.internal_object()` at the C2 size with torch in the process, as in bench.py (step, then the time a
device synchronisation still waits)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
torch.cuda.set_device(0)
import sasktran2_amd as sk
from bench_frontend import case
from tests import scenarios as sc
c = case(10_000)
cfg = sk.Config(); cfg.num_streams = 16; cfg.num_singlescatter_moments = 16
geo = sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.Spherical)
atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, thermal=True)
for i in range(8):
    t = time.perf_counter(); fe = atmo.internal_object(); a = time.perf_counter(); torch.cuda.synchronize(); b = time.perf_counter()
    print(i, f"step {1e3*(a-t):.3f} ms, sync {1e3*(b-a):.3f} ms, reused {fe.spectra_reused}", flush=True)
# the same again with the cyclic collector off: a one-off step of tens of ms in the first loop that is gone here is a generation-2 collection
# walking torch's module objects, not the front end
import gc
atmo2 = sc.frontend_product_atmosphere(sk, c, geo, cfg, thermal=True)
gc.collect()
gc.disable()
for i in range(8):
    t = time.perf_counter(); fe = atmo2.internal_object(); a = time.perf_counter(); torch.cuda.synchronize(); b = time.perf_counter()
    print("gc off", i, f"step {1e3*(a-t):.3f} ms, sync {1e3*(b-a):.3f} ms, reused {fe.spectra_reused}", flush=True)
gc.enable()
t = time.perf_counter(); n = gc.collect(); print(f"a full collection takes {1e3*(time.perf_counter()-t):.1f} ms ({n} objects freed)")
