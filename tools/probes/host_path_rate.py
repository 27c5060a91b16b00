"""PCIe-inclusive rate of the reference-compatible host entry point (sk_engine_calculate_radiance with host buffers)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import scenarios as sc
from tests.gpu_harness import build_engine, run_product

s = sc.limb_c2()
eng = build_engine(s)
run_product(s, engine=eng)
t = time.perf_counter()
for _ in range(3):
    run_product(s, engine=eng)
dt = (time.perf_counter() - t) / 3
print(f"host-buffer path: {dt*1e3:.1f} ms per 10000-wavelength call -> {10000/dt:.0f} wavelengths/s (includes H2D of 74 MB, D2H of 208 MB, layout mapping on the host)")
