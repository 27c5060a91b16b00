"""PCIe-inclusive rate of the reference-compatible host entry point for the two-stream benchmark (sk_engine_calculate_radiance with host
buffers: 220 000 wavelengths, 20 layers, 16 stored Legendre moments, weighting functions)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import sasktran2_amd as sk
from bench_ts import scenario
from tests.gpu_harness import build_ts_engine

NW = 220_000
s = scenario(NW)
engine, config, geometry = build_ts_engine(s)
atmosphere = sk.Atmosphere(geometry, config, numwavel=NW, calculate_derivatives=True)
atmosphere.storage.total_extinction[:] = s["extinction"]
atmosphere.storage.ssa[:] = s["ssa"]
atmosphere.leg_coeff.a1[:] = s["leg"]
atmosphere.surface.albedo[:] = s["albedo"]
engine.calculate_radiance(atmosphere, as_dataset=False)
t = time.perf_counter()
for _ in range(3):
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
dt = (time.perf_counter() - t) / 3
mb_in = (s["extinction"].nbytes * 2 + s["leg"].nbytes) / 1e6
mb_out = sum(res[k].nbytes for k in ("radiance", "wf_extinction", "wf_ssa", "wf_albedo")) / 1e6
print(f"host-buffer path: {dt*1e3:.1f} ms per {NW}-wavelength call -> {NW/dt/1e6:.2f} M wavelengths/s (includes H2D of {mb_in:.0f} MB, D2H of {mb_out:.0f} MB, "
      f"the Python mirror's copies of the storage and the user derivative mappings)")
