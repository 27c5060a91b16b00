"""What the device front end buys at the boundary: one `Engine.calculate_radiance` call of the two-stream source (21 levels, one line of
sight, weighting functions) through the raw interface -- [altitude x wavelength] arrays prepared on the host and uploaded -- and through the
constituent interface, where they are assembled in HBM.  Usage: python tools/probes/frontend_end_to_end.py [wavelengths]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sasktran2_amd as sk  # noqa: E402
from tests import scenarios as sc  # noqa: E402

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 220_000
c = sc.frontend_case(nwavel=NW, ngrid=21)
cfg = sk.Config()
cfg.num_streams = 2
cfg.num_singlescatter_moments = 16
cfg.multiple_scatter_source = sk.MultipleScatterSource.TwoStream
cfg.single_scatter_source = sk.SingleScatterSource.NoSource
geo = sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.PlaneParallel)
vg = sk.ViewingGeometry()
vg.add_ray(sk.GroundViewingSolar(0.6, 1.0, 0.8, 200000.0))
engine = sk.Engine(cfg, geo, vg)

atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, pressure_derivative=False, temperature_derivative=False)
atmo.surface.albedo[:] = 0.3
for _ in range(2):
    got = engine.calculate_radiance(atmo, as_dataset=False)
t = time.perf_counter()
for _ in range(3):
    got = engine.calculate_radiance(atmo, as_dataset=False)
dt_fe = (time.perf_counter() - t) / 3

st = atmo.storage
raw = sk.Atmosphere(geo, cfg, numwavel=NW)
raw.storage.total_extinction[:] = st.total_extinction
raw.storage.ssa[:] = st.ssa
raw.leg_coeff.a1[:] = st.leg_coeff
raw.surface.albedo[:] = 0.3
for _ in range(2):
    ref = engine.calculate_radiance(raw, as_dataset=False)
t = time.perf_counter()
for _ in range(3):
    ref = engine.calculate_radiance(raw, as_dataset=False)
dt_raw = (time.perf_counter() - t) / 3
print(f"{NW} wavelengths x 21 levels, two-stream with weighting functions: raw interface (host arrays, wf_extinction / wf_ssa) {dt_raw * 1e3:.1f} ms, "
      f"constituent interface (assembled in HBM, {len(atmo.frontend.mappings)} mappings: {sorted(atmo.frontend.mappings)}) {dt_fe * 1e3:.1f} ms; "
      f"radiance max rel diff {float(np.max(np.abs(got['radiance'] / ref['radiance'] - 1))):.2e}")
