"""Where a front-end step spends its host time (cProfile of Atmosphere.internal_object at the C2 size)."""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import sasktran2_amd as sk  # noqa: E402
from bench_frontend import case  # noqa: E402
from tests import scenarios as sc  # noqa: E402

c = case(10_000)
cfg = sk.Config()
cfg.num_streams = 16
cfg.num_singlescatter_moments = 16
geo = sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.Spherical)
atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, thermal=True)
for _ in range(3):
    atmo.internal_object()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    atmo.internal_object()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
