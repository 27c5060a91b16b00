"""One host-buffer call at the C2 size under different caller patterns: results dropped before the next call / kept alive during it."""
import os
import sys
import time

os.environ["SASKTRAN2_AMD_HOST_TIMING"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import sasktran2_amd as sk  # noqa: E402
from tests import scenarios as sc  # noqa: E402
from tests.gpu_harness import build_engine  # noqa: E402

NW = 10_000
s = sc.limb_c2(num_wavelengths=NW)
engine, config, geometry = build_engine(s)
atmosphere = sk.Atmosphere(geometry, config, numwavel=NW, calculate_derivatives=True)
atmosphere.storage.total_extinction[:] = s["extinction"]
atmosphere.storage.ssa[:] = s["ssa"]
atmosphere.leg_coeff.a1[:] = s["leg"]
atmosphere.surface.albedo[:] = s["albedo"]
engine.calculate_radiance(atmosphere, as_dataset=False)
pattern = sys.argv[1]
res = None
for i in range(4):
    if pattern == "drop":
        res = None
    t = time.perf_counter()
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
    print(f"[{pattern}] whole call {1e3 * (time.perf_counter() - t):.1f} ms", file=sys.stderr, flush=True)
