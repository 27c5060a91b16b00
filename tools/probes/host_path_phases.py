"""Where one host-buffer call (Engine.calculate_radiance, C2 size) spends its time: the phases of sk_engine_calculate_radiance
(SASKTRAN2_AMD_HOST_TIMING=1, stderr) and the Python mirror's own share."""
import os
import sys
import time

os.environ["SASKTRAN2_AMD_HOST_TIMING"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import sasktran2_amd as sk  # noqa: E402
from tests import scenarios as sc  # noqa: E402
from tests.gpu_harness import build_engine  # noqa: E402

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
s = sc.limb_c2(num_wavelengths=NW)
engine, config, geometry = build_engine(s)
atmosphere = sk.Atmosphere(geometry, config, numwavel=NW, calculate_derivatives=True)
atmosphere.storage.total_extinction[:] = s["extinction"]
atmosphere.storage.ssa[:] = s["ssa"]
atmosphere.leg_coeff.a1[:] = s["leg"]
atmosphere.surface.albedo[:] = s["albedo"]
engine.calculate_radiance(atmosphere, as_dataset=False)
for _ in range(3):
    t = time.perf_counter()
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
    print(f"whole call {1e3 * (time.perf_counter() - t):.1f} ms", file=sys.stderr, flush=True)
    del res
