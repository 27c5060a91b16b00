"""Where one host-buffer call (Engine.calculate_radiance, C2 size) spends its time: the phases of sk_engine_calculate_radiance
(SASKTRAN2_AMD_HOST_TIMING=1, stderr) and the Python mirror's own share."""
import os
import sys
import time

os.environ["SASKTRAN2_AMD_HOST_TIMING"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (before the library: two HIP runtimes in one process, torch's must be the one that initialises)

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

# the same solve with inputs and outputs resident in HBM: one LONE step (queue empty before and after) against back-to-back steps,
# and the host time the enqueue of one step takes
from sasktran2_amd.device import DeviceSolver  # noqa: E402

dev = torch.device("cuda", 0)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
ext, ssa = to(s["extinction"].T), to(s["ssa"].T)
leg, alb = to(np.transpose(s["leg"], (2, 1, 0))), to(s["albedo"])
solver = DeviceSolver(engine)
out = solver.allocate_outputs(NW, derivatives=True, device=dev)
solver.solve(ext, ssa, leg, alb, out)
torch.cuda.synchronize()
for _ in range(3):
    t = time.perf_counter()
    solver.solve(ext, ssa, leg, alb, out)
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"lone device-resident step {1e3 * (time.perf_counter() - t):.1f} ms (enqueue returned after {1e3 * t_enq:.1f} ms)", file=sys.stderr, flush=True)
    time.sleep(0.02)
t = time.perf_counter()
for _ in range(5):
    solver.solve(ext, ssa, leg, alb, out)
torch.cuda.synchronize()
print(f"back-to-back device-resident steps {1e3 * (time.perf_counter() - t) / 5:.1f} ms each", file=sys.stderr, flush=True)
