"""Two-stream source on the limb geometry of BASELINE configs[1] (25 tangent heights, 50 layers, three solar-zenith columns, 10 000 wavelengths):
device-resident rate with and without weighting functions, per-kernel times."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
torch.cuda.init()
from sasktran2_amd.device import TsDeviceSolver
from tests import scenarios as sc
from tests.gpu_harness import build_ts_engine

NW = int(os.environ.get("NW", "10000"))
c2 = sc.limb_c2(num_wavelengths=NW, nlyr=50, nstr=16)
s = dict(c2, leg=c2["leg"][:2].copy(), config=dict(nstr=2, num_sza=3))
engine, config, geometry = build_ts_engine(s)
solver = TsDeviceSolver(engine)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ext, ssa = to(s["extinction"].T), to(s["ssa"].T)
leg, alb = to(np.transpose(s["leg"], (2, 1, 0))), to(s["albedo"])
for derivs in (False, True):
    out = solver.allocate_outputs(NW, derivatives=derivs)
    solver.solve(ext, ssa, leg, alb, out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        solver.solve(ext, ssa, leg, alb, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    solver.set_profiling(True)
    solver.solve(ext, ssa, leg, alb, out)
    torch.cuda.synchronize()
    kt = solver.kernel_times()
    solver.set_profiling(False)
    print(f"derivatives={derivs}: {dt*1e3:.2f} ms per {NW} wavelengths x 25 lines of sight -> {NW/dt/1e3:.0f}k wavelengths/s; kernels {kt}; launch {solver.launch_shape()}")
