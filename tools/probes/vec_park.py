"""Eigen stage of the polarised path in its parked layout against the 8-matrix layout: kernel times and a digest of the outputs of one solve
(saved per process: the layout knob SASKTRAN2_AMD_VEC_PARK is read once).  usage: vec_park.py TAG [nstr,nlyr,nw;...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from tests import scenarios as sc
from tests.gpu_harness import build_engine, run_product
from sasktran2_amd.device import DeviceSolver

tag = sys.argv[1]
cases = tuple(tuple(int(x) for x in c.split(',')) for c in (sys.argv[2] if len(sys.argv) > 2 else "32,60,512").split(';'))
out_all = {}
for nstr, nlyr, nw in cases:
    s = sc.to_vector(sc.limb_c2(num_wavelengths=nw, nlyr=nlyr, nstr=nstr), rayleigh_fraction=0.6)
    eng = build_engine(s)
    run_product(s, derivatives=True, engine=eng)
    t = time.time()
    out = run_product(s, derivatives=True, engine=eng)
    dt = time.time() - t
    ds = DeviceSolver(eng[0])
    ds.set_profiling(True)
    run_product(s, derivatives=True, engine=eng)
    kt = ds.kernel_times()
    ds.set_profiling(False)
    print(f"[{tag}] nstr {nstr} nlyr {nlyr} nw {nw}: {dt*1e3:.1f} ms/call, {nw/dt:.1f} wavelengths/s", flush=True)
    print("   kernels (ms):", {k: round(v[0], 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}, flush=True)
    for k in ("radiance", "d_extinction", "d_ssa", "d_albedo"):
        a = np.ascontiguousarray(out[k])
        out_all[f"{nstr}_{k}"] = np.array([int.from_bytes(__import__("hashlib").sha256(a.tobytes()).digest()[:7], "big"), float(np.abs(a).max())])
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/park_{tag}.npz", **out_all)
