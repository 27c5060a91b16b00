"""Throughput of the polarised (NSTOKES = 3) path on one GPU: limb geometry of BASELINE configs[3] at a few sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from tests import scenarios as sc
from tests.gpu_harness import build_engine, run_product

import os
cases = ((16, 50, 64), (16, 50, 1024), (16, 50, 4096), (32, 60, 64), (32, 60, 512))
if os.environ.get('VEC_CASES'):
    cases = tuple(tuple(int(x) for x in c.split(',')) for c in os.environ['VEC_CASES'].split(';'))
for nstr, nlyr, nw in cases:
    s = sc.to_vector(sc.limb_c2(num_wavelengths=nw, nlyr=nlyr, nstr=nstr), rayleigh_fraction=0.6)
    eng = build_engine(s)
    DER = bool(int(os.environ.get('VEC_DERIVS', '0')))
    run_product(s, derivatives=DER, engine=eng)
    t = time.time()
    out = run_product(s, derivatives=DER, engine=eng)
    dt = time.time() - t
    from sasktran2_amd.device import DeviceSolver
    ds = DeviceSolver(eng[0])
    ds.set_profiling(True)
    run_product(s, derivatives=DER, engine=eng)
    kt = ds.kernel_times()
    ds.set_profiling(False)
    print("   kernels (ms):", {k: round(v[0], 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])}, flush=True)
    print(f"nstr {nstr} nlyr {nlyr} nw {nw}: {dt*1e3:.1f} ms/call, {nw/dt:.1f} wavelengths/s, I range {out['radiance'][...,0].min():.3e}..{out['radiance'][...,0].max():.3e}", flush=True)
