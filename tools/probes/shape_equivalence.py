"""Which launch-shape knob changes bits?  (results are expected to be identical for every setting)"""
import os, sys, copy
import numpy as np
sys.path.insert(0, os.getcwd())
from tests import scenarios as sc
from tests.gpu_harness import run_product

s = sc.limb_c2(num_wavelengths=64)
def run(**env):
    for k in ("SASKTRAN2_AMD_SPLIT", "SASKTRAN2_AMD_CHUNK", "SASKTRAN2_AMD_STREAMS"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    return run_product(s)
base = run(SASKTRAN2_AMD_SPLIT=1)
for env in ({"SASKTRAN2_AMD_SPLIT": 2}, {"SASKTRAN2_AMD_SPLIT": 4}, {"SASKTRAN2_AMD_SPLIT": 8}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_CHUNK": 16},
            {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_CHUNK": 16, "SASKTRAN2_AMD_STREAMS": 3}):
    o = run(**env)
    print(env, {k: float(np.max(np.abs(o[k] - base[k]) / (np.abs(base[k]) + 1e-300))) for k in ("radiance", "d_extinction", "d_ssa", "d_albedo")})
