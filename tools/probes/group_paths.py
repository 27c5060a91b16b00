"""Scattering-group weighting functions of a 16-stream limb scene from the matrix-core kernels and from the run-time-n kernels
(SASKTRAN2_AMD_GROUPS=vector): largest difference relative to the largest weighting function.  usage: python tools/probes/group_paths.py [nw]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_parity import _group_solve   # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for mode in ("matrix", "vector"):
    os.environ["SASKTRAN2_AMD_GROUPS"] = mode
    res[mode] = _group_solve(nw=nw, nstr=16, ngroups=2)
for k in res["matrix"]:
    a, b = res["matrix"][k], res["vector"][k]
    print(f"{k:24s} max |matrix - vector| / max |vector| = {np.max(np.abs(a - b)) / max(np.abs(b).max(), 1e-300):.3e}   (max {np.abs(b).max():.3e})")
