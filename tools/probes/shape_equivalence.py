"""Which launch-shape knob changes bits?  (results are expected to be identical for every setting -- except the cooperative Jacobi of
SASKTRAN2_AMD_KJ=0, a different eigen-solver that agrees to rounding).  One child process per setting: several knobs are read once per process."""
import os, subprocess, sys, tempfile
import numpy as np

CASES = [{"SASKTRAN2_AMD_SPLIT": 1}, {"SASKTRAN2_AMD_SPLIT": 2}, {"SASKTRAN2_AMD_SPLIT": 4}, {"SASKTRAN2_AMD_SPLIT": 8},
         {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_CHUNK": 16}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_CHUNK": 16, "SASKTRAN2_AMD_STREAMS": 3},
         # round 3: orders in flight (also moves the live-flag kernel between streams), no look-ahead, unfused adjoint, cooperative Jacobi
         {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_DEPTH": 2}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_DEPTH": 3}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_DEPTH": 4},
         {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_LOOKAHEAD": 0}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_FUSE": 0}, {"SASKTRAN2_AMD_SPLIT": 1, "SASKTRAN2_AMD_KJ": 0}]
KEYS = ("radiance", "d_extinction", "d_ssa", "d_albedo")

if len(sys.argv) > 1:   # child: solve and save
    sys.path.insert(0, os.getcwd())
    from tests import scenarios as sc
    from tests.gpu_harness import run_product
    o = run_product(sc.limb_c2(num_wavelengths=64))
    np.savez(sys.argv[1], **{k: o[k] for k in KEYS})
    sys.exit(0)

base = None
with tempfile.TemporaryDirectory() as tmp:
    for i, env in enumerate(CASES):
        out = os.path.join(tmp, f"c{i}.npz")
        e = {k: v for k, v in os.environ.items() if not k.startswith("SASKTRAN2_AMD_")}
        e.update({k: str(v) for k, v in env.items()})
        subprocess.run([sys.executable, __file__, out], env=e, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        o = dict(np.load(out))
        if base is None:
            base = o
            continue
        print(env, {k: float(np.max(np.abs(o[k] - base[k]) / (np.abs(base[k]) + 1e-300))) for k in KEYS}, flush=True)
