"""Observed errors of the finite-difference criteria of tests/test_gpu_brdf.py (their asserted tolerance is 1e-6 of the largest entry)."""
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
import sasktran2_amd as sk
from tests import scenarios as sc
from tests.gpu_harness import build_engine, run_product
import tests.test_gpu_brdf as T

for kind, args0, steps in (("modis", [0.1, 0.1, 0.1], [1e-3] * 3), ("kokhanovsky", [0.04], [2e-5])):
    s = T._scene(1, 4)
    nw = len(s["albedo"])
    args = T._args(args0, nw)
    eng = build_engine(s)
    s["brdf"] = (kind, args)
    base = run_product(s, derivatives=True, engine=eng)
    for a, h in enumerate(steps):
        ap, am = args.copy(), args.copy()
        ap[:, a] += h
        am[:, a] -= h
        fd = (run_product({**s, "brdf": (kind, ap)}, derivatives=False, engine=eng)["radiance"] - run_product({**s, "brdf": (kind, am)}, derivatives=False, engine=eng)["radiance"]) / (2 * h)
        print(f"raw {kind} arg {a}: {np.max(np.abs(fd - base['d_brdf'][..., a])) / np.abs(fd).max():.2e}")
surface = sk.constituent.MODIS(isotropic=0.1, volumetric=0.1, geometric=0.1)
engine, atmo = T._constituent_scene(sk, surface)
base = engine.calculate_radiance(atmo, as_dataset=False)
for part in ("isotropic", "volumetric", "geometric"):
    vals = {}
    for sign in (+1, -1):
        setattr(surface, part, getattr(surface, part) + sign * 0.001)
        vals[sign] = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
        setattr(surface, part, getattr(surface, part) - sign * 0.001)
    fd = (vals[+1] - vals[-1]) / 0.002
    print(f"constituent modis {part}: {np.max(np.abs(base[f'wf_surface_{part}'][0] - fd)) / np.abs(fd).max():.2e}")
ice = lambda wl: 1.31 - 1j * 2e-9 * (np.asarray(wl) / 500.0) ** 2   # noqa: E731
surface = sk.constituent.SnowKokhanovsky(L=3.6e6, M=5.5e-8, refractive_index_fn=ice)
engine, atmo = T._constituent_scene(sk, surface)
base = engine.calculate_radiance(atmo, as_dataset=False)
for part, h in (("L", 3.6e3), ("M", 5.5e-11)):
    vals = {}
    for sign in (+1, -1):
        setattr(surface, part, getattr(surface, part) + sign * h)
        vals[sign] = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
        setattr(surface, part, getattr(surface, part) - sign * h)
    fd = (vals[+1] - vals[-1]) / (2 * h)
    print(f"constituent kokhanovsky {part}: {np.max(np.abs(base[f'wf_surface_{part}'][0] - fd)) / np.abs(fd).max():.2e}")
