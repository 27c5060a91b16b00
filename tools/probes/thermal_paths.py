"""Thermal source inside the discrete-ordinates equations (plane parallel): matrix-core kernels against the run-time-n kernels
(SASKTRAN2_AMD_RUNTIME_N=1), output by output, for sun only / emission only / both.  usage: python tools/probes/thermal_paths.py [nstr]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import sasktran2_amd as sk  # noqa: E402

nstr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ng, nw = 11, 3
alt = np.linspace(0.0, 20000.0, ng)
config = sk.Config()
config.num_streams = nstr
config.num_singlescatter_moments = 16
config.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
config.single_scatter_source = sk.SingleScatterSource.DiscreteOrdinates
if os.environ.get("PROBE_NO_EMISSION_SOURCE") != "1":
    config.emission_source = sk.EmissionSource.DiscreteOrdinates
geometry = sk.Geometry1D(0.6, 0.0, 6372000.0, alt, sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.PlaneParallel)
viewing = sk.ViewingGeometry()
viewing.add_ray(sk.GroundViewingSolar(0.6, 0.4, 0.9, 200000.0))
viewing.add_ray(sk.GroundViewingSolar(0.6, 1.7, 0.5, 200000.0))
engine = sk.Engine(config, geometry, viewing)
base = dict(ext=(6e-5 * np.exp(-alt / 7000.0))[:, None] * np.array([0.6, 1.0, 2.5])[None, :], ssa=np.linspace(0.35, 0.8, ng)[:, None] * np.ones((1, nw)),
            em=(3.0 * np.exp(-alt / 9000.0) + 0.4)[:, None] * np.array([1.0, 0.7, 1.3])[None, :], alb=np.array([0.1, 0.25, 0.4]), sem=np.array([2.0, 1.5, 2.6]))
leg = ((2 * np.arange(16) + 1) * 0.55 ** np.arange(16))[:, None, None] * np.ones((1, ng, nw))


def solve(sun, emis, surf):
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = base["ext"]
    atmosphere.storage.ssa[:] = base["ssa"]
    atmosphere.leg_coeff.a1[:] = leg
    atmosphere.storage.solar_irradiance[:] = sun
    atmosphere.storage.emission_source[:] = emis * base["em"]
    atmosphere.surface.emission[:] = surf * base["sem"]
    atmosphere.surface.albedo[:] = base["alb"]
    return engine.calculate_radiance(atmosphere, as_dataset=False)


cases = (("sun only", (1.0, 0.0, 0.0)), ("emission only", (0.0, 1.0, 0.0)), ("surface emission only", (0.0, 0.0, 1.0)), ("all", (1.0, 1.0, 1.0)))
if os.environ.get("PROBE_NO_EMISSION_SOURCE") == "1":
    cases = cases[:1]
for label, args in cases:
    os.environ.pop("SASKTRAN2_AMD_RUNTIME_N", None)
    a = solve(*args)
    os.environ["SASKTRAN2_AMD_RUNTIME_N"] = "1"
    b = solve(*args)
    print(label)
    for k in ("radiance", "wf_extinction", "wf_ssa", "wf_albedo"):
        d = np.abs(a[k] - b[k])
        print(f"   {k:16s} max diff / max = {d.max() / max(np.abs(a[k]).max(), 1e-300):.2e}   worst altitude index {np.unravel_index(d.argmax(), d.shape)[0] if d.ndim == 4 else '-'}")
