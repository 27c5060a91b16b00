"""Cost of one scattering group at the C2 size: the "solve" phase of sk_engine_calculate_radiance (SASKTRAN2_AMD_HOST_TIMING=1, stderr) for the
plain request and for the request with one derivative mapping that carries d_legendre / scat_factor, and the per-kernel times of the second.
usage: python tools/probes/c2_group_cost.py [nwavel] [ngroups]"""
import ctypes
import os
import sys
import time

os.environ["SASKTRAN2_AMD_HOST_TIMING"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import sasktran2_amd as sk  # noqa: E402
from sasktran2_amd._lib_loader import lib  # noqa: E402
from tests import scenarios as sc  # noqa: E402
from tests.gpu_harness import build_engine  # noqa: E402

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = sc.limb_c2(num_wavelengths=NW)
engine, config, geometry = build_engine(s)
ngrid, nlos, nleg = s["extinction"].shape[0], len(s["rays"]), s["leg"].shape[0]
dp, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
l = np.arange(nleg)
keep = []


def hook(L, h_st, h_out, out):
    for gi in range(NG):
        name = f"wf_aerosol_{gi}".encode()
        m = vp()
        L.sk_atmosphere_storage_get_derivative_mapping(h_st, name, ctypes.byref(m))
        for which, val in (("d_extinction", 1.0), ("d_ssa", 0.1)):
            ptr = dp()
            getattr(L, f"sk_deriv_mapping_get_{which}")(m, ctypes.byref(ptr))
            np.ctypeslib.as_array(ptr, shape=(ngrid * NW,))[:] = val
        f = L.sk_deriv_mapping_get_scat_factor
        f.argtypes = [vp, ctypes.POINTER(dp)]
        ptr = dp()
        f(m, ctypes.byref(ptr))
        np.ctypeslib.as_array(ptr, shape=(ngrid * NW,))[:] = 0.5
        f = L.sk_deriv_mapping_get_d_legendre
        f.argtypes = [vp, ctypes.POINTER(dp)]
        ptr = dp()
        f(m, ctypes.byref(ptr))
        # constituent - mixture coefficients: a Henyey-Greenstein aerosol against whatever the scene holds ([wavel][q][l] in memory)
        d = ((2 * l + 1) * (0.7 - 0.1 * gi) ** l)[:, None, None] - s["leg"]
        np.ctypeslib.as_array(ptr, shape=(nleg * ngrid * NW,))[:] = np.asfortranarray(d).ravel(order="F")
        out[name.decode()] = sk._host.empty((ngrid, NW, nlos, 1))
        L.sk_output_assign_derivative_memory(h_out, name, out[name.decode()].ctypes.data_as(dp), NW * nlos, 1, ngrid)


atmosphere = sk.Atmosphere(geometry, config, numwavel=NW, calculate_derivatives=True)
atmosphere.storage.total_extinction[:] = s["extinction"]
atmosphere.storage.ssa[:] = s["ssa"]
atmosphere.leg_coeff.a1[:] = s["leg"]
atmosphere.surface.albedo[:] = s["albedo"]
for label, h in (("plain", None), (f"{NG} group(s)", hook), ("plain", None), (f"{NG} group(s)", hook)):
    print(f"---- {label}", file=sys.stderr, flush=True)
    t = time.perf_counter()
    res = engine.calculate_radiance(atmosphere, as_dataset=False, _mapping_hook=h)
    print(f"whole call {1e3 * (time.perf_counter() - t):.1f} ms", file=sys.stderr, flush=True)
    del res
L = lib()
L.sk_amd_engine_set_profiling(engine._h, 1)
engine.calculate_radiance(atmosphere, as_dataset=False, _mapping_hook=hook)
buf = ctypes.create_string_buffer(8192)
L.sk_amd_engine_kernel_names(engine._h, buf, 8192)
for name in buf.value.decode().split(","):
    ms, n = ctypes.c_double(), ctypes.c_long()
    L.sk_amd_engine_kernel_time(engine._h, name.encode(), ctypes.byref(ms), ctypes.byref(n))
    print(f"{name:28s} {ms.value:9.3f} ms in {n.value} launches (serialised)", file=sys.stderr)
