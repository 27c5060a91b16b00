"""ctypes front end of tests/emulate_kernels.cpp: the HIP kernel bodies executed on the CPU (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_ROOT = _HERE.parent
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_LIB = None
# SASKTRAN2_AMD_EMU_SANITIZE=1: AddressSanitizer + UBSan builds of the emulation libraries (tools/sanitize_emulation.sh preloads the runtime)
_SAN = os.environ.get("SASKTRAN2_AMD_EMU_SANITIZE", "") == "1"
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if _SAN else []
_SAN_TAG = "_asan" if _SAN else ""


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "_build" / f"libemu{_SAN_TAG}.so"
        srcs = [_HERE / "emulate_kernels.cpp", _ROOT / "sasktran2_amd/csrc/hip/disco_core.hpp",
                _ROOT / "sasktran2_amd/csrc/host/geo_tables.hpp", _ROOT / "sasktran2_amd/csrc/host/geometry.hpp"]
        if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
            so.parent.mkdir(exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *_SAN_FLAGS, "-o", str(so), str(srcs[0])])
        L = ctypes.CDLL(str(so))
        L.emu_create.restype = ctypes.c_void_p
        L.emu_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp, ctypes.c_int]
        L.emu_set_delta_f.argtypes = [_dp]
        L.emu_set_thermal.argtypes = [_dp, _dp, _dp, _dp]
        L.emu_destroy.argtypes = [ctypes.c_void_p]
        L.emu_run.restype = ctypes.c_int
        L.emu_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_int, _dp,
                              _dp, _dp, _dp, _dp, _ip]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def run_emulated(s, derivatives=True):
    g = s["geometry"]
    alt = np.ascontiguousarray(g["altitudes"], dtype=np.float64)
    rays = s["rays"]
    kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
    params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
    h = lib().emu_create(g["cos_sza"], g["saa"], g["earth_radius"], _p(alt), len(alt), g["interp_method"], g["geotype"],
                         s["config"]["nstr"], s["config"]["num_sza"], len(rays), kinds.ctypes.data_as(_ip), _p(params),
                         int(s["config"].get("ss_in_do", True)))
    assert h
    try:
        k = np.asfortranarray(s["extinction"], dtype=np.float64)
        w = np.asfortranarray(s["ssa"], dtype=np.float64)
        lg = np.asfortranarray(s["leg"], dtype=np.float64)
        nw, nlos, ngrid = k.shape[1], len(rays), len(alt)
        alb = np.ascontiguousarray(s["albedo"], dtype=np.float64)
        rad, od = np.zeros((nw, nlos)), np.zeros((nw, nlos))
        dk, dw, da = np.zeros((nw, nlos, ngrid)), np.zeros((nw, nlos, ngrid)), np.zeros((nw, nlos))
        status = np.zeros(nw, dtype=np.int32)
        df = None if s.get("delta_f") is None else np.asfortranarray(s["delta_f"], dtype=np.float64)
        lib().emu_set_delta_f(_p(df))
        # thermal source inside the discrete-ordinates equations (plane-parallel scenes with an "emission" profile)
        em = sem = wem = wsem = None
        if s.get("emission") is not None:
            em = np.asfortranarray(s["emission"], dtype=np.float64)
            sem = None if s.get("surface_emission") is None else np.ascontiguousarray(s["surface_emission"], dtype=np.float64)
            wem, wsem = np.zeros((ngrid, nw, nlos)), np.zeros((nw, nlos))
        irr = None if (em is None or s.get("solar_irradiance") is None) else np.ascontiguousarray(s["solar_irradiance"], dtype=np.float64)
        lib().emu_set_thermal(_p(em), _p(sem), _p(wem), _p(wsem))
        try:
            rc = lib().emu_run(h, nw, lg.shape[0], _p(k), _p(w), _p(lg), _p(irr), _p(alb), int(derivatives), _p(rad), _p(dk),
                               _p(dw), _p(da), _p(od), status.ctypes.data_as(_ip))
        finally:
            lib().emu_set_thermal(None, None, None, None)
        assert rc == 0
        out = {"radiance": rad, "los_optical_depth": od, "d_extinction": dk, "d_ssa": dw, "d_albedo": da, "status": status}
        if wem is not None:
            out["d_emission"], out["d_surface_emission"] = np.transpose(wem, (1, 2, 0)), wsem
        return out
    finally:
        lib().emu_destroy(h)


# ---------------------------------------------------------------------------------------------- polarised path
_LIBV = None


def libv():
    global _LIBV
    if _LIBV is None:
        so = _HERE / "_build" / f"libemuv{_SAN_TAG}.so"
        srcs = [_HERE / "emulate_vector.cpp", _ROOT / "sasktran2_amd/csrc/hip/disco_vec_core.hpp",
                _ROOT / "sasktran2_amd/csrc/hip/disco_core.hpp", _ROOT / "sasktran2_amd/csrc/host/geo_tables.hpp",
                _ROOT / "sasktran2_amd/csrc/host/geometry.hpp"]
        if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
            so.parent.mkdir(exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *_SAN_FLAGS, "-o", str(so), str(srcs[0])])
        L = ctypes.CDLL(str(so))
        L.emuv_create.restype = ctypes.c_void_p
        L.emuv_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.emuv_destroy.argtypes = [ctypes.c_void_p]
        L.emuv_run.restype = ctypes.c_int
        L.emuv_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip,
                               ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _dp, _dp]
        _LIBV = L
    return _LIBV


def run_emulated_vector(s, derivatives=False, d_legendre=None, brdf=None):
    """d_legendre: optional [G, nleg (*4), ngrid, nwavel] directions of scattering derivative groups (deriv_mapping.h d_legendre)."""
    """Radiances [nwavel, nlos, nstokes] of the polarised task bodies (disco_vec_core.hpp) executed on the CPU."""
    g = s["geometry"]
    alt = np.ascontiguousarray(g["altitudes"], dtype=np.float64)
    rays = s["rays"]
    kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
    params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
    S = s["config"].get("nstokes", 1)
    L = libv()
    h = L.emuv_create(g["cos_sza"], g["saa"], g["earth_radius"], _p(alt), len(alt), g["interp_method"], g["geotype"],
                      s["config"]["nstr"], s["config"]["num_sza"], len(rays), kinds.ctypes.data_as(_ip), _p(params),
                      int(s["config"].get("ss_in_do", True)), S, int(s["config"].get("stokes_basis", 0)))
    assert h
    try:
        k = np.asfortranarray(s["extinction"], dtype=np.float64)
        w = np.asfortranarray(s["ssa"], dtype=np.float64)
        lg = np.asfortranarray(s["leg"], dtype=np.float64)
        nw, nlos = k.shape[1], len(rays)
        alb = np.ascontiguousarray(s["albedo"], dtype=np.float64)
        df = None if s.get("delta_f") is None else np.asfortranarray(s["delta_f"], dtype=np.float64)
        rad, od = np.zeros((nw, nlos, S)), np.zeros((nw, nlos))
        status = np.zeros(nw, dtype=np.int32)
        ngrid = len(alt)
        dk = dw = da = None
        if derivatives:
            dk, dw, da = np.zeros((nw, nlos, S, ngrid)), np.zeros((nw, nlos, S, ngrid)), np.zeros((nw, nlos, S))
        G, dl, dsc = 0, None, None
        if derivatives and d_legendre is not None:
            G = len(d_legendre)
            dl = np.ascontiguousarray(np.stack([np.asfortranarray(x).ravel(order="F") for x in d_legendre]))   # [G][w][q][leg]
            dsc = np.zeros((nw, nlos, S, G, ngrid))
        dbr = None
        if brdf is not None:   # (kind, args [nwavel, nargs]): non-Lambertian surface, plane-parallel / pseudo-spherical post-processing
            kind = {"kokhanovsky": 1, "modis": 2, "legacy_test_function": 3}[brdf[0]]
            bargs = np.ascontiguousarray(brdf[1], dtype=np.float64).reshape(nw, -1)
            dbr = np.zeros((nw, nlos, S, bargs.shape[1])) if derivatives else None
            L.emuv_set_brdf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp]
            assert L.emuv_set_brdf(h, kind, bargs.shape[1], _p(bargs), _p(dbr)) == 0
        rc = L.emuv_run(h, nw, lg.shape[0] // (4 if S == 3 else 1), _p(k), _p(w), _p(lg), None, _p(alb), _p(df), _p(rad), _p(od),
                        status.ctypes.data_as(_ip), int(derivatives), _p(dk), _p(dw), _p(da), G, _p(dl), _p(dsc))
        assert rc == 0, rc
        out = {"radiance": rad if S == 3 else rad[..., 0], "los_optical_depth": od, "status": status}
        if derivatives:
            sq = (lambda a: a) if S == 3 else (lambda a: a[:, :, 0])
            out.update({"d_extinction": sq(dk), "d_ssa": sq(dw), "d_albedo": sq(da)})
            if dbr is not None:
                out["d_brdf"] = dbr if S == 3 else dbr[:, :, 0]
            if dsc is not None:
                out["d_scat"] = dsc if S == 3 else dsc[:, :, 0]
        return out
    finally:
        L.emuv_destroy(h)


# ---- two-stream source (tests/emulate_twostream.cpp runs sasktran2_amd/csrc/hip/ts_core.hpp on the CPU)
_LIB_TS = None


def lib_ts():
    global _LIB_TS
    if _LIB_TS is None:
        so = _HERE / "_build" / f"libemuts{_SAN_TAG}.so"
        srcs = [_HERE / "emulate_twostream.cpp", _ROOT / "sasktran2_amd/csrc/hip/ts_core.hpp",
                _ROOT / "sasktran2_amd/csrc/host/ts_geometry.hpp", _ROOT / "sasktran2_amd/csrc/host/geometry.hpp"]
        if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
            so.parent.mkdir(exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *_SAN_FLAGS, "-o", str(so), str(srcs[0])])
        L = ctypes.CDLL(str(so))
        L.emts_create.restype = ctypes.c_void_p
        L.emts_create.argtypes = [ctypes.c_double] * 3 + [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp] + [ctypes.c_int] * 3
        L.emts_destroy.argtypes = [ctypes.c_void_p]
        L.emts_run.restype = ctypes.c_int
        L.emts_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [_dp] * 15 + [ctypes.c_int, ctypes.c_int]
        _LIB_TS = L
    return _LIB_TS


def run_emulated_twostream(s, derivatives=True, solar=True, thermal=False, dual_adjoint=False, allow_fused=True):
    """The two-stream kernels on the CPU.  Returns the oracle's output names and layouts: radiance [nw, nlos], d_extinction / d_ssa /
    d_b1 / d_emission [nw, nlos, ngrid], d_albedo / d_surface_emission [nw, nlos]."""
    g = s["geometry"]
    alt = np.ascontiguousarray(g["altitudes"], dtype=np.float64)
    rays = s["rays"]
    kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
    params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
    L = lib_ts()
    h = L.emts_create(g["cos_sza"], g["saa"], g["earth_radius"], _p(alt), len(alt), g["interp_method"], g["geotype"], len(rays),
                      kinds.ctypes.data_as(_ip), _p(params), int(s["config"]["num_sza"]), int(solar), int(thermal))
    assert h
    try:
        k = np.asfortranarray(s["extinction"], dtype=np.float64)
        w = np.asfortranarray(s["ssa"], dtype=np.float64)
        lg = np.asfortranarray(s["leg"], dtype=np.float64)
        nw, nlos, ngrid = k.shape[1], len(rays), len(alt)
        alb = np.ascontiguousarray(s["albedo"], dtype=np.float64)
        opt = lambda key, f: None if s.get(key) is None else f(s[key], dtype=np.float64)
        df, em = opt("delta_f", np.asfortranarray), opt("emission", np.asfortranarray)
        sem, irr = opt("surface_emission", np.ascontiguousarray), opt("solar_irradiance", np.ascontiguousarray)
        rad = np.zeros((nw, nlos))
        wf = [None] * 6
        if derivatives:
            wf = [np.zeros((ngrid, nw, nlos)) for _ in range(4)] + [np.zeros((nw, nlos)) for _ in range(2)]
        rc = L.emts_run(h, nw, lg.shape[0], _p(k), _p(w), _p(lg), _p(df), _p(em), _p(sem), _p(irr), _p(alb), _p(rad), *[_p(a) for a in wf], int(dual_adjoint), int(allow_fused))
        assert rc == 0, rc
        out = {"radiance": rad}
        if derivatives:
            for name, a in zip(("d_extinction", "d_ssa", "d_b1", "d_emission"), wf[:4]):
                out[name] = np.ascontiguousarray(a.transpose(1, 2, 0))
            out["d_albedo"], out["d_surface_emission"] = wf[4], wf[5]
        return out
    finally:
        L.emts_destroy(h)
