"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY, see disco_oracle.hpp)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

TANGENT_ALTITUDE_SOLAR = 0
GROUND_VIEWING_SOLAR = 1

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> Path:
    # SASKTRAN2_ORACLE_NATIVE=1 (bench.py's cpu_baseline leg): the oracle compiled with -O3 -march=native ON the machine that times it
    native = os.environ.get("SASKTRAN2_ORACLE_NATIVE", "") not in ("", "0")
    if native:
        try:
            subprocess.check_call(["make", "-C", str(_HERE), "-s", "native"])
            return _HERE / "liboracle_native.so"
        except (subprocess.CalledProcessError, OSError):
            pass   # (no compiler on this host: the portable build below)
    so = _HERE / "liboracle.so"
    srcs = [_HERE / "oracle_capi.cpp", _HERE / "disco_oracle.hpp", _HERE / "linalg.hpp",
            _HERE.parent / "sasktran2_amd" / "csrc" / "host" / "geometry.hpp"]
    if force or not so.exists() or any(s.exists() and s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_HERE), "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(str(so))
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_run.restype = ctypes.c_int
        L.orc_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_int,
                              ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _ip]
        L.orc_set_delta_f.argtypes = [ctypes.c_void_p, _dp]
        L.orc_set_emission.argtypes = [ctypes.c_void_p, _dp, _dp]
        L.orc_set_forced_azimuth.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_set_flux_observers.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _dp]
        L.orc_num_sza.argtypes = [ctypes.c_void_p]
        L.orc_chapman.argtypes = [ctypes.c_void_p, _dp]
        L.orc_streams.argtypes = [ctypes.c_void_p, _dp, _dp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class OracleEngine:
    """rays: list of (kind, cos_sza, relative_azimuth, observer_altitude, tangent_altitude|cos_viewing_zenith)."""

    def __init__(self, cos_sza, saa, earth_radius, altitudes, interp_method, geotype, nstr, num_sza, rays,
                 ss_in_do=True, nstokes=1, stokes_basis=0):
        self.alt = np.ascontiguousarray(altitudes, dtype=np.float64)
        self.nlos = len(rays)
        self.ngrid = len(self.alt)
        self.nstr = nstr
        self.nstokes = nstokes
        kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
        params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
        self._h = lib().orc_create(cos_sza, saa, earth_radius, _p(self.alt), self.ngrid, interp_method, geotype, nstr,
                                   num_sza, self.nlos, kinds.ctypes.data_as(_ip), _p(params), int(ss_in_do), int(nstokes), int(stokes_basis))
        if not self._h:
            raise RuntimeError("oracle engine construction failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def chapman(self):
        nsza = lib().orc_num_sza(self._h)
        out = np.zeros((nsza, self.ngrid - 1, self.ngrid - 1))
        lib().orc_chapman(self._h, _p(out))
        return out

    def ray_tables(self, ray):
        """diffuse-table stencils of one line of sight as the shared geometry precompute holds them: {"ground_is_hit", "ground_saa",
        "ground": [(angle, sza, w)], "layers": [{"distance", "azi", "table": [(sza, alt, angle, w)]}] (far end -> observer)}"""
        L = lib()
        L.orc_ray_tables.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.c_int]
        n = L.orc_ray_tables(self._h, ray, None, 0)
        buf = np.zeros(n)
        L.orc_ray_tables(self._h, ray, _p(buf), n)
        nl, hit, gsaa, ng = int(buf[0]), bool(buf[1]), buf[2], int(buf[3])
        pos = 4
        ground = [(int(buf[pos + 3 * i]), int(buf[pos + 3 * i + 1]), buf[pos + 3 * i + 2]) for i in range(ng)]
        pos += 3 * ng
        layers = []
        for _ in range(nl):
            dist, azi, k = buf[pos], buf[pos + 1], int(buf[pos + 2])
            pos += 3
            table = [(int(buf[pos + 4 * i]), int(buf[pos + 4 * i + 1]), int(buf[pos + 4 * i + 2]), buf[pos + 4 * i + 3]) for i in range(k)]
            pos += 4 * k
            layers.append({"distance": dist, "azi": azi, "table": table})
        return {"ground_is_hit": hit, "ground_saa": gsaa, "ground": ground, "layers": layers}

    def streams(self):
        mu = np.zeros(self.nstr)
        wt = np.zeros(self.nstr)
        lib().orc_streams(self._h, _p(mu), _p(wt))
        return mu, wt

    def run(self, extinction, ssa, leg, albedo, solar_irradiance=None, derivatives=False, nthreads=1, delta_f=None, emission=None,
            surface_emission=None, forced_azimuth=0, flux_altitudes=None, brdf=None):
        """brdf: None (Lambertian `albedo`) or (kind, args [nwavel, nargs]) with kind "kokhanovsky" (1 argument) / "modis" (isotropic,
        volumetric, geometric); plane-parallel / pseudo-spherical post-processing only; adds "d_brdf" [nwavel, nlos, (3,) nargs].
        extinction, ssa: [ngrid, nwavel]; leg: [nleg, ngrid, nwavel] (nstokes = 3: [4 * nleg, ngrid, nwavel] with
        a1, a2, a3, b1 per order); albedo: [nwavel] (Fortran-style axes as in the reference storage: location fastest).
        Outputs carry a trailing Stokes axis for nstokes = 3 (radiance [nwavel, nlos, 3], d_* [nwavel, nlos, 3, ngrid]).
        Plane-parallel post-processing only: `forced_azimuth` > 0 solves exactly that many azimuth orders (Config.num_forced_azimuth);
        `flux_altitudes` adds "upwelling_flux", "downwelling_flux", "actinic_flux" [nwavel, observer] at those altitudes."""
        k = np.asfortranarray(extinction, dtype=np.float64)
        w = np.asfortranarray(ssa, dtype=np.float64)
        lg = np.asfortranarray(leg, dtype=np.float64)
        nwavel = k.shape[1]
        df = None if delta_f is None else np.asfortranarray(delta_f, dtype=np.float64)
        lib().orc_set_delta_f(self._h, _p(df))
        em = None if emission is None else np.asfortranarray(emission, dtype=np.float64)
        sem = None if surface_emission is None else np.ascontiguousarray(surface_emission, dtype=np.float64)
        lib().orc_set_emission(self._h, _p(em), _p(sem))   # EmissionSource::discrete_ordinates (plane parallel, radiances only)
        alb = np.ascontiguousarray(albedo, dtype=np.float64)
        irr = None if solar_irradiance is None else np.ascontiguousarray(solar_irradiance, dtype=np.float64)
        S = self.nstokes
        sshape = (nwavel, self.nlos) if S == 1 else (nwavel, self.nlos, S)
        rad = np.zeros(sshape)
        od = np.zeros((nwavel, self.nlos))
        nazi = np.zeros(nwavel, dtype=np.int32)
        dk = dw = da = None
        if derivatives:
            dk = np.zeros(sshape + (self.ngrid,))
            dw = np.zeros(sshape + (self.ngrid,))
            da = np.zeros(sshape)
        lib().orc_set_forced_azimuth(self._h, int(forced_azimuth))
        bargs = dbr = None
        if brdf is not None:
            kind = {"kokhanovsky": 1, "modis": 2, "legacy_test_function": 3}[brdf[0]]
            bargs = np.ascontiguousarray(brdf[1], dtype=np.float64).reshape(nwavel, -1)
            dbr = np.zeros(sshape + (bargs.shape[1],)) if derivatives else None
            L = lib()
            L.orc_set_brdf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp]
            if L.orc_set_brdf(self._h, kind, bargs.shape[1], _p(bargs), _p(dbr)) != 0:
                raise RuntimeError("oracle: bad BRDF arguments")
        fa = flux = None
        if flux_altitudes is not None and len(flux_altitudes) > 0:
            fa = np.ascontiguousarray(flux_altitudes, dtype=np.float64)
            flux = np.zeros((nwavel, len(fa), 3))
        if lib().orc_set_flux_observers(self._h, _p(fa), 0 if fa is None else len(fa), _p(flux)) != 0:
            raise RuntimeError("oracle: flux observers need plane-parallel post-processing")
        rc = lib().orc_run(self._h, nwavel, lg.shape[0] // (4 if S == 3 else 1), _p(k), _p(w), _p(lg), _p(irr), _p(alb), int(derivatives),
                           nthreads, _p(rad), _p(dk), _p(dw), _p(da), _p(od), nazi.ctypes.data_as(_ip))
        if brdf is not None:
            lib().orc_set_brdf(self._h, 0, 1, None, None)
        if rc != 0:
            raise RuntimeError(f"oracle run failed with code {rc}")
        lib().orc_set_flux_observers(self._h, None, 0, None)
        out = {"radiance": rad, "los_optical_depth": od, "num_azimuth": nazi}
        if dbr is not None:
            out["d_brdf"] = dbr
        if flux is not None:
            out.update({"upwelling_flux": flux[:, :, 0], "downwelling_flux": flux[:, :, 1], "actinic_flux": flux[:, :, 2]})
        if derivatives:
            out.update({"d_extinction": dk, "d_ssa": dw, "d_albedo": da})
        return out


class SoOracleEngine:
    """Exact single scatter + legacy successive orders (oracle/so_oracle.hpp).  rays as for OracleEngine."""

    def __init__(self, cos_sza, saa, earth_radius, altitudes, interp_method, geotype, rays, nstokes=1, stokes_basis=0,
                 num_ss_moments=16, num_ms_moments=16, num_incoming=110, num_outgoing=110, num_sza=1, num_iterations=50,
                 single_scatter=True, multiple_scatter=True, emission=0, occultation=False, fixed_point=None):
        """fixed_point: None (legacy passes) or dict(relative_tolerance, absolute_tolerance, damping, anderson_depth) for
        MultipleScatterSource::SuccessiveOrders (num_iterations is then the iteration limit)"""
        L = lib()
        L.orso_create.restype = ctypes.c_void_p
        L.orso_create.argtypes = [ctypes.c_double] * 3 + [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp] + [ctypes.c_int] * 12
        L.orso_set_emission.argtypes = [ctypes.c_void_p, _dp, _dp]
        L.orso_destroy.argtypes = [ctypes.c_void_p]
        L.orso_sizes.argtypes = [ctypes.c_void_p, _ip]
        L.orso_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        self.alt = np.ascontiguousarray(altitudes, dtype=np.float64)
        self.nlos, self.ngrid, self.nstokes = len(rays), len(self.alt), nstokes
        kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
        params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
        self._h = L.orso_create(cos_sza, saa, earth_radius, _p(self.alt), self.ngrid, interp_method, geotype, self.nlos,
                                kinds.ctypes.data_as(_ip), _p(params), int(nstokes), int(stokes_basis), int(num_ss_moments),
                                int(num_ms_moments), int(num_incoming), int(num_outgoing), int(num_sza), int(num_iterations),
                                int(single_scatter), int(multiple_scatter), int(emission), int(occultation))   # single_scatter: 0 none, 1 exact, 2 solar table
        if not self._h:
            raise RuntimeError("successive-orders oracle construction failed")
        if fixed_point is not None:
            L.orso_set_fixed_point.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int]
            if L.orso_set_fixed_point(self._h, 1, float(fixed_point.get("relative_tolerance", 1e-6)), float(fixed_point.get("absolute_tolerance", 1e-12)),
                                      float(fixed_point.get("damping", 1.0)), int(fixed_point.get("anderson_depth", 3))) != 0:
                raise ValueError("invalid fixed-point settings")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orso_destroy(self._h)
            self._h = None

    def sizes(self):
        out = np.zeros(4, dtype=np.int32)
        lib().orso_sizes(self._h, out.ctypes.data_as(_ip))
        return dict(points=int(out[0]), incoming_rays=int(out[1]), outgoing_directions=int(out[2]), accumulation_entries=int(out[3]))

    def run(self, extinction, ssa, leg, albedo, solar_irradiance=None, nthreads=1, derivatives=False, emission=None, surface_emission=None, brdf=None):
        """derivatives: native weighting functions of the exact single-scatter source, d_extinction / d_ssa [nwavel, nlos, S, ngrid],
        d_albedo [nwavel, nlos, S] (only without the diffuse source).  brdf = (kind, args [nwavel, nargs]): a non-Lambertian surface under the
        lines of sight's direct bounce (single scatter only); adds d_brdf [nwavel, nlos, S, nargs]."""
        k = np.asfortranarray(extinction, dtype=np.float64)
        w = np.asfortranarray(ssa, dtype=np.float64)
        lg = np.asfortranarray(leg, dtype=np.float64)
        nwavel, S = k.shape[1], self.nstokes
        alb = np.ascontiguousarray(albedo, dtype=np.float64)
        irr = None if solar_irradiance is None else np.ascontiguousarray(solar_irradiance, dtype=np.float64)
        rad = np.zeros((nwavel, self.nlos, S))
        od = np.zeros((nwavel, self.nlos))
        em = None if emission is None else np.asfortranarray(emission, dtype=np.float64)
        sem = None if surface_emission is None else np.ascontiguousarray(surface_emission, dtype=np.float64)
        lib().orso_set_emission(self._h, _p(em), _p(sem))
        dk = dw = da = de = ds = None
        if derivatives:
            dk, dw, da = np.zeros((nwavel, self.nlos, S, self.ngrid)), np.zeros((nwavel, self.nlos, S, self.ngrid)), np.zeros((nwavel, self.nlos, S))
            de, ds = np.zeros((nwavel, self.nlos, S, self.ngrid)), np.zeros((nwavel, self.nlos, S))
        los_alb = d_los_alb = None
        L = lib()
        L.orso_set_brdf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, _dp]
        if brdf is not None:
            kind = {"kokhanovsky": 1, "modis": 2, "legacy_test_function": 3}[brdf[0]]
            bargs = np.ascontiguousarray(brdf[1], dtype=np.float64).reshape(nwavel, -1)
            los_alb, d_los_alb = np.zeros((nwavel, self.nlos)), np.zeros((nwavel, self.nlos, bargs.shape[1]))
            if L.orso_set_brdf(self._h, kind, bargs.shape[1], _p(bargs), nwavel, _p(los_alb), _p(d_los_alb)) != 0:
                raise RuntimeError("successive-orders oracle: bad BRDF arguments")
        rc = L.orso_run(self._h, nwavel, lg.shape[0] // (4 if S == 3 else 1), _p(k), _p(w), _p(lg), _p(irr), _p(alb), nthreads, _p(rad), _p(od),
                        _p(dk), _p(dw), _p(da), _p(de), _p(ds))
        L.orso_set_brdf(self._h, 0, 1, None, 0, None, None)
        if rc != 0:
            raise RuntimeError(f"successive-orders oracle run failed with code {rc}")
        out = {"radiance": rad, "los_optical_depth": od}
        if derivatives:
            out.update({"d_extinction": dk, "d_ssa": dw, "d_albedo": da, "d_emission": de, "d_surface_emission": ds})
            if brdf is not None:   # the single scatter is linear in pi x brdf: d I / d argument = (d I / d albedo) x pi d brdf / d argument
                out["d_brdf"] = da[..., None] * d_los_alb[:, :, None, :]
        return out


class TsOracleEngine:
    """Two-stream source (oracle/ts_oracle.hpp): multiple_scatter_source = twostream (`solar`) and / or emission_source = twostream
    (`thermal`).  rays as for OracleEngine."""

    def __init__(self, cos_sza, saa, earth_radius, altitudes, interp_method, geotype, rays, num_sza=1, solar=True, thermal=False):
        L = lib()
        L.orts_create.restype = ctypes.c_void_p
        L.orts_create.argtypes = [ctypes.c_double] * 3 + [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _dp] + [ctypes.c_int] * 3
        L.orts_destroy.argtypes = [ctypes.c_void_p]
        L.orts_sizes.argtypes = [ctypes.c_void_p, _ip]
        L.orts_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [_dp] * 8 + [ctypes.c_int] + [_dp] * 7
        self.alt = np.ascontiguousarray(altitudes, dtype=np.float64)
        self.nlos, self.ngrid = len(rays), len(self.alt)
        kinds = np.ascontiguousarray([r[0] for r in rays], dtype=np.int32)
        params = np.ascontiguousarray([[r[1], r[2], r[3], r[4]] for r in rays], dtype=np.float64)
        self._h = L.orts_create(cos_sza, saa, earth_radius, _p(self.alt), self.ngrid, interp_method, geotype, self.nlos,
                                kinds.ctypes.data_as(_ip), _p(params), int(num_sza), int(solar), int(thermal))
        if not self._h:
            raise RuntimeError("two-stream oracle construction failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orts_destroy(self._h)
            self._h = None

    def sizes(self):
        out = np.zeros(3, dtype=np.int32)
        lib().orts_sizes(self._h, out.ctypes.data_as(_ip))
        return dict(columns=int(out[0]), segments=int(out[1]), od_entries=int(out[2]))

    def run(self, extinction, ssa, leg, albedo, solar_irradiance=None, derivatives=False, nthreads=1, delta_f=None, emission=None,
            surface_emission=None):
        """extinction, ssa, emission, delta_f: [ngrid, nwavel]; leg: [nleg, ngrid, nwavel]; albedo, surface_emission: [nwavel].
        Returns radiance [nwavel, nlos] and, with derivatives, d_extinction / d_ssa / d_b1 / d_emission [nwavel, nlos, ngrid] (b1: the
        first Legendre coefficient after the delta-M correction) and d_albedo / d_surface_emission [nwavel, nlos]."""
        k = np.asfortranarray(extinction, dtype=np.float64)
        w = np.asfortranarray(ssa, dtype=np.float64)
        lg = np.asfortranarray(leg, dtype=np.float64)
        nwavel = k.shape[1]
        df = None if delta_f is None else np.asfortranarray(delta_f, dtype=np.float64)
        em = None if emission is None else np.asfortranarray(emission, dtype=np.float64)
        sem = None if surface_emission is None else np.ascontiguousarray(surface_emission, dtype=np.float64)
        alb = np.ascontiguousarray(albedo, dtype=np.float64)
        irr = None if solar_irradiance is None else np.ascontiguousarray(solar_irradiance, dtype=np.float64)
        rad = np.zeros((nwavel, self.nlos))
        d = [None] * 6
        if derivatives:
            d = [np.zeros((nwavel, self.nlos, self.ngrid)) for _ in range(4)] + [np.zeros((nwavel, self.nlos)) for _ in range(2)]
        rc = lib().orts_run(self._h, nwavel, lg.shape[0], _p(k), _p(w), _p(lg), _p(df), _p(em), _p(sem), _p(irr), _p(alb), nthreads,
                            _p(rad), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), _p(d[5]))
        if rc != 0:
            raise RuntimeError(f"two-stream oracle run failed with code {rc}")
        out = {"radiance": rad}
        if derivatives:
            out.update(dict(zip(("d_extinction", "d_ssa", "d_b1", "d_emission", "d_albedo", "d_surface_emission"), d)))
        return out
