"""sasktran2_amd -- MI355X-native drop-in for the DISCO hot path of SASKTRAN2.

The Python surface mirrors the reference's `sasktran2` package for this path (same class / property names and
argument order): Config, Geometry1D, ViewingGeometry (+ TangentAltitudeSolar, GroundViewingSolar), Atmosphere (raw
interface) and Engine.calculate_radiance().  Reference: src/sasktran2/{engine,atmosphere,config,geometry}.py,
src/sasktran2/viewinggeo/wrappers.py.  Everything numerical runs in the HIP library behind the C ABI of
include/sasktran2_amd.h.
"""
from __future__ import annotations

import contextlib
import ctypes
import enum

import numpy as np

from ._lib_loader import last_error, lib
from .hostmem import pool as _host

_dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


class InterpolationMethod(enum.IntEnum):  # cpp/include/sasktran2/grids.h:24
    ShellInterpolation = 0
    LinearInterpolation = 1
    LowerInterpolation = 2


class GeometryType(enum.IntEnum):  # cpp/include/sasktran2/geometry.h:11-16
    PlaneParallel = 0
    PseudoSpherical = 1
    Spherical = 2
    Ellipsoidal = 3


class MultipleScatterSource(enum.IntEnum):  # cpp/include/sasktran2/config.h:65-71
    DiscreteOrdinates = 0
    SuccessiveOrdersLegacy = 1
    TwoStream = 2
    NoSource = 3
    SuccessiveOrders = 4


class SingleScatterSource(enum.IntEnum):  # cpp/include/sasktran2/config.h:41-46
    Exact = 0
    SolarTable = 1
    DiscreteOrdinates = 2
    NoSource = 3


class EmissionSource(enum.IntEnum):  # cpp/include/sasktran2/config.h:122-128
    Standard = 0
    NoSource = 1
    DiscreteOrdinates = 2
    VolumeEmissionRate = 3
    TwoStream = 4


class OccultationSource(enum.IntEnum):  # cpp/include/sasktran2/config.h:104
    Standard = 0
    NoSource = 1


class FluxType(enum.IntEnum):  # cpp/include/sasktran2/config.h (FluxType)
    Upwelling = 0
    Downwelling = 1
    Actinic = 2
    Divergence = 3


class StokesBasis(enum.IntEnum):  # cpp/include/sasktran2/config.h:145
    Standard = 0
    Solar = 1
    Observer = 2


class Config:
    """src/sasktran2/config.py:21-765 (the keys this path reads)."""

    _KEYS = {"num_stokes": "num_stokes", "num_streams": "num_streams", "num_threads": "num_threads",
             "num_sza": "num_do_sza", "num_singlescatter_moments": "num_singlescatter_moments",
             "wavelength_batch_size": "wavelength_batch_size", "output_los_optical_depth": "output_los_optical_depth",
             "multiple_scatter_source": "multiple_scatter_source", "single_scatter_source": "single_scatter_source",
             "wf_enabled": "wf_enabled", "delta_m_scaling": "apply_delta_scaling", "stokes_basis": "stokes_basis",
             "emission_source": "emission_source", "occultation_source": "occultation_source", "do_backprop": "do_backprop",
             "num_forced_azimuth": "num_do_forced_azimuth",   # src/sasktran2/config.py:541-554
             # successive orders (src/sasktran2/config.py:410-418,505-514,607-634 -> the C API's `hr` keys)
             "num_successive_orders_iterations": "num_hr_spherical_iterations", "num_successive_orders_incoming": "num_hr_incoming",
             "num_successive_orders_outgoing": "num_hr_outgoing", "init_successive_orders_with_discrete_ordinates": "initialize_hr_with_do",
             # MultipleScatterSource.SuccessiveOrders: the fixed-point solver's settings (src/sasktran2/config.py:426-503)
             "successive_orders_anderson_depth": "successive_orders_anderson_depth"}
    _FLOAT_KEYS = ("successive_orders_relative_tolerance", "successive_orders_absolute_tolerance", "successive_orders_damping")

    def __init__(self):
        object.__setattr__(self, "_h", lib().sk_config_create())

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):   # (at interpreter shutdown the module globals may be gone already)
            lib().sk_config_destroy(self._h)

    def __getattr__(self, name):
        if name == "flux_types":   # src/sasktran2/config.py:745-765
            n = ctypes.c_int()
            lib().sk_config_get_num_flux_types(self._h, ctypes.byref(n))
            t = (ctypes.c_int * max(n.value, 1))()
            lib().sk_config_get_flux_types(self._h, t)
            return [FluxType(t[i]) for i in range(n.value)]
        if name in Config._FLOAT_KEYS:
            d = ctypes.c_double()
            getattr(lib(), f"sk_config_get_{name}")(self._h, ctypes.byref(d))
            return d.value
        key = Config._KEYS.get(name)
        if key is None:
            raise AttributeError(name)
        v = ctypes.c_int()
        getattr(lib(), f"sk_config_get_{key}")(self._h, ctypes.byref(v))
        return v.value

    def __setattr__(self, name, value):
        if name == "flux_types":
            t = (ctypes.c_int * max(len(value), 1))(*[int(v) for v in value])
            if lib().sk_config_set_flux_types(self._h, t, len(value)) != 0:
                raise RuntimeError(last_error())
            return
        if name in Config._FLOAT_KEYS:
            if getattr(lib(), f"sk_config_set_{name}")(self._h, float(value)) != 0:
                raise RuntimeError(last_error())
            return
        key = Config._KEYS.get(name)
        if key is None:
            raise AttributeError(f"Config has no option {name!r} on the MI355X path")
        if getattr(lib(), f"sk_config_set_{key}")(self._h, int(value)) != 0:
            raise RuntimeError(last_error())


class Geometry1D:
    """src/sasktran2/geometry.py:13-56."""

    def __init__(self, cos_sza, solar_azimuth, earth_radius_m, altitude_grid_m,
                 interpolation_method=InterpolationMethod.LinearInterpolation, geometry_type=GeometryType.Spherical):
        self._alt = np.ascontiguousarray(np.atleast_1d(altitude_grid_m), dtype=np.float64)
        self._h = lib().sk_geometry1d_create(cos_sza, solar_azimuth, earth_radius_m, _p(self._alt), len(self._alt),
                                             int(interpolation_method), int(geometry_type))
        if not self._h:
            raise ValueError("invalid Geometry1D")

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):   # (at interpreter shutdown the module globals may be gone already)
            lib().sk_geometry1d_destroy(self._h)

    def altitudes(self):
        return self._alt.copy()


class TangentAltitudeSolar:
    def __init__(self, tangent_altitude_m, relative_azimuth, observer_altitude_m, cos_sza):
        self.args = (tangent_altitude_m, relative_azimuth, observer_altitude_m, cos_sza)


class GroundViewingSolar:
    def __init__(self, cos_sza, relative_azimuth, cos_viewing_zenith, observer_altitude_m):
        self.args = (cos_sza, relative_azimuth, cos_viewing_zenith, observer_altitude_m)


class TangentAltitude:
    """src/sasktran2/viewinggeo/wrappers.py:38-172 (1D use: tangent point at `horizontal_angle` from the reference point in the
    solar plane, `viewing_azimuth` in its local horizontal plane)."""

    def __init__(self, tangent_altitude_m, observer_altitude_m, horizontal_angle=0.0, viewing_azimuth=0.0):
        self.args = (tangent_altitude_m, observer_altitude_m, horizontal_angle, viewing_azimuth)


class SolarAnglesObserverLocation:
    """src/sasktran2/viewinggeo/wrappers.py:255-295: observer at an altitude, look direction from solar angles at the observer."""

    def __init__(self, cos_sza, relative_azimuth, cos_viewing_zenith, observer_altitude_m):
        self.args = (cos_sza, relative_azimuth, cos_viewing_zenith, observer_altitude_m)


class FluxObserverSolar:
    """src/sasktran2/viewinggeo/wrappers.py:297-320: a horizontal surface at an altitude that the fluxes are reported on."""

    def __init__(self, cos_sza, observer_altitude_m):
        self.args = (cos_sza, observer_altitude_m)


class ViewingGeometry:
    """src/sasktran2/viewinggeo/wrappers.py:13-36."""

    def __init__(self):
        self._h = lib().sk_viewing_geometry_create()
        self._rays = []
        self._flux_observers = []

    def add_flux_observer(self, observer):
        if not isinstance(observer, FluxObserverSolar):
            raise TypeError("unsupported flux observer type on the MI355X path")
        if lib().sk_viewing_geometry_add_flux_observer_solar(self._h, *map(float, observer.args)) != 0:
            raise ValueError("invalid flux observer")
        self._flux_observers.append(observer)

    @property
    def flux_observers(self):
        return self._flux_observers

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):   # (at interpreter shutdown the module globals may be gone already)
            lib().sk_viewing_geometry_destroy(self._h)

    def add_ray(self, ray):
        if isinstance(ray, TangentAltitudeSolar):
            lib().sk_viewing_geometry_add_tangent_altitude_solar(self._h, *map(float, ray.args))
        elif isinstance(ray, GroundViewingSolar):
            c, az, cv, alt = map(float, ray.args)
            lib().sk_viewing_geometry_add_ground_viewing_solar(self._h, c, az, alt, cv)
        elif isinstance(ray, TangentAltitude):
            if lib().sk_viewing_geometry_add_tangent_altitude(self._h, *map(float, ray.args)) != 0:
                raise ValueError("invalid TangentAltitude ray")
        elif isinstance(ray, SolarAnglesObserverLocation):
            lib().sk_viewing_geometry_add_solar_angles_observer_location(self._h, *map(float, ray.args))
        else:
            raise TypeError("unsupported ray type on the MI355X path")
        self._rays.append(ray)

    @property
    def observer_rays(self):
        return self._rays


class _Storage:
    def __init__(self, ngrid, nwavel, nleg, nstokes=1):
        # (large arrays in page-locked memory, hostmem.py: the engine's uploads then run at the DMA engines' rate)
        self.total_extinction = _host.zeros((ngrid, nwavel), order="F")
        self.ssa = _host.zeros((ngrid, nwavel), order="F")
        self.emission_source = _host.zeros((ngrid, nwavel), order="F")
        # NSTOKES = 3: four Greek coefficients per order, order-major (cpp/include/c_api/atmosphere.h:77-83)
        self.leg_coeff = _host.zeros((nleg * (4 if nstokes == 3 else 1), ngrid, nwavel), order="F")
        self.solar_irradiance = np.ones(nwavel)


class _LegCoeff:
    """Views of the Greek coefficients (src/sasktran2/atmosphere.py `leg_coeff.a1 / a2 / a3 / b1`)."""

    def __init__(self, storage, nstokes=1):
        self._s = storage
        self._step = 4 if nstokes == 3 else 1

    @property
    def a1(self):
        return self._s.leg_coeff[0::self._step]

    def _polarised(self, k):
        if self._step == 1:
            raise AttributeError("only a1 exists for NSTOKES = 1")
        return self._s.leg_coeff[k::4]

    @property
    def a2(self):
        return self._polarised(1)

    @property
    def a3(self):
        return self._polarised(2)

    @property
    def b1(self):
        return self._polarised(3)


class _Surface:
    """Surface of an atmosphere: a Lambertian `albedo[wavelength]` by default.  `set_brdf("kokhanovsky" | "modis")` selects one of the reference's
    other BRDFs (cpp/include/sasktran2/atmosphere/surface.h:151-352) and allocates `brdf_args[argument, wavelength]` as the reference's
    `Surface.brdf_args` (SnowKokhanovsky: (chi + M) L / wavelength; MODIS: isotropic, volumetric, geometric); their weighting functions come
    back as `wf_brdf[argument, wavelength, los, stokes]`.  Served by the discrete-ordinates source in plane-parallel / pseudo-spherical geometry."""
    _NARGS = {"lambertian": 1, "kokhanovsky": 1, "modis": 3, "legacy_test_function": 1}   # (the last: the reference's functional-BRDF known answer, parity suite only)
    _CREATE = {"legacy_test_function": "sk_amd_brdf_create_legacy_test_function"}

    def __init__(self, nwavel):
        self.albedo = np.zeros(nwavel)
        self.emission = np.zeros(nwavel)
        self.brdf = "lambertian"
        self.brdf_args = None

    def set_brdf(self, kind):
        if kind not in self._NARGS:
            raise ValueError(f"unknown BRDF {kind!r}: lambertian, kokhanovsky, modis")
        self.brdf = kind
        self.brdf_args = None if kind == "lambertian" else np.zeros((self._NARGS[kind], self.albedo.size))

    @property
    def num_brdf_args(self):
        return self._NARGS[self.brdf]


class EquationOfState:
    """Ideal-gas number densities and their derivatives (src/sasktran2/util/state.py:8-114)."""
    K_BOLTZMANN = 1.380649e-23

    def __init__(self):
        self.pressure_pa = None
        self.temperature_k = None
        self.specific_humidity = None

    @property
    def air_numberdensity(self):
        kt = self.K_BOLTZMANN * self.temperature_k
        return {"N": self.pressure_pa / kt, "dN_dP": 1 / kt, "dN_dT": -self.pressure_pa / (kt * self.temperature_k)}

    @property
    def dry_air_numberdensity(self):
        wet = self.air_numberdensity
        q = 0 if self.specific_humidity is None else self.specific_humidity
        return {"N": wet["N"] * (1 - q), "dN_dP": wet["dN_dP"] * (1 - q), "dN_dT": wet["dN_dT"] * (1 - q), "dN_dsh": -wet["N"]}


class Atmosphere:
    """src/sasktran2/atmosphere.py:176-290, 719-863.  Two ways to fill it, as in the reference:

    * raw interface: fill `storage.total_extinction`, `storage.ssa` [altitude, wavelength], `leg_coeff.a1` [moment, altitude,
      wavelength] and `surface.albedo`; weighting functions `wf_extinction`, `wf_ssa`, `wf_albedo`;
    * constituents: `atmosphere["name"] = sk.constituent.X(...)` with `wavelengths_nm`, `pressure_pa`, `temperature_k` set.  The storage
      and every derivative mapping are then assembled in HBM by the device front end (`sasktran2_amd.frontend`); `storage` reads them
      back on access, and the weighting functions carry the constituents' names (`wf_<name>_vmr`, `wf_temperature_k`, ...)."""

    def __init__(self, model_geometry, config, wavelengths_nm=None, wavenumber_cminv=None, numwavel=None,
                 calculate_derivatives=True, pressure_derivative=True, temperature_derivative=True, **_):
        if wavelengths_nm is None and wavenumber_cminv is not None:
            wavelengths_nm = 1e7 / np.asarray(wavenumber_cminv, dtype=float)
        if numwavel is None:
            if wavelengths_nm is None:
                raise ValueError("One of wavelengths_nm, wavenumber_cminv, numwavel must be set")
            numwavel = len(wavelengths_nm)
        self.wavelengths_nm = None if wavelengths_nm is None else np.asarray(wavelengths_nm, dtype=float)
        self._ngrid = len(model_geometry.altitudes())
        self._nwavel = int(numwavel)
        self._calculate_derivatives = bool(calculate_derivatives)
        self._pressure_derivative = bool(pressure_derivative)
        self._temperature_derivative = bool(temperature_derivative)
        self.nstokes = int(config.num_stokes)
        self._num_moments = int(config.num_singlescatter_moments)
        self._raw_storage = _Storage(self._ngrid, self._nwavel, config.num_singlescatter_moments, self.nstokes)
        self.leg_coeff = _LegCoeff(self._raw_storage, self.nstokes)
        self.surface = _Surface(self._nwavel)
        self._altitudes = model_geometry.altitudes()
        self._model_geometry = model_geometry
        self._config = config
        self.state_equation = EquationOfState()
        self._constituents = {}
        self._frontend = None
        self._surface_mappings = {}

    # ---- state and constituents (src/sasktran2/atmosphere.py:291-520, 860-864)
    model_geometry = property(lambda self: self._model_geometry)
    num_wavel = property(lambda self: self._nwavel)
    num_locations = property(lambda self: self._ngrid)
    num_moments = property(lambda self: self._num_moments)
    num_leg = property(lambda self: self._num_moments * (4 if self.nstokes == 3 else 1))
    calculate_pressure_derivative = property(lambda self: self._calculate_derivatives and self._pressure_derivative)
    calculate_temperature_derivative = property(lambda self: self._calculate_derivatives and self._temperature_derivative)

    @property
    def wavenumbers_cminv(self):
        return None if self.wavelengths_nm is None else 1e7 / self.wavelengths_nm

    @property
    def pressure_pa(self):
        return self.state_equation.pressure_pa

    @pressure_pa.setter
    def pressure_pa(self, value):
        self.state_equation.pressure_pa = self._state(value, "pressure_pa")

    @property
    def temperature_k(self):
        return self.state_equation.temperature_k

    @temperature_k.setter
    def temperature_k(self, value):
        self.state_equation.temperature_k = self._state(value, "temperature_k")

    def _state(self, value, name):
        value = np.asarray(value, dtype=float)
        if value.shape != (self._ngrid,):
            raise ValueError(f"{name} must have one value per grid altitude")
        return value

    def __setitem__(self, item, value):
        self._constituents[item] = value

    def __getitem__(self, item):
        return self._constituents.get(item)

    @property
    def frontend(self):
        """the device front end the constituents write their terms and mappings into"""
        if self._frontend is None:
            from .frontend import FrontEnd
            self._frontend = FrontEnd(self._ngrid, self._nwavel, self.num_leg, self.nstokes)
        return self._frontend

    @property
    def storage(self):
        if self._constituents:
            from .frontend import DeviceStorage
            if self._frontend is None or not self._frontend.assembled:
                self.internal_object()   # the reference exposes the storage at any time: assemble on first use
            return DeviceStorage(self._frontend, self._raw_storage.solar_irradiance)
        return self._raw_storage

    def register_surface_mapping(self, name, interpolator, interp_dim, d_brdf=1.0, d_emission=None):
        """a derivative with respect to a surface parameter: d(BRDF argument) and d(surface emission) per wavelength, and the
        [wavelength, parameter] interpolator its output goes through (None: one value per wavelength, src/sasktran2/engine.py:19-32)"""
        d_brdf = np.asarray(d_brdf, dtype=float)
        # one column per BRDF argument as the reference's mapping (num_wavel x num_args); a vector / scalar is the single argument of a Lambertian or Kokhanovsky surface
        d_brdf = d_brdf.T if d_brdf.ndim == 2 else np.broadcast_to(d_brdf, (self._nwavel,))[None, :]
        self._surface_mappings[name] = (None if interpolator is None else np.asarray(interpolator, dtype=float), interp_dim,
                                        np.ascontiguousarray(d_brdf),
                                        None if d_emission is None else np.broadcast_to(np.asarray(d_emission, dtype=float), (self._nwavel,)))

    def internal_object(self):
        """Assemble the atmosphere from its constituents in HBM: add every constituent, normalise, then let every constituent register
        its derivative mappings (src/sasktran2/atmosphere.py:730-771).  Returns the front end; a raw-interface atmosphere has none."""
        if not self._constituents:
            return None
        fe = self.frontend
        fe.reset()
        self._surface_mappings = {}
        if any(getattr(c, "_adds_surface_emission", False) for c in self._constituents.values()):
            self.surface.emission[:] = 0.0      # constituents add to it (the reference zeroes the surface before a rebuild)
        try:
            for constituent in self._constituents.values():
                constituent.add_to_atmosphere(self)
            fe.set_solar_irradiance(self._raw_storage.solar_irradiance)
            fe.assemble()
            if self._calculate_derivatives:
                for name, constituent in self._constituents.items():
                    constituent.register_derivative(self, name)
                fe.finalize()
        except Exception:
            fe.reset()
            raise
        return fe

    @property
    def calculate_derivatives(self):
        return self._calculate_derivatives

    def _native_handles(self, L, nstokes, direct=False):
        """The storage / surface / atmosphere handles of this object (created once, as `Atmosphere.internal_object()` of the
        reference).  `direct`: the handle maps the caller-visible storage arrays themselves, as the reference's handle does
        (cpp/include/sasktran2/atmosphere/grid_storage.h:77-102) -- no host copy per call; otherwise private Fortran-ordered buffers that
        every calculation refills (delta-M scaling rewrites the storage in place and the caller's arrays keep the unscaled properties)."""
        handles = self.__dict__.setdefault("_handles", {})
        st = self._raw_storage
        H = handles.get(direct)
        if H is not None and H.get("brdf_kind", "lambertian") != self.surface.brdf:
            self._destroy_handles(handles.pop(direct))   # another BRDF was selected: the surface handle is built for one
            H = None
        if H is not None and direct and not (H["ext"] is st.total_extinction and H["ssa"] is st.ssa and H["em"] is st.emission_source and
                                             H["leg"] is st.leg_coeff and H["irr"] is st.solar_irradiance):
            self._destroy_handles(handles.pop(direct))   # the caller rebound a storage array: map the new one
            H = None
        if H is None:
            if direct:
                for name in ("total_extinction", "ssa", "emission_source", "leg_coeff", "solar_irradiance"):
                    a = getattr(st, name)
                    if a.dtype != np.float64 or not a.flags.f_contiguous:
                        setattr(st, name, np.asfortranarray(a, dtype=np.float64))
                H = {"ext": st.total_extinction, "ssa": st.ssa, "em": st.emission_source, "leg": st.leg_coeff, "irr": st.solar_irradiance,
                     "alb": np.zeros(self._nwavel), "emis": np.zeros(self._nwavel), "mappings_set": False}
            else:
                H = {"ext": _host.zeros(st.total_extinction.shape, order="F"), "ssa": _host.zeros(st.ssa.shape, order="F"),
                     "em": _host.zeros(st.emission_source.shape, order="F"), "leg": _host.zeros(st.leg_coeff.shape, order="F"),
                     "irr": np.ones(self._nwavel), "alb": np.zeros(self._nwavel), "emis": np.zeros(self._nwavel)}
            H["storage"] = L.sk_atmosphere_storage_create(self._ngrid, self._nwavel, H["leg"].shape[0], nstokes, _p(H["ssa"]),
                                                          _p(H["ext"]), _p(H["em"]), _p(H["leg"]), _p(H["irr"]))
            H["brdf_kind"] = self.surface.brdf
            H["alb"] = np.zeros((self._nwavel, self.surface.num_brdf_args))   # (num_args x num_wavel) column-major, surface.h:366-372
            H["brdf"] = getattr(L, _Surface._CREATE.get(self.surface.brdf, f"sk_brdf_create_{self.surface.brdf}"))(nstokes)
            H["surface"] = L.sk_surface_create(self._nwavel, nstokes, _p(H["emis"]))
            L.sk_surface_set_brdf(H["surface"], H["brdf"], _p(H["alb"]))
            H["atmosphere"] = L.sk_atmosphere_create(H["storage"], H["surface"], int(self._calculate_derivatives), 0)
            H["mappings_set"] = False
            handles[direct] = H
        return H

    @staticmethod
    def _destroy_handles(H):
        L = lib()
        L.sk_atmosphere_destroy(H["atmosphere"])
        L.sk_surface_destroy(H["surface"])
        L.sk_brdf_destroy(H["brdf"])
        L.sk_atmosphere_storage_destroy(H["storage"])

    def __del__(self):
        handles = self.__dict__.get("_handles")
        if handles and callable(lib):   # (at interpreter shutdown the module globals may be gone already)
            for H in handles.values():
                self._destroy_handles(H)
            handles.clear()


class Communicator:
    """RCCL communicator of a wavelength-sharded run (one process per GPU; additive to the reference's surface).  Rank 0 creates the
    128-byte id with `Communicator.unique_id()`, the application distributes it (MPI, a file, a torch store), every rank constructs
    `Communicator(id, rank, world)` after selecting its GPU."""

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        if lib().sk_amd_comm_unique_id(buf) != 0:
            raise RuntimeError(f"sk_amd_comm_unique_id failed: {last_error()}")
        return buf.raw

    def __init__(self, unique_id: bytes, rank: int, world: int):
        self.rank, self.world = rank, world
        self._h = lib().sk_amd_comm_create(ctypes.create_string_buffer(bytes(unique_id), 128), rank, world)
        if not self._h:
            raise RuntimeError(f"sk_amd_comm_create failed: {last_error()}")

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):
            lib().sk_amd_comm_destroy(self._h)
            self._h = None


class Engine:
    """src/sasktran2/engine.py:70-168,474-612 -- `calculate_radiance(atmosphere)` returns the reference's variables
    (`radiance[wavelength, los, stokes]`, `wf_extinction` / `wf_ssa` `[altitude, wavelength, los, stokes]`,
    `wf_albedo[wavelength, los, stokes]`, `los_optical_depth[wavelength, los]`) as a dict of ndarrays, wrapped in an
    xarray.Dataset when xarray is importable."""

    def __init__(self, config, model_geometry, viewing_geo):
        self._config, self._geometry, self._viewing = config, model_geometry, viewing_geo
        self._h = lib().sk_engine_create(config._h, model_geometry._h, viewing_geo._h)
        if not self._h:
            raise RuntimeError(f"Engine construction failed: {last_error()}")
        self.nlos = len(viewing_geo.observer_rays)

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):   # (at interpreter shutdown the module globals may be gone already)
            lib().sk_engine_destroy(self._h)

    def _prepare(self, atmosphere, derivs):
        """Handle flow of the reference (src/sasktran2/atmosphere.py:719-858): ONE storage / surface / atmosphere handle per Atmosphere
        object, and before every calculation set_zero -> refill -> derivative mappings -> apply_delta_m_scaling on that handle."""
        L = lib()
        st = atmosphere._raw_storage
        nw, ngrid = atmosphere._nwavel, atmosphere._ngrid
        S = int(self._config.num_stokes)
        per = 4 if S == 3 else 1
        delta_order = int(self._config.num_streams) if self._config.delta_m_scaling and st.leg_coeff.shape[0] // per > self._config.num_streams else None
        # delta-M scaling rewrites the storage in place (src/sasktran2/atmosphere.py:845-854 applies it with order = num_streams
        # when more moments than streams are stored): the handle then maps private copies, the caller's arrays keep the unscaled
        # properties.  Without it the handle maps the caller-visible arrays themselves and nothing is copied on the host.
        direct = delta_order is None
        H = atmosphere._native_handles(L, S, direct)
        if not direct:
            L.sk_atmosphere_storage_set_zero(H["storage"])
            H["ext"][...] = st.total_extinction
            H["ssa"][...] = st.ssa
            H["em"][...] = st.emission_source
            H["leg"][...] = st.leg_coeff
            H["irr"][...] = st.solar_irradiance
        if atmosphere.surface.brdf == "lambertian":
            H["alb"][:, 0] = atmosphere.surface.albedo
        else:
            H["alb"][...] = np.asarray(atmosphere.surface.brdf_args, dtype=float).T
        H["emis"][...] = atmosphere.surface.emission
        if derivs and not (direct and H["mappings_set"]):
            # raw-interface mappings, src/sasktran2/atmosphere.py:784-806 (identity; nothing rescales them without delta-M: set once)
            for name, which in (("wf_extinction", "d_extinction"), ("wf_ssa", "d_ssa")):
                m = ctypes.c_void_p()
                L.sk_atmosphere_storage_get_derivative_mapping(H["storage"], name.encode(), ctypes.byref(m))
                ptr = _dp()
                getattr(L, f"sk_deriv_mapping_get_{which}")(m, ctypes.byref(ptr))
                np.ctypeslib.as_array(ptr, shape=(ngrid * nw,))[:] = 1.0
            nargs = atmosphere.surface.num_brdf_args
            for a in range(nargs):   # one mapping per BRDF argument: d_brdf (num_wavel x num_args, column-major) is its unit column
                m = ctypes.c_void_p()
                name = b"wf_albedo" if atmosphere.surface.brdf == "lambertian" else f"wf_brdf_{a}".encode()
                L.sk_surface_get_derivative_mapping(H["surface"], name, ctypes.byref(m))
                ptr = _dp()
                L.sk_surface_deriv_mapping_get_d_brdf(m, ctypes.byref(ptr))
                d = np.ctypeslib.as_array(ptr, shape=(nargs, nw))
                d[:] = 0.0
                d[a] = 1.0
            if direct:
                H["mappings_set"] = True
        return H, delta_order

    def _apply_delta_m(self, H, delta_order):
        if delta_order is not None:
            rc = lib().sk_atmosphere_apply_delta_m_scaling(H["atmosphere"], delta_order)
            if rc != 0:
                raise RuntimeError(f"sk_atmosphere_apply_delta_m_scaling returned {rc}")

    @staticmethod
    def _surface_handles(L, atmosphere, S):
        """BRDF and surface handles of a constituent atmosphere for one call (keeps the argument / emission arrays alive in the tuple)."""
        nw = atmosphere._nwavel
        if atmosphere.surface.brdf == "lambertian":
            alb = np.ascontiguousarray(atmosphere.surface.albedo, dtype=float)
        else:
            alb = np.ascontiguousarray(np.asarray(atmosphere.surface.brdf_args, dtype=float).T)   # (num_args x num_wavel) column-major
        emis = np.ascontiguousarray(atmosphere.surface.emission, dtype=float)
        h_brdf = getattr(L, _Surface._CREATE.get(atmosphere.surface.brdf, f"sk_brdf_create_{atmosphere.surface.brdf}"))(S)
        h_su = L.sk_surface_create(nw, S, _p(emis))
        L.sk_surface_set_brdf(h_su, h_brdf, _p(alb))
        return h_brdf, h_su, (alb, emis)

    @staticmethod
    def _fill_surface_mapping(L, h_su, atmosphere, name):
        nw, nargs = atmosphere._nwavel, atmosphere.surface.num_brdf_args
        _, _, d_brdf, d_emission = atmosphere._surface_mappings[name]
        if d_brdf.shape != (nargs, nw):
            raise ValueError(f"surface mapping {name!r} holds {d_brdf.shape[0]} BRDF argument(s), the surface has {nargs}")
        m = ctypes.c_void_p()
        L.sk_surface_get_derivative_mapping(h_su, name.encode(), ctypes.byref(m))
        ptr = _dp()
        L.sk_surface_deriv_mapping_get_d_brdf(m, ctypes.byref(ptr))
        np.ctypeslib.as_array(ptr, shape=(nargs, nw))[:] = d_brdf
        if d_emission is not None:
            L.sk_surface_deriv_mapping_get_d_emission(m, ctypes.byref(ptr))
            np.ctypeslib.as_array(ptr, shape=(nw,))[:] = d_emission

    @contextlib.contextmanager
    def _constituent_atmosphere(self, atmosphere):
        """Atmosphere handle of a constituent atmosphere for one call: the storage assembled in HBM by the front end, a surface with the
        surface derivative mappings of the constituents, delta-M scaling applied as `calculate_radiance` does.  Yields (front end, surface
        handle, atmosphere handle) and destroys the per-call handles afterwards."""
        L = lib()
        fe = atmosphere.internal_object()
        nw, S = atmosphere._nwavel, int(self._config.num_stokes)
        h_brdf, h_su, _keep = self._surface_handles(L, atmosphere, S)
        h_atm = L.sk_atmosphere_create(fe.storage_handle(), h_su, int(atmosphere.calculate_derivatives), 0)
        try:
            for name in atmosphere._surface_mappings:
                self._fill_surface_mapping(L, h_su, atmosphere, name)
            if self._config.delta_m_scaling and atmosphere.num_moments > self._config.num_streams:
                rc = L.sk_atmosphere_apply_delta_m_scaling(h_atm, int(self._config.num_streams))
                if rc != 0:
                    raise RuntimeError(f"sk_atmosphere_apply_delta_m_scaling returned {rc}: {last_error()}")
            yield fe, h_su, h_atm
        finally:
            L.sk_atmosphere_destroy(h_atm)
            L.sk_surface_destroy(h_su)
            L.sk_brdf_destroy(h_brdf)
            del _keep

    def calculate_jvp(self, atmosphere, tangents, surface_tangents=None):
        """Jacobian-vector product (src/sasktran2/engine.py:170-312, `sk_engine_calculate_jvp`): `tangents` maps a derivative-mapping
        name ("wf_extinction", "wf_ssa") to a vector over its outputs, `surface_tangents` ("wf_albedo") to one value per wavelength.
        Returns (radiance, jvp), both [wavelength, los, stokes]."""
        L = lib()
        nw, S = atmosphere._nwavel, int(self._config.num_stokes)
        with contextlib.ExitStack() as stack:
            if atmosphere._constituents:
                # constituent interface: tangents are keyed by the mapping names of the constituents ("wf_o3_vmr", ...), one value per
                # output of the mapping; the storage stays in HBM
                if not atmosphere.calculate_derivatives:
                    raise ValueError("calculate_jvp needs an Atmosphere with calculate_derivatives=True")
                _, _, h_atm = stack.enter_context(self._constituent_atmosphere(atmosphere))
            else:
                H, delta_order = self._prepare(atmosphere, True)
                self._apply_delta_m(H, delta_order)
                h_atm = H["atmosphere"]
            return self._jvp_on_handle(L, h_atm, nw, S, tangents, surface_tangents)

    def _jvp_on_handle(self, L, h_atm, nw, S, tangents, surface_tangents):
        H = {"atmosphere": h_atm}
        rad, jvp = np.zeros((nw, self.nlos, S)), np.zeros((nw, self.nlos, S))
        h = L.sk_output_jvp_create(_p(rad), _p(jvp), nw * self.nlos, S)
        keep = []
        try:
            for name, t in tangents.items():
                t = np.ascontiguousarray(t, dtype=float)
                keep.append(t)
                L.sk_output_jvp_assign_derivative_tangent(h, name.encode(), _p(t), t.size)
            for name, t in (surface_tangents or {}).items():
                t = np.ascontiguousarray(t, dtype=float)
                keep.append(t)
                L.sk_output_jvp_assign_surface_tangent(h, name.encode(), _p(t), t.size)
            rc = L.sk_engine_calculate_jvp(self._h, H["atmosphere"], h)
            if rc != 0:
                raise RuntimeError(f"sk_engine_calculate_jvp returned {rc}: {last_error()}")
        finally:
            L.sk_output_jvp_destroy(h)
        return rad, jvp

    def calculate_vjp(self, atmosphere, cotangent, names=("wf_extinction", "wf_ssa"), surface_names=("wf_albedo",)):
        """Vector-Jacobian product (`sk_engine_calculate_vjp`): cotangent [wavelength, los, stokes] -> (radiance, {name: gradient})."""
        L = lib()
        nw, S, ngrid = atmosphere._nwavel, int(self._config.num_stokes), atmosphere._ngrid
        with contextlib.ExitStack() as stack:
            if atmosphere._constituents:
                # constituent interface: one gradient per derivative mapping of the constituents (all of them unless `names` /
                # `surface_names` were given explicitly), of the mapping's own output size
                if not atmosphere.calculate_derivatives:
                    raise ValueError("calculate_vjp needs an Atmosphere with calculate_derivatives=True")
                fe, _, h_atm = stack.enter_context(self._constituent_atmosphere(atmosphere))
                if names == ("wf_extinction", "wf_ssa"):
                    names = tuple(fe.mappings)
                if surface_names == ("wf_albedo",):
                    surface_names = tuple(atmosphere._surface_mappings)
                sizes = {}
                for name in names:
                    if name not in fe.mappings:
                        raise KeyError(f"unknown derivative mapping {name!r}; the constituents registered {sorted(fe.mappings)}")
                    sizes[name] = fe.mappings[name].num_output
            else:
                H, delta_order = self._prepare(atmosphere, True)
                self._apply_delta_m(H, delta_order)
                h_atm = H["atmosphere"]
                sizes = {name: ngrid for name in names}
                if surface_names == ("wf_albedo",) and atmosphere.surface.brdf != "lambertian":   # raw interface: one mapping per BRDF argument
                    surface_names = tuple(f"wf_brdf_{a}" for a in range(atmosphere.surface.num_brdf_args))
            return self._vjp_on_handle(L, h_atm, nw, S, cotangent, names, surface_names, sizes)

    def _vjp_on_handle(self, L, h_atm, nw, S, cotangent, names, surface_names, sizes):
        H = {"atmosphere": h_atm}
        rad = np.zeros((nw, self.nlos, S))
        cot = np.ascontiguousarray(cotangent, dtype=float).reshape(nw, self.nlos, S)
        h = L.sk_output_vjp_create(_p(rad), _p(cot), nw * self.nlos, S)
        grads = {}
        try:
            for name in names:
                grads[name] = np.zeros(sizes[name])
                L.sk_output_vjp_assign_derivative_gradient(h, name.encode(), _p(grads[name]), sizes[name])
            for name in surface_names:
                grads[name] = np.zeros(nw)
                L.sk_output_vjp_assign_surface_gradient(h, name.encode(), _p(grads[name]), nw)
            rc = L.sk_engine_calculate_vjp(self._h, H["atmosphere"], h)
            if rc != 0:
                raise RuntimeError(f"sk_engine_calculate_vjp returned {rc}: {last_error()}")
            L.sk_output_vjp_finalize(h)
        finally:
            L.sk_output_vjp_destroy(h)
        return rad, grads

    def calculate_radiance(self, atmosphere, as_dataset=True, _mapping_hook=None, shard=None):
        """`_mapping_hook(L, storage_handle, output_handle, out_dict)` lets a caller register further derivative mappings through
        the C ABI before the solve (used by the tests for interpolated / log-space mappings).
        `shard=(communicator, rank, world)` (additive; `sasktran2_amd.Communicator`): one process per GPU, every rank passes the full
        atmosphere, solves its wavelength range and receives the full result through one RCCL all-gather
        (`sk_amd_engine_calculate_radiance_sharded`); `shard=("emulate", world, fail_rank)` plays the ranks of that partition in turn in
        this process (no collective: what one GPU can check of the sharded entry point; `fail_rank = -1`: no simulated failure)."""
        if atmosphere._constituents:
            if shard is not None:
                raise ValueError("shard= serves the raw interface (constituent atmospheres live in HBM: use sasktran2_amd.distributed.ShardedSolver)")
            if _mapping_hook is not None:
                raise ValueError("_mapping_hook belongs to the raw interface")
            return self._calculate_radiance_constituents(atmosphere, as_dataset)
        L = lib()
        nw, ngrid, nlos = atmosphere._nwavel, atmosphere._ngrid, self.nlos
        S = int(self._config.num_stokes)
        derivs = atmosphere.calculate_derivatives
        H, delta_order = self._prepare(atmosphere, derivs)
        h_st, h_su, h_atm = H["storage"], H["surface"], H["atmosphere"]
        rad = _host.zeros((nw, nlos, S))
        flux, flux_names = self._flux_output(nw)
        h_out = L.sk_output_create(_p(rad), nw * nlos, S, _p(flux) if flux is not None else None, 0 if flux is None else flux.size)
        out = {"radiance": rad}
        try:
            if derivs:
                for name in ("wf_extinction", "wf_ssa"):
                    out[name] = _host.empty((ngrid, nw, nlos, S))   # (written in full by the call; page-locked: see hostmem.py)
                    L.sk_output_assign_derivative_memory(h_out, name.encode(), _p(out[name]), nw * nlos, S, ngrid)
                if atmosphere.surface.brdf == "lambertian":
                    out["wf_albedo"] = _host.zeros((nw, nlos, S))
                    L.sk_output_assign_surface_derivative_memory(h_out, b"wf_albedo", _p(out["wf_albedo"]), nw * nlos, S)
                else:
                    out["wf_brdf"] = np.zeros((atmosphere.surface.num_brdf_args, nw, nlos, S))
                    for a in range(atmosphere.surface.num_brdf_args):
                        L.sk_output_assign_surface_derivative_memory(h_out, f"wf_brdf_{a}".encode(), _p(out["wf_brdf"][a]), nw * nlos, S)
            if _mapping_hook is not None:
                out["_surface_handle"] = h_su
                _mapping_hook(L, h_st, h_out, out)
                out.pop("_surface_handle", None)
            self._apply_delta_m(H, delta_order)
            if shard is not None and shard[0] == "emulate":
                # the partition of `world` ranks played in turn by this process, no collective (sk_amd_engine_calculate_radiance_sharded_emulated)
                _, world, fail_rank = shard
                rc = L.sk_amd_engine_calculate_radiance_sharded_emulated(self._h, h_atm, h_out, int(world), int(fail_rank))
            elif shard is not None:
                comm, rank, world = shard
                rc = L.sk_amd_engine_calculate_radiance_sharded(self._h, h_atm, h_out, comm._h, int(rank), int(world))
            else:
                rc = L.sk_engine_calculate_radiance(self._h, h_atm, h_out, 0)
            if rc != 0:
                raise RuntimeError(f"sk_engine_calculate_radiance returned {rc}: {last_error()}")
            od = _dp()
            L.sk_output_get_los_optical_depth(h_out, ctypes.byref(od))
            out["los_optical_depth"] = np.ctypeslib.as_array(od, shape=(nlos, nw)).T.copy()
            for i, name in enumerate(flux_names):   # src/sasktran2/engine.py:493-509
                out[f"{name}_flux"] = flux[i]
        finally:
            L.sk_output_destroy(h_out)
        if as_dataset:
            try:
                import xarray as xr
            except ImportError:
                return out
            dims = {"radiance": ("wavelength", "los", "stokes"), "wf_extinction": ("altitude", "wavelength", "los", "stokes"),
                    "wf_ssa": ("altitude", "wavelength", "los", "stokes"), "wf_albedo": ("wavelength", "los", "stokes"),
                    "los_optical_depth": ("wavelength", "los")}
            dims.update({f"{name}_flux": ("wavelength", "flux_location") for name in flux_names})
            return xr.Dataset({k: (dims[k], v) for k, v in out.items()}, coords={"altitude": atmosphere._altitudes})
        return out

    def _flux_output(self, nw):
        """The flux array [flux type][wavelength][flux observer] of one call and the names of its types (src/sasktran2/engine.py:493-509),
        or (None, []) without flux observers."""
        nobs = len(self._viewing.flux_observers)
        if nobs == 0:
            return None, []
        names = {0: "upwelling", 1: "downwelling", 2: "actinic", 3: "divergence"}
        types = [names[int(t)] for t in self._config.flux_types]
        return np.zeros((len(types), nw, nobs)), types

    def _calculate_radiance_constituents(self, atmosphere, as_dataset):
        """Constituent interface: the atmosphere is assembled in HBM (`Atmosphere.internal_object`), the storage handle of the device
        front end goes to the engine without an upload, and the weighting functions come back under the reference's names: one array
        per derivative mapping, mappings with the same `assign_name` added up (src/sasktran2/engine.py:340-372)."""
        L = lib()
        fe = atmosphere.internal_object()
        nw, nlos, S = atmosphere._nwavel, self.nlos, int(self._config.num_stokes)
        derivs = atmosphere.calculate_derivatives
        h_brdf, h_su, _keep = self._surface_handles(L, atmosphere, S)
        h_atm = L.sk_atmosphere_create(fe.storage_handle(), h_su, int(derivs), 0)
        rad = _host.zeros((nw, nlos, S))
        flux, flux_names = self._flux_output(nw)
        h_out = L.sk_output_create(_p(rad), nw * nlos, S, _p(flux) if flux is not None else None, 0 if flux is None else flux.size)
        raw, raw_surface = {}, {}
        try:
            if derivs:
                for name, m in fe.mappings.items():
                    raw[name] = _host.empty((m.num_output, nw, nlos, S))
                    L.sk_output_assign_derivative_memory(h_out, name.encode(), _p(raw[name]), nw * nlos, S, m.num_output)
                for name in atmosphere._surface_mappings:
                    self._fill_surface_mapping(L, h_su, atmosphere, name)
                    raw_surface[name] = _host.zeros((nw, nlos, S))
                    L.sk_output_assign_surface_derivative_memory(h_out, name.encode(), _p(raw_surface[name]), nw * nlos, S)
            # src/sasktran2/atmosphere.py:845-854: scale when more moments are stored than streams are used
            if self._config.delta_m_scaling and atmosphere.num_moments > self._config.num_streams:
                rc = L.sk_atmosphere_apply_delta_m_scaling(h_atm, int(self._config.num_streams))
                if rc != 0:
                    raise RuntimeError(f"sk_atmosphere_apply_delta_m_scaling returned {rc}: {last_error()}")
            rc = L.sk_engine_calculate_radiance(self._h, h_atm, h_out, 0)
            if rc != 0:
                raise RuntimeError(f"sk_engine_calculate_radiance returned {rc}: {last_error()}")
            od = _dp()
            L.sk_output_get_los_optical_depth(h_out, ctypes.byref(od))
            los_od = np.ctypeslib.as_array(od, shape=(nlos, nw)).T.copy()
        finally:
            L.sk_output_destroy(h_out)
            L.sk_atmosphere_destroy(h_atm)
            L.sk_surface_destroy(h_su)
            L.sk_brdf_destroy(h_brdf)
        out, dims = {"radiance": rad, "los_optical_depth": los_od}, {"radiance": ("wavelength", "los", "stokes"), "los_optical_depth": ("wavelength", "los")}
        for i, name in enumerate(flux_names):
            out[f"{name}_flux"], dims[f"{name}_flux"] = flux[i], ("wavelength", "flux_location")
        for name, m in fe.mappings.items():
            key = m.assign_name or name
            d = (m.interp_dim, "wavelength", "los", "stokes")
            if key in out:
                if dims[key] != d or out[key].shape != raw[name].shape:
                    raise ValueError(f"Radiance derivative mappings assigned to {key!r} have incompatible parameter dimensions")
                out[key] = out[key] + raw[name]
            else:
                out[key], dims[key] = raw[name], d
        for name, (interp, interp_dim, _, _) in atmosphere._surface_mappings.items():
            if interp is None:      # one value per wavelength (engine.py:19-22, 540-549)
                out[name], dims[name] = raw_surface[name], ("wavelength", "los", "stokes")
            else:
                out[name] = np.einsum("ij...,il->lij...", raw_surface[name], interp)
                dims[name] = (interp_dim, "wavelength", "los", "stokes")
        if as_dataset:
            try:
                import xarray as xr
            except ImportError:
                return out
            return xr.Dataset({k: (dims[k], v) for k, v in out.items()})
        return out


from . import climatology, constituent, optical  # noqa: E402,F401  (the constituent interface: sk.constituent.Rayleigh, sk.optical.HenyeyGreenstein, ...)
