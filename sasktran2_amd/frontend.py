"""Host handle of the device front end (`sk_amd_frontend_*`, include/sasktran2_amd.h): what `Atmosphere.internal_object()` drives when
constituents are set.  The reference's counterpart is the Rust `PyAtmosphere` storage the constituents write into
(src/sasktran2/atmosphere.py:262-272, 730-771); here the constituents hand over profiles and spectra, and the [location x wavelength]
arrays exist only in HBM."""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib_loader import last_error, lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _frozen_key(a):
    """identity of a spectrum that cannot have changed since it was uploaded: (address, shape) of a read-only float64 array (what the
    optical properties' spectrum caches hand out), None for an absent one, False for anything writeable"""
    if a is None:
        return None
    if not isinstance(a, np.ndarray) or a.flags.writeable or a.dtype != np.float64 or not a.flags.c_contiguous:
        return False
    return (a.__array_interface__["data"][0], a.shape)


class DeviceMapping:
    """Read-only view of one derivative mapping held in HBM (`atmosphere.storage.get_derivative_mapping(name)` of the reference)."""

    def __init__(self, fe, name, scatterer, emission, interpolator, interp_dim, assign_name, log_radiance_space=False):
        self._fe, self.name, self.scatterer, self.emission = fe, name, scatterer, emission
        self.interpolator, self.interp_dim, self.assign_name, self.log_radiance_space = interpolator, interp_dim, assign_name, log_radiance_space

    def _get(self, field, shape):
        return self._fe.download(f"{self.name}:{field}", shape)

    @property
    def d_extinction(self):
        return self._get("d_extinction", (self._fe.nwavel, self._fe.ngrid)).T

    @property
    def d_ssa(self):
        return self._get("d_ssa", (self._fe.nwavel, self._fe.ngrid)).T

    @property
    def scat_factor(self):
        return self._get("scat_factor", (self._fe.nwavel, self._fe.ngrid)).T if self.scatterer else None

    @property
    def d_leg_coeff(self):
        return self._get("d_legendre", (self._fe.nwavel, self._fe.ngrid, self._fe.nleg)).transpose(2, 1, 0) if self.scatterer else None

    @property
    def d_emission(self):
        return self._get("d_emission", (self._fe.nwavel, self._fe.ngrid)).T if self.emission else None

    @property
    def num_output(self):
        return self._fe.ngrid if self.interpolator is None else self.interpolator.shape[1]


class FrontEnd:
    def __init__(self, ngrid, nwavel, nleg, nstokes):
        self.ngrid, self.nwavel, self.nleg, self.nstokes = int(ngrid), int(nwavel), int(nleg), int(nstokes)
        self._h = lib().sk_amd_frontend_create(self.ngrid, self.nwavel, self.nleg, self.nstokes)
        if not self._h:
            raise RuntimeError(f"sk_amd_frontend_create failed: {last_error()}")
        self.mappings = {}
        self.assembled = False
        self._nterms = 0
        self.spectra_reused = 0   # terms of the current assembly whose spectra were already in HBM
        self._slots = []      # per term position: (key of the spectra in HBM, the arrays themselves -- held so the addresses stay theirs)

    def __del__(self):
        if getattr(self, "_h", None) and callable(lib):
            lib().sk_amd_frontend_destroy(self._h)
            self._h = None

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} returned {rc}: {last_error()}")
        return rc

    def reset(self):
        self._check(lib().sk_amd_frontend_reset(self._h), "sk_amd_frontend_reset")
        self.mappings = {}
        self.assembled = False
        self._nterms = 0
        self.spectra_reused = 0

    def add_term(self, profile, ext, scat=None, leg=None):
        profile, ext, scat, leg = _c(profile), _c(ext), _c(scat), _c(leg)
        if profile.shape != (self.ngrid,) or ext.shape != (self.nwavel,):
            raise ValueError("a term is profile [location] x spectrum [wavelength]")
        if scat is not None and scat.shape != (self.nwavel,):
            raise ValueError("the scattering spectrum has one value per wavelength")
        if leg is not None and leg.shape != (self.nwavel, self.nleg):
            raise ValueError("phase moments are [wavelength, first dimension of leg_coeff]")
        # spectra that are the same read-only arrays as at this position in the previous assembly are still in HBM: only the profile moves
        slot, key = self._nterms, (_frozen_key(ext), _frozen_key(scat), _frozen_key(leg))
        reusable = False not in key
        if reusable and slot < len(self._slots) and self._slots[slot][0] == key:
            index = lib().sk_amd_frontend_readd_term(self._h, _vp(profile))
            self.spectra_reused += index == slot
        else:
            index = -2
        if index == -2:
            del self._slots[slot:]
            self._slots.extend([(None, None)] * (slot - len(self._slots)))
            index = lib().sk_amd_frontend_add_term(self._h, _vp(profile), _vp(ext), _vp(scat), _vp(leg))
            if index >= 0:
                self._slots.append((key, (ext, scat, leg)) if reusable else (None, None))
        if index != slot:
            self._slots = []
            self._check(min(index, -1), "sk_amd_frontend_add_term")
        self._nterms += 1
        return index

    def set_profile(self, term, profile):
        profile = _c(profile)
        self._check(lib().sk_amd_frontend_set_profile(self._h, int(term), _vp(profile)), "sk_amd_frontend_set_profile")
        self.assembled = False

    def set_planck_emission(self, temperature_k, wavelengths_nm):
        t, w = _c(temperature_k), _c(wavelengths_nm)
        self._check(lib().sk_amd_frontend_set_planck_emission(self._h, _vp(t), _vp(w)), "sk_amd_frontend_set_planck_emission")

    def set_solar_irradiance(self, irradiance):
        v = _c(irradiance)
        self._check(lib().sk_amd_frontend_set_solar_irradiance(self._h, _vp(v)), "sk_amd_frontend_set_solar_irradiance")

    def assemble(self):
        self._check(lib().sk_amd_frontend_assemble(self._h), "sk_amd_frontend_assemble")
        self.mappings = {}
        self.assembled = True

    def add_mapping(self, name, terms=(), weights=None, scatterer=False, planck=False, interpolator=None, interp_dim="altitude", assign_name="",
                    log_radiance_space=False):
        """interpolator: [location, nout] (rows = model grid), as the reference's mappings hold it; log_radiance_space: the engine divides
        this weighting function by the radiance (`DerivativeMapping.log_radiance_space` of the reference)"""
        terms = np.ascontiguousarray(terms, dtype=np.int32)
        w = None
        if weights is not None:
            w = _c(weights)
            if w.shape != (len(terms), self.ngrid):
                raise ValueError("weights are [term, location]")
        interp, nout = None, 0
        if interpolator is not None:
            interpolator = np.asarray(interpolator, dtype=np.float64)
            if interpolator.shape[0] != self.ngrid:
                raise ValueError("the interpolator has one row per grid location")
            nout = interpolator.shape[1]
            interp = np.asfortranarray(interpolator)
        self._check(lib().sk_amd_frontend_add_mapping(self._h, name.encode(), len(terms), terms.ctypes.data_as(ctypes.c_void_p), _vp(w), int(scatterer),
                                                      int(planck), _vp(interp), nout, interp_dim.encode(), assign_name.encode()),
                    "sk_amd_frontend_add_mapping")
        if log_radiance_space:
            m = ctypes.c_void_p()
            self._check(lib().sk_atmosphere_storage_get_derivative_mapping(self.storage_handle(), name.encode(), ctypes.byref(m)), "sk_atmosphere_storage_get_derivative_mapping")
            self._check(lib().sk_deriv_mapping_set_log_radiance_space(m, 1), "sk_deriv_mapping_set_log_radiance_space")
        self.mappings[name] = DeviceMapping(self, name, bool(scatterer), bool(planck), None if interpolator is None else interpolator.copy(),
                                            interp_dim, assign_name, bool(log_radiance_space))

    def finalize(self):
        """form the arrays of every mapping added so far (two launches for all of them)"""
        self._check(lib().sk_amd_frontend_finalize(self._h), "sk_amd_frontend_finalize")

    def storage_handle(self):
        h = lib().sk_amd_frontend_storage(self._h)
        if not h:
            raise RuntimeError("the front end has not been assembled")
        return h

    def download(self, what, shape):
        out = np.zeros(shape)
        self._check(lib().sk_amd_frontend_download(self._h, what.encode(), _vp(out)), "sk_amd_frontend_download")
        return out

    def set_profiling(self, on):
        self._check(lib().sk_amd_frontend_set_profiling(self._h, int(bool(on))), "sk_amd_frontend_set_profiling")

    def kernel_times(self):
        """{kernel: (total ms, launches)} since profiling was switched on"""
        buf = ctypes.create_string_buffer(1024)
        lib().sk_amd_frontend_kernel_names(self._h, buf, 1024)
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, n = ctypes.c_double(), ctypes.c_long()
            if lib().sk_amd_frontend_kernel_time(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(n)) == 0:
                out[name] = (ms.value, n.value)
        return out

    def device_pointer(self, what):
        p, n = ctypes.c_void_p(), ctypes.c_ulonglong()
        self._check(lib().sk_amd_frontend_device_pointer(self._h, what.encode(), ctypes.byref(p), ctypes.byref(n)), "sk_amd_frontend_device_pointer")
        return p.value, n.value


class DeviceStorage:
    """`atmosphere.storage` of an atmosphere assembled on the device: the reference's array names, copied from HBM on access."""

    def __init__(self, fe, solar_irradiance):
        self._fe = fe
        self.solar_irradiance = solar_irradiance

    @property
    def total_extinction(self):
        return self._fe.download("extinction", (self._fe.nwavel, self._fe.ngrid)).T

    @property
    def ssa(self):
        return self._fe.download("ssa", (self._fe.nwavel, self._fe.ngrid)).T

    @property
    def emission_source(self):
        return self._fe.download("emission", (self._fe.nwavel, self._fe.ngrid)).T

    @property
    def leg_coeff(self):
        return self._fe.download("leg_coeff", (self._fe.nwavel, self._fe.ngrid, self._fe.nleg)).transpose(2, 1, 0)

    def derivative_mapping_names(self):
        return list(self._fe.mappings)

    def get_derivative_mapping(self, name):
        return self._fe.mappings[name]
