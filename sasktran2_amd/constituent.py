"""Constituents of the reference's atmosphere interface (`atmosphere["o3"] = sk.constituent.VMRAltitudeAbsorber(...)`), for the classes
whose contribution is a sum of profile x spectrum terms.  Same names, arguments and weighting-function names as the reference; the
arithmetic on [location x wavelength] arrays runs in `frontend.hip` (see `sasktran2_amd.frontend`), these classes only prepare what is
small.

Reference behaviour followed (nothing copied):
  src/sasktran2/constituent/rayleigh.py, rust/sasktran2-rs/src/constituent/types/rayleigh.rs:110-258
  src/sasktran2/constituent/vmraltitudeabsorber.py, rust/.../vmr_alt_absorber.rs:64-260, 339-399
  src/sasktran2/constituent/numdenscatterer.py, rust/.../number_density_scatterer.rs:218-440
  src/sasktran2/constituent/collisioninducedabsorber.py
  src/sasktran2/constituent/gaussianheight.py
  src/sasktran2/constituent/amf.py
  src/sasktran2/constituent/manual.py, rust/.../manual.rs:25-111
  src/sasktran2/constituent/emission.py (ThermalEmission, SurfaceThermalEmission), rust/.../emission.rs:9-116
  src/sasktran2/constituent/brdf/lambertiansurface.py, brdf/__init__.py:11-63
"""
from __future__ import annotations

import numpy as np

from .optical import OpticalProperty, _SpectrumCache, linear_interpolating_matrix, rayleigh_cross_section_bates


class Constituent:
    """src/sasktran2/constituent/base.py"""

    def add_to_atmosphere(self, atmo):
        raise NotImplementedError

    def register_derivative(self, atmo, name):
        raise NotImplementedError

    def _altitude_matrix(self, atmo):
        """[grid altitude, own altitude] interpolation matrix; both grids are fixed for the life of their objects, so it is built once per
        model geometry"""
        held = getattr(self, "_matrix_for", None)
        if held is None or held[0] is not atmo.model_geometry:
            held = (atmo.model_geometry, linear_interpolating_matrix(self._altitudes_m, atmo.model_geometry.altitudes(), self._mode))
            held[1].setflags(write=False)
            self._matrix_for = held
        return held[1]


def _state_derivatives(atmo):
    """(name, dN/d name) of the state variables weighting functions are asked for (vmr_alt_absorber.rs:183-209)"""
    air = atmo.state_equation.dry_air_numberdensity
    out = []
    if atmo.calculate_pressure_derivative:
        out.append(("pressure_pa", air["dN_dP"]))
    if atmo.calculate_temperature_derivative:
        out.append(("temperature_k", air["dN_dT"]))
    return out


def _require_state(atmo, who):
    if atmo.wavelengths_nm is None:
        raise ValueError(f"It is required to give the Atmosphere object wavelengths to use the {who} constituent")
    if atmo.pressure_pa is None:
        raise ValueError(f"It is required to set the pressure_pa property in the Atmosphere object to use the {who} constituent")
    if atmo.temperature_k is None:
        raise ValueError(f"It is required to set the temperature_k property in the Atmosphere object to use the {who} constituent")


class Rayleigh(Constituent):
    def __init__(self, method="bates", n2_percentage=78.084, o2_percentage=20.946, ar_percentage=0.934, co2_percentage=0.036,
                 wavelengths_nm=None, xs=None, king_factor=None):
        method = method.lower()
        if method not in ("bates", "manual"):
            raise ValueError("Method must be bates or manual")
        if method == "manual" and (wavelengths_nm is None or xs is None or king_factor is None):
            raise ValueError("the manual method needs wavelengths_nm, xs and king_factor")
        self._method = method
        self._mix = (n2_percentage, o2_percentage, ar_percentage, co2_percentage)
        self._table = None if method == "bates" else tuple(np.asarray(v, dtype=float) for v in (wavelengths_nm, xs, king_factor))
        self._term = None
        self._cache = _SpectrumCache()

    def _cross_section(self, wavelengths_nm):
        if self._method == "bates":
            return rayleigh_cross_section_bates(wavelengths_nm / 1000.0, *self._mix)
        wl, xs, king = self._table          # held outside the table (rayleigh.rs:92-105)
        return np.interp(wavelengths_nm, wl, xs), np.interp(wavelengths_nm, wl, king)

    def add_to_atmosphere(self, atmo):
        _require_state(atmo, "Rayleigh")
        shape = (atmo.num_wavel, atmo.num_leg, atmo.nstokes)

        def build(_):
            sigma, king = self._cross_section(atmo.wavelengths_nm)
            delta = 6.0 * (king - 1.0) / (3.0 + 7.0 * king)
            factor = (1.0 - delta) / (2.0 + delta)
            leg = np.zeros(shape[:2])
            if atmo.nstokes == 1:
                leg[:, 0], leg[:, 2] = 1.0, factor
            else:   # a1, a2, a3, b1 per order (rayleigh.rs:145-158)
                leg[:, 0], leg[:, 8], leg[:, 9], leg[:, 11] = 1.0, factor, 6.0 * factor, np.sqrt(6.0) * factor
            return (np.ascontiguousarray(sigma, dtype=float), leg, shape)
        sigma, leg, built_for = self._cache.get(atmo.wavenumbers_cminv, build)
        if built_for != shape:       # the same spectral grid with another moment count / Stokes dimension
            self._cache = _SpectrumCache()
            sigma, leg, _ = self._cache.get(atmo.wavenumbers_cminv, build)
        self._term = atmo.frontend.add_term(atmo.state_equation.air_numberdensity["N"], sigma, sigma, leg)

    def register_derivative(self, atmo, name):
        air = atmo.state_equation.air_numberdensity
        for key, flag, factor in (("pressure_pa", atmo.calculate_pressure_derivative, air["dN_dP"]),
                                  ("temperature_k", atmo.calculate_temperature_derivative, air["dN_dT"])):
            if flag:
                atmo.frontend.add_mapping(f"wf_{name}_{key}", [self._term], scatterer=True, interpolator=np.diag(factor), interp_dim="altitude",
                                          assign_name=f"wf_{key}")


class VMRAltitudeAbsorber(Constituent):
    def __init__(self, optical_property: OpticalProperty, altitudes_m, vmr, out_of_bounds_mode="zero"):
        if optical_property.is_scatterer:
            raise ValueError("VMRAltitudeAbsorber takes a purely absorbing optical property")
        self._optical_property = optical_property
        self._altitudes_m = np.asarray(altitudes_m, dtype=float)
        self.vmr = np.asarray(vmr, dtype=float)
        self._mode = out_of_bounds_mode.lower()
        self._terms = []

    @property
    def altitudes_m(self):
        return self._altitudes_m

    def _profile(self, atmo):
        m = self._altitude_matrix(atmo)
        return m, m @ self.vmr

    def add_to_atmosphere(self, atmo):
        _require_state(atmo, "VMRAltitudeAbsorber")
        _, vmr = self._profile(atmo)
        n = atmo.state_equation.dry_air_numberdensity["N"]
        self._spectral = self._optical_property.spectral_terms(atmo)
        self._terms = [atmo.frontend.add_term(vmr * n * weight, spec["ext"]) for weight, _, spec in self._spectral]

    def register_derivative(self, atmo, name):
        m, vmr = self._profile(atmo)
        n = atmo.state_equation.dry_air_numberdensity["N"]
        weights = np.array([w for w, _, _ in self._spectral])
        fe = atmo.frontend
        fe.add_mapping(f"wf_{name}_vmr", self._terms, weights, interpolator=m * n[:, np.newaxis], interp_dim=f"{name}_altitude")
        state = _state_derivatives(atmo)
        for key, factor in state:
            fe.add_mapping(f"wf_{name}_{key}", self._terms, weights, interpolator=np.diag(vmr * factor), interp_dim="altitude", assign_name=f"wf_{key}")
        if state:
            # d(cross section)/d(table parameter) through the interpolation weights (vmr_alt_absorber.rs:233-256)
            for key in sorted({k for _, slopes, _ in self._spectral for k in slopes}):
                slopes = np.array([s.get(key, np.zeros(atmo.num_locations)) for _, s, _ in self._spectral])
                fe.add_mapping(f"wf_{name}_{key}_xs", self._terms, slopes, interpolator=np.diag(vmr * n), interp_dim="altitude", assign_name=f"wf_{key}")


class CollisionInducedAbsorber(Constituent):
    """Collision-induced absorption of a pair with constant mole fractions: extinction = cross section [m^5] x (fraction product) x N^2
    (src/sasktran2/constituent/collisioninducedabsorber.py).  `name`: "O2O2"."""

    def __init__(self, optical_property: OpticalProperty, name: str):
        if optical_property.is_scatterer:
            raise ValueError("CollisionInducedAbsorber takes a purely absorbing optical property")
        if name.lower() != "o2o2":
            raise ValueError(f"Unknown name '{name}'")
        self._optical_property = optical_property
        self._fraction_product = 0.20964 ** 2
        self._terms = []

    def add_to_atmosphere(self, atmo):
        _require_state(atmo, "O2O2")
        n = atmo.state_equation.air_numberdensity["N"]
        self._spectral = self._optical_property.spectral_terms(atmo)
        self._terms = [atmo.frontend.add_term(self._fraction_product * n ** 2 * weight, spec["ext"]) for weight, _, spec in self._spectral]

    def register_derivative(self, atmo, name):
        air = atmo.state_equation.air_numberdensity
        n = air["N"]
        weights = np.array([w for w, _, _ in self._spectral])
        state = [(key, d) for key, flag, d in (("pressure_pa", atmo.calculate_pressure_derivative, air["dN_dP"]),
                                               ("temperature_k", atmo.calculate_temperature_derivative, air["dN_dT"])) if flag]
        for key, dn in state:   # through the number density: d(N^2) = 2 N dN
            atmo.frontend.add_mapping(f"wf_{name}_{key}", self._terms, weights * (2.0 * self._fraction_product * n * dn)[np.newaxis, :],
                                      interp_dim="altitude", assign_name=f"wf_{key}")
        if state:
            for key in sorted({k for _, slopes, _ in self._spectral for k in slopes}):   # through the table's own parameters
                slopes = np.array([s.get(key, np.zeros(atmo.num_locations)) for _, s, _ in self._spectral])
                atmo.frontend.add_mapping(f"wf_{name}_{key}_xs", self._terms, slopes * (self._fraction_product * n ** 2)[np.newaxis, :],
                                          interp_dim="altitude", assign_name=f"wf_{key}")


class NumberDensityScatterer(Constituent):
    _wf_name = "number_density"

    def __init__(self, optical_property: OpticalProperty, altitudes_m, number_density, out_of_bounds_mode="zero", **kwargs):
        if kwargs:   # the reference passes these to parameterised scattering databases (particle size axes), which this build does not hold
            raise ValueError(f"auxiliary optical parameters {sorted(kwargs)} need a parameterised scattering database, which is not part of this build")
        if not optical_property.is_scatterer:
            raise ValueError("NumberDensityScatterer takes a scattering optical property")
        self._optical_property = optical_property
        self._altitudes_m = np.asarray(altitudes_m, dtype=float)
        self.number_density = np.zeros_like(self._altitudes_m) if number_density is None else np.asarray(number_density, dtype=float)
        self._mode = out_of_bounds_mode.lower()
        self._vertical_deriv_factor = np.ones_like(self._altitudes_m)
        self._terms = []

    @property
    def altitudes_m(self):
        return self._altitudes_m

    def add_to_atmosphere(self, atmo):
        if atmo.wavelengths_nm is None:
            raise ValueError("the atmosphere needs wavelengths for a scatterer defined by an optical property")
        m = self._altitude_matrix(atmo)
        nd = m @ self.number_density
        self._spectral = self._optical_property.spectral_terms(atmo)
        self._terms = [atmo.frontend.add_term(nd * weight, spec["ext"], spec["scat"], spec["leg"]) for weight, _, spec in self._spectral]

    def register_derivative(self, atmo, name):
        m = self._altitude_matrix(atmo)
        weights = np.array([w for w, _, _ in self._spectral])
        atmo.frontend.add_mapping(f"wf_{name}_{self._wf_name}", self._terms, weights, scatterer=True,
                                  interpolator=m * self._vertical_deriv_factor[np.newaxis, :], interp_dim=f"{name}_altitude")


class ExtinctionScatterer(NumberDensityScatterer):
    """An extinction profile [1/m] at one wavelength; the number density follows from the optical property's cross section there
    (src/sasktran2/constituent/numdenscatterer.py:117-197)."""
    _wf_name = "extinction"

    def __init__(self, optical_property, altitudes_m, extinction_per_m, extinction_wavelength_nm, out_of_bounds_mode="zero", **kwargs):
        super().__init__(optical_property, altitudes_m, None, out_of_bounds_mode, **kwargs)
        self.extinction_per_m = np.asarray(extinction_per_m, dtype=float)
        self._extinction_wavelength_nm = float(extinction_wavelength_nm)

    def _update_numberdensity(self):
        xs = self._optical_property.cross_sections(np.array([self._extinction_wavelength_nm]))[0]
        if not xs > 0.0:
            raise ValueError("the optical property has no cross section at the extinction wavelength")
        self._vertical_deriv_factor = np.full(len(self._altitudes_m), 1.0 / xs)
        self.number_density = self.extinction_per_m / xs

    def add_to_atmosphere(self, atmo):
        self._update_numberdensity()
        super().add_to_atmosphere(atmo)


class GaussianHeightExtinction(NumberDensityScatterer):
    """A scatterer whose extinction profile is a gaussian in altitude (src/sasktran2/constituent/gaussianheight.py): centre `height_m`,
    full width at half maximum `width_fwhm_m`, and the vertical optical depth at one wavelength.  Weighting functions
    `wf_<name>_height_m`, `wf_<name>_width_fwhm_m`, `wf_<name>_vertical_optical_depth` (one parameter each)."""

    def __init__(self, optical_property, height_m, width_fwhm_m, vertical_optical_depth, vertical_optical_depth_wavel_nm, altitudes_m,
                 out_of_bounds_mode="zero", **kwargs):
        super().__init__(optical_property, altitudes_m, None, out_of_bounds_mode, **kwargs)
        self.height_m, self.width_fwhm_m, self.vertical_optical_depth = float(height_m), float(width_fwhm_m), float(vertical_optical_depth)
        self._vertical_optical_depth_wavel_nm = float(vertical_optical_depth_wavel_nm)

    def _update_numberdensity(self):
        a = self._altitudes_m
        self._xs_at_wavel = self._optical_property.cross_sections(np.array([self._vertical_optical_depth_wavel_nm]))[0]
        if not self._xs_at_wavel > 0.0:
            raise ValueError("the optical property has no cross section at the wavelength of the vertical optical depth")
        self._gaussian = np.exp(-4.0 * np.log(2.0) * (a - self.height_m) ** 2 / self.width_fwhm_m ** 2)
        self._gaussian_od = np.trapezoid(self._gaussian, a)
        self.number_density = self._gaussian * self.vertical_optical_depth / self._gaussian_od / self._xs_at_wavel

    def add_to_atmosphere(self, atmo):
        self._update_numberdensity()
        super().add_to_atmosphere(atmo)

    def register_derivative(self, atmo, name):
        a, g, ln2 = self._altitudes_m, self._gaussian, np.log(2.0)
        dg_dh = g * 8.0 * ln2 * (a - self.height_m) / self.width_fwhm_m ** 2
        dg_dw = g * 8.0 * ln2 * (a - self.height_m) ** 2 / self.width_fwhm_m ** 3
        outer = self.vertical_optical_depth / self._gaussian_od / self._xs_at_wavel
        # the integral in the denominator moves with the parameter as well (gaussianheight.py:171-199)
        columns = {"height_m": outer * (dg_dh - g * np.trapezoid(dg_dh, a) / self._gaussian_od),
                   "width_fwhm_m": outer * (dg_dw - g * np.trapezoid(dg_dw, a) / self._gaussian_od),
                   "vertical_optical_depth": g / self._gaussian_od / self._xs_at_wavel}
        m = self._altitude_matrix(atmo)
        weights = np.array([w for w, _, _ in self._spectral])
        for key, column in columns.items():
            atmo.frontend.add_mapping(f"wf_{name}_{key}", self._terms, weights, scatterer=True, interpolator=m @ column[:, np.newaxis],
                                      interp_dim=f"{name}_{key}")


class Manual(Constituent):
    """Extinction [1/m] and single scatter albedo given directly on the model grid, `[altitude, wavelength]`, optionally with phase moments
    `[moment, altitude, wavelength]` (first dimension of leg_coeff); mixes with the other constituents; no weighting functions of its own
    (src/sasktran2/constituent/manual.py, rust/.../manual.rs).  On the device: one term per level, its profile that level alone.  The
    arrays are copied at construction (and stay in HBM from one assembly to the next)."""

    def __init__(self, extinction, ssa, legendre_moments=None, delta_scale=False):
        extinction, ssa = np.array(extinction, dtype=float), np.array(ssa, dtype=float)
        if extinction.shape != ssa.shape:
            raise ValueError("extinction and ssa must have the same shape")
        if extinction.ndim == 3:
            raise NotImplementedError("(horizontal, altitude, wavelength) Manual constituents belong to the 2-D geometry, which is not part of this build")
        if extinction.ndim != 2:
            raise ValueError("extinction and ssa must have shape (altitude, wavelength)")
        leg = None
        if legendre_moments is not None:
            leg = np.array(legendre_moments, dtype=float)
            if leg.ndim != 3 or leg.shape[1:] != extinction.shape:
                raise ValueError(f"legendre_moments must have shape (moment, *extinction.shape); got {leg.shape} for extinction shape {extinction.shape}")
        if delta_scale:
            if leg is None:
                raise ValueError("legendre_moments must be provided when delta_scale is enabled")
            # manual.py:87-100 (four Greek coefficients per order): truncation fraction from the last stored a1
            norder = leg.shape[0] // 4
            f = leg[::4][-1] / (2.0 * norder + 1)
            extinction *= 1.0 - ssa * f
            ssa *= (1.0 - f) / (1.0 - ssa * f)
            for i in range(norder):
                for c in range(3):
                    leg[4 * i + c] = (leg[4 * i + c] - f * (2.0 * i + 1)) / (1.0 - f)
                leg[4 * i + 3] /= 1.0 - f
        self._extinction, self._ssa = extinction, ssa
        self._scat = extinction * ssa
        self._leg = None if leg is None else np.ascontiguousarray(leg.transpose(1, 2, 0))    # [altitude, wavelength, moment]
        for a in (self._extinction, self._ssa, self._scat, self._leg):
            if a is not None:
                a.setflags(write=False)

    extinction = property(lambda self: self._extinction)
    ssa = property(lambda self: self._ssa)
    leg_coeff = property(lambda self: None if self._leg is None else self._leg.transpose(2, 0, 1))

    def add_to_atmosphere(self, atmo):
        if self._extinction.shape[1] != atmo.num_wavel:
            raise ValueError(f"Manual constituent wavelength dimension does not match the atmosphere: {self._extinction.shape[1]} != {atmo.num_wavel}")
        if self._extinction.shape[0] != atmo.num_locations:
            raise ValueError(f"Altitude-profile Manual constituent does not match the atmosphere: {self._extinction.shape[0]} altitudes != {atmo.num_locations}")
        if self._leg is not None and self._leg.shape[2] != atmo.num_leg:
            raise ValueError(f"Manual constituent Legendre moment dimension does not match the atmosphere: {self._leg.shape[2]} != {atmo.num_leg}")
        level = np.eye(atmo.num_locations)
        for j in range(atmo.num_locations):
            atmo.frontend.add_term(level[j], self._extinction[j], self._scat[j], None if self._leg is None else self._leg[j])

    def register_derivative(self, atmo, name):
        pass


class AirMassFactor(Constituent):
    """Adds nothing to the atmosphere; asks for the weighting function `air_mass_factor` [altitude, wavelength, los, stokes]: the derivative
    of the log radiance with respect to absorber extinction in a layer per unit of layer thickness, sign of an air mass factor
    (src/sasktran2/constituent/amf.py).  On the device this is the mapping of an absorber term with a unit spectrum and no amount."""
    _unit_spectrum = None

    def add_to_atmosphere(self, atmo):
        if self._unit_spectrum is None or len(self._unit_spectrum) != atmo.num_wavel:
            self._unit_spectrum = np.ones(atmo.num_wavel)
            self._unit_spectrum.setflags(write=False)
        self._term = atmo.frontend.add_term(np.zeros(atmo.num_locations), self._unit_spectrum)

    def register_derivative(self, atmo, name):
        factors = -1.0 / np.gradient(atmo.model_geometry.altitudes())
        factors[0] *= 2.0     # half layers at the bottom and the top (amf.py:36-40)
        factors[-1] *= 2.0
        atmo.frontend.add_mapping("air_mass_factor", [self._term], factors[np.newaxis, :], interp_dim="altitude", log_radiance_space=True)


class ThermalEmission(Constituent):
    """Planck emission of the atmosphere at its own temperature (src/sasktran2/constituent/emission.py `ThermalEmission`)."""

    def add_to_atmosphere(self, atmo):
        if atmo.wavelengths_nm is None or atmo.temperature_k is None:
            raise ValueError("ThermalEmission needs wavelengths_nm and temperature_k in the Atmosphere object")
        atmo.frontend.set_planck_emission(atmo.temperature_k, atmo.wavelengths_nm)

    def register_derivative(self, atmo, name):
        atmo.frontend.add_mapping(f"wf_{name}_temperature_k", planck=True, interp_dim="altitude", assign_name="wf_temperature_k")


def _planck(temperature_k, wavelengths_nm):
    """Planck radiance in W / (m^2 sr nm) and its temperature derivative (emission.rs:9-30)"""
    h, c, kb = 6.62607015e-34, 299792458.0, 1.380649e-23
    wl = np.asarray(wavelengths_nm, dtype=float) * 1e-9
    ex = np.exp(h * c / (wl * kb * temperature_k))
    return 2.0 * h * c ** 2 / wl ** 5 / (ex - 1.0) * 1e-9, 2.0 * h ** 2 * c ** 3 * ex / (wl ** 6 * kb * temperature_k ** 2 * (ex - 1.0) ** 2) * 1e-9


class SurfaceThermalEmission(Constituent):
    """A surface emitting emissivity x Planck radiance at its own temperature (src/sasktran2/constituent/emission.py:113-190): emissivity
    constant in wavelength (scalar) or on the atmosphere's wavelength grid.  Weighting functions `wf_<name>_temperature_k` and
    `wf_<name>_emissivity`, one value per wavelength.  What the surface holds is one number per wavelength: host arithmetic."""
    _adds_surface_emission = True

    def __init__(self, temperature_k, emissivity):
        self.temperature_k = float(temperature_k)
        self.emissivity = np.atleast_1d(np.asarray(emissivity, dtype=float))

    def _emissivity_on_grid(self, atmo):
        if atmo.wavelengths_nm is None:
            raise ValueError("SurfaceThermalEmission needs wavelengths_nm in the Atmosphere object")
        if len(self.emissivity) not in (1, atmo.num_wavel):
            raise ValueError("the emissivity is a scalar or has one value per wavelength of the atmosphere")
        return np.broadcast_to(self.emissivity, (atmo.num_wavel,))

    def add_to_atmosphere(self, atmo):
        atmo.surface.emission[:] += self._emissivity_on_grid(atmo) * _planck(self.temperature_k, atmo.wavelengths_nm)[0]

    def register_derivative(self, atmo, name):
        b, db = _planck(self.temperature_k, atmo.wavelengths_nm)
        atmo.register_surface_mapping(f"wf_{name}_temperature_k", None, "dummy", d_brdf=0.0, d_emission=self._emissivity_on_grid(atmo) * db)
        atmo.register_surface_mapping(f"wf_{name}_emissivity", None, "dummy", d_brdf=0.0, d_emission=b)


class LambertianSurface(Constituent):
    """Albedo constant in wavelength (scalar), on the atmosphere's own wavelength grid, or interpolated from `wavelengths_nm` /
    `wavenumbers_cminv` (src/sasktran2/constituent/brdf/lambertiansurface.py)."""

    def __init__(self, albedo, wavelengths_nm=None, wavenumbers_cminv=None, out_of_bounds_mode="zero"):
        self.albedo = np.atleast_1d(np.asarray(albedo, dtype=float))
        self._mode = out_of_bounds_mode.lower()
        if wavelengths_nm is not None:
            self._x, self._var = np.atleast_1d(np.asarray(wavelengths_nm, dtype=float)), "wavelengths_nm"
        elif wavenumbers_cminv is not None:
            self._x, self._var = np.atleast_1d(np.asarray(wavenumbers_cminv, dtype=float)), "wavenumbers_cminv"
        else:
            self._x, self._var = None, ("constant" if len(self.albedo) == 1 else "native")

    def _interpolating_matrix(self, atmo):
        if self._var == "constant":
            return np.ones((atmo.num_wavel, 1))
        if self._var == "native":
            if len(self.albedo) != atmo.num_wavel:
                raise ValueError("The number of albedo values must match the number of wavelengths in the atmosphere")
            return np.eye(atmo.num_wavel)
        grid = getattr(atmo, self._var)
        if grid is None:
            raise ValueError(f"The atmosphere does not have {self._var} defined, cannot interpolate the user albedo")
        return linear_interpolating_matrix(self._x, grid, self._mode)

    def add_to_atmosphere(self, atmo):
        if atmo.surface.brdf != "lambertian":
            atmo.surface.set_brdf("lambertian")
        atmo.surface.albedo[:] = self._interpolating_matrix(atmo) @ self.albedo

    def register_derivative(self, atmo, name):
        atmo.register_surface_mapping(f"wf_{name}_albedo", self._interpolating_matrix(atmo), f"{name}_wavelength")


class _WavelengthParameters:
    """Parameters given as scalars, on the atmosphere's wavelength grid, or interpolated from `wavelengths_nm`
    (WavelengthInterpolatorMixin, src/sasktran2/constituent/brdf/__init__.py)."""

    def _setup_wavelengths(self, wavelengths_nm, out_of_bounds_mode, param_length):
        self._mode = out_of_bounds_mode.lower()
        self._x = None if wavelengths_nm is None else np.atleast_1d(np.asarray(wavelengths_nm, dtype=float))
        self._param_length = param_length

    def _interpolating_matrix(self, atmo):
        if self._x is None:
            if self._param_length == 1:
                return np.ones((atmo.num_wavel, 1))
            if self._param_length != atmo.num_wavel:
                raise ValueError("The number of surface parameter values must match the number of wavelengths in the atmosphere")
            return np.eye(atmo.num_wavel)
        if atmo.wavelengths_nm is None:
            raise ValueError("The atmosphere does not have wavelengths_nm defined, cannot interpolate the surface parameters")
        return linear_interpolating_matrix(self._x, atmo.wavelengths_nm, self._mode)


class MODIS(Constituent, _WavelengthParameters):
    """Kernel-driven BRDF of the MODIS products: isotropic + volumetric (Ross-thick) + geometric (Li-sparse-R) weights
    (src/sasktran2/constituent/brdf/modis.py; cpp/include/sasktran2/atmosphere/surface.h:236-352).  Weighting functions
    `wf_{name}_isotropic`, `wf_{name}_volumetric`, `wf_{name}_geometric`."""

    def __init__(self, isotropic, volumetric=0.0, geometric=0.0, wavelengths_nm=None, out_of_bounds_mode="zero"):
        self.isotropic = np.atleast_1d(np.asarray(isotropic, dtype=float))
        self.volumetric = np.atleast_1d(np.asarray(volumetric, dtype=float))
        self.geometric = np.atleast_1d(np.asarray(geometric, dtype=float))
        self._setup_wavelengths(wavelengths_nm, out_of_bounds_mode, len(self.isotropic))

    def add_to_atmosphere(self, atmo):
        atmo.surface.set_brdf("modis")
        m = self._interpolating_matrix(atmo)
        for a, p in enumerate((self.isotropic, self.volumetric, self.geometric)):
            atmo.surface.brdf_args[a, :] = m @ (p if len(p) == m.shape[1] else np.broadcast_to(p, (m.shape[1],)))

    def register_derivative(self, atmo, name):
        m = self._interpolating_matrix(atmo)
        for a, part in enumerate(("isotropic", "volumetric", "geometric")):
            d = np.zeros((atmo.num_wavel, 3))
            d[:, a] = 1.0
            atmo.register_surface_mapping(f"wf_{name}_{part}", m, f"{name}_wavelength", d_brdf=d)


class SnowKokhanovsky(Constituent, _WavelengthParameters):
    """Snow BRDF of Kokhanovsky with grain parameters L and M (src/sasktran2/constituent/brdf/kokhanovsky.py; surface.h:151-229): the BRDF
    argument is (chi + M) L / wavelength_nm with chi the imaginary part of the refractive index of ice.  The reference reads chi from its ice
    data base, which this build does not carry: `refractive_index_fn(wavelengths_nm)` has to return the complex refractive index (imaginary
    part negative, as the reference's `RefractiveIndex.refractive_index`).  Weighting functions `wf_{name}_L`, `wf_{name}_M`."""

    def __init__(self, L=3_600_000.0, M=5.5e-8, refractive_index_fn=None, wavelengths_nm=None, out_of_bounds_mode="zero"):
        if refractive_index_fn is None:
            raise ValueError("SnowKokhanovsky: pass refractive_index_fn (the reference's ice refractive-index table is not part of this build)")
        self.L = np.atleast_1d(np.asarray(L, dtype=float))
        self.M = np.atleast_1d(np.asarray(M, dtype=float))
        self._n = refractive_index_fn.refractive_index if hasattr(refractive_index_fn, "refractive_index") else refractive_index_fn
        self._setup_wavelengths(wavelengths_nm, out_of_bounds_mode, len(self.L))

    def _factors(self, atmo):
        if atmo.wavelengths_nm is None:
            raise ValueError("Atmosphere must have wavelengths defined before using SnowKokhanovsky")
        m = self._interpolating_matrix(atmo)
        chi = -np.imag(np.asarray(self._n(atmo.wavelengths_nm)))
        expand = lambda p: m @ (p if len(p) == m.shape[1] else np.broadcast_to(p, (m.shape[1],)))  # noqa: E731
        return m, chi, expand(self.L), expand(self.M)

    def add_to_atmosphere(self, atmo):
        atmo.surface.set_brdf("kokhanovsky")
        _, chi, L_, M_ = self._factors(atmo)
        atmo.surface.brdf_args[0, :] = (chi + M_) * L_ / atmo.wavelengths_nm

    def register_derivative(self, atmo, name):
        m, chi, L_, M_ = self._factors(atmo)
        atmo.register_surface_mapping(f"wf_{name}_L", m, f"{name}_wavelength", d_brdf=(chi + M_) / atmo.wavelengths_nm)
        atmo.register_surface_mapping(f"wf_{name}_M", m, f"{name}_wavelength", d_brdf=L_ / atmo.wavelengths_nm)

