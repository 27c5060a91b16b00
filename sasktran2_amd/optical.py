"""Optical properties of the constituent interface (mirror of `sasktran2.optical` for the classes that need no data files).

An optical property here answers one question of the device front end (`frontend.hip` via `sasktran2_amd.frontend`): which SPECTRA
[wavelength] and which per-location WEIGHTS [location] make up its cross sections -- the [location x wavelength] arrays the reference
materialises on the host (`OpticalProperty.atmosphere_quantities`, src/sasktran2/optical/base.py:23-47) are formed in HBM from them.

Reference behaviour followed (nothing copied):
  rust/sasktran2-rs/src/optical/types/xsec_dbase.rs:66-210     cross-section tables: linear in wavenumber (0 outside the table), linear in
                                                               the parameter (held outside), d/d(parameter) from the weights' slopes
  src/sasktran2/optical/database.py:62-153                     OpticalDatabaseGenericAbsorber: the netCDF front door of the same tables
                                                               (absent here: no xarray / netCDF in this image; the in-memory classes
                                                               below are what it constructs, `AbsorberDatabaseDim1/Dim2`)
  src/sasktran2/optical/henyey.py:21-152, rust/.../scat_dbase.rs:120-212   HenyeyGreenstein: moments (2l+1) g^l at the table nodes,
                                                               interpolated in wavenumber; the `ssa` container holds the SCATTERING
                                                               CROSS SECTION when used by a constituent (tests/constituent/test_scatterer_2d.py:80-99)
  rust/sasktran2-rs/src/interpolation/linear.rs:93-140         interpolation weights
  rust/sasktran2-rs/src/optical/rayleigh.rs:1-120              Bates (1984) Rayleigh cross section / King factor
"""
from __future__ import annotations

import numpy as np


def interp_weights(grid, x, mode):
    """Vectorised `interp1_weights`: (i0, w0, dw0, i1, w1, dw1) with value = y[i0] w0 + y[i1] w1; mode "zero" or "extend"."""
    grid = np.asarray(grid, dtype=float)
    x = np.atleast_1d(np.asarray(x, dtype=float))
    if mode not in ("zero", "extend"):
        raise ValueError("out_of_bounds_mode must be 'zero' or 'extend'")
    n = len(grid)
    idx = np.searchsorted(grid, x, side="left")   # first element >= x
    below, above = x < grid[0], x > grid[-1]
    interior = ~below & ~above & (idx > 0)
    lo, hi = np.clip(idx - 1, 0, n - 1), np.clip(idx, 0, n - 1)
    span = np.where(interior, grid[hi] - grid[lo], 1.0)
    w = np.where(interior, (x - grid[lo]) / span, 0.0)
    dw = np.where(interior, 1.0 / span, 0.0)
    i0 = np.where(interior, lo, 0)
    i1 = np.where(interior, hi, 0)
    w0 = np.where(interior, 1.0 - w, 1.0)     # x == grid[0]: the first node alone
    w1 = w
    edge = 1.0 if mode == "extend" else 0.0
    w0 = np.where(below | above, edge, w0)
    i0 = np.where(above, n - 1, i0)
    i1 = np.where(above, n - 1, i1)
    return i0, w0, -dw, i1, w1, dw


def linear_interpolating_matrix(from_grid, to_grid, mode):
    """[len(to), len(from)] (rust/sasktran2-rs/src/interpolation/linear.rs:4-25)."""
    i0, w0, _, i1, w1, _ = interp_weights(from_grid, to_grid, mode)
    m = np.zeros((len(np.atleast_1d(to_grid)), len(from_grid)))
    rows = np.arange(m.shape[0])
    np.add.at(m, (rows, i0), w0)
    np.add.at(m, (rows, i1), w1)
    return m


def rayleigh_cross_section_bates(wavelengths_um, n2_percentage=78.084, o2_percentage=20.946, ar_percentage=0.934, co2_percentage=0.036):
    """(cross section [m^2 / molecule], effective King factor) of dry air after Bates (1984), `sasktran2.optical.rayleigh`."""
    wl = np.atleast_1d(np.asarray(wavelengths_um, dtype=float))
    inv2 = wl ** -2.0

    def piecewise(ranges, pole):
        out = np.zeros_like(wl)
        for (a, b), (lo, hi) in ranges:
            sel = (wl > lo) & (wl <= hi)
            out[sel] = a + b / (pole - inv2[sel])
        return out

    o2_n = piecewise((((23796.7, 168988.4), (0.0, 0.221)), ((22120.4, 203187.6), (0.221, 0.288)), ((20564.8, 248089.9), (0.288, 0.546)),
                      ((21351.1, 218567.0), (0.546, np.inf))), 40.9)
    d = 0.468 - wl
    n2_n = piecewise((((6998.749, 3233582.0), (0.0, 0.254)), ((5989.242, 3363266.3), (0.254, 0.468)), ((6855.200, 3243157.0), (0.468, np.inf))),
                     144.0) + 2.27684009 * np.sign(d) * np.exp(-np.abs(d) / 0.003)
    ar_n = (np.sqrt(5.547e-4 * (1.0 + 5.15e-3 * inv2 + 4.19e-5 * inv2 ** 2) + 1.0) - 1.0) * 1.0e8
    co2_n = 22822.1 + 117.8 * inv2 + 2406030.0 / (130.0 - inv2) + 15997.0 / (38.9 - inv2)
    o2_k = 1.096 + 1.385e-3 * inv2 + 1.448e-4 * inv2 ** 2
    n2_k = 1.034 + 3.17e-4 * inv2
    f = np.array([o2_percentage, n2_percentage, ar_percentage, co2_percentage]) / 100.0
    lorenz = f[0] * o2_n ** 2 * o2_k + f[1] * n2_n ** 2 * n2_k + f[2] * ar_n ** 2 + f[3] * co2_n ** 2 * 1.15
    king = f[0] * o2_k + f[1] * n2_k + f[2] + f[3] * 1.15
    loschmidt = 2.686_780_111_798_444e25
    return 32.0 * np.pi ** 3 / (3.0 * loschmidt ** 2 * wl ** 4) * lorenz * 1e8, king


class OpticalProperty:
    """A set of spectral terms: `spectral_terms(atmo)` -> [(weight [location], d weight / d parameter {name: [location]}, spectra)], with
    spectra = {"ext": [wavelength], "scat": [wavelength] or None, "leg": [wavelength, first dimension of leg_coeff] or None}."""

    is_scatterer = False

    def spectral_terms(self, atmo):
        raise NotImplementedError


def _table_spectrum(wvnum_grid, row, wavenumbers):
    i0, w0, _, i1, w1, _ = interp_weights(wvnum_grid, wavenumbers, "zero")
    return row[i0] * w0 + row[i1] * w1


class _SpectrumCache:
    """The table interpolated to the atmosphere's spectral grid, kept while that grid stays the same (a retrieval changes profiles step
    after step, not the wavelengths).  What it hands out is read-only."""

    def __init__(self):
        self._key, self._value = None, None

    def get(self, wavenumbers, build):
        if self._key is None or self._key.shape != wavenumbers.shape or not np.array_equal(self._key, wavenumbers):
            self._key, self._value = wavenumbers.copy(), _freeze(build(wavenumbers))
        return self._value


def _freeze(value):
    """read-only arrays: the front end keeps a spectrum it was handed in HBM for as long as the same read-only array comes back"""
    for a in (value.values() if isinstance(value, dict) else value if isinstance(value, tuple) else (value,)):
        if isinstance(a, np.ndarray):
            a.setflags(write=False)
    return value


def _sorted_wavenumbers(wavenumber_cminv, wavelength_nm):
    if (wavenumber_cminv is None) == (wavelength_nm is None):
        raise ValueError("give exactly one of wavenumber_cminv and wavelength_nm")
    wv = np.asarray(wavenumber_cminv if wavenumber_cminv is not None else 1e7 / np.asarray(wavelength_nm, dtype=float), dtype=float)
    order = np.argsort(wv)
    return wv[order], order


class AbsorberDatabaseDim1(OpticalProperty):
    """Absorption cross sections [m^2] on a spectral grid (the reference's `AbsorberDatabaseDim1(wavenumber_cminv, xs)`)."""

    def __init__(self, wavenumber_cminv=None, xs=None, wavelength_nm=None):
        self._wvnum, order = _sorted_wavenumbers(wavenumber_cminv, wavelength_nm)
        self._xs = np.asarray(xs, dtype=float)[order]
        if self._xs.shape != self._wvnum.shape:
            raise ValueError("xs must have one value per spectral grid point")
        self._cache = _SpectrumCache()

    def spectral_terms(self, atmo):
        ones = np.ones(atmo.num_locations)
        return [(ones, {}, {"ext": self._cache.get(atmo.wavenumbers_cminv, lambda wv: _table_spectrum(self._wvnum, self._xs, wv)), "scat": None, "leg": None})]


class AbsorberDatabaseDim2(OpticalProperty):
    """Cross sections [parameter, spectral grid] with one atmospheric parameter, by default the temperature
    (`AbsorberDatabaseDim2(wavenumber_cminv, param, xs, names)`): one spectral term per table node the profile touches."""

    def __init__(self, wavenumber_cminv=None, param=None, xs=None, names=("temperature_k",), wavelength_nm=None):
        self._wvnum, order = _sorted_wavenumbers(wavenumber_cminv, wavelength_nm)
        param = np.asarray(param, dtype=float)
        porder = np.argsort(param)
        self._param = param[porder]
        self._xs = np.asarray(xs, dtype=float)[porder][:, order]
        self._name = names[0]
        if self._xs.shape != (len(self._param), len(self._wvnum)):
            raise ValueError("xs must be [parameter, spectral grid]")
        if self._name not in ("temperature_k", "pressure_pa"):
            raise ValueError("the table parameter must be temperature_k or pressure_pa")
        self._cache = _SpectrumCache()

    def spectral_terms(self, atmo):
        values = getattr(atmo, self._name)
        if values is None:
            raise ValueError(f"the atmosphere has no {self._name}")
        i0, w0, dw0, i1, w1, dw1 = interp_weights(self._param, values, "extend")
        nloc = len(values)
        weight = np.zeros((len(self._param), nloc))
        slope = np.zeros_like(weight)
        loc = np.arange(nloc)
        np.add.at(weight, (i0, loc), w0)
        np.add.at(weight, (i1, loc), w1)
        np.add.at(slope, (i0, loc), dw0)
        np.add.at(slope, (i1, loc), dw1)
        rows = self._cache.get(atmo.wavenumbers_cminv, lambda wv: np.array([_table_spectrum(self._wvnum, r, wv) for r in self._xs]))
        out = []
        for t in range(len(self._param)):
            if not (weight[t].any() or slope[t].any()):
                continue
            out.append((weight[t], {self._name: slope[t]}, {"ext": rows[t], "scat": None, "leg": None}))
        return out


class AbsorberDatabaseDim3(OpticalProperty):
    """Cross sections [parameter 0, parameter 1, spectral grid] -- the pressure / temperature tables of line absorbers
    (`AbsorberDatabaseDim3(wavenumber_cminv, param0, param1, xs, names)`; xsec_dbase.rs, Ix3): bilinear in the two parameters (held outside),
    one spectral term per table node pair the profile touches, derivative weights for both parameters."""

    def __init__(self, wavenumber_cminv=None, param0=None, param1=None, xs=None, names=("pressure_pa", "temperature_k"), wavelength_nm=None):
        self._wvnum, order = _sorted_wavenumbers(wavenumber_cminv, wavelength_nm)
        p0, p1 = np.asarray(param0, dtype=float), np.asarray(param1, dtype=float)
        o0, o1 = np.argsort(p0), np.argsort(p1)
        self._p0, self._p1 = p0[o0], p1[o1]
        self._xs = np.asarray(xs, dtype=float)[o0][:, o1][:, :, order]
        self._names = tuple(names)
        if self._xs.shape != (len(p0), len(p1), len(self._wvnum)):
            raise ValueError("xs must be [parameter 0, parameter 1, spectral grid]")
        if len(self._names) != 2 or any(n not in ("temperature_k", "pressure_pa") for n in self._names) or self._names[0] == self._names[1]:
            raise ValueError("the table parameters must be temperature_k and pressure_pa")
        self._cache = _SpectrumCache()

    def spectral_terms(self, atmo):
        v0, v1 = getattr(atmo, self._names[0]), getattr(atmo, self._names[1])
        if v0 is None or v1 is None:
            raise ValueError("the atmosphere needs pressure_pa and temperature_k for this table")
        a = interp_weights(self._p0, v0, "extend")
        b = interp_weights(self._p1, v1, "extend")
        nloc, n0, n1 = len(v0), len(self._p0), len(self._p1)
        w = np.zeros((n0, n1, nloc))
        d0, d1 = np.zeros_like(w), np.zeros_like(w)
        loc = np.arange(nloc)
        for ia, wa, da in ((a[0], a[1], a[2]), (a[3], a[4], a[5])):
            for ib, wb, db in ((b[0], b[1], b[2]), (b[3], b[4], b[5])):
                np.add.at(w, (ia, ib, loc), wa * wb)
                np.add.at(d0, (ia, ib, loc), da * wb)
                np.add.at(d1, (ia, ib, loc), wa * db)
        rows = self._cache.get(atmo.wavenumbers_cminv, lambda wv: np.array([[_table_spectrum(self._wvnum, r, wv) for r in plane] for plane in self._xs]))
        out = []
        for i in range(n0):
            for j in range(n1):
                if w[i, j].any() or d0[i, j].any() or d1[i, j].any():
                    out.append((w[i, j], {self._names[0]: d0[i, j], self._names[1]: d1[i, j]}, {"ext": rows[i, j], "scat": None, "leg": None}))
        return out


class HenyeyGreenstein(OpticalProperty):
    """Scatterer with a Henyey-Greenstein phase function; build with `from_parameters` (src/sasktran2/optical/henyey.py:22-59).
    As in the reference's constituent pipeline `ssa` carries the scattering cross section [m^2]."""

    is_scatterer = True

    def __init__(self, wavelength_nm, xs_total, ssa, g, max_num_moments=128):
        self._wvnum, order = _sorted_wavenumbers(None, wavelength_nm)
        self._xs = np.asarray(xs_total, dtype=float)[order]
        self._scat = np.asarray(ssa, dtype=float)[order]
        g = np.asarray(g, dtype=float)[order]
        self._moments = np.stack([(2.0 * l + 1.0) * g ** l for l in range(int(max_num_moments))], axis=1)   # [node, l]
        self._cache = _SpectrumCache()

    @classmethod
    def from_parameters(cls, wavelength_nm, xs_total, ssa, g, max_num_moments=128):
        return cls(wavelength_nm, xs_total, ssa, g, max_num_moments)

    def cross_sections(self, wavelengths_nm):
        """total cross section at the given wavelengths (`OpticalProperty.cross_sections(...).extinction`)"""
        return _table_spectrum(self._wvnum, self._xs, 1e7 / np.atleast_1d(np.asarray(wavelengths_nm, dtype=float)))

    def spectral_terms(self, atmo):
        per = 1 if atmo.nstokes == 1 else 4
        nmoments = atmo.num_moments

        def build(wv):
            norder = min(self._moments.shape[1], nmoments)
            i0, w0, _, i1, w1, _ = interp_weights(self._wvnum, wv, "zero")
            leg = np.zeros((len(wv), nmoments * per))
            leg[:, 0:norder * per:per] = self._moments[i0, :norder] * w0[:, None] + self._moments[i1, :norder] * w1[:, None]
            return {"ext": self._xs[i0] * w0 + self._xs[i1] * w1, "scat": self._scat[i0] * w0 + self._scat[i1] * w1, "leg": leg, "shape": (nmoments, per)}
        spec = self._cache.get(atmo.wavenumbers_cminv, build)
        if spec["shape"] != (nmoments, per):     # the same spectral grid asked for with another moment count / Stokes dimension
            self._cache = _SpectrumCache()
            spec = self._cache.get(atmo.wavenumbers_cminv, build)
        return [(np.ones(atmo.num_locations), {}, spec)]
