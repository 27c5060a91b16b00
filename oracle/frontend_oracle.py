"""CPU restatement (numpy) of the reference's optical-property front end -- SURVEY section 8(f) row 4: constituents -> (total extinction,
single scatter albedo, Legendre coefficients, emission) and the derivative mappings that turn the engine's native weighting functions
into weighting functions of the constituents' own parameters.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by the `cpu_baseline` leg of bench_frontend.py; the product
(sasktran2_amd.constituent / sasktran2_amd.optical -> frontend.hip) never touches it.

Every array is materialised [location, wavelength] exactly the way the reference does it, one constituent after the other, so the
floating-point order of the sums is the reference's; the device assembler (one pass, every term summed in registers) is compared with a
relative tolerance, not bit for bit.

Reference (behaviour restated, nothing copied):
  src/sasktran2/atmosphere.py:719-857                               Atmosphere.internal_object: add -> normalise -> register derivatives
  rust/sasktran2-rs/src/bindings/atmosphere_storage.rs:113-153      normalize_by_extinctions
  src/sasktran2/util/state.py:72-114                                ideal-gas number density and its derivatives
  rust/sasktran2-rs/src/optical/rayleigh.rs:1-120                   Bates (1984) Rayleigh cross section and King factor
  rust/sasktran2-rs/src/constituent/types/rayleigh.rs:110-258       Rayleigh constituent
  rust/sasktran2-rs/src/constituent/types/vmr_alt_absorber.rs:14-260, 339-399   VMRAltitudeAbsorber
  rust/sasktran2-rs/src/constituent/types/number_density_scatterer.rs:218-440   NumberDensityScatterer (density derivative)
  src/sasktran2/constituent/numdenscatterer.py:117-197              ExtinctionScatterer
  src/sasktran2/constituent/gaussianheight.py:108-232               GaussianHeightExtinction
  src/sasktran2/constituent/emission.py:113-190                     SurfaceThermalEmission
  src/sasktran2/constituent/amf.py:24-50                            AirMassFactor
  rust/sasktran2-rs/src/constituent/types/manual.rs:25-111          Manual
  rust/sasktran2-rs/src/constituent/types/emission.rs:9-116         ThermalEmission
  rust/sasktran2-rs/src/optical/types/xsec_dbase.rs:66-210          cross-section tables: linear in wavenumber (zero outside), linear in the
                                                                    parameter (held outside)
  rust/sasktran2-rs/src/optical/types/scat_dbase.rs:120-290         Henyey-Greenstein table: moments (2l+1) g^l AT THE TABLE NODES, then
                                                                    interpolated in wavenumber
  rust/sasktran2-rs/src/interpolation/linear.rs:4-140, grid1d.rs:99-135   interpolation weights
  src/sasktran2/engine.py:340-372                                   output assembly (mappings with the same assign_name add up)

Pin: the reference holds no stored numbers for this row; its own tests pin it by identities this module is tested against
(tests/test_frontend_oracle.py): analytic mappings equal finite differences of the assembled storage (tests/constituent/
test_vmr_absorber_2d.py:266-317, test_scatterer_2d.py:257-293), weighting functions equal finite differences of the radiance
(tests/weightingfunctions/test_pressure_temperature.py), Henyey-Greenstein moments (tests/optical/test_henyey.py:43-49), the interpolation
matrices of rust/sasktran2-rs/src/interpolation/linear.rs:158-176.
"""
from __future__ import annotations

import numpy as np

K_BOLTZMANN = 1.380649e-23
PLANCK = 6.62607015e-34
SPEED_OF_LIGHT = 299792458.0


# ---------------------------------------------------------------------------------------------------------------- interpolation
def interp_weights(grid, x, mode):
    """[(index, weight, d weight / dx)] * 2 of linear interpolation of `grid` at x (linear.rs:93-140)."""
    n = len(grid)
    if x < grid[0]:
        return ((0, 0.0, 0.0), (0, 0.0, 0.0)) if mode == "zero" else ((0, 1.0, 0.0), (0, 0.0, 0.0))
    if x > grid[-1]:
        return ((0, 0.0, 0.0), (0, 0.0, 0.0)) if mode == "zero" else ((n - 1, 1.0, 0.0), (n - 1, 0.0, 0.0))
    idx = int(np.searchsorted(grid, x, side="left"))   # first element >= x
    if idx == 0:
        return ((0, 1.0, 0.0), (0, 0.0, 0.0))
    w = (x - grid[idx - 1]) / (grid[idx] - grid[idx - 1])
    dw = 1.0 / (grid[idx] - grid[idx - 1])
    return ((idx - 1, 1.0 - w, -dw), (idx, w, dw))


def interp_matrix(from_grid, to_grid, mode):
    """linear_interpolating_matrix (linear.rs:4-25): [len(to), len(from)]."""
    m = np.zeros((len(to_grid), len(from_grid)))
    for i, x in enumerate(to_grid):
        for idx, w, _ in interp_weights(from_grid, x, mode):
            m[i, idx] += w
    return m


# ---------------------------------------------------------------------------------------------------------------- optical properties
def _o2_refrac(wl):
    out = np.zeros_like(wl)
    for (a, b), (lo, hi) in (((23796.7, 168988.4), (0.0, 0.221)), ((22120.4, 203187.6), (0.221, 0.288)),
                             ((20564.8, 248089.9), (0.288, 0.546)), ((21351.1, 218567.0), (0.546, np.inf))):
        sel = (wl > lo) & (wl <= hi)
        out[sel] = a + b / (40.9 - wl[sel] ** -2.0)
    return out


def _n2_refrac(wl):
    out = np.zeros_like(wl)
    for (a, b), (lo, hi) in (((6998.749, 3233582.0), (0.0, 0.254)), ((5989.242, 3363266.3), (0.254, 0.468)),
                             ((6855.200, 3243157.0), (0.468, np.inf))):
        sel = (wl > lo) & (wl <= hi)
        d = 0.468 - wl[sel]
        out[sel] = a + b / (144.0 - wl[sel] ** -2.0) + 2.27684009 * np.sign(d) * np.exp(-np.abs(d) / 0.003)
    return out


def rayleigh_bates(wavelength_um, n2=78.084, o2=20.946, ar=0.934, co2=0.036):
    """(cross section [m^2], effective King factor) -- optical/rayleigh.rs:90-120."""
    wl = np.atleast_1d(np.asarray(wavelength_um, dtype=float))
    ar_n = (np.sqrt(5.547e-4 * (1.0 + 5.15e-3 * wl ** -2.0 + 4.19e-5 * wl ** -4.0) + 1.0) - 1.0) * 1.0e8
    co2_n = 22822.1 + 117.8 * wl ** -2.0 + 2406030.0 / (130.0 - wl ** -2.0) + 15997.0 / (38.9 - wl ** -2.0)
    o2_k = 1.096 + 1.385e-3 * wl ** -2.0 + 1.448e-4 * wl ** -4.0
    n2_k = 1.034 + 3.17e-4 * wl ** -2.0
    lorenz = (o2 / 100.0 * _o2_refrac(wl) ** 2 * o2_k + n2 / 100.0 * _n2_refrac(wl) ** 2 * n2_k + ar / 100.0 * ar_n ** 2 * 1.0
              + co2 / 100.0 * co2_n ** 2 * 1.15)
    king = o2 / 100.0 * o2_k + n2 / 100.0 * n2_k + ar / 100.0 * 1.0 + co2 / 100.0 * 1.15
    loschmidt = 2.686_780_111_798_444e25
    return 32.0 * np.pi ** 3 / (3.0 * loschmidt ** 2 * wl ** 4) * lorenz * 1e8, king


class AbsorberTable:
    """Cross sections [m^2] on a wavenumber grid, optionally at several temperatures (xsec_dbase.rs Ix1 / Ix2 with temperature_k)."""

    def __init__(self, wavenumber_cminv, xs, temperature_k=None):
        self.wvnum = np.asarray(wavenumber_cminv, dtype=float)
        self.xs = np.asarray(xs, dtype=float)
        self.temperature = None if temperature_k is None else np.asarray(temperature_k, dtype=float)
        assert self.xs.shape == ((len(self.wvnum),) if self.temperature is None else (len(self.temperature), len(self.wvnum)))

    def _spectral(self, row, wavenumbers):
        out = np.zeros(len(wavenumbers))
        for j, wv in enumerate(wavenumbers):
            (i0, w0, _), (i1, w1, _) = interp_weights(self.wvnum, wv, "zero")
            out[j] = row[i0] * w0 + row[i1] * w1
        return out

    def cross_section(self, wavenumbers, temperature_k):
        """xs [location, wavelength] and d xs / dT (or None)."""
        nloc = len(temperature_k)
        if self.temperature is None:
            return np.tile(self._spectral(self.xs, wavenumbers), (nloc, 1)), None
        rows = [self._spectral(r, wavenumbers) for r in self.xs]
        xs = np.zeros((nloc, len(wavenumbers)))
        dxs = np.zeros_like(xs)
        for q, t in enumerate(temperature_k):
            for i, w, dw in interp_weights(self.temperature, t, "extend"):
                xs[q] += rows[i] * w
                dxs[q] += rows[i] * dw
        return xs, dxs


class AbsorberTable2:
    """Cross sections on a wavenumber grid at (pressure, temperature) nodes: bilinear in the two parameters, held outside (xsec_dbase.rs, Ix3)."""

    def __init__(self, wavenumber_cminv, pressure_pa, temperature_k, xs):
        self.wvnum, self.pressure, self.temperature = (np.asarray(v, dtype=float) for v in (wavenumber_cminv, pressure_pa, temperature_k))
        self.xs = np.asarray(xs, dtype=float)
        assert self.xs.shape == (len(self.pressure), len(self.temperature), len(self.wvnum))

    def cross_section(self, wavenumbers, temperature_k, pressure_pa):
        """xs, d xs / dP, d xs / dT  [location, wavelength]"""
        one = AbsorberTable(self.wvnum, self.xs[0, 0])
        rows = [[one._spectral(self.xs[i, j], wavenumbers) for j in range(len(self.temperature))] for i in range(len(self.pressure))]
        nloc = len(temperature_k)
        xs, dp, dt = (np.zeros((nloc, len(wavenumbers))) for _ in range(3))
        for q in range(nloc):
            for i, wi, dwi in interp_weights(self.pressure, pressure_pa[q], "extend"):
                for j, wj, dwj in interp_weights(self.temperature, temperature_k[q], "extend"):
                    xs[q] += rows[i][j] * wi * wj
                    dp[q] += rows[i][j] * dwi * wj
                    dt[q] += rows[i][j] * wi * dwj
        return xs, dp, dt


class HenyeyGreensteinTable:
    """Total cross section, scattering cross section (the reference keeps it in the `ssa` container) and asymmetry parameter on a
    wavelength grid; moments (2l+1) g^l are formed at the nodes and interpolated in wavenumber (scat_dbase.rs:120-157, 161-212)."""

    def __init__(self, wavelength_nm, xs_total, xs_scattering, g, max_num_moments=128):
        wvnum = 1e7 / np.asarray(wavelength_nm, dtype=float)
        order = np.argsort(wvnum)
        self.wvnum = wvnum[order]
        self.xs = np.asarray(xs_total, dtype=float)[order]
        self.scat = np.asarray(xs_scattering, dtype=float)[order]
        g = np.asarray(g, dtype=float)[order]
        self.moments = np.stack([(2.0 * l + 1.0) * g ** l for l in range(max_num_moments)], axis=1)   # [node, l]

    def quantities(self, wavenumbers, nmoments, nstokes):
        """(xs [w], scattering xs [w], legendre [w, first dimension of leg_coeff]) -- altitude independent."""
        per = 1 if nstokes == 1 else 4
        norder = min(self.moments.shape[1], nmoments)
        xs = np.zeros(len(wavenumbers))
        sc = np.zeros(len(wavenumbers))
        leg = np.zeros((len(wavenumbers), nmoments * per))
        for j, wv in enumerate(wavenumbers):
            (i0, w0, _), (i1, w1, _) = interp_weights(self.wvnum, wv, "zero")
            xs[j] = self.xs[i0] * w0 + self.xs[i1] * w1
            sc[j] = self.scat[i0] * w0 + self.scat[i1] * w1
            leg[j, 0:norder * per:per] = self.moments[i0, :norder] * w0 + self.moments[i1, :norder] * w1
        return xs, sc, leg


# ---------------------------------------------------------------------------------------------------------------- atmosphere
class Mapping:
    def __init__(self, nloc, nwavel, nleg, scatterer=False):
        self.d_extinction = np.zeros((nloc, nwavel))
        self.d_ssa = np.zeros((nloc, nwavel))
        self.d_emission = np.zeros((nloc, nwavel))
        self.scat_factor = np.zeros((nloc, nwavel)) if scatterer else None
        self.d_legendre = np.zeros((nleg, nloc, nwavel)) if scatterer else None
        self.interpolator = None       # [nloc, nout]
        self.interp_dim = "altitude"
        self.assign_name = ""
        self.log_radiance_space = False    # the output is divided by the radiance (cpp/lib/output/outputc.cpp:153-162)


class Atmosphere:
    """State + storage of one assembly (the Python reference object reduced to what the constituents read and write)."""

    def __init__(self, altitudes_m, wavelengths_nm, nmoments, nstokes=1, pressure_pa=None, temperature_k=None,
                 pressure_derivative=True, temperature_derivative=True):
        self.altitudes = np.asarray(altitudes_m, dtype=float)
        self.wavelengths = np.asarray(wavelengths_nm, dtype=float)
        self.wavenumbers = 1e7 / self.wavelengths
        self.nstokes = nstokes
        self.nmoments = nmoments
        self.nleg = nmoments * (1 if nstokes == 1 else 4)
        self.pressure = None if pressure_pa is None else np.asarray(pressure_pa, dtype=float)
        self.temperature = None if temperature_k is None else np.asarray(temperature_k, dtype=float)
        self.pressure_derivative = pressure_derivative
        self.temperature_derivative = temperature_derivative
        nloc, nw = len(self.altitudes), len(self.wavelengths)
        self.ext = np.zeros((nloc, nw))
        self.ssa = np.zeros((nloc, nw))
        self.leg = np.zeros((self.nleg, nloc, nw))
        self.emission = np.zeros((nloc, nw))
        self.mappings = {}
        self.constituents = {}

    def air(self):
        """state.py:72-92"""
        return {"N": self.pressure / (K_BOLTZMANN * self.temperature), "dN_dP": 1 / (K_BOLTZMANN * self.temperature),
                "dN_dT": -self.pressure / (K_BOLTZMANN * self.temperature ** 2)}

    def mapping(self, name, scatterer=False):
        if name not in self.mappings:
            self.mappings[name] = Mapping(len(self.altitudes), len(self.wavelengths), self.nleg, scatterer)
        return self.mappings[name]

    def assemble(self, derivatives=True):
        """atmosphere.py:730-771"""
        self.ext[:] = 0
        self.ssa[:] = 0
        self.leg[:] = 0
        self.emission[:] = 0
        self.mappings = {}
        for c in self.constituents.values():
            c.add(self)
        # normalize_by_extinctions (atmosphere_storage.rs:113-153)
        self.leg /= np.where(self.ssa == 0.0, 1.0, self.ssa)[np.newaxis]
        positive = self.ext > 0.0
        self.ssa = np.where(positive, np.minimum(self.ssa / np.where(positive, self.ext, 1.0), 1.0), 0.0)
        if derivatives:
            for name, c in self.constituents.items():
                c.register(self, name)
        return self


def rayleigh_moment(atmo, factor):
    """spectrum of every coefficient of the Rayleigh phase matrix in the storage's first dimension (rayleigh.rs:141-160)"""
    vals = np.zeros((atmo.nleg, len(factor)))
    if atmo.nstokes == 1:
        vals[0] = 1.0
        vals[2] = factor
    else:
        vals[0] = 1.0
        vals[8] = factor
        vals[9] = 6.0 * factor
        vals[11] = np.sqrt(6.0) * factor
    return vals


class Rayleigh:
    def __init__(self, method="bates", n2=78.084, o2=20.946, ar=0.934, co2=0.036, wavelengths_nm=None, xs=None, king_factor=None):
        self.method, self.mix = method, (n2, o2, ar, co2)
        self.table = (wavelengths_nm, xs, king_factor)

    def cross_section(self, wavelengths_nm):
        if self.method == "bates":
            return rayleigh_bates(wavelengths_nm / 1000.0, *self.mix)
        wl, xs, king = self.table      # held outside the table (rayleigh.rs:92-105)
        return np.interp(wavelengths_nm, wl, xs), np.interp(wavelengths_nm, wl, king)

    def add(self, atmo):
        sigma, king = self.cross_section(atmo.wavelengths)
        delta = 6.0 * (king - 1.0) / (3.0 + 7.0 * king)
        ext = sigma[np.newaxis, :] * atmo.air()["N"][:, np.newaxis]
        atmo.ext += ext
        atmo.ssa += ext
        atmo.leg += rayleigh_moment(atmo, (1.0 - delta) / (2.0 + delta))[:, np.newaxis, :] * ext[np.newaxis]

    def register(self, atmo, name):
        sigma, king = self.cross_section(atmo.wavelengths)
        delta = 6.0 * (king - 1.0) / (3.0 + 7.0 * king)
        air = atmo.air()
        for key, flag, factor in (("pressure_pa", atmo.pressure_derivative, air["dN_dP"]), ("temperature_k", atmo.temperature_derivative, air["dN_dT"])):
            if not flag:
                continue
            m = atmo.mapping(f"wf_{name}_{key}", scatterer=True)
            m.d_extinction += sigma[np.newaxis, :]
            m.d_ssa += sigma[np.newaxis, :] * (1.0 - atmo.ssa) / atmo.ext
            m.d_legendre += rayleigh_moment(atmo, (1.0 - delta) / (2.0 + delta))[:, np.newaxis, :] - atmo.leg
            m.scat_factor += sigma[np.newaxis, :] / (atmo.ssa * atmo.ext)
            m.interp_dim = "altitude"
            m.interpolator = np.diag(factor)
            m.assign_name = f"wf_{key}"


def _absorber_mapping(m, d_ext, atmo):
    """assign_absorber_derivatives (vmr_alt_absorber.rs:14-62) for a pure absorber (species ssa 0)"""
    m.d_extinction[:] = d_ext
    with np.errstate(divide="ignore", invalid="ignore"):
        m.d_ssa[:] = np.where(atmo.ext == 0.0, 0.0, (0.0 - atmo.ssa) * d_ext / atmo.ext)


class VMRAltitudeAbsorber:
    def __init__(self, table, altitudes_m, vmr, out_of_bounds_mode="zero"):
        self.table, self.alts, self.vmr, self.mode = table, np.asarray(altitudes_m, float), np.asarray(vmr, float), out_of_bounds_mode

    def _xs(self, atmo):
        """cross section and its derivatives with respect to the state variables the table depends on"""
        if isinstance(self.table, AbsorberTable2):
            xs, dp, dt = self.table.cross_section(atmo.wavenumbers, atmo.temperature, atmo.pressure)
            return xs, {"pressure_pa": dp, "temperature_k": dt}
        xs, dt = self.table.cross_section(atmo.wavenumbers, atmo.temperature)
        return xs, ({} if dt is None else {"temperature_k": dt})

    def add(self, atmo):
        m = interp_matrix(self.alts, atmo.altitudes, self.mode)
        vmr = m @ self.vmr
        xs, _ = self._xs(atmo)
        atmo.ext += xs * vmr[:, np.newaxis] * atmo.air()["N"][:, np.newaxis]

    def register(self, atmo, name):
        m = interp_matrix(self.alts, atmo.altitudes, self.mode)
        vmr = m @ self.vmr
        air = atmo.air()
        xs, dxs = self._xs(atmo)
        mp = atmo.mapping(f"wf_{name}_vmr")
        _absorber_mapping(mp, xs, atmo)
        mp.interp_dim = f"{name}_altitude"
        mp.interpolator = m * air["N"][:, np.newaxis]
        state = []
        if atmo.pressure_derivative:
            state.append(("pressure_pa", air["dN_dP"]))
        if atmo.temperature_derivative:
            state.append(("temperature_k", air["dN_dT"]))
        for key, factor in state:
            mp = atmo.mapping(f"wf_{name}_{key}")
            _absorber_mapping(mp, xs, atmo)
            mp.interp_dim = "altitude"
            mp.assign_name = f"wf_{key}"
            mp.interpolator = np.diag(vmr * factor)
        if state:
            for key, d in dxs.items():   # vmr_alt_absorber.rs:233-256: one mapping per parameter of the table
                mp = atmo.mapping(f"wf_{name}_{key}_xs")
                _absorber_mapping(mp, d, atmo)
                mp.interp_dim = "altitude"
                mp.assign_name = f"wf_{key}"
                mp.interpolator = np.diag(vmr * air["N"])


class CollisionInducedAbsorber:
    """collisioninducedabsorber.py:60-126: extinction = xs [m^5] x fraction product x N^2; state derivatives through N and through the table"""

    def __init__(self, table, fraction_product=0.20964 ** 2):
        self.table, self.fp = table, fraction_product

    def _xs(self, atmo):
        xs, dt = self.table.cross_section(atmo.wavenumbers, atmo.temperature)
        return xs, ({} if dt is None else {"temperature_k": dt})

    def add(self, atmo):
        xs, _ = self._xs(atmo)
        atmo.ext += xs * (self.fp * atmo.air()["N"] ** 2)[:, np.newaxis]

    def register(self, atmo, name):
        air = atmo.air()
        xs, dxs = self._xs(atmo)
        state = []
        if atmo.pressure_derivative:
            state.append(("pressure_pa", air["dN_dP"]))
        if atmo.temperature_derivative:
            state.append(("temperature_k", air["dN_dT"]))
        for key, dn in state:
            mp = atmo.mapping(f"wf_{name}_{key}")
            _absorber_mapping(mp, 2 * self.fp * (air["N"] * dn)[:, np.newaxis] * xs, atmo)
            mp.interp_dim, mp.assign_name = "altitude", f"wf_{key}"
        if state:
            for key, d in dxs.items():
                mp = atmo.mapping(f"wf_{name}_{key}_xs")
                _absorber_mapping(mp, d * (self.fp * air["N"] ** 2)[:, np.newaxis], atmo)
                mp.interp_dim, mp.assign_name = "altitude", f"wf_{key}"


class NumberDensityScatterer:
    wf_name = "number_density"

    def __init__(self, table, altitudes_m, number_density, out_of_bounds_mode="zero"):
        self.table, self.alts, self.mode = table, np.asarray(altitudes_m, float), out_of_bounds_mode
        self.number_density = np.asarray(number_density, float)
        self.vertical_factor = np.ones_like(self.number_density)

    def add(self, atmo):
        nd = interp_matrix(self.alts, atmo.altitudes, self.mode) @ self.number_density
        xs, sc, leg = self.table.quantities(atmo.wavenumbers, atmo.nmoments, atmo.nstokes)
        atmo.ext += xs[np.newaxis, :] * nd[:, np.newaxis]
        atmo.ssa += sc[np.newaxis, :] * nd[:, np.newaxis]
        atmo.leg += (sc[np.newaxis, :] * nd[:, np.newaxis])[np.newaxis] * leg.T[:, np.newaxis, :]

    def register(self, atmo, name):
        m = interp_matrix(self.alts, atmo.altitudes, self.mode)
        xs, sc, leg = self.table.quantities(atmo.wavenumbers, atmo.nmoments, atmo.nstokes)
        with np.errstate(divide="ignore", invalid="ignore"):
            species_ssa = sc / xs
        species_ssa[~np.isfinite(species_ssa)] = 1.0
        mp = atmo.mapping(f"wf_{name}_{self.wf_name}", scatterer=True)
        mp.d_extinction += xs[np.newaxis, :]
        mp.scat_factor += (species_ssa * xs)[np.newaxis, :] / (atmo.ssa * atmo.ext)
        mp.d_ssa += xs[np.newaxis, :] * (species_ssa[np.newaxis, :] - atmo.ssa) / atmo.ext
        mp.d_legendre += leg.T[:, np.newaxis, :] - atmo.leg
        mp.interpolator = m * self.vertical_factor[np.newaxis, :]
        mp.interp_dim = f"{name}_altitude"


class ExtinctionScatterer(NumberDensityScatterer):
    """numdenscatterer.py:117-197: an extinction profile at one wavelength, turned into a number density with the table's cross section there"""
    wf_name = "extinction"

    def __init__(self, table, altitudes_m, extinction_per_m, extinction_wavelength_nm, out_of_bounds_mode="zero"):
        super().__init__(table, altitudes_m, np.zeros(len(altitudes_m)), out_of_bounds_mode)
        self.extinction = np.asarray(extinction_per_m, float)
        self.reference_wavelength = float(extinction_wavelength_nm)

    def _update(self):
        xs_ref = self.table.quantities(np.array([1e7 / self.reference_wavelength]), 1, 1)[0][0]
        self.vertical_factor = np.full(len(self.alts), 1.0 / xs_ref)
        self.number_density = self.extinction / xs_ref

    def add(self, atmo):
        self._update()
        super().add(atmo)

    def register(self, atmo, name):
        self._update()
        super().register(atmo, name)


class GaussianHeightExtinction(NumberDensityScatterer):
    """src/sasktran2/constituent/gaussianheight.py:108-232: number density = gaussian(height, FWHM) x vertical optical depth /
    (trapezoid integral of the gaussian x cross section at the reference wavelength) on the constituent's own altitude grid; one
    scatterer mapping per parameter (height_m, width_fwhm_m, vertical_optical_depth), each with a one-column interpolator."""

    def __init__(self, table, height_m, width_fwhm_m, vertical_optical_depth, vertical_optical_depth_wavel_nm, altitudes_m, out_of_bounds_mode="zero"):
        super().__init__(table, altitudes_m, np.zeros(len(altitudes_m)), out_of_bounds_mode)
        self.height_m, self.width_fwhm_m, self.vertical_optical_depth = float(height_m), float(width_fwhm_m), float(vertical_optical_depth)
        self.reference_wavelength = float(vertical_optical_depth_wavel_nm)

    def _update(self):
        a = self.alts
        self.xs_ref = self.table.quantities(np.array([1e7 / self.reference_wavelength]), 1, 1)[0][0]
        self.gaussian = np.exp(-4.0 * np.log(2.0) * (a - self.height_m) ** 2 / self.width_fwhm_m ** 2)
        self.gaussian_od = np.trapezoid(self.gaussian, a)
        self.number_density = self.gaussian * self.vertical_optical_depth / self.gaussian_od / self.xs_ref

    def add(self, atmo):
        self._update()
        super().add(atmo)

    def register(self, atmo, name):
        self._update()
        a, g = self.alts, self.gaussian
        m = interp_matrix(self.alts, atmo.altitudes, self.mode)
        xs, sc, leg = self.table.quantities(atmo.wavenumbers, atmo.nmoments, atmo.nstokes)
        with np.errstate(divide="ignore", invalid="ignore"):
            species_ssa = sc / xs
        species_ssa[~np.isfinite(species_ssa)] = 1.0
        dg_dh = g * 8.0 * np.log(2.0) * (a - self.height_m) / self.width_fwhm_m ** 2
        dg_dw = g * 8.0 * np.log(2.0) * (a - self.height_m) ** 2 / self.width_fwhm_m ** 3
        outer = self.vertical_optical_depth / self.gaussian_od / self.xs_ref
        columns = {"height_m": outer * (dg_dh - g * np.trapezoid(dg_dh, a) / self.gaussian_od),
                   "width_fwhm_m": outer * (dg_dw - g * np.trapezoid(dg_dw, a) / self.gaussian_od),
                   "vertical_optical_depth": g / self.gaussian_od / self.xs_ref}
        for key, column in columns.items():
            mp = atmo.mapping(f"wf_{name}_{key}", scatterer=True)
            mp.d_extinction += xs[np.newaxis, :]
            mp.scat_factor += (species_ssa * xs)[np.newaxis, :] / (atmo.ssa * atmo.ext)
            mp.d_ssa += xs[np.newaxis, :] * (species_ssa[np.newaxis, :] - atmo.ssa) / atmo.ext
            mp.d_legendre += leg.T[:, np.newaxis, :] - atmo.leg
            mp.interpolator = m @ column[:, np.newaxis]
            mp.interp_dim = f"{name}_{key}"


class Manual:
    """rust/sasktran2-rs/src/constituent/types/manual.rs:25-111: extinction and single scatter albedo [location, wavelength] (and phase
    moments [moment, location, wavelength]) given on the model grid, added like any other constituent; no derivatives"""

    def __init__(self, extinction, ssa, legendre_moments=None):
        self.extinction, self.ssa = np.asarray(extinction, float), np.asarray(ssa, float)
        self.legendre = None if legendre_moments is None else np.asarray(legendre_moments, float)

    def add(self, atmo):
        atmo.ext += self.extinction
        atmo.ssa += self.extinction * self.ssa
        if self.legendre is not None:
            atmo.leg += (self.extinction * self.ssa)[np.newaxis] * self.legendre

    def register(self, atmo, name):
        pass


class AirMassFactor:
    """src/sasktran2/constituent/amf.py:24-50: adds nothing to the atmosphere; its mapping `air_mass_factor` is the derivative of the log
    radiance with respect to absorber extinction in a layer, per unit of layer thickness, with the sign of an air mass factor"""

    def add(self, atmo):
        pass

    def register(self, atmo, name):
        f = -1.0 / np.gradient(atmo.altitudes)
        f[0] *= 2.0       # half layers at the bottom and the top
        f[-1] *= 2.0
        m = atmo.mapping("air_mass_factor")
        m.d_extinction += f[:, np.newaxis]
        m.d_ssa += f[:, np.newaxis] * (0.0 - atmo.ssa) / atmo.ext
        m.log_radiance_space = True
        m.interp_dim = "altitude"


def planck(temperature_k, wavelength_nm):
    """emission.rs:9-16: W / (m^2 sr nm)"""
    wl = wavelength_nm * 1e-9
    return (2.0 * PLANCK * SPEED_OF_LIGHT ** 2 / wl ** 5) / (np.exp(PLANCK * SPEED_OF_LIGHT / (wl * K_BOLTZMANN * temperature_k)) - 1.0) * 1e-9


def d_planck_dT(temperature_k, wavelength_nm):
    """emission.rs:18-30"""
    wl = wavelength_nm * 1e-9
    ex = np.exp(PLANCK * SPEED_OF_LIGHT / (wl * K_BOLTZMANN * temperature_k))
    return (2.0 * PLANCK ** 2 * SPEED_OF_LIGHT ** 3 * ex) / (wl ** 6 * K_BOLTZMANN * temperature_k ** 2 * (ex - 1.0) ** 2) * 1e-9


class ThermalEmission:
    def add(self, atmo):
        atmo.emission += planck(atmo.temperature[:, np.newaxis], atmo.wavelengths[np.newaxis, :])

    def register(self, atmo, name):
        m = atmo.mapping(f"wf_{name}_temperature_k")
        m.d_emission += d_planck_dT(atmo.temperature[:, np.newaxis], atmo.wavelengths[np.newaxis, :])
        m.interp_dim = "altitude"
        m.assign_name = "wf_temperature_k"


class SurfaceThermalEmission:
    """src/sasktran2/constituent/emission.py:113-190: surface emission [wavelength] += emissivity x B(T_surface); surface mappings
    wf_<name>_temperature_k (d_emission = emissivity x dB/dT) and wf_<name>_emissivity (d_emission = B), one value per wavelength
    (interp_dim "dummy" / none: src/sasktran2/engine.py:540-549)."""

    def __init__(self, temperature_k, emissivity):
        self.temperature_k, self.emissivity = float(temperature_k), np.atleast_1d(np.asarray(emissivity, dtype=float))

    def emission(self, wavelengths_nm):
        return self.emissivity * planck(self.temperature_k, wavelengths_nm)

    def surface_mappings(self, name, wavelengths_nm):
        """{mapping name: d_emission [wavelength]}"""
        return {f"wf_{name}_temperature_k": self.emissivity * d_planck_dT(self.temperature_k, wavelengths_nm),
                f"wf_{name}_emissivity": planck(self.temperature_k, wavelengths_nm)}


# ---------------------------------------------------------------------------------------------------------------- output assembly
def assemble_weighting_functions(atmo, native):
    """engine.py:340-372 / cpp/lib/output/outputc.cpp:38-236: native = {"ext", "ssa" [nloc, nw, nlos], optional "leg" [nleg, nloc, nw, nlos],
    "emission" [nloc, nw, nlos], "radiance" [nw, nlos] for log-radiance mappings} -> {output name: [nout, nw, nlos]}; mappings with the same assign_name add up."""
    out = {}
    for name, m in atmo.mappings.items():
        v = m.d_extinction[:, :, np.newaxis] * native["ext"] + m.d_ssa[:, :, np.newaxis] * native["ssa"]
        if m.scat_factor is not None and "leg" in native:
            v = v + m.scat_factor[:, :, np.newaxis] * np.einsum("lqw,lqwr->qwr", m.d_legendre, native["leg"])
        if "emission" in native:
            v = v + m.d_emission[:, :, np.newaxis] * native["emission"]
        if m.interpolator is not None:
            v = np.einsum("qj,qwr->jwr", m.interpolator, v)
        if m.log_radiance_space:
            v = v / native["radiance"][np.newaxis]
        key = m.assign_name or name
        out[key] = out[key] + v if key in out else v
    return out
