"""Optical-property front end, CPU side: the numpy oracle (oracle/frontend_oracle.py) against the identities the
reference's own tests use for this row -- it stores no numbers for it -- and the product's host logic (sasktran2_amd.optical /
.constituent, which prepare spectra and profiles for the device) against the oracle.

  * analytic derivative mappings == finite differences of the assembled storage
    (reference: tests/constituent/test_vmr_absorber_2d.py:266-317, test_scatterer_2d.py:257-293, 364-400)
  * Henyey-Greenstein moments (tests/optical/test_henyey.py:43-49), interpolation matrices (rust/.../interpolation/linear.rs:158-176)
  * Rayleigh cross section: Bates / Bucholtz value at 550 nm, King factor, NSTOKES = 3 coefficients (rayleigh.rs:141-160)
"""
from __future__ import annotations

import numpy as np
import pytest

from oracle import frontend_oracle as fo
from tests import scenarios as sc


def _storage_derivative(atmo, key):
    """d(ext), d(ssa), d(leg) per unit of every output of the mappings assigned to `key`, from the analytic mapping arrays"""
    d_ext = d_ssa = d_leg = None
    for name, m in atmo.mappings.items():
        if (m.assign_name or name) != key:
            continue
        interp = m.interpolator if m.interpolator is not None else np.eye(len(atmo.altitudes))
        e = np.einsum("qj,qw->jqw", interp, m.d_extinction)
        s = np.einsum("qj,qw->jqw", interp, m.d_ssa)
        g = np.zeros((interp.shape[1],) + atmo.leg.shape)
        if m.scat_factor is not None:
            g = np.einsum("qj,lqw->jlqw", interp, m.d_legendre * m.scat_factor[np.newaxis])
        d_ext = e if d_ext is None else d_ext + e
        d_ssa = s if d_ssa is None else d_ssa + s
        d_leg = g if d_leg is None else d_leg + g
    return d_ext, d_ssa, d_leg


def _check_against_finite_difference(case_kwargs, key, perturb, j, step, rtol):
    c = sc.frontend_case(nwavel=12)
    base = sc.frontend_oracle_atmosphere(c, 8, **case_kwargs).assemble()
    d_ext, d_ssa, d_leg = _storage_derivative(base, key)
    results = []
    for sign in (+1, -1):
        cp = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
        perturb(cp, j, sign * step)
        a = sc.frontend_oracle_atmosphere(cp, 8, **case_kwargs).assemble(derivatives=False)
        results.append((a.ext.copy(), a.ssa.copy(), a.leg.copy()))
    for analytic, (plus, minus), scale in ((d_ext[j], (results[0][0], results[1][0]), base.ext), (d_ssa[j], (results[0][1], results[1][1]), None),
                                           (d_leg[j], (results[0][2], results[1][2]), None)):
        numeric = (plus - minus) / (2 * step)
        floor = np.max(np.abs(analytic)) * rtol + 8e-16 * np.max(np.abs(plus)) / step + 1e-300   # rounding of the difference quotient
        np.testing.assert_allclose(numeric, analytic, rtol=rtol, atol=floor)
    assert np.any(d_ext[j] != 0.0)


def test_vmr_mapping_matches_storage_difference():
    def perturb(c, j, h):
        c["o3_vmr"][j] += h
    for j in (2, 6, 11):
        _check_against_finite_difference({}, "wf_o3_vmr", perturb, j, 1e-9, 2e-6)


def test_number_density_mapping_matches_storage_difference():
    def perturb(c, j, h):
        c["aer_nd"][j] += h
    for j in (0, 2, 4):
        _check_against_finite_difference({}, "wf_aerosol_number_density", perturb, j, 1e3, 2e-6)


def test_extinction_mapping_matches_storage_difference():
    def perturb(c, j, h):
        c["aer_ext"][j] += h
    for j in (1, 3):
        _check_against_finite_difference({}, "wf_cloud_extinction", perturb, j, 1e-12, 2e-6)


@pytest.mark.parametrize("pt_table", [False, True], ids=["temperature table", "+ pressure-temperature table"])
@pytest.mark.parametrize("key,field,step", [("wf_temperature_k", "temperature", 1e-3), ("wf_pressure_pa", "pressure", 1e-2)])
def test_state_mappings_match_storage_difference(key, field, step, pt_table):
    """every constituent that depends on the state adds a mapping with the same assign_name (Rayleigh through N, the absorber through N
    and through its temperature-dependent table); their sum is the total derivative"""
    def perturb(c, j, h):
        c[field][j] += h
    for j in (1, 9, 20):
        _check_against_finite_difference({"pt_table": pt_table, "cia": pt_table}, key, perturb, j, step, 5e-6)


def test_thermal_emission_and_its_temperature_mapping():
    c = sc.frontend_case(nwavel=6)
    c["wavelengths"] = np.linspace(8000.0, 12000.0, 6)
    a = sc.frontend_oracle_atmosphere(c, 4, thermal=True).assemble()
    # Planck radiance at 300 K, 10 um: 9.92 W m^-2 sr^-1 um^-1 (textbook value) = 9.92e-3 per nm
    np.testing.assert_allclose(fo.planck(np.array(300.0), 10000.0), 9.92e-3, rtol=2e-3)
    m = a.mappings["wf_thermal_temperature_k"]
    h = 1e-3
    fd = (fo.planck(c["temperature"][:, None] + h, c["wavelengths"][None, :]) - fo.planck(c["temperature"][:, None] - h, c["wavelengths"][None, :])) / (2 * h)
    np.testing.assert_allclose(m.d_emission, fd, rtol=1e-7)
    assert m.assign_name == "wf_temperature_k" and np.all(a.emission > 0)


def test_normalisation_and_transparent_cells():
    c = sc.frontend_case(nwavel=5)
    a = sc.frontend_oracle_atmosphere(c, 8).assemble()
    assert np.all(a.ssa <= 1.0) and np.all(a.ssa > 0.0) and np.all(a.ext > 0.0)
    np.testing.assert_allclose(a.leg[0], 1.0, rtol=1e-14)    # beta_0 = 1 wherever something scatters
    # only an absorber, placed where its profile is zero: ext = 0 -> ssa = 0, finite mappings (test_vmr_absorber_2d.py:330-345)
    b = fo.Atmosphere(c["altitudes"], c["wavelengths"], 4, 1, c["pressure"], c["temperature"])
    wv = np.sort(1e7 / c["tab_wl"])
    b.constituents["x"] = fo.VMRAltitudeAbsorber(fo.AbsorberTable(wv, c["tab_xs"][0][::-1]), c["o3_alt"], np.zeros(14), "zero")
    b.assemble()
    assert np.all(b.ext == 0.0) and np.all(b.ssa == 0.0)
    for m in b.mappings.values():
        assert np.all(np.isfinite(m.d_ssa)) and np.all(np.isfinite(m.d_extinction))


def test_henyey_greenstein_moments_and_interpolation_matrix():
    wl = np.arange(300.0, 801.0, 10.0)
    hg = fo.HenyeyGreensteinTable(wl, np.full_like(wl, 1e-4), np.full_like(wl, 0.9e-4), np.full_like(wl, 0.75), 16)
    xs, scat, leg = hg.quantities(1e7 / np.arange(350.0, 500.0, 7.3), 16, 1)
    np.testing.assert_allclose(xs, 1e-4, rtol=1e-12)
    np.testing.assert_allclose(scat / xs, 0.9, rtol=1e-12)
    np.testing.assert_allclose(leg[:, 0], 1.0, rtol=1e-12)
    np.testing.assert_allclose(leg[:, 5], 0.75 ** 5 * 11, rtol=1e-12)
    m = fo.interp_matrix(np.array([0.0, 1.0, 2.0]), np.array([-0.5, 0.5, 1.5, 2.5]), "extend")
    np.testing.assert_array_equal(m, [[1, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0.5], [0, 0, 1]])
    z = fo.interp_matrix(np.array([0.0, 1.0, 2.0]), np.array([-0.5, 2.5]), "zero")
    assert np.all(z == 0.0)


def test_rayleigh_cross_section_and_polarised_coefficients():
    sigma, king = fo.rayleigh_bates(np.array([0.55, 0.35]))
    np.testing.assert_allclose(sigma[0], 4.513e-31, rtol=3e-3)     # Bucholtz (1995) table 2: 4.513e-27 cm^2 at 0.55 um
    assert 1.04 < king[0] < 1.06 and sigma[1] / sigma[0] > (0.55 / 0.35) ** 4
    c = sc.frontend_case(nwavel=4)
    a = fo.Atmosphere(c["altitudes"], c["wavelengths"], 4, 3, c["pressure"], c["temperature"])
    a.constituents["rayleigh"] = fo.Rayleigh()
    a.assemble()
    _, king = fo.rayleigh_bates(c["wavelengths"] / 1000.0)
    delta = 6 * (king - 1) / (3 + 7 * king)
    f = (1 - delta) / (2 + delta)
    np.testing.assert_allclose(a.ssa, 1.0, rtol=1e-15)
    np.testing.assert_allclose(a.leg[8], np.tile(f, (25, 1)), rtol=1e-14)          # a1, order 2
    np.testing.assert_allclose(a.leg[9], np.tile(6 * f, (25, 1)), rtol=1e-14)      # a2
    np.testing.assert_allclose(a.leg[11], np.tile(np.sqrt(6) * f, (25, 1)), rtol=1e-14)   # b1


# ---------------------------------------------------------------------------------------------------------------- host logic of the product
def test_product_host_logic_matches_oracle():
    """what the product prepares on the host for the device: interpolation weights, Bates spectra, the terms of the tables"""
    import sasktran2_amd.optical as opt
    rng = np.random.default_rng(5)
    grid = np.sort(rng.random(9)) * 10
    x = np.concatenate([[-1.0, grid[0], grid[3], grid[-1], 11.0], rng.random(20) * 10])
    for mode in ("zero", "extend"):
        i0, w0, dw0, i1, w1, dw1 = opt.interp_weights(grid, x, mode)
        y = rng.random(9)
        for k, xv in enumerate(x):
            (a0, b0, c0), (a1, b1, c1) = fo.interp_weights(grid, xv, mode)
            assert y[i0[k]] * w0[k] + y[i1[k]] * w1[k] == pytest.approx(y[a0] * b0 + y[a1] * b1, rel=1e-15, abs=1e-300)
            assert y[i0[k]] * dw0[k] + y[i1[k]] * dw1[k] == pytest.approx(y[a0] * c0 + y[a1] * c1, rel=1e-15, abs=1e-300)
        np.testing.assert_allclose(opt.linear_interpolating_matrix(grid, x, mode), fo.interp_matrix(grid, x, mode), rtol=1e-15)
    wl = np.linspace(0.2, 1.1, 300)
    for got, want in zip(opt.rayleigh_cross_section_bates(wl), fo.rayleigh_bates(wl)):
        np.testing.assert_allclose(got, want, rtol=1e-14)

    class _Atmo:   # the attributes an optical property reads
        pass
    c = sc.frontend_case(nwavel=17)
    a = _Atmo()
    a.wavenumbers_cminv, a.temperature_k, a.pressure_pa = 1e7 / c["wavelengths"], c["temperature"], c["pressure"]
    a.num_locations, a.num_moments, a.nstokes = len(c["altitudes"]), 8, 1
    terms = opt.AbsorberDatabaseDim2(wavelength_nm=c["tab_wl"], param=c["tab_t"], xs=c["tab_xs"]).spectral_terms(a)
    xs = sum(w[:, None] * s["ext"][None, :] for w, _, s in terms)
    dxs = sum(d["temperature_k"][:, None] * s["ext"][None, :] for _, d, s in terms)
    wv = 1e7 / c["tab_wl"]
    order = np.argsort(wv)
    want, dwant = fo.AbsorberTable(wv[order], c["tab_xs"][:, order], c["tab_t"]).cross_section(a.wavenumbers_cminv, c["temperature"])
    np.testing.assert_allclose(xs, want, rtol=1e-14)
    np.testing.assert_allclose(dxs, dwant, rtol=1e-13, atol=1e-45)
    terms = opt.AbsorberDatabaseDim3(wavelength_nm=c["tab_wl"], param0=c["tab_p"], param1=c["tab_t"], xs=c["pt_xs"]).spectral_terms(a)
    want, dp, dt = fo.AbsorberTable2(wv[order], c["tab_p"], c["tab_t"], c["pt_xs"][:, :, order]).cross_section(a.wavenumbers_cminv, c["temperature"], c["pressure"])
    np.testing.assert_allclose(sum(w[:, None] * s["ext"][None, :] for w, _, s in terms), want, rtol=1e-14)
    np.testing.assert_allclose(sum(d["pressure_pa"][:, None] * s["ext"][None, :] for _, d, s in terms), dp, rtol=1e-13, atol=1e-45)
    np.testing.assert_allclose(sum(d["temperature_k"][:, None] * s["ext"][None, :] for _, d, s in terms), dt, rtol=1e-13, atol=1e-45)
    (w, _, s), = opt.HenyeyGreenstein.from_parameters(c["hg_wl"], c["hg_xs"], c["hg_scat"], c["hg_g"], 64).spectral_terms(a)
    oxs, osc, oleg = fo.HenyeyGreensteinTable(c["hg_wl"], c["hg_xs"], c["hg_scat"], c["hg_g"], 64).quantities(a.wavenumbers_cminv, 8, 1)
    np.testing.assert_allclose(s["ext"], oxs, rtol=1e-14)
    np.testing.assert_allclose(s["scat"], osc, rtol=1e-14)
    np.testing.assert_allclose(s["leg"], oleg, rtol=1e-14)


def test_us76_standard_atmosphere_follows_the_reference_table():
    """`add_us76_standard_atmosphere` of the reference interpolates a 21-node table (src/sasktran2/climatology/us76.py:87-116; the nodes
    are the fixture tests/golden/us76_nodes.json): linear temperature, log-linear pressure, clamped below -1 km, temperature held and the
    pressure continued with the last log-slope above 80 km.  Checked at the nodes and at off-node altitudes, where the table and the
    standard's own equations differ by kelvins."""
    import json
    from pathlib import Path
    from sasktran2_amd.climatology import us76
    g = json.loads((Path(__file__).parent / "golden" / "us76_nodes.json").read_text())
    zn, tn, pn = np.array(g["altitude_m"]), np.array(g["temperature_c"]) + 273.15, np.array(g["pressure_1e4_pa"]) * 1e4
    t, p = us76.temperature_pressure(zn)
    np.testing.assert_allclose(t, tn, rtol=0, atol=1e-12)
    np.testing.assert_allclose(p, pn, rtol=1e-14)
    # off-node: 12 km (between the 10 and 15 km nodes, across the tropopause), 45 km (across the stratopause), 90 km (above the table), -2 km
    z = np.array([12000.0, 45000.0, 90000.0, -2000.0])
    t, p = us76.temperature_pressure(z)
    f12, f45 = (12000.0 - 10000.0) / 5000.0, (45000.0 - 40000.0) / 10000.0
    np.testing.assert_allclose(t, [tn[11] + f12 * (tn[12] - tn[11]), tn[16] + f45 * (tn[17] - tn[16]), tn[-1], tn[0]], rtol=1e-14)
    slope = (np.log(pn[-1]) - np.log(pn[-2])) / 10000.0
    np.testing.assert_allclose(p, [pn[11] ** (1 - f12) * pn[12] ** f12, pn[16] ** (1 - f45) * pn[17] ** f45,
                                   pn[-1] * np.exp(slope * 10000.0), pn[0]], rtol=1e-13)
    # the standard's equations are something else between the nodes: 216.65 K at 12 km against 220.6 K from the table
    ta, pa = us76.analytic_temperature_pressure(z[:2])
    assert abs(ta[0] - 216.65) < 1e-9 and abs(t[0] - ta[0]) > 3.0
    np.testing.assert_allclose(us76.analytic_temperature_pressure(zn[1:-3])[1], pn[1:-3], rtol=6e-4)   # printed precision of the table

    class _Geo:
        def altitudes(self):
            return z

    class _Atmo:
        model_geometry = _Geo()
    a = _Atmo()
    us76.add_us76_standard_atmosphere(a)
    np.testing.assert_array_equal(a.temperature_k, t)
    np.testing.assert_array_equal(a.pressure_pa, p)


def test_cached_spectra_are_read_only_and_keyed_by_identity():
    """what lets the front end keep a spectrum in HBM from one assembly to the next (sasktran2_amd/frontend.py `_frozen_key`): the
    optical properties hand out the same read-only arrays while the spectral grid stays the same, new ones when it changes"""
    import sasktran2_amd.optical as opt
    from sasktran2_amd.frontend import _frozen_key

    class _Atmo:
        pass
    c = sc.frontend_case(nwavel=17)
    a = _Atmo()
    a.wavenumbers_cminv, a.temperature_k, a.pressure_pa = 1e7 / c["wavelengths"], c["temperature"], c["pressure"]
    a.num_locations, a.num_moments, a.nstokes = len(c["altitudes"]), 8, 1
    table = opt.AbsorberDatabaseDim2(wavelength_nm=c["tab_wl"], param=c["tab_t"], xs=c["tab_xs"])
    hg = opt.HenyeyGreenstein.from_parameters(c["hg_wl"], c["hg_xs"], c["hg_scat"], c["hg_g"], 64)
    first = [s for _, _, s in table.spectral_terms(a)] + [hg.spectral_terms(a)[0][2]]
    a.temperature_k = c["temperature"] + 0.01
    again = [s for _, _, s in table.spectral_terms(a)] + [hg.spectral_terms(a)[0][2]]
    assert len(first) == len(again)
    for s0, s1 in zip(first, again):
        for field in ("ext", "scat", "leg"):
            k0, k1 = _frozen_key(s0[field]), _frozen_key(s1[field])
            assert k0 is not False and k0 == k1
            if s0[field] is not None:
                assert not s0[field].flags.writeable
                with pytest.raises(ValueError):
                    s0[field][...] = 0.0
    a.wavenumbers_cminv = a.wavenumbers_cminv * 1.001
    moved = [s for _, _, s in table.spectral_terms(a)]
    held = first   # (the front end holds the arrays it uploaded, so a new array cannot take the address of an old one)
    assert all(_frozen_key(s1["ext"]) != _frozen_key(s0["ext"]) for s0, s1 in zip(held, moved))
    assert _frozen_key(np.zeros(3)) is False and _frozen_key(None) is None
    ro = np.zeros((4, 3))
    ro.setflags(write=False)
    assert _frozen_key(ro[:, 0]) is False and _frozen_key(ro[1]) == (ro[1].__array_interface__["data"][0], (3,))


def test_surface_thermal_emission_oracle():
    """emission.py:113-190: emissivity x Planck at the surface temperature; its two surface mappings are the derivatives of that product"""
    wl = np.linspace(4000.0, 14000.0, 7)
    eps = np.linspace(0.8, 1.0, 7)
    s = fo.SurfaceThermalEmission(285.0, eps)
    maps = s.surface_mappings("ground", wl)
    h = 1e-3
    fd = (fo.SurfaceThermalEmission(285.0 + h, eps).emission(wl) - fo.SurfaceThermalEmission(285.0 - h, eps).emission(wl)) / (2 * h)
    np.testing.assert_allclose(maps["wf_ground_temperature_k"], fd, rtol=1e-8)
    np.testing.assert_allclose(maps["wf_ground_emissivity"] * eps, s.emission(wl), rtol=1e-15)
    # a black body at 300 K radiates 9.92 W / (m^2 sr um) at 10 um (Planck's law with the CODATA constants)
    assert fo.SurfaceThermalEmission(300.0, 1.0).emission(np.array([10000.0]))[0] * 1e3 == pytest.approx(9.924, rel=1e-3)


@pytest.mark.parametrize("index,key,step", [(0, "height_m", 1.0), (1, "width_fwhm_m", 1.0), (2, "vertical_optical_depth", 1e-6)])
def test_gaussian_height_mappings_match_storage_difference(index, key, step):
    """gaussianheight.py:163-232: the three one-parameter mappings of GaussianHeightExtinction equal finite differences of the assembled
    storage (the criterion of tests/constituent/test_gaussian_height.py of the reference, on the storage instead of the radiance)"""
    c = sc.frontend_case(nwavel=12)
    params = [17000.0, 5000.0, 0.08]
    base = sc.frontend_oracle_atmosphere(c, 8, gaussian=tuple(params)).assemble()
    assert np.trapezoid(base.constituents["layer"].number_density, base.constituents["layer"].alts) * base.constituents["layer"].xs_ref == pytest.approx(0.08, rel=1e-13)
    d_ext, d_ssa, d_leg = _storage_derivative(base, f"wf_layer_{key}")
    results = []
    for sign in (+1, -1):
        p = list(params)
        p[index] += sign * step
        a = sc.frontend_oracle_atmosphere(c, 8, gaussian=tuple(p)).assemble(derivatives=False)
        results.append((a.ext.copy(), a.ssa.copy(), a.leg.copy()))
    for analytic, plus, minus in ((d_ext[0], results[0][0], results[1][0]), (d_ssa[0], results[0][1], results[1][1]), (d_leg[0], results[0][2], results[1][2])):
        numeric = (plus - minus) / (2 * step)
        floor = np.max(np.abs(analytic)) * 2e-6 + 8e-16 * np.max(np.abs(plus)) / step + 1e-300
        np.testing.assert_allclose(numeric, analytic, rtol=2e-6, atol=floor)
    assert np.any(d_ext[0] != 0.0) and d_ext.shape[0] == 1


def test_air_mass_factor_mapping_is_an_absorber_perturbation():
    """amf.py:24-50: the `air_mass_factor` mapping is (-1 / layer thickness) x the storage derivative of adding absorber extinction to a
    level -- extinction + 1, scattering extinction unchanged -- and is flagged for the log radiance"""
    from oracle import frontend_oracle as fo
    c = sc.frontend_case(nwavel=7)
    a = sc.frontend_oracle_atmosphere(c, 8)
    a.constituents["amf"] = fo.AirMassFactor()
    a = a.assemble()
    m = a.mappings["air_mass_factor"]
    assert m.log_radiance_space and m.interp_dim == "altitude" and m.interpolator is None and m.scat_factor is None
    dz = np.gradient(c["altitudes"])
    dz[0] /= 2.0
    dz[-1] /= 2.0
    h = 1e-7 * a.ext
    ssa_plus = a.ssa * a.ext / (a.ext + h)            # the same scattering extinction in a larger total
    np.testing.assert_allclose(m.d_extinction * -dz[:, None], 1.0, rtol=1e-14)
    np.testing.assert_allclose(m.d_ssa * -dz[:, None], (ssa_plus - a.ssa) / h, rtol=1e-6, atol=1e-300)   # (forward difference, step 1e-7 of the extinction)
    native = {"ext": np.ones((len(dz), 7, 2)), "ssa": np.zeros((len(dz), 7, 2)), "radiance": np.full((7, 2), 4.0)}
    np.testing.assert_allclose(fo.assemble_weighting_functions(a, native)["air_mass_factor"], np.broadcast_to((-1.0 / dz / 4.0)[:, None, None], (len(dz), 7, 2)), rtol=1e-14)
