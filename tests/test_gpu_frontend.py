"""Optical-property front end on the device: parity of frontend.hip -- through the C ABI and the Python constituent
interface -- with the numpy oracle (oracle/frontend_oracle.py), and the reference's own criterion for this row: weighting functions of
the constituents' parameters equal finite differences of the radiance (tests/weightingfunctions/test_pressure_temperature.py,
tests/constituent/test_scatterer_extinction.py:86-160).

Tolerances: storage and mapping arrays 2e-13 relative to the largest element of the array at that wavelength (the device sums all terms in
one pass, the oracle adds constituent after constituent: a different order of the same sums); radiances 1e-10; weighting functions against
finite differences 2e-5 (central differences of an FP64 forward model)."""
from __future__ import annotations

import numpy as np
import pytest

from tests import scenarios as sc

pytestmark = pytest.mark.gpu


def _geometry(sk, c, geotype=None):
    return sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation,
                         sk.GeometryType.Spherical if geotype is None else geotype)


def _config(sk, nstokes=1, streams=4, moments=8, **kw):
    cfg = sk.Config()
    cfg.num_stokes = nstokes
    cfg.num_streams = streams
    cfg.num_singlescatter_moments = moments
    cfg.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
    cfg.single_scatter_source = sk.SingleScatterSource.DiscreteOrdinates
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _close(got, want, tol=2e-13):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    scale = np.max(np.abs(want), axis=tuple(range(want.ndim - 1)), keepdims=True) if want.ndim > 1 else np.max(np.abs(want))
    assert np.all(np.abs(got - want) <= tol * scale + 1e-300), float(np.max(np.abs(got - want) / (scale + 1e-300)))


@pytest.mark.parametrize("nstokes", [1, 3])
def test_storage_and_mappings_match_oracle(nstokes):
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=33)
    cfg = _config(sk, nstokes=nstokes, moments=8)
    atmo = sc.frontend_product_atmosphere(sk, c, _geometry(sk, c), cfg, thermal=True, pt_table=True, cia=True)
    atmo.internal_object()
    want = sc.frontend_oracle_atmosphere(c, 8, nstokes, thermal=True, pt_table=True, cia=True).assemble()
    st = atmo.storage
    _close(st.total_extinction, want.ext)
    _close(st.ssa, want.ssa)
    _close(st.leg_coeff, want.leg)
    _close(st.emission_source, want.emission)
    assert set(st.derivative_mapping_names()) == set(want.mappings)
    for name, w in want.mappings.items():
        m = st.get_derivative_mapping(name)
        assert (m.assign_name, m.interp_dim) == (w.assign_name, w.interp_dim), name
        _close(m.d_extinction, w.d_extinction)
        # d_ssa is a difference (species ssa - total ssa) scaled by 1/k: compare against the size of its two parts
        part = np.abs(w.d_extinction) / np.where(want.ext > 0, want.ext, 1.0)
        assert np.all(np.abs(m.d_ssa - w.d_ssa) <= 1e-12 * np.max(part, axis=0, keepdims=True) + 1e-300), name
        if w.scat_factor is not None:
            _close(m.scat_factor, w.scat_factor)
            assert np.all(np.abs(m.d_leg_coeff - w.d_legendre) <= 2e-13 * (1.0 + np.abs(want.leg))), name
        else:
            assert m.scat_factor is None
        if name == "wf_thermal_temperature_k":
            _close(m.d_emission, w.d_emission)
        if w.interpolator is None:
            assert m.interpolator is None
        else:
            np.testing.assert_allclose(m.interpolator, w.interpolator, rtol=1e-14, atol=0)


def test_c_abi_terms_profile_update_and_refusals():
    """the C ABI directly: one term, its derivative mapping, a profile update, and what a device-resident mapping refuses"""
    import ctypes

    from sasktran2_amd._lib_loader import last_error, lib
    L = lib()
    ngrid, nw, nleg = 6, 5, 4
    rng = np.random.default_rng(1)
    prof, ext, scat = rng.random(ngrid) + 0.1, rng.random(nw) + 0.5, rng.random(nw) * 0.4
    leg = rng.random((nw, nleg))
    leg[:, 0] = 1.0
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    f = L.sk_amd_frontend_create(ngrid, nw, nleg, 1)
    assert f
    try:
        assert L.sk_amd_frontend_storage(f) is None                     # nothing assembled yet
        t = L.sk_amd_frontend_add_term(f, vp(prof), vp(ext), vp(scat), vp(leg))
        assert t == 0 and L.sk_amd_frontend_assemble(f) == 0
        k = np.zeros((nw, ngrid))
        assert L.sk_amd_frontend_download(f, b"extinction", vp(k)) == 0
        np.testing.assert_allclose(k, ext[:, None] * prof[None, :], rtol=1e-15)
        terms = np.array([0], dtype=np.int32)
        assert L.sk_amd_frontend_add_mapping(f, b"wf_x", 1, vp(terms), None, 1, 0, None, 0, b"altitude", b"") == 0
        assert L.sk_amd_frontend_add_mapping(f, b"wf_x", 1, vp(terms), None, 1, 0, None, 0, b"altitude", b"") == -3 and b"exists" in L.sk_amd_last_error()
        d = np.zeros((nw, ngrid))
        assert L.sk_amd_frontend_download(f, b"wf_x:d_ssa", vp(d)) == 0
        np.testing.assert_allclose(d, 0.0, atol=1e-15)                  # a single species: its ssa IS the total ssa
        dl = np.zeros((nw, ngrid, nleg))
        assert L.sk_amd_frontend_download(f, b"wf_x:d_legendre", vp(dl)) == 0
        np.testing.assert_allclose(dl, 0.0, atol=1e-15)
        # a host accessor of a device mapping is refused, loudly
        st = L.sk_amd_frontend_storage(f)
        m = ctypes.c_void_p()
        assert L.sk_atmosphere_storage_get_derivative_mapping(st, b"wf_x", ctypes.byref(m)) == 0
        p = ctypes.POINTER(ctypes.c_double)()
        assert L.sk_deriv_mapping_get_d_ssa(m, ctypes.byref(p)) == -2 and "device" in last_error()
        assert L.sk_atmosphere_storage_set_zero(st) == -2
        # amounts change, spectra stay
        prof2 = prof * 2.0
        assert L.sk_amd_frontend_set_profile(f, 0, vp(prof2)) == 0 and L.sk_amd_frontend_assemble(f) == 0
        assert L.sk_amd_frontend_download(f, b"extinction", vp(k)) == 0
        np.testing.assert_allclose(k, ext[:, None] * prof2[None, :], rtol=1e-15)
        assert L.sk_amd_frontend_download(f, b"nothing", vp(k)) == -3
        assert L.sk_amd_frontend_add_term(f, vp(prof), vp(ext), None, vp(leg)) == -3     # moments without a scattering cross section
    finally:
        L.sk_amd_frontend_destroy(f)


def _engine_case(sk, c, cfg, rays, **atmo_kw):
    geo = _geometry(sk, c, atmo_kw.pop("geotype", None))
    vg = sk.ViewingGeometry()
    for r in rays:
        vg.add_ray(r)
    return sk.Engine(cfg, geo, vg), geo


def _perturbed(c, key, j, h):
    cp = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    cp[key][j] += h
    return cp


@pytest.mark.parametrize("key,field,j,step", [
    ("wf_o3_vmr", "o3_vmr", 5, 2e-8), ("wf_aerosol_number_density", "aer_nd", 2, 2e4), ("wf_cloud_extinction", "aer_ext", 1, 2e-10),
    ("wf_temperature_k", "temperature", 6, 5e-2), ("wf_pressure_pa", "pressure", 3, 5.0)])
def test_weighting_functions_match_finite_differences(key, field, j, step):
    """radiance through the whole chain (constituents -> device assembly -> discrete ordinates), limb and nadir lines of sight"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=6)
    cfg = _config(sk, streams=8, moments=8)
    rays = [sk.TangentAltitudeSolar(15000.0, 0.3, 200000.0, 0.6), sk.TangentAltitudeSolar(32000.0, 0.3, 200000.0, 0.6),
            sk.GroundViewingSolar(0.6, 0.5, 0.8, 200000.0)]
    engine, geo = _engine_case(sk, c, cfg, rays)

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, calculate_derivatives=derivs, pt_table=True, cia=True)
        atmo["surface"] = sk.constituent.LambertianSurface(0.3)
        return engine.calculate_radiance(atmo, as_dataset=False)
    base = radiance(c, True)
    assert base[key].shape[1:] == (6, 3, 1)
    fd = (radiance(_perturbed(c, field, j, step), False)["radiance"] - radiance(_perturbed(c, field, j, -step), False)["radiance"]) / (2 * step)
    got = base[key][j]
    assert np.max(np.abs(got)) > 0
    np.testing.assert_allclose(got, fd, rtol=2e-5, atol=2e-6 * np.max(np.abs(fd)))
    # the surface constituent's weighting function: one parameter (constant albedo) x wavelength x los
    assert base["wf_surface_albedo"].shape == (1, 6, 3, 1)


def test_constituent_path_equals_raw_interface():
    """the same atmosphere through the raw interface (host arrays from the oracle, uploaded) and through the device front end: radiances
    agree to rounding, and the absorber weighting function equals interpolator^T (d_extinction x wf_extinction + d_ssa x wf_ssa)"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=9)
    for delta_m in (False, True):
        cfg = _config(sk, streams=4, moments=8, delta_m_scaling=delta_m)
        rays = [sk.TangentAltitudeSolar(20000.0, 0.0, 200000.0, 0.6), sk.GroundViewingSolar(0.6, 1.0, 0.7, 200000.0)]
        engine, geo = _engine_case(sk, c, cfg, rays)
        atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg)
        atmo.surface.albedo[:] = 0.25
        got = engine.calculate_radiance(atmo, as_dataset=False)
        want = sc.frontend_oracle_atmosphere(c, 8).assemble()
        raw = sk.Atmosphere(geo, cfg, numwavel=9)
        raw.storage.total_extinction[:] = want.ext
        raw.storage.ssa[:] = want.ssa
        raw.leg_coeff.a1[:] = want.leg
        raw.surface.albedo[:] = 0.25
        ref = engine.calculate_radiance(raw, as_dataset=False)
        np.testing.assert_allclose(got["radiance"], ref["radiance"], rtol=1e-10)
        np.testing.assert_allclose(got["los_optical_depth"], ref["los_optical_depth"], rtol=1e-12)
        if not delta_m:
            m = want.mappings["wf_o3_vmr"]
            v = m.d_extinction[:, :, None, None] * ref["wf_extinction"] + m.d_ssa[:, :, None, None] * ref["wf_ssa"]
            np.testing.assert_allclose(got["wf_o3_vmr"], np.einsum("qj,qwls->jwls", m.interpolator, v), rtol=1e-8, atol=1e-10 * np.max(np.abs(got["wf_o3_vmr"])))   # (the raw run takes the matrix-core kernels, the constituent run -- scattering groups -- the run-time-n ones)


def test_delta_m_scaling_on_the_device():
    """delta-M scaling of storage and mappings in HBM (k_fe_delta_m_*) against the host scaling of sk_atmosphere_apply_delta_m_scaling
    (same atmosphere through the raw interface, the oracle's mapping arrays written into a host mapping), and the absorber's weighting
    function against finite differences.  (A scatterer's weighting function under delta-M is NOT compared with finite differences: the
    discrete-ordinates source takes a layer's scattering direction from one grid point of its stencil, as the reference does --
    sktran_do_layerarray.cpp:759-860 -- which is exact only while the truncation fraction does not vary over the layer.)"""
    import ctypes

    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=4)
    ng, nw, moments = 25, 4, 16
    cfg = _config(sk, streams=4, moments=moments, delta_m_scaling=True)
    engine, geo = _engine_case(sk, c, cfg, [sk.GroundViewingSolar(0.6, 0.5, 0.8, 200000.0), sk.TangentAltitudeSolar(15000.0, 0.3, 200000.0, 0.6)])
    kw = dict(extinction_scatterer=False, pressure_derivative=False, temperature_derivative=False)

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, calculate_derivatives=derivs, **kw)
        atmo.surface.albedo[:] = 0.1
        return engine.calculate_radiance(atmo, as_dataset=False)
    base = radiance(c, True)
    want = sc.frontend_oracle_atmosphere(c, moments, **kw).assemble()
    mp = want.mappings["wf_aerosol_number_density"]
    dp, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p

    def hook(L, h_st, h_out, out):
        m = vp()
        L.sk_atmosphere_storage_get_derivative_mapping(h_st, b"wf_scatterer", ctypes.byref(m))
        for which, arr in (("d_extinction", mp.d_extinction.T), ("d_ssa", mp.d_ssa.T), ("scat_factor", mp.scat_factor.T),
                           ("d_legendre", mp.d_legendre.transpose(2, 1, 0))):
            ptr = dp()
            assert getattr(L, f"sk_deriv_mapping_get_{which}")(m, ctypes.byref(ptr)) == 0
            np.ctypeslib.as_array(ptr, shape=(arr.size,))[:] = arr.reshape(-1)
        out["wf_scatterer"] = np.zeros((ng, nw, 2, 1))
        L.sk_output_assign_derivative_memory(h_out, b"wf_scatterer", out["wf_scatterer"].ctypes.data_as(dp), nw * 2, 1, ng)
    raw = sk.Atmosphere(geo, cfg, numwavel=nw)
    raw.storage.total_extinction[:] = want.ext
    raw.storage.ssa[:] = want.ssa
    raw.leg_coeff.a1[:] = want.leg
    raw.surface.albedo[:] = 0.1
    host = engine.calculate_radiance(raw, as_dataset=False, _mapping_hook=hook)
    np.testing.assert_allclose(base["radiance"], host["radiance"], rtol=1e-10)
    ref = np.einsum("qj,qwls->jwls", mp.interpolator, host["wf_scatterer"])
    np.testing.assert_allclose(base["wf_aerosol_number_density"], ref, rtol=1e-8, atol=1e-10 * np.max(np.abs(ref)))
    j, step = 4, 2e-8
    fd = (radiance(_perturbed(c, "o3_vmr", j, step), False)["radiance"] - radiance(_perturbed(c, "o3_vmr", j, -step), False)["radiance"]) / (2 * step)
    np.testing.assert_allclose(base["wf_o3_vmr"][j], fd, rtol=2e-5, atol=2e-6 * np.max(np.abs(fd)))


def test_two_stream_thermal_source_from_constituents():
    """emission_source = twostream with the Planck term of ThermalEmission: wf_temperature_k collects the Planck derivative, the number
    density of every gas and the temperature dependence of the absorber's table"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=5)
    c["wavelengths"] = np.linspace(9000.0, 11000.0, 5)
    c["tab_wl"] = np.linspace(8000.0, 12000.0, 111)
    c["tab_xs"] = c["tab_xs"] * 1e3
    cfg = sk.Config()
    cfg.num_streams = 2
    cfg.num_singlescatter_moments = 4
    cfg.multiple_scatter_source = sk.MultipleScatterSource.NoSource
    cfg.single_scatter_source = sk.SingleScatterSource.NoSource
    cfg.emission_source = sk.EmissionSource.TwoStream
    engine, geo = _engine_case(sk, c, cfg, [sk.GroundViewingSolar(0.6, 0.0, 0.9, 200000.0)], geotype=sk.GeometryType.PlaneParallel)

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, thermal=True, calculate_derivatives=derivs, extinction_scatterer=False)
        atmo.surface.albedo[:] = 0.02
        atmo.surface.emission[:] = 0.0
        return engine.calculate_radiance(atmo, as_dataset=False)
    base = radiance(c, True)
    assert np.all(base["radiance"] > 0)
    for j in (0, 4, 12):
        step = 2e-2
        fd = (radiance(_perturbed(c, "temperature", j, step), False)["radiance"] - radiance(_perturbed(c, "temperature", j, -step), False)["radiance"]) / (2 * step)
        np.testing.assert_allclose(base["wf_temperature_k"][j], fd, rtol=2e-5, atol=2e-6 * np.max(np.abs(fd)))


def test_exact_single_scatter_from_constituents_and_input_validation():
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=7)
    cfg = sk.Config()
    cfg.num_singlescatter_moments = 16
    cfg.multiple_scatter_source = sk.MultipleScatterSource.NoSource
    cfg.single_scatter_source = sk.SingleScatterSource.Exact
    engine, geo = _engine_case(sk, c, cfg, [sk.TangentAltitudeSolar(18000.0, 0.2, 200000.0, 0.6)])
    atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, calculate_derivatives=False)
    got = engine.calculate_radiance(atmo, as_dataset=False)
    want = sc.frontend_oracle_atmosphere(c, 16).assemble(derivatives=False)
    raw = sk.Atmosphere(geo, cfg, numwavel=7, calculate_derivatives=False)
    raw.storage.total_extinction[:] = want.ext
    raw.storage.ssa[:] = want.ssa
    raw.leg_coeff.a1[:] = want.leg
    np.testing.assert_allclose(got["radiance"], engine.calculate_radiance(raw, as_dataset=False)["radiance"], rtol=1e-10)
    # a negative mixing ratio makes a negative extinction: found by the device-side scan, reported like the reference does
    bad = _perturbed(c, "o3_vmr", 6, -1.0)
    with pytest.raises(RuntimeError, match="total extinction contains values less than"):
        engine.calculate_radiance(sc.frontend_product_atmosphere(sk, bad, geo, cfg, calculate_derivatives=False), as_dataset=False)


def test_polarised_engine_from_constituents():
    """NSTOKES = 3: Rayleigh's a1 / a2 / b1 coefficients and the scatterers' a1 through the run-time-n discrete-ordinates kernels; the
    same atmosphere through the raw interface (oracle arrays) gives the same Stokes vectors, and wf_o3_vmr equals finite differences"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=3)
    cfg = _config(sk, nstokes=3, streams=4, moments=8)
    rays = [sk.TangentAltitudeSolar(22000.0, 0.4, 200000.0, 0.6), sk.GroundViewingSolar(0.6, 0.8, 0.7, 200000.0)]
    engine, geo = _engine_case(sk, c, cfg, rays)

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, calculate_derivatives=derivs, pressure_derivative=False, temperature_derivative=False)
        atmo.surface.albedo[:] = 0.2
        return engine.calculate_radiance(atmo, as_dataset=False)
    got = radiance(c, True)
    want = sc.frontend_oracle_atmosphere(c, 8, 3).assemble(derivatives=False)
    raw = sk.Atmosphere(geo, cfg, numwavel=3, calculate_derivatives=False)
    raw.storage.total_extinction[:] = want.ext
    raw.storage.ssa[:] = want.ssa
    raw.storage.leg_coeff[:] = want.leg
    raw.surface.albedo[:] = 0.2
    ref = engine.calculate_radiance(raw, as_dataset=False)
    assert np.max(np.abs(got["radiance"][..., 1])) > 1e-3 * np.max(np.abs(got["radiance"][..., 0]))     # Q is alive
    np.testing.assert_allclose(got["radiance"], ref["radiance"], rtol=1e-9, atol=1e-12 * np.max(np.abs(ref["radiance"])))
    j, step = 6, 2e-8
    fd = (radiance(_perturbed(c, "o3_vmr", j, step), False)["radiance"] - radiance(_perturbed(c, "o3_vmr", j, -step), False)["radiance"]) / (2 * step)
    np.testing.assert_allclose(got["wf_o3_vmr"][j], fd, rtol=5e-5, atol=5e-6 * np.max(np.abs(fd)))


def test_line_of_sight_emission_and_jvp_vjp_with_device_mappings():
    """emission_source = standard over the Planck term of ThermalEmission (exact line-of-sight integration): wf_temperature_k -- Planck
    derivative + number densities + the absorber's table -- against finite differences; then the JVP / VJP entry points contract the same
    device-resident mappings (sk_engine_calculate_jvp / _vjp)"""
    import ctypes

    import sasktran2_amd as sk
    from sasktran2_amd._lib_loader import last_error, lib
    c = sc.frontend_case(nwavel=4)
    c["wavelengths"] = np.linspace(9000.0, 11000.0, 4)
    c["tab_wl"] = np.linspace(8000.0, 12000.0, 111)
    c["tab_xs"] = c["tab_xs"] * 3e3
    cfg = sk.Config()
    cfg.num_singlescatter_moments = 4
    cfg.multiple_scatter_source = sk.MultipleScatterSource.NoSource
    cfg.single_scatter_source = sk.SingleScatterSource.NoSource
    cfg.emission_source = sk.EmissionSource.Standard
    engine, geo = _engine_case(sk, c, cfg, [sk.TangentAltitudeSolar(12000.0, 0.0, 200000.0, 0.6), sk.GroundViewingSolar(0.6, 0.0, 0.9, 200000.0)])

    def radiance(case, derivs):
        # an absorbing, emitting atmosphere
        atmo = sk.Atmosphere(geo, cfg, wavelengths_nm=case["wavelengths"], calculate_derivatives=derivs)
        atmo.pressure_pa, atmo.temperature_k = case["pressure"], case["temperature"]
        table = sk.optical.AbsorberDatabaseDim2(wavelength_nm=case["tab_wl"], param=case["tab_t"], xs=case["tab_xs"])
        atmo["o3"] = sk.constituent.VMRAltitudeAbsorber(table, case["o3_alt"], case["o3_vmr"], "extend")
        atmo["thermal"] = sk.constituent.ThermalEmission()
        return engine.calculate_radiance(atmo, as_dataset=False), atmo
    base, atmo = radiance(c, True)
    assert np.all(base["radiance"] > 0)
    for j in (1, 5, 14):
        step = 2e-2
        fd = (radiance(_perturbed(c, "temperature", j, step), False)[0]["radiance"] - radiance(_perturbed(c, "temperature", j, -step), False)[0]["radiance"]) / (2 * step)
        np.testing.assert_allclose(base["wf_temperature_k"][j], fd, rtol=2e-5, atol=2e-6 * np.max(np.abs(fd)))
    # JVP / VJP through the C ABI on the storage handle of the front end
    L = lib()
    fe = atmo.frontend
    nw, nlos = 4, 2
    alb, emis = np.zeros(nw), np.zeros(nw)
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)   # noqa: E731
    h_brdf = L.sk_brdf_create_lambertian(1)
    h_su = L.sk_surface_create(nw, 1, p(emis))
    L.sk_surface_set_brdf(h_su, h_brdf, p(alb))
    h_atm = L.sk_atmosphere_create(fe.storage_handle(), h_su, 1, 0)
    try:
        name = "wf_o3_vmr"
        nout = fe.mappings[name].num_output
        rng = np.random.default_rng(2)
        tangent = rng.random(nout)
        rad, jvp = np.zeros((nw, nlos, 1)), np.zeros((nw, nlos, 1))
        h = L.sk_output_jvp_create(p(rad), p(jvp), nw * nlos, 1)
        L.sk_output_jvp_assign_derivative_tangent(h, name.encode(), p(tangent), nout)
        assert L.sk_engine_calculate_jvp(engine._h, h_atm, h) == 0, last_error()
        L.sk_output_jvp_destroy(h)
        np.testing.assert_allclose(jvp, np.einsum("jwls,j->wls", base[name], tangent), rtol=1e-10)
        cot = rng.random((nw, nlos, 1))
        grad = np.zeros(nout)
        h = L.sk_output_vjp_create(p(rad), p(cot), nw * nlos, 1)
        L.sk_output_vjp_assign_derivative_gradient(h, name.encode(), p(grad), nout)
        assert L.sk_engine_calculate_vjp(engine._h, h_atm, h) == 0, last_error()
        L.sk_output_vjp_destroy(h)
        np.testing.assert_allclose(grad, np.einsum("jwls,wls->j", base[name], cot), rtol=1e-10)
    finally:
        L.sk_atmosphere_destroy(h_atm)
        L.sk_surface_destroy(h_su)
        L.sk_brdf_destroy(h_brdf)
    # ... and through the Python surface: Engine.calculate_jvp / calculate_vjp of a CONSTITUENT atmosphere route through the front end
    # (mapping names of the constituents, the storage stays in HBM) instead of computing on the empty raw storage
    names = [n for n in base if n.startswith("wf_")]
    tang = {n: rng.standard_normal(base[n].shape[0]) for n in names if base[n].ndim == 4 and n in fe.mappings}
    rad_j, jvp_py = engine.calculate_jvp(atmo, tang)
    np.testing.assert_allclose(rad_j, base["radiance"], rtol=1e-13)
    want = sum(np.einsum("jwls,j->wls", base[n], t) for n, t in tang.items())
    np.testing.assert_allclose(jvp_py, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    cot = rng.standard_normal(base["radiance"].shape)
    rad_v, grads = engine.calculate_vjp(atmo, cot)
    np.testing.assert_allclose(rad_v, base["radiance"], rtol=1e-13)
    for n in tang:
        np.testing.assert_allclose(grads[n], np.einsum("jwls,wls->j", base[n], cot), rtol=1e-9, atol=1e-12 * np.abs(grads[n]).max())
    with pytest.raises(KeyError):
        engine.calculate_vjp(atmo, cot, names=("wf_extinction",))


@pytest.mark.parametrize("ms,nstokes", [("none", 1), ("none", 3), ("discrete_ordinates", 1), ("twostream", 1), ("successive_orders_legacy", 1), ("successive_orders", 1),
                                        ("successive_orders_legacy", 3)],
                         ids=["exact single scatter alone", "exact single scatter alone, NSTOKES 3", "+ discrete ordinates", "+ two-stream",
                              "+ successive orders (legacy)", "+ successive orders (Anderson driver)", "+ successive orders (legacy), NSTOKES 3"])
def test_scatterer_weighting_functions_with_exact_single_scatter(ms, nstokes):
    """single_scatter_source = exact -- the reference's limb configuration -- with scatterer constituents: the single-scatter source
    differentiates along the scattering groups' d_legendre directions (k_so_group_phase / k_so_los_deriv), next to the multiple-scatter
    engine's own group derivatives; weighting functions of number density, extinction, pressure and vmr against finite differences"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=4)
    cfg = sk.Config()
    cfg.num_stokes = nstokes
    cfg.num_streams = 2 if ms == "twostream" else 4
    cfg.num_singlescatter_moments = 16
    cfg.single_scatter_source = sk.SingleScatterSource.Exact
    cfg.multiple_scatter_source = {"none": sk.MultipleScatterSource.NoSource, "discrete_ordinates": sk.MultipleScatterSource.DiscreteOrdinates,
                                   "twostream": sk.MultipleScatterSource.TwoStream, "successive_orders_legacy": sk.MultipleScatterSource.SuccessiveOrdersLegacy,
                                   "successive_orders": sk.MultipleScatterSource.SuccessiveOrders}[ms]
    if ms.startswith("successive_orders"):
        # the diffuse field's share of the scattering-group rows: first-order phase values on the incoming rays and the points' scattering
        # matrices (so_adjoint.inc); the Anderson driver converged tightly so that its adjoint about the final state is the derivative
        cfg.num_streams = 8
        cfg.num_successive_orders_incoming = cfg.num_successive_orders_outgoing = 26
        cfg.init_successive_orders_with_discrete_ordinates = False
        cfg.num_successive_orders_iterations = 3 if ms.endswith("legacy") else 60
        if ms == "successive_orders":
            cfg.successive_orders_relative_tolerance = 1e-11
            cfg.successive_orders_absolute_tolerance = 0.0
    rays = [sk.TangentAltitudeSolar(14000.0, 0.5, 200000.0, 0.6), sk.TangentAltitudeSolar(28000.0, 0.5, 200000.0, 0.6),
            sk.GroundViewingSolar(0.6, 0.5, 0.8, 200000.0)]
    engine, geo = _engine_case(sk, c, cfg, rays)

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, calculate_derivatives=derivs, temperature_derivative=False)
        atmo.surface.albedo[:] = 0.2
        return engine.calculate_radiance(atmo, as_dataset=False)
    base = radiance(c, True)
    for key, field, j, step in (("wf_aerosol_number_density", "aer_nd", 2, 2e4), ("wf_cloud_extinction", "aer_ext", 1, 2e-10),
                                ("wf_pressure_pa", "pressure", 4, 5.0), ("wf_o3_vmr", "o3_vmr", 5, 2e-8)):
        fd = (radiance(_perturbed(c, field, j, step), False)["radiance"] - radiance(_perturbed(c, field, j, -step), False)["radiance"]) / (2 * step)
        assert np.max(np.abs(base[key][j])) > 0
        np.testing.assert_allclose(base[key][j], fd, rtol=3e-5, atol=3e-6 * np.max(np.abs(fd)), err_msg=key)


def test_line_of_sight_emission_next_to_discrete_ordinates():
    """emission_source = standard (Planck term of ThermalEmission integrated along the line of sight) next to the discrete-ordinates solar
    source with the exact single scatter, scatterers present: wf_temperature_k gathers the Planck derivative, the number densities
    (scattering groups of Rayleigh) and the absorber's table; against finite differences"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=3)
    c["wavelengths"] = np.linspace(9500.0, 10500.0, 3)
    c["tab_wl"] = np.linspace(8000.0, 12000.0, 111)
    c["tab_xs"] = c["tab_xs"] * 3e3
    c["hg_wl"] = np.linspace(8000.0, 12000.0, 6)
    cfg = _config(sk, streams=4, moments=8)
    cfg.single_scatter_source = sk.SingleScatterSource.Exact
    cfg.emission_source = sk.EmissionSource.Standard
    engine, geo = _engine_case(sk, c, cfg, [sk.TangentAltitudeSolar(16000.0, 0.3, 200000.0, 0.6), sk.GroundViewingSolar(0.6, 0.3, 0.8, 200000.0)])

    def radiance(case, derivs):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, thermal=True, calculate_derivatives=derivs, extinction_scatterer=False)
        atmo.surface.albedo[:] = 0.1
        return engine.calculate_radiance(atmo, as_dataset=False)
    base = radiance(c, True)
    for key, field, j, step in (("wf_temperature_k", "temperature", 3, 2e-2), ("wf_temperature_k", "temperature", 12, 2e-2),
                                ("wf_aerosol_number_density", "aer_nd", 1, 2e4), ("wf_o3_vmr", "o3_vmr", 4, 2e-8)):
        fd = (radiance(_perturbed(c, field, j, step), False)["radiance"] - radiance(_perturbed(c, field, j, -step), False)["radiance"]) / (2 * step)
        np.testing.assert_allclose(base[key][j], fd, rtol=3e-5, atol=3e-6 * np.max(np.abs(fd)), err_msg=key)


def test_constituent_run_is_independent_of_the_spectrum_it_sits_in():
    """130 wavelengths (three wavefront slabs, the last one padded) through the exact single scatter + discrete ordinates with every
    constituent weighting function: a few of those wavelengths solved alone give the same numbers (blocks of the spectrum are independent;
    padding lanes and the group buffers of the single-scatter derivative kernel do not leak)"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=130)
    cfg = _config(sk, streams=4, moments=8)
    cfg.single_scatter_source = sk.SingleScatterSource.Exact
    rays = [sk.TangentAltitudeSolar(18000.0, 0.4, 200000.0, 0.6), sk.GroundViewingSolar(0.6, 0.4, 0.75, 200000.0)]
    engine, geo = _engine_case(sk, c, cfg, rays)

    def run(case):
        atmo = sc.frontend_product_atmosphere(sk, case, geo, cfg, pt_table=True, cia=True)
        atmo.surface.albedo[:] = 0.15
        return engine.calculate_radiance(atmo, as_dataset=False)
    full = run(c)
    pick = np.array([0, 63, 64, 127, 129])
    part = run(dict(c, wavelengths=c["wavelengths"][pick]))
    for key, val in part.items():
        ref = full[key][pick] if key in ("radiance", "los_optical_depth") else full[key][:, pick]
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(val - ref)) <= 1e-12 * scale, key
    assert all(np.all(np.isfinite(v)) for v in full.values())


def test_spectra_stay_in_hbm_between_assemblies():
    """A retrieval rebuilds its atmosphere step after step with new amounts and the same cross sections: from the second assembly on the
    terms keep the spectra they uploaded (read-only arrays of the optical properties' caches) and move only their profiles.  The result is
    that of a front end that never saw the earlier state; a spectrum that changed is uploaded again."""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=33)
    cfg = _config(sk, moments=8)
    geo = _geometry(sk, c)
    atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, thermal=True, pt_table=True, cia=True)
    fe = atmo.internal_object()
    nterms = fe._nterms
    assert nterms > 4 and fe.spectra_reused == 0
    c2 = dict(c, o3_vmr=c["o3_vmr"] * 1.3, aer_nd=c["aer_nd"] * 0.7, temperature=c["temperature"] + 0.01)
    atmo["o3"].vmr = c2["o3_vmr"]
    atmo["aerosol"].number_density = c2["aer_nd"]
    atmo.temperature_k = c2["temperature"]
    fe = atmo.internal_object()
    assert fe._nterms == nterms and fe.spectra_reused == nterms
    want = sc.frontend_oracle_atmosphere(c2, 8, 1, thermal=True, pt_table=True, cia=True).assemble()
    st = atmo.storage
    _close(st.total_extinction, want.ext)
    _close(st.ssa, want.ssa)
    _close(st.leg_coeff, want.leg)
    _close(st.emission_source, want.emission)
    for name, w in want.mappings.items():
        m = st.get_derivative_mapping(name)
        _close(m.d_extinction, w.d_extinction)
        _close(m.d_ssa, w.d_ssa)
        if w.d_legendre is not None:
            _close(m.d_leg_coeff, w.d_legendre)
    fresh = sc.frontend_product_atmosphere(sk, c2, geo, cfg, thermal=True, pt_table=True, cia=True)
    fresh.internal_object()
    assert np.array_equal(fresh.storage.total_extinction, st.total_extinction) and np.array_equal(fresh.storage.leg_coeff, st.leg_coeff)
    # another spectral grid: everything is uploaded again
    atmo.wavelengths_nm = c["wavelengths"] * 1.001
    fe = atmo.internal_object()
    assert fe.spectra_reused == 0
    want = sc.frontend_oracle_atmosphere(dict(c2, wavelengths=c["wavelengths"] * 1.001), 8, 1, thermal=True, pt_table=True, cia=True).assemble()
    _close(atmo.storage.total_extinction, want.ext)
    _close(atmo.storage.leg_coeff, want.leg)
    # the C ABI alone: a term added again keeps its spectra, a position that never held one is refused
    from sasktran2_amd.frontend import FrontEnd
    raw = FrontEnd(4, 5, 2, 1)
    prof, ext = np.array([1.0, 2.0, 3.0, 4.0]), np.linspace(1.0, 2.0, 5)
    raw.add_term(prof, ext)
    raw.assemble()
    first = raw.download("extinction", (5, 4))
    raw.reset()
    from sasktran2_amd._lib_loader import lib
    assert lib().sk_amd_frontend_readd_term(raw._h, (2 * prof).ctypes.data) == 0
    assert lib().sk_amd_frontend_readd_term(raw._h, prof.ctypes.data) == -2
    raw.assemble()
    assert np.array_equal(raw.download("extinction", (5, 4)), 2 * first)


@pytest.mark.parametrize("source", ["twostream", "discrete_ordinates", "line_of_sight"])
def test_surface_thermal_emission_constituent(source):
    """SurfaceThermalEmission (src/sasktran2/constituent/emission.py:113-190): the surface emits emissivity x Planck(T_surface); what the
    surface holds equals the oracle, and wf_<name>_temperature_k / wf_<name>_emissivity (one value per wavelength) equal finite differences
    of the radiance, for each emission source that sees the surface.  Emissivity per wavelength and as a scalar."""
    import sasktran2_amd as sk
    from oracle import frontend_oracle as fo
    c = sc.frontend_case(nwavel=5)
    c["wavelengths"] = np.linspace(9000.0, 11000.0, 5)
    c["tab_wl"] = np.linspace(8000.0, 12000.0, 111)
    c["tab_xs"] = c["tab_xs"] * 1e3
    c["hg_wl"] = np.linspace(8000.0, 12000.0, 6)
    cfg = sk.Config()
    cfg.num_singlescatter_moments = 4
    cfg.single_scatter_source = sk.SingleScatterSource.NoSource
    cfg.multiple_scatter_source = sk.MultipleScatterSource.NoSource
    geotype = sk.GeometryType.PlaneParallel
    if source == "twostream":
        cfg.num_streams = 2
        cfg.emission_source = sk.EmissionSource.TwoStream
    elif source == "discrete_ordinates":
        cfg.num_streams = 4
        cfg.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
        cfg.single_scatter_source = sk.SingleScatterSource.DiscreteOrdinates
        cfg.emission_source = sk.EmissionSource.DiscreteOrdinates
    else:
        cfg.emission_source = sk.EmissionSource.Standard
        geotype = sk.GeometryType.Spherical
    engine, geo = _engine_case(sk, c, cfg, [sk.GroundViewingSolar(0.6, 0.0, 0.9, 200000.0), sk.GroundViewingSolar(0.6, 0.5, 0.6, 200000.0)], geotype=geotype)
    emissivity = np.linspace(0.9, 0.98, 5)

    def radiance(t_surface, eps, derivs=False):
        # (the thermal pass of the discrete-ordinates engine does not serve scattering-group mappings: Rayleigh without state derivatives)
        state = source != "discrete_ordinates"
        atmo = sk.Atmosphere(geo, cfg, wavelengths_nm=c["wavelengths"], calculate_derivatives=derivs, pressure_derivative=state, temperature_derivative=state)
        atmo.pressure_pa, atmo.temperature_k = c["pressure"], c["temperature"]
        atmo["rayleigh"] = sk.constituent.Rayleigh()
        atmo["o3"] = sk.constituent.VMRAltitudeAbsorber(sk.optical.AbsorberDatabaseDim2(wavelength_nm=c["tab_wl"], param=c["tab_t"], xs=c["tab_xs"]),
                                                        c["o3_alt"], c["o3_vmr"], "extend")
        atmo["thermal"] = sk.constituent.ThermalEmission()
        atmo.surface.albedo[:] = 0.03
        atmo["ground"] = sk.constituent.SurfaceThermalEmission(t_surface, eps)
        return engine.calculate_radiance(atmo, as_dataset=False), atmo
    base, atmo = radiance(291.0, emissivity, True)
    want = fo.SurfaceThermalEmission(291.0, emissivity)
    np.testing.assert_allclose(atmo.surface.emission, want.emission(c["wavelengths"]), rtol=1e-14)
    again = engine.calculate_radiance(atmo, as_dataset=False)       # a rebuild starts from a surface that emits nothing
    assert np.array_equal(again["radiance"], base["radiance"])
    wf_t, wf_e = base["wf_ground_temperature_k"], base["wf_ground_emissivity"]
    assert wf_t.shape == base["radiance"].shape == wf_e.shape
    assert np.all(wf_t > 0) and np.all(wf_e > 0)
    h = 1e-2
    fd = (radiance(291.0 + h, emissivity)[0]["radiance"] - radiance(291.0 - h, emissivity)[0]["radiance"]) / (2 * h)
    np.testing.assert_allclose(wf_t, fd, rtol=2e-6, atol=1e-9 * np.max(np.abs(fd)))   # (opaque wavelengths: rounding of the difference)
    h = 1e-3
    for j in (0, 3):
        de = np.zeros(5)
        de[j] = h
        fd = (radiance(291.0, emissivity + de)[0]["radiance"] - radiance(291.0, emissivity - de)[0]["radiance"]) / (2 * h)
        np.testing.assert_allclose(wf_e[j], fd[j], rtol=1e-7, atol=1e-9 * np.max(np.abs(wf_e)))
        assert np.all(np.delete(fd, j, axis=0) == 0)
    # the surface mappings are those of the oracle; a scalar emissivity gives the same numbers as the same value on the grid
    for name, d_em in want.surface_mappings("ground", c["wavelengths"]).items():
        np.testing.assert_allclose(atmo._surface_mappings[name][3], d_em, rtol=1e-14)
    flat, _ = radiance(291.0, 0.95, True)
    grid, _ = radiance(291.0, np.full(5, 0.95), True)
    for key in ("radiance", "wf_ground_temperature_k", "wf_ground_emissivity"):
        assert np.array_equal(flat[key], grid[key])
    with pytest.raises(ValueError):
        radiance(291.0, np.ones(3))


def test_gaussian_height_extinction_constituent():
    """GaussianHeightExtinction (src/sasktran2/constituent/gaussianheight.py): storage and the three one-parameter mappings equal the
    oracle's; wf_<name>_height_m / _width_fwhm_m / _vertical_optical_depth equal finite differences of the radiance (the criterion of
    the reference's tests/constituent/test_gaussian_height.py), limb and nadir, discrete ordinates + exact single scatter"""
    import sasktran2_amd as sk
    c = sc.frontend_case(nwavel=6)
    cfg = _config(sk, streams=8, moments=8)
    cfg.single_scatter_source = sk.SingleScatterSource.Exact
    rays = [sk.TangentAltitudeSolar(15000.0, 0.3, 200000.0, 0.6), sk.TangentAltitudeSolar(24000.0, 0.3, 200000.0, 0.6),
            sk.GroundViewingSolar(0.6, 0.5, 0.8, 200000.0)]
    engine, geo = _engine_case(sk, c, cfg, rays)
    params = (17000.0, 5000.0, 0.08)

    def radiance(p, derivs):
        atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, calculate_derivatives=derivs, gaussian=p)
        atmo.surface.albedo[:] = 0.2
        return engine.calculate_radiance(atmo, as_dataset=False), atmo
    base, atmo = radiance(params, True)
    want = sc.frontend_oracle_atmosphere(c, 8, gaussian=params).assemble()
    st = atmo.storage
    _close(st.total_extinction, want.ext)
    _close(st.ssa, want.ssa)
    _close(st.leg_coeff, want.leg)
    for index, (key, step) in enumerate((("height_m", 2.0), ("width_fwhm_m", 2.0), ("vertical_optical_depth", 1e-5))):
        name = f"wf_layer_{key}"
        m, w = st.get_derivative_mapping(name), want.mappings[name]
        assert m.interp_dim == w.interp_dim == f"layer_{key}"
        _close(m.d_extinction, w.d_extinction)
        _close(m.d_ssa, w.d_ssa)
        _close(m.scat_factor, w.scat_factor)
        _close(m.d_leg_coeff, w.d_legendre)
        np.testing.assert_allclose(m.interpolator, w.interpolator, rtol=1e-13, atol=1e-300)
        assert base[name].shape == (1, 6, 3, 1)
        plus, minus = list(params), list(params)
        plus[index] += step
        minus[index] -= step
        fd = (radiance(tuple(plus), False)[0]["radiance"] - radiance(tuple(minus), False)[0]["radiance"]) / (2 * step)
        assert np.max(np.abs(fd)) > 0
        np.testing.assert_allclose(base[name][0], fd, rtol=3e-5, atol=3e-6 * np.max(np.abs(fd)), err_msg=key)


def test_air_mass_factor_constituent():
    """AirMassFactor (src/sasktran2/constituent/amf.py; the reference's tests/weightingfunctions/test_amf.py only runs it): the mapping
    arrays equal the oracle's, the output equals the oracle's assembly of the raw-interface weighting functions divided by the radiance, and
    it equals -(1 / layer thickness) d ln I / d(absorber extinction of a level) by finite differences of a grey probe absorber."""
    import sasktran2_amd as sk
    from oracle import frontend_oracle as fo
    c = sc.frontend_case(nwavel=5)
    cfg = _config(sk, streams=4, moments=8)
    rays = [sk.GroundViewingSolar(0.6, 0.0, 0.8, 200000.0), sk.TangentAltitudeSolar(14000.0, 0.2, 200000.0, 0.6)]
    engine, geo = _engine_case(sk, c, cfg, rays)
    ngrid = len(c["altitudes"])
    sigma = 1e-24

    def radiance(probe, derivs):
        atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, calculate_derivatives=derivs, pressure_derivative=False, temperature_derivative=False)
        atmo.surface.albedo[:] = 0.3
        atmo["amf"] = sk.constituent.AirMassFactor()
        if probe is not None:   # grey absorber on the model grid: extinction `probe` [1/m] per level
            grey = sk.optical.AbsorberDatabaseDim1(wavelength_nm=np.array([200.0, 2000.0]), xs=np.array([sigma, sigma]))
            air = atmo.state_equation.dry_air_numberdensity["N"]
            atmo["probe"] = sk.constituent.VMRAltitudeAbsorber(grey, c["altitudes"], probe / (sigma * air), "zero")
        return engine.calculate_radiance(atmo, as_dataset=False), atmo
    base, atmo = radiance(None, True)
    oracle = sc.frontend_oracle_atmosphere(c, 8, pressure_derivative=False, temperature_derivative=False)
    oracle.constituents["amf"] = fo.AirMassFactor()
    want = oracle.assemble()
    m, w = atmo.storage.get_derivative_mapping("air_mass_factor"), want.mappings["air_mass_factor"]
    assert m.log_radiance_space and m.interp_dim == "altitude"
    _close(m.d_extinction, w.d_extinction)
    _close(m.d_ssa, w.d_ssa)
    _close(atmo.storage.total_extinction, want.ext)      # the constituent itself adds nothing
    amf = base["air_mass_factor"]
    assert amf.shape == (ngrid, 5, 2, 1)
    # raw interface with the oracle's arrays -> native weighting functions -> the oracle's output assembly
    raw = sk.Atmosphere(geo, cfg, numwavel=5)
    raw.storage.total_extinction[:], raw.storage.ssa[:] = want.ext, want.ssa
    raw.leg_coeff.a1[:] = want.leg
    raw.surface.albedo[:] = 0.3
    ref = engine.calculate_radiance(raw, as_dataset=False)
    only = fo.Atmosphere(c["altitudes"], c["wavelengths"], 8)
    only.mappings = {"air_mass_factor": w}
    expect = fo.assemble_weighting_functions(only, {"ext": ref["wf_extinction"][..., 0], "ssa": ref["wf_ssa"][..., 0], "radiance": ref["radiance"][..., 0]})
    # (the two runs take different kernels -- scattering groups send the constituent run to the run-time-n ones -- and the extinction and
    # albedo terms of this mapping cancel where the air is thin: their rounding difference, 1e-12 of each term, is what is left)
    np.testing.assert_allclose(amf[..., 0], expect["air_mass_factor"], rtol=5e-6, atol=1e-8 * np.max(np.abs(amf)))
    # a nadir box air mass factor is positive and of the order of the geometric one, 1 / mu_sun + 1 / mu_view, where the air is thin
    assert 1.5 < amf[ngrid - 3, -1, 0, 0] < 1.0 / 0.6 + 1.0 / 0.8 + 1.5
    dz = np.gradient(c["altitudes"])
    dz[0] /= 2.0
    dz[-1] /= 2.0
    for q in (0, 7, 19):
        h = 0.02 * ((1.0 - want.ssa[q]) * want.ext[q]).min()      # (the minus side must leave an absorbing atmosphere at every wavelength)
        probe = np.zeros(ngrid)
        probe[q] = h
        plus, minus = radiance(probe, False)[0]["radiance"], radiance(-probe, False)[0]["radiance"]
        fd = -(plus - minus) / (2 * h) / base["radiance"] / dz[q]
        np.testing.assert_allclose(amf[q], fd, rtol=2e-5, atol=2e-6 * np.max(np.abs(fd)), err_msg=f"level {q}")


@pytest.mark.parametrize("nstokes", [1, 3])
def test_manual_constituent(nstokes):
    """Manual (src/sasktran2/constituent/manual.py, rust/.../manual.rs:25-111): extinction / single scatter albedo / phase moments given on
    the model grid add to the other constituents exactly as the oracle adds them; the other constituents' mappings see the totals; the
    arrays stay in HBM from one assembly to the next; delta_scale follows the reference's formula; shape errors are the reference's."""
    import sasktran2_amd as sk
    from oracle import frontend_oracle as fo
    c = sc.frontend_case(nwavel=9)
    cfg = _config(sk, nstokes=nstokes, moments=8)
    geo = _geometry(sk, c)
    rng = np.random.default_rng(11)
    ngrid, nleg = len(c["altitudes"]), 8 * (4 if nstokes == 3 else 1)
    ext = 2e-5 * np.exp(-((c["altitudes"][:, None] - 9000.0) / 2500.0) ** 2) * (1.0 + 0.2 * rng.random((ngrid, 9)))
    ssa = 0.6 + 0.39 * rng.random((ngrid, 9))
    leg = rng.random((nleg, ngrid, 9))
    leg[0] = 1.0
    atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg)
    atmo["cloud_layer"] = sk.constituent.Manual(ext, ssa, leg)
    fe = atmo.internal_object()
    oracle = sc.frontend_oracle_atmosphere(c, 8, nstokes)
    oracle.constituents["cloud_layer"] = fo.Manual(ext, ssa, leg)
    want = oracle.assemble()
    st = atmo.storage
    _close(st.total_extinction, want.ext)
    _close(st.ssa, want.ssa)
    _close(st.leg_coeff, want.leg)
    assert set(st.derivative_mapping_names()) == set(want.mappings)      # none of its own
    for name, w in want.mappings.items():       # (2e-12: next to a layer a thousand times denser than the gas the differences species - total cancel further)
        m = st.get_derivative_mapping(name)
        _close(m.d_extinction, w.d_extinction, 2e-12)
        _close(m.d_ssa, w.d_ssa, 2e-12)
        if w.d_legendre is not None:
            _close(m.d_leg_coeff, w.d_legendre, 2e-12)
            _close(m.scat_factor, w.scat_factor, 2e-12)
    nterms = fe._nterms
    fe = atmo.internal_object()
    assert fe.spectra_reused == nterms and nterms >= ngrid
    _close(atmo.storage.leg_coeff, want.leg)
    # without phase moments the layer scatters with no phase function of its own: it adds to extinction and scattering extinction only
    plain = sc.frontend_product_atmosphere(sk, c, geo, cfg)
    plain["cloud_layer"] = sk.constituent.Manual(ext, np.zeros_like(ext))
    plain.internal_object()
    oracle = sc.frontend_oracle_atmosphere(c, 8, nstokes)
    oracle.constituents["cloud_layer"] = fo.Manual(ext, np.zeros_like(ext))
    want0 = oracle.assemble()
    _close(plain.storage.total_extinction, want0.ext)
    _close(plain.storage.ssa, want0.ssa)
    _close(plain.storage.leg_coeff, want0.leg)
    # delta_scale (manual.py:87-100; four coefficients per order)
    if nstokes == 3:
        scaled = sk.constituent.Manual(ext, ssa, leg, delta_scale=True)
        f = leg[::4][-1] / (2.0 * 8 + 1)
        np.testing.assert_allclose(scaled.extinction, ext * (1.0 - ssa * f), rtol=1e-15)
        np.testing.assert_allclose(scaled.ssa, ssa * (1.0 - f) / (1.0 - ssa * f), rtol=1e-15)
        np.testing.assert_allclose(scaled.leg_coeff[4 * 3 + 1], (leg[4 * 3 + 1] - 7.0 * f) / (1.0 - f), rtol=1e-14)
        np.testing.assert_allclose(scaled.leg_coeff[4 * 3 + 3], leg[4 * 3 + 3] / (1.0 - f), rtol=1e-14)
        assert ext[3, 2] != scaled.extinction[3, 2]       # the caller's arrays are left alone
    with pytest.raises(ValueError, match="same shape"):
        sk.constituent.Manual(ext, ssa[:, :4])
    with pytest.raises(ValueError, match="legendre_moments must have shape"):
        sk.constituent.Manual(ext, ssa, leg[:, :3])
    with pytest.raises(ValueError, match="delta_scale"):
        sk.constituent.Manual(ext, ssa, delta_scale=True)
    bad = sc.frontend_product_atmosphere(sk, c, geo, cfg)
    bad["cloud_layer"] = sk.constituent.Manual(ext[:, :5], ssa[:, :5])
    with pytest.raises(ValueError, match="wavelength dimension"):
        bad.internal_object()
