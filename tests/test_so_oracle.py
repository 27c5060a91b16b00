"""CPU oracle of the exact single-scatter + legacy successive-orders sources (oracle/so_oracle.hpp) against the reference's
own pinned values and against physical properties.  No GPU needed."""
import numpy as np
import pytest

from tests import scenarios as sc


def make_so_oracle(s, **override):
    from oracle.oracle_py import SoOracleEngine
    g, c = s["geometry"], dict(s["config"], **override)
    return SoOracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"], s["rays"],
                          nstokes=c["nstokes"], stokes_basis=c.get("stokes_basis", 0), num_ss_moments=c["num_singlescatter_moments"],
                          num_ms_moments=c["num_streams"], num_incoming=c["num_incoming"], num_outgoing=c["num_outgoing"],
                          num_sza=c["num_sza"], num_iterations=c["num_iterations"], single_scatter={"none": 0, "exact": 1, "solartable": 2}[c["single_scatter"]],
                          multiple_scatter=c["multiple_scatter"] in ("successive_orders_legacy", "successive_orders"),
                          emission={"none": 0, "standard": 1, "volume_emission_rate": 2}[c.get("emission", "none")],
                          occultation=bool(c.get("occultation", False)),
                          # MultipleScatterSource::SuccessiveOrders: the fixed-point driver (config keys successive_orders_*)
                          fixed_point=(dict(relative_tolerance=c.get("relative_tolerance", 1e-6), absolute_tolerance=c.get("absolute_tolerance", 1e-12),
                                            damping=c.get("damping", 1.0), anderson_depth=c.get("anderson_depth", 3))
                                       if c["multiple_scatter"] == "successive_orders" else None))


def run_so_oracle(s, nthreads=2, **override):
    return make_so_oracle(s, **override).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], nthreads=nthreads)


def test_lebedev_generator_is_a_quadrature():
    """Weights sum to one and every rule integrates the spherical harmonics it is designed for (checked through x^2, x^4, x^2 y^2:
    1/3, 1/5, 1/15) -- the orbit parameters of sasktran2_amd/csrc/host/lebedev.hpp are the published Lebedev-Laikov ones."""
    import ctypes
    from oracle.oracle_py import lib, _dp
    L = lib()
    L.orso_lebedev.argtypes = [ctypes.c_int, _dp]
    for n in (6, 14, 26, 38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590, 770, 974):
        xyzw = np.zeros((n, 4))
        assert L.orso_lebedev(n, xyzw.ctypes.data_as(_dp)) == 0
        x, y, z, w = xyzw.T
        np.testing.assert_allclose(x * x + y * y + z * z, 1.0, atol=2e-15)
        assert abs(w.sum() - 1) < 2e-14
        assert abs((w * x * x).sum() - 1 / 3) < 1e-14
        if n >= 14:
            assert abs((w * x ** 4).sum() - 1 / 5) < 1e-14
            assert abs((w * x * x * y * y).sum() - 1 / 15) < 1e-14
    # the point order is the one the reference's interpolation tie-breaks depend on: first orbit = the six axis points
    assert np.array_equal(xyzw[:6, :3], [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])


def test_successive_orders_regression_pin():
    """tests/engine/test_1d_solver_regression.py:140-240: NSTOKES = 3 Stokes vectors of 4 rays x 2 wavelengths at rtol 5e-7."""
    s = sc.regression_so()
    got = run_so_oracle(s)
    np.testing.assert_allclose(got["radiance"], s["expected_radiance"], rtol=s["rtol"], atol=s["atol"])
    np.testing.assert_allclose(got["los_optical_depth"], s["expected_od"], rtol=5e-7, atol=1e-13)


def test_scalar_intensity_is_close_to_the_polarised_one_and_orders_add_up():
    s3, s1 = sc.regression_so(), sc.regression_so(nstokes=1)
    v, sca = run_so_oracle(s3), run_so_oracle(s1)
    # neglecting polarisation changes Rayleigh-like intensities by a few per cent at most
    assert np.max(np.abs(sca["radiance"][..., 0] / v["radiance"][..., 0] - 1)) < 0.05
    # every further order of scattering adds a positive, shrinking amount
    series = [run_so_oracle(s1, num_iterations=n)["radiance"][..., 0] for n in (0, 1, 2, 3, 6)]
    steps = [b - a for a, b in zip(series[:-1], series[1:])]
    assert all(np.all(d > 0) for d in steps)
    assert np.all(steps[1] < steps[0]) and np.all(steps[2] < steps[1])
    # no multiple-scatter source at all == zero iterations
    only_ss = run_so_oracle(s1, multiple_scatter="none")
    np.testing.assert_allclose(only_ss["radiance"], series[0][..., None], rtol=1e-14)


def test_single_scatter_is_linear_in_the_solar_irradiance_and_black_without_scatterers():
    s = sc.regression_so(nstokes=1)
    o = make_so_oracle(s)
    one = o.run(s["extinction"], s["ssa"], s["leg"], s["albedo"])["radiance"]
    two = o.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], solar_irradiance=np.full(2, 2.5))["radiance"]
    np.testing.assert_allclose(two, 2.5 * one, rtol=1e-13)
    dark = o.run(s["extinction"], np.zeros_like(s["ssa"]), s["leg"], np.zeros(2))["radiance"]
    assert np.all(dark == 0)


def test_successive_orders_converges_to_the_discrete_ordinates_field():
    """Independent check of the whole diffuse machinery: many orders of scattering on 110 directions approach the
    discrete-ordinates multiple-scatter solution of the same scene (different discretisations: agreement to a few per cent; measured 2-5.4 %)."""
    from tests.test_oracle_golden import run_oracle
    s = sc.regression_so(nstokes=1)
    s["config"].update(num_incoming=110, num_outgoing=110, num_iterations=25)
    so = run_so_oracle(s)["radiance"][..., 0]
    d = sc.regression_1d(2)
    d["config"]["nstr"] = 16
    do = run_oracle(d, derivatives=False)["radiance"]
    ss = run_so_oracle(s, multiple_scatter="none")["radiance"][..., 0]
    # in spherical geometry the reference's DO source (and the oracle's) delivers the multiple-scatter term only
    ms_so = so - ss
    assert np.max(np.abs(ms_so / do - 1)) < 0.07


@pytest.mark.parametrize("nstokes", [1, 3])
def test_exact_single_scatter_weighting_functions_match_finite_differences(nstokes):
    """The reference's own criterion for weighting functions (tests/weightingfunctions/test_lowlevel.py:318-486, src/sasktran2/test_util/wf.py:9-24):
    analytic == central finite differences, relative to the largest value of the profile."""
    import copy
    s = sc.regression_so(nstokes=nstokes)
    o = make_so_oracle(s, multiple_scatter="none")
    r = o.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    for name, key in (("extinction", "d_extinction"), ("ssa", "d_ssa")):
        scale = np.abs(r[key]).max(axis=3)
        scale = np.where(scale > 0, scale, 1.0)
        for q in (0, 3, 7, 12, 20, 24):
            h = 1e-5 * s[name][q]
            plus, minus = copy.deepcopy(s), copy.deepcopy(s)
            plus[name][q] += h
            minus[name][q] -= h
            fd = (o.run(plus["extinction"], plus["ssa"], plus["leg"], plus["albedo"])["radiance"]
                  - o.run(minus["extinction"], minus["ssa"], minus["leg"], minus["albedo"])["radiance"]) / (2 * h[:, None, None])
            assert np.max(np.abs(fd - r[key][..., q]) / scale) < 5e-6, (name, q)
    h = 1e-6
    fd = (o.run(s["extinction"], s["ssa"], s["leg"], s["albedo"] + h)["radiance"] - o.run(s["extinction"], s["ssa"], s["leg"], s["albedo"] - h)["radiance"]) / (2 * h)
    np.testing.assert_allclose(r["d_albedo"], fd, atol=1e-9 * np.abs(r["radiance"]).max() / h * 1e-3 + 1e-10)


def test_emission_and_occultation_sources_have_their_closed_forms():
    """Line-of-sight sources without scattering (cpp/lib/emission/emission_source.cpp, cpp/lib/solar/occultation.cpp): an occultation gives
    the transmission e^-tau of rays that leave the atmosphere; an isothermal, non-scattering atmosphere emits B (1 - e^-tau) (+ the
    attenuated surface emission behind rays that end on the ground); a uniform volume emission rate in a transparent atmosphere
    integrates to rate x path length."""
    s = sc.regression_so(nstokes=1)
    nothing = dict(single_scatter="none", multiple_scatter="none")
    od = make_so_oracle(s, **nothing, occultation=True).run(s["extinction"], s["ssa"], s["leg"], s["albedo"])
    hits_ground = np.array([True, True, False, False])
    np.testing.assert_allclose(od["radiance"][:, ~hits_ground, 0], np.exp(-od["los_optical_depth"][:, ~hits_ground]), rtol=1e-13)
    assert np.all(od["radiance"][:, hits_ground, 0] == 0)
    B, Bs = 3.0e-4, 7.0e-4
    em = make_so_oracle(s, **nothing, emission="standard").run(s["extinction"], np.zeros_like(s["ssa"]), s["leg"], s["albedo"],
                                                              emission=np.full_like(s["ssa"], B), surface_emission=np.full(2, Bs))
    tau = em["los_optical_depth"]
    want = B * (1 - np.exp(-tau)) + np.where(hits_ground[None, :], Bs * np.exp(-tau), 0.0)
    np.testing.assert_allclose(em["radiance"][..., 0], want, rtol=1e-12)
    # scattering removes (1 - ssa) of the thermal source
    half = make_so_oracle(s, **nothing, emission="standard").run(s["extinction"], np.full_like(s["ssa"], 0.25), s["leg"], s["albedo"],
                                                                emission=np.full_like(s["ssa"], B), surface_emission=np.zeros(2))
    np.testing.assert_allclose(half["radiance"][..., 0], 0.75 * B * (1 - np.exp(-tau)), rtol=1e-12)
    thin = make_so_oracle(s, **nothing, emission="volume_emission_rate").run(s["extinction"] * 1e-12, s["ssa"], s["leg"], s["albedo"],
                                                                            emission=np.full_like(s["ssa"], 2.0), surface_emission=np.zeros(2))
    alt = s["geometry"]["altitudes"]
    R, top = s["geometry"]["earth_radius"], alt[-1]
    # nadir rays: path = chord from the ground to the top of the atmosphere at the viewing zenith angle
    for r, cvz in ((0, 0.32), (1, 0.78)):
        path = -R * cvz + np.sqrt((R * cvz) ** 2 + 2 * R * top + top * top)
        np.testing.assert_allclose(thin["radiance"][:, r, 0], 2.0 * path, rtol=1e-9)


@pytest.mark.parametrize("config", [dict(emission="standard"), dict(emission="volume_emission_rate"), dict(occultation=True),
                                    dict(emission="standard", single_scatter="exact")],
                         ids=["standard emission", "volume emission rate", "occultation", "emission + exact single scatter"])
def test_emission_and_occultation_weighting_functions_match_finite_differences(config):
    """Weighting functions of the line-of-sight sources without scattering (EmissionSource::integrated_source / end_of_ray_source with their
    derivative rows, cpp/lib/emission/emission_source.cpp:100-173; the occultation source is differentiated through the path transmission):
    analytic == central finite differences with respect to extinction, single scatter albedo, the emission profile and the surface emission."""
    s = sc.regression_so(nstokes=1)
    base = dict(single_scatter="none", multiple_scatter="none")
    base.update(config)
    eng = make_so_oracle(s, **base)
    nw = s["extinction"].shape[1]
    alt = s["geometry"]["altitudes"]
    em = (2.0e-4 + 1.0e-4 * np.exp(-alt / 20_000.0))[:, None] * np.linspace(1.0, 1.3, nw)[None, :]
    sem = np.linspace(5e-4, 7e-4, nw)
    args = dict(extinction=s["extinction"], ssa=s["ssa"], emission=em, surface_emission=sem)

    def run(derivatives=False, **over):
        a = dict(args, **over)
        return eng.run(a["extinction"], a["ssa"], s["leg"], s["albedo"], derivatives=derivatives, emission=a["emission"],
                       surface_emission=a["surface_emission"])

    ref = run(True)
    scale = np.max(np.abs(ref["radiance"]))
    for name, key in (("d_extinction", "extinction"), ("d_ssa", "ssa"), ("d_emission", "emission")):
        for q in (0, 3, 11, 24):
            x = np.array(args[key], copy=True)
            h = (0.1 if key == "emission" else 1e-5) * x[q, 1] if x[q, 1] != 0 else 1e-9   # (the radiance is linear in the emission: any step)
            xp, xm = x.copy(), x.copy()
            xp[q, 1] += h
            xm[q, 1] -= h
            fd = (run(**{key: xp})["radiance"][1] - run(**{key: xm})["radiance"][1]) / (2 * h)
            an = ref[name][1, :, :, q]
            assert np.max(np.abs(fd - an)) <= 2e-6 * max(np.max(np.abs(an)), scale / abs(x[q, 1]) * 1e-6), (name, q)
    h = 1e-4
    fd = (run(surface_emission=sem + h)["radiance"] - run(surface_emission=sem - h)["radiance"]) / (2 * h)
    np.testing.assert_allclose(ref["d_surface_emission"], fd, rtol=1e-6, atol=1e-9)
    if "emission" not in config:
        assert np.all(ref["d_emission"] == 0) and np.all(ref["d_surface_emission"] == 0)


def test_solar_table_single_scatter_is_close_to_the_exact_source():
    # SingleScatterSource<SolarTransmissionTable>: 100 cosines x the altitude grid, bilinear in optical depth (solartransmissiontable.cpp:113-238)
    s = sc.regression_so(nstokes=1)
    exact = make_so_oracle(s, multiple_scatter="none").run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    table = make_so_oracle(s, multiple_scatter="none", single_scatter="solartable").run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    np.testing.assert_allclose(table["radiance"], exact["radiance"], rtol=2e-4)
    assert np.max(np.abs(table["radiance"] / exact["radiance"] - 1)) > 1e-9
    assert np.max(np.abs(table["d_extinction"] - exact["d_extinction"])) < 1e-3 * np.max(np.abs(exact["d_extinction"]))
    # analytic == finite differences for the tabulated source as well
    k = s["extinction"]
    eng = make_so_oracle(s, multiple_scatter="none", single_scatter="solartable")
    for q in (2, 9, 17):
        h = 1e-5 * k[q, 0]
        kp, km = k.copy(), k.copy()
        kp[q, 0] += h
        km[q, 0] -= h
        fd = (eng.run(kp, s["ssa"], s["leg"], s["albedo"])["radiance"][0] - eng.run(km, s["ssa"], s["leg"], s["albedo"])["radiance"][0]) / (2 * h)
        assert np.max(np.abs(fd - table["d_extinction"][0, :, :, q])) < 1e-6 * np.max(np.abs(table["d_extinction"][0]))


# ---- MultipleScatterSource::SuccessiveOrders: the reference's fixed-point / Anderson driver (cpp/lib/successive_orders/fixed_point.h) on
# the operators of the legacy discretisation
def test_fixed_point_without_mixing_is_the_legacy_iteration():
    """depth 0, damping 1, zero tolerances: every step is x <- S (first + T x), i.e. one legacy pass -- bit for bit"""
    s = sc.regression_so()
    legacy = run_so_oracle(s, num_iterations=4)
    fp = run_so_oracle(s, num_iterations=4, multiple_scatter="successive_orders", relative_tolerance=0.0, absolute_tolerance=0.0, anderson_depth=0)
    np.testing.assert_array_equal(fp["radiance"], legacy["radiance"])


def test_anderson_fixed_point_reaches_the_limit_of_the_orders_sooner():
    """the converged field is the limit of the successive orders; Anderson(3) gets there in fewer applications of the map"""
    s = sc.regression_so()
    limit = run_so_oracle(s, num_iterations=60)["radiance"]
    scale = np.abs(limit[..., 0]).max()
    tight = run_so_oracle(s, num_iterations=60, multiple_scatter="successive_orders", relative_tolerance=1e-11, absolute_tolerance=0.0)["radiance"]
    assert np.max(np.abs(tight - limit)) < 1e-9 * scale
    errs = {}
    for n in (6, 10):
        errs["legacy", n] = np.max(np.abs(run_so_oracle(s, num_iterations=n)["radiance"] - limit)) / scale
        errs["anderson", n] = np.max(np.abs(run_so_oracle(s, num_iterations=n, multiple_scatter="successive_orders", relative_tolerance=0.0,
                                                          absolute_tolerance=0.0)["radiance"] - limit)) / scale
    assert errs["anderson", 10] < 0.1 * errs["legacy", 10], errs
    assert errs["anderson", 10] < errs["anderson", 6], errs


def test_fixed_iteration_solution_agrees_with_the_legacy_source():
    """the reference's own pin of the new source (tests/engine/test_successive_orders_cpp.py:158-186, its scene and its settings: three
    iterations, zero tolerances, Anderson depth 0, black surface): new and legacy source agree within rtol 2e-3 / atol 2e-7.  Here both run
    on the same discretisation, so a fixed number of unmixed steps IS the legacy iteration and the two agree to rounding."""
    for nstokes in (1, 3):
        new = run_so_oracle(sc.successive_orders_reference_scene(nstokes, 3, "successive_orders"))["radiance"]
        legacy = run_so_oracle(sc.successive_orders_reference_scene(nstokes, 3, "successive_orders_legacy"))["radiance"]
        np.testing.assert_allclose(new, legacy, rtol=2e-3, atol=2e-7)
        np.testing.assert_allclose(new, legacy, rtol=1e-13, atol=1e-18)
        assert np.all(new[..., 0] > 1e-6)
        if nstokes == 3:
            assert np.any(np.abs(new[..., 1]) > 1e-8) and np.any(np.abs(new[..., 2]) > 1e-8)


def test_zero_tolerances_use_the_iteration_count_and_a_huge_tolerance_stops_at_once():
    """tests/engine/test_successive_orders_cpp.py:225-247"""
    one = run_so_oracle(sc.successive_orders_reference_scene(1, 1, albedo=0.12))["radiance"]
    two = run_so_oracle(sc.successive_orders_reference_scene(1, 2, albedo=0.12))["radiance"]
    early = run_so_oracle(sc.successive_orders_reference_scene(1, 8, albedo=0.12, absolute_tolerance=1e100))["radiance"]
    np.testing.assert_allclose(early, one, rtol=2e-13, atol=2e-14)
    assert np.linalg.norm(two - one) > 1e-7 * np.linalg.norm(one)


# ---------------------------------------------------------------------------------------------------------------------------
# Weighting functions through the diffuse field (oracle/so_oracle.hpp, ms_adjoint): the discrete adjoint of the passes against central
# finite differences of the oracle's own radiances.  (The reference's legacy source has no weighting functions to compare with.)
# ---------------------------------------------------------------------------------------------------------------------------
def _fd_check(s, engine, base, tol):
    k, w = s["extinction"], s["ssa"]
    run = lambda kk, ww, aa: engine.run(kk, ww, s["leg"], aa, nthreads=4)["radiance"]   # noqa: E731
    for key, arr, which in (("d_extinction", k, 0), ("d_ssa", w, 1)):
        scale = np.abs(base[key]).max(axis=-1)
        for q in range(0, k.shape[0] - 6, 4):   # (the top levels carry too little extinction for a clean difference quotient)
            h = 1e-6 * arr[q].max() if which == 0 else 1e-6
            up, dn = arr.copy(), arr.copy()
            up[q] += h
            dn[q] -= h
            fd = (run(up, w, s["albedo"]) - run(dn, w, s["albedo"])) / (2 * h) if which == 0 else (run(k, up, s["albedo"]) - run(k, dn, s["albedo"])) / (2 * h)
            assert np.max(np.abs(fd - base[key][..., q]) / scale) < tol, (key, q)
    h = 1e-6
    fd = (run(k, w, s["albedo"] + h) - run(k, w, s["albedo"] - h)) / (2 * h)
    np.testing.assert_allclose(base["d_albedo"], fd, rtol=1e-6, atol=1e-9 * np.abs(fd).max())


@pytest.mark.parametrize("nstokes", [1, 3])
def test_diffuse_field_adjoint_against_finite_differences(nstokes):
    s = sc.regression_so(nstokes=nstokes)
    s["albedo"] = np.full_like(s["albedo"], 0.3)
    e = make_so_oracle(s, num_iterations=3)
    base = e.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], nthreads=4, derivatives=True)
    _fd_check(s, e, base, 5e-6)


def test_diffuse_field_adjoint_of_the_anderson_driver_against_finite_differences():
    s = sc.regression_so(nstokes=1)
    s["albedo"] = np.full_like(s["albedo"], 0.3)
    e = make_so_oracle(s, num_iterations=60, multiple_scatter="successive_orders", relative_tolerance=1e-10, absolute_tolerance=0.0, anderson_depth=3)
    base = e.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], nthreads=4, derivatives=True)
    _fd_check(s, e, base, 5e-6)
