"""GPU parity of the exact single-scatter + successive-orders path (Python mirror -> C ABI -> HIP kernels on cuda:0) against the
reference's pinned Stokes vectors and against the CPU oracle (oracle/so_oracle.hpp).  Tolerance: BASELINE.json's 1e-6 relative on
Stokes I; Q and U to 1e-6 of the intensity."""
import numpy as np
import pytest

from tests import scenarios as sc
from tests.test_so_oracle import run_so_oracle

pytestmark = pytest.mark.gpu


def run_product_so(*a, **k):
    from tests.gpu_harness import run_product_so as rp
    return rp(*a, **k)


def assert_so_parity(got, ref, rtol=1e-6):
    np.testing.assert_allclose(got["los_optical_depth"], ref["los_optical_depth"], rtol=1e-12)
    np.testing.assert_allclose(got["radiance"][..., 0], ref["radiance"][..., 0], rtol=rtol, atol=1e-300)
    if got["radiance"].shape[-1] == 3:
        scale = np.abs(ref["radiance"][..., :1])
        assert np.max(np.abs(got["radiance"][..., 1:] - ref["radiance"][..., 1:]) / scale) < rtol


def test_reference_regression_pin_on_the_device():
    """tests/engine/test_1d_solver_regression.py:140-240 (NSTOKES = 3, exact single scatter + legacy successive orders) through the C ABI."""
    s = sc.regression_so()
    got = run_product_so(s)
    np.testing.assert_allclose(got["radiance"], s["expected_radiance"], rtol=s["rtol"], atol=s["atol"])
    np.testing.assert_allclose(got["los_optical_depth"], s["expected_od"], rtol=5e-7, atol=1e-13)
    assert_so_parity(got, run_so_oracle(s))


@pytest.mark.parametrize("nstokes", [1, 3])
@pytest.mark.parametrize("sources", [("exact", "none"), ("none", "successive_orders_legacy"), ("exact", "successive_orders_legacy")],
                         ids=["single scatter only", "multiple scatter only", "both"])
def test_source_combinations_match_oracle(nstokes, sources):
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=5)
    o = dict(single_scatter=sources[0], multiple_scatter=sources[1])
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


@pytest.mark.parametrize("num_sza,ndir,niter", [(1, 26, 1), (1, 50, 4), (3, 14, 2), (1, 110, 6)])
def test_discretisations_match_oracle(num_sza, ndir, niter):
    s = sc.regression_so(nstokes=1, num_wavelengths=3)
    o = dict(num_sza=num_sza, num_incoming=ndir, num_outgoing=ndir, num_iterations=niter)
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


def test_different_incoming_and_outgoing_spheres_polarised():
    s = sc.regression_so(nstokes=3, num_wavelengths=2)
    o = dict(num_incoming=38, num_outgoing=26, num_iterations=2, num_sza=1)
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


def test_c5_shape_limb_scene_matches_oracle_and_is_block_invariant(monkeypatch):
    """BASELINE configs[4] geometry at a reduced wavelength count: 50 layers, 25 limb rays, 110 / 110 directions, 10 orders, aerosol
    phase functions with all 16 moments; a spectrum that does not fill the last wavefront; blocks of wavelengths give identical bits."""
    s = sc.limb_c5(num_wavelengths=200)
    ref = run_so_oracle(s, nthreads=8)
    got = run_product_so(s)
    assert_so_parity(got, ref)
    monkeypatch.setenv("SASKTRAN2_AMD_SO_CHUNK", "128")
    blocks = run_product_so(s)
    assert np.array_equal(blocks["radiance"], got["radiance"])


def test_solar_irradiance_scales_the_radiance():
    from tests.gpu_harness import build_so_engine
    s = sc.regression_so(nstokes=1)
    eng = build_so_engine(s)
    one = run_product_so(s, engine=eng)["radiance"]
    two = run_product_so(s, engine=eng, solar_irradiance=np.array([2.0, 3.0]))["radiance"]
    np.testing.assert_allclose(two, one * np.array([2.0, 3.0])[:, None, None], rtol=1e-13)


def test_exact_single_scatter_on_top_of_discrete_ordinates():
    """single_scatter_source = exact with multiple_scatter_source = discrete_ordinates: in spherical geometry the DO source gives the
    multiple-scatter term, the exact source the single scatter (engine.cpp:217-230, 280-369)."""
    from tests.test_oracle_golden import run_oracle
    s = sc.regression_so(nstokes=1)
    got = run_product_so(s, multiple_scatter="discrete_ordinates")["radiance"][..., 0]
    ss = run_so_oracle(s, multiple_scatter="none")["radiance"][..., 0]
    d = sc.regression_1d(2)
    do = run_oracle(d, derivatives=False)["radiance"]
    np.testing.assert_allclose(got, ss + do, rtol=1e-6)


def assert_so_wf_parity(got, ref, tol=1e-4):
    """Weighting functions relative to the largest one of their profile (the reference's criterion, BASELINE tolerance 1e-4)."""
    for key in ("d_extinction", "d_ssa"):
        scale = np.abs(ref[key]).max(axis=3, keepdims=True)
        # a Stokes component that vanishes in the chosen basis (rounding-level profile) is compared on the scale of the intensity's profile
        scale = np.maximum(scale, 1e-6 * scale[:, :, :1])
        assert np.max(np.abs(got[key] - ref[key]) / scale) < tol, key
    np.testing.assert_allclose(got["d_albedo"], ref["d_albedo"], rtol=tol, atol=1e-12 * np.abs(ref["radiance"]).max())


@pytest.mark.parametrize("nstokes,basis", [(1, 0), (3, 0), (3, 1), (3, 2)], ids=["scalar", "standard basis", "solar basis", "observer basis"])
def test_exact_single_scatter_weighting_functions_match_oracle(nstokes, basis):
    """single_scatter_source = exact alone: wf_extinction / wf_ssa / wf_albedo (reverse mode on the device) against the oracle's dense
    forward-mode linearisation, which matches central finite differences to 1e-6 (tests/test_so_oracle.py)."""
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=3)
    s["config"]["stokes_basis"] = basis
    o = dict(multiple_scatter="none")
    from tests.test_so_oracle import make_so_oracle
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref)


def test_exact_single_scatter_weighting_functions_on_the_c5_limb_geometry():
    s = sc.limb_c5(num_wavelengths=130)
    o = dict(multiple_scatter="none")
    from tests.test_so_oracle import make_so_oracle
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=8)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref)


@pytest.mark.parametrize("nstokes", [1, 3])
def test_weighting_functions_of_discrete_ordinates_plus_exact_single_scatter(nstokes):
    """multiple_scatter_source = discrete_ordinates with single_scatter_source = exact: native weighting functions of both sources are
    added on the device before the user mappings; checked against oracle(DO, dense forward mode) + oracle(exact single scatter)."""
    from tests.test_oracle_golden import run_oracle
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=2)
    got = run_product_so(s, derivatives=True, multiple_scatter="discrete_ordinates")
    ss = make_so_oracle(s, multiple_scatter="none").run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    d = sc.regression_1d(2)
    if nstokes == 3:
        d = dict(d, leg=s["leg"], config=dict(d["config"], nstokes=3))
    do = run_oracle(d, derivatives=True)
    shape = ss["radiance"].shape
    ref = {"radiance": ss["radiance"] + do["radiance"].reshape(shape), "los_optical_depth": ss["los_optical_depth"],
           "d_extinction": ss["d_extinction"] + do["d_extinction"].reshape(ss["d_extinction"].shape),
           "d_ssa": ss["d_ssa"] + do["d_ssa"].reshape(ss["d_ssa"].shape), "d_albedo": ss["d_albedo"] + do["d_albedo"].reshape(shape)}
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref)


@pytest.mark.parametrize("nstokes", [1, 3])
@pytest.mark.parametrize("emission", ["standard", "volume_emission_rate"])
def test_emission_and_occultation_sources_match_oracle(nstokes, emission):
    """emission_source = standard / volume_emission_rate and occultation_source = standard next to the exact single scatter and the
    successive orders (cpp/lib/emission/emission_source.cpp, cpp/lib/solar/occultation.cpp): the line-of-sight terms add up."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=3)
    alt = s["geometry"]["altitudes"]
    em = 3.0e-4 * np.exp(-alt / 12_000.0)[:, None] * np.linspace(0.7, 1.4, 3)[None, :]
    sem = np.linspace(1.0e-4, 7.0e-4, 3)
    o = dict(emission=emission, occultation=True)
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], emission=em, surface_emission=sem)
    got = run_product_so(s, emission_source=em, surface_emission=sem, **o)
    assert_so_parity(got, ref)
    # the sources alone (no scattering source at all): thermal emission + transmission
    alone = dict(o, single_scatter="none", multiple_scatter="none")
    ref = make_so_oracle(s, **alone).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], emission=em, surface_emission=sem)
    got = run_product_so(s, emission_source=em, surface_emission=sem, **alone)
    assert_so_parity(got, ref)
    # the line-of-sight sources are linearised (emission_source.cpp:100-173): extinction, ssa, emission profile, surface emission
    ref = make_so_oracle(s, **alone).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], emission=em, surface_emission=sem, derivatives=True)
    got = run_product_so(s, emission_source=em, surface_emission=sem, derivatives=True, **alone)
    assert_so_parity(got, ref)
    for k in ("d_extinction", "d_ssa", "d_emission"):
        scale = np.maximum(np.abs(ref[k]).max(axis=3, keepdims=True), 1e-300)
        assert np.max(np.abs(got[k] - ref[k]) / scale) < 1e-10, k
    np.testing.assert_allclose(got["d_surface_emission"], ref["d_surface_emission"], rtol=1e-12, atol=1e-300)
    # ... and next to the exact single scatter
    both = dict(o, multiple_scatter="none")
    ref = make_so_oracle(s, **both).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], emission=em, surface_emission=sem, derivatives=True)
    got = run_product_so(s, emission_source=em, surface_emission=sem, derivatives=True, **both)
    assert_so_parity(got, ref)
    for k in ("d_extinction", "d_ssa", "d_emission"):
        scale = np.maximum(np.abs(ref[k]).max(axis=3, keepdims=True), 1e-6 * np.abs(ref["d_extinction"]).max())
        assert np.max(np.abs(got[k] - ref[k]) / scale) < 1e-9, k
    np.testing.assert_allclose(got["d_albedo"], ref["d_albedo"], rtol=1e-10, atol=1e-300)
    # ... and with the successive-orders diffuse field on top (its discrete adjoint, so_adjoint.inc)
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], emission=em, surface_emission=sem, derivatives=True)
    got = run_product_so(s, emission_source=em, surface_emission=sem, derivatives=True, **o)
    assert_so_parity(got, ref)
    for k in ("d_extinction", "d_ssa", "d_emission"):
        scale = np.maximum(np.abs(ref[k]).max(axis=3, keepdims=True), 1e-6 * np.abs(ref["d_extinction"]).max())
        assert np.max(np.abs(got[k] - ref[k]) / scale) < 1e-6, k
    np.testing.assert_allclose(got["d_albedo"], ref["d_albedo"], rtol=1e-6, atol=1e-300)


@pytest.mark.parametrize("nstokes", [1, 3])
def test_solar_table_single_scatter(nstokes):
    """single_scatter_source = solartable (SingleScatterSource<SolarTransmissionTable>, cpp/lib/solar/solartransmissiontable.cpp:113-238): the
    solar optical depths of the line-of-sight boundaries come from the (altitude, cos sza) table the successive-orders source uses for its own
    first order (pinned with it on the reference's regression) instead of a ray trace per boundary: device == oracle, and both within the
    table's interpolation error of the exact source."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=3)
    o = dict(single_scatter="solartable", multiple_scatter="none")
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    for k in ("d_extinction", "d_ssa"):
        scale = np.maximum(np.abs(ref[k]).max(axis=3, keepdims=True), 1e-6 * np.abs(ref[k]).max())
        assert np.max(np.abs(got[k] - ref[k]) / scale) < 1e-9, k
    exact = run_product_so(s, single_scatter="exact", multiple_scatter="none")
    np.testing.assert_allclose(got["radiance"][..., 0], exact["radiance"][..., 0], rtol=2e-4)
    assert np.max(np.abs(got["radiance"][..., 0] / exact["radiance"][..., 0] - 1)) > 1e-9   # (it IS another source)
    # with the successive orders on top
    both = dict(single_scatter="solartable", multiple_scatter="successive_orders_legacy")
    assert_so_parity(run_product_so(s, **both), make_so_oracle(s, **both).run(s["extinction"], s["ssa"], s["leg"], s["albedo"]))


# ---- MultipleScatterSource.SuccessiveOrders (enum 4): the reference's fixed-point / Anderson driver (cpp/lib/successive_orders/fixed_point.h)
# on the operators of the legacy discretisation (k_so_fixed_point)
@pytest.mark.parametrize("nstokes", [1, 3])
def test_new_source_at_fixed_iterations_agrees_with_the_legacy_source(nstokes):
    """the reference's pin of its new source (tests/engine/test_successive_orders_cpp.py:158-186: its scene, three iterations, zero
    tolerances, Anderson depth 0, black surface; rtol 2e-3, atol 2e-7).  On this build's shared discretisation unmixed steps ARE the legacy
    passes: identical bits."""
    new = run_product_so(sc.successive_orders_reference_scene(nstokes, 3, "successive_orders"))
    legacy = run_product_so(sc.successive_orders_reference_scene(nstokes, 3, "successive_orders_legacy"))
    np.testing.assert_allclose(new["radiance"], legacy["radiance"], rtol=2e-3, atol=2e-7)
    np.testing.assert_array_equal(new["radiance"], legacy["radiance"])
    assert np.all(np.isfinite(new["radiance"])) and np.min(new["radiance"][..., 0]) > 1e-6
    assert_so_parity(new, run_so_oracle(sc.successive_orders_reference_scene(nstokes, 3, "successive_orders")))


@pytest.mark.parametrize("case", [dict(anderson_depth=3, num_iterations=6), dict(anderson_depth=2, damping=0.8, num_iterations=7),
                                  dict(anderson_depth=3, relative_tolerance=1e-7, num_iterations=40),
                                  dict(anderson_depth=0, relative_tolerance=1e-5, num_iterations=40)],
                         ids=["anderson 3, six steps", "anderson 2 damped", "anderson 3 to 1e-7", "plain to 1e-5"])
def test_fixed_point_driver_matches_the_oracle(case):
    """Anderson mixing, damping and the per-wavelength stop rule on the device against the oracle's restatement of the reference's solver
    (mixing amplifies rounding differences of the two implementations: 1e-6 relative still holds with margin)"""
    s = sc.regression_so(nstokes=1, num_wavelengths=5)
    o = dict(multiple_scatter="successive_orders", relative_tolerance=0.0, absolute_tolerance=0.0)
    o.update(case)
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


def test_zero_tolerances_use_the_iteration_count_and_a_huge_tolerance_stops_after_one_step():
    """tests/engine/test_successive_orders_cpp.py:225-247"""
    one = run_product_so(sc.successive_orders_reference_scene(1, 1, albedo=0.12))["radiance"]
    two = run_product_so(sc.successive_orders_reference_scene(1, 2, albedo=0.12))["radiance"]
    early = run_product_so(sc.successive_orders_reference_scene(1, 8, albedo=0.12, absolute_tolerance=1e100))["radiance"]
    np.testing.assert_allclose(early, one, rtol=2e-13, atol=2e-14)
    assert np.linalg.norm(two - one) > 1e-7 * np.linalg.norm(one)


def test_anderson_converges_to_the_limit_of_the_orders_and_settings_are_validated():
    import sasktran2_amd as sk
    from tests.gpu_harness import build_so_engine
    s = sc.regression_so(nstokes=1, num_wavelengths=4)
    limit = run_product_so(s, num_iterations=60)["radiance"]
    got = run_product_so(s, num_iterations=60, multiple_scatter="successive_orders", relative_tolerance=1e-10, absolute_tolerance=0.0)["radiance"]
    assert np.max(np.abs(got / limit - 1)) < 1e-8
    with pytest.raises(RuntimeError):   # damping outside (0, 1] is refused when the engine is built (fixed_point.h:19-39)
        build_so_engine(s, multiple_scatter="successive_orders", damping=1.5)


def test_full_size_c5_properties_with_the_new_source():
    """BASELINE configs[4] at full size with multiple_scatter_source = successive_orders (10 fixed-point steps, Anderson depth 3, zero
    tolerances): finite, positive, smooth in wavelength; a subset of wavelengths solved alone agrees bit for bit (blocks and slabs do not
    couple wavelengths); oracle spot check"""
    nw = 20_000
    s = sc.limb_c5(num_wavelengths=nw)
    o = dict(multiple_scatter="successive_orders", relative_tolerance=0.0, absolute_tolerance=0.0, anderson_depth=3)
    from tests.gpu_harness import build_so_engine
    eng = build_so_engine(s, **o)
    full = run_product_so(s, engine=eng)["radiance"]
    assert full.shape == (nw, 25, 1) and np.all(np.isfinite(full)) and np.all(full > 0)
    d2 = np.abs(np.diff(full[..., 0], 2, axis=0)) / full[1:-1, :, 0]
    assert d2.max() < 1e-5
    idx = np.array([0, 63, 64, 9_999, 10_000, 19_999])
    sub = dict(s, extinction=s["extinction"][:, idx], ssa=s["ssa"][:, idx], leg=s["leg"][:, :, idx], albedo=s["albedo"][idx])
    part = run_product_so(sub, engine=eng)["radiance"]
    np.testing.assert_array_equal(part, full[idx])
    ref = run_so_oracle(sub, nthreads=6, **o)["radiance"]
    np.testing.assert_allclose(part[..., 0], ref[..., 0], rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------------
# Weighting functions through the diffuse field: the discrete adjoint of the passes (sasktran2_amd/csrc/hip/so_adjoint.inc).  The reference's
# legacy source has none; the checkers are the oracle's own adjoint (validated against finite differences on the CPU, test_so_oracle.py)
# and central finite differences of the DEVICE radiances.
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nstokes,iterations", [(1, 1), (1, 3), (3, 2)])
def test_successive_orders_weighting_functions_match_the_oracle_adjoint(nstokes, iterations):
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=3)
    s["albedo"] = np.array([0.0, 0.3, 0.8])
    o = dict(num_iterations=iterations)
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=4)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-6)


def test_successive_orders_weighting_functions_against_finite_differences_of_the_device_radiance():
    from tests.gpu_harness import build_so_engine
    s = sc.regression_so(nstokes=1, num_wavelengths=2)
    s["albedo"] = np.array([0.3, 0.6])
    eng = build_so_engine(s, num_iterations=4)
    got = run_product_so(s, engine=eng, derivatives=True)
    k, w = s["extinction"], s["ssa"]
    for key, arr, name in (("d_extinction", k, "extinction"), ("d_ssa", w, "ssa")):
        scale = np.abs(got[key]).max(axis=3)
        for q in (0, 5, 11, 17):
            h = 1e-5 * arr[q].max() if name == "extinction" else 1e-5
            up, dn = arr.copy(), arr.copy()
            up[q] += h
            dn[q] -= h
            rp = run_product_so(dict(s, **{name: up}), engine=eng)["radiance"]
            rm = run_product_so(dict(s, **{name: dn}), engine=eng)["radiance"]
            fd = (rp - rm) / (2 * h)
            assert np.max(np.abs(fd - got[key][..., q]) / scale) < 2e-5, (name, q)
    h = 1e-5
    rp = run_product_so(dict(s, albedo=s["albedo"] + h), engine=eng)["radiance"]
    rm = run_product_so(dict(s, albedo=s["albedo"] - h), engine=eng)["radiance"]
    np.testing.assert_allclose(got["d_albedo"], (rp - rm) / (2 * h), rtol=1e-6)


def test_new_source_weighting_functions_fixed_iterations_and_anderson():
    """multiple_scatter_source = successive_orders: with depth 0, damping 1 and zero tolerances the driver IS the legacy iteration (exact
    transpose); the Anderson-mixed, tolerance-stopped driver is differentiated about its final state (adjoint Neumann series)."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=1, num_wavelengths=2)
    s["albedo"] = np.array([0.2, 0.5])
    legacy = run_product_so(s, derivatives=True, num_iterations=3)
    same = run_product_so(s, derivatives=True, num_iterations=3, multiple_scatter="successive_orders", relative_tolerance=0.0, absolute_tolerance=0.0,
                          damping=1.0, anderson_depth=0)
    for key in ("radiance", "d_extinction", "d_ssa", "d_albedo"):
        np.testing.assert_allclose(same[key], legacy[key], rtol=1e-12, atol=1e-300)
    o = dict(num_iterations=40, multiple_scatter="successive_orders", relative_tolerance=1e-10, absolute_tolerance=0.0, damping=1.0, anderson_depth=3)
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=4)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-6)


def test_successive_orders_weighting_functions_on_the_c5_limb_geometry():
    """BASELINE configs[4]'s discretisation (110 / 110 directions, 50 layers, 25 limb lines of sight, ten passes) at 130 wavelengths: the device
    rows against the oracle's adjoint, relative to the largest value of each profile."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.limb_c5(num_wavelengths=130)
    ref = make_so_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=16)
    got = run_product_so(s, derivatives=True)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-8)


@pytest.mark.parametrize("source,single", [("successive_orders", "exact"), ("successive_orders_legacy", "solartable"), ("successive_orders_legacy", "none")])
def test_weighting_functions_on_the_reference_scene_of_the_new_source(source, single):
    """The reference's own scene for its new source (a nadir ray that ends on the ground and a limb ray, seven levels, albedo 0.35): device rows
    against the oracle's adjoint -- the ground point's outgoing source behind the nadir ray, the table form of the lines of sight's solar
    transmission and the diffuse field alone."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.successive_orders_reference_scene(nstokes=1, iterations=4, source=source, albedo=0.35)
    s["config"]["single_scatter"] = single
    ref = make_so_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=4)
    got = run_product_so(s, derivatives=True)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-8)


def test_weighting_functions_with_three_solar_zenith_profiles_and_upward_rays():
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(nstokes=1, num_wavelengths=2)
    s["albedo"] = np.array([0.15, 0.4])
    s["rays"] = list(s["rays"]) + [(sc.VIEWING_UP_SOLAR, 0.6, 0.3, 9_000.0, 0.55)]
    o = dict(num_sza=3, num_iterations=3)
    ref = make_so_oracle(s, **o).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=4)
    got = run_product_so(s, derivatives=True, **o)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-8)


def test_weighting_functions_scale_with_the_solar_irradiance_at_two_thousand_wavelengths():
    """Size-independent property on a block of the C5 spectrum: every row is linear in the solar irradiance, and the albedo row of a
    black surface vanishes only where no line of sight or diffuse ray reaches the ground (limb rays above it still see it through the
    diffuse field)."""
    from tests.gpu_harness import build_so_engine
    s = sc.limb_c5(num_wavelengths=2048, num_iterations=3)
    eng = build_so_engine(s)
    one = run_product_so(s, engine=eng, derivatives=True)
    irr = np.linspace(0.5, 2.0, 2048)
    two = run_product_so(s, engine=eng, derivatives=True, solar_irradiance=irr)
    for key in ("radiance", "d_extinction", "d_ssa", "d_albedo"):
        scale = irr.reshape((-1,) + (1,) * (one[key].ndim - 1))
        # (rows change sign along the profile: entries next to a zero crossing are sums with cancellation -- the tolerance is on the profile's scale)
        np.testing.assert_allclose(two[key], one[key] * scale, rtol=1e-11, atol=1e-11 * np.abs(one[key]).max())
    assert np.all(np.isfinite(one["d_extinction"])) and np.all(one["d_albedo"] > 0)


def test_jvp_and_vjp_with_the_successive_orders_source():
    """sk_engine_calculate_jvp / _vjp (contractions of the native derivatives on the device) with the diffuse field of the successive orders
    in the Jacobian: equal to the contracted weighting functions, and <cotangent, J t> = <J^T cotangent, t>."""
    import sasktran2_amd as sk
    from tests.gpu_harness import build_so_engine
    s = sc.successive_orders_reference_scene(nstokes=1, iterations=3, source="successive_orders", albedo=0.3)
    engine, config, geometry = build_so_engine(s)
    nw = s["extinction"].shape[1]
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.storage.leg_coeff[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    full = engine.calculate_radiance(atmosphere, as_dataset=False)
    rng = np.random.default_rng(5)
    ngrid = s["extinction"].shape[0]
    t_k, t_w, t_a = rng.standard_normal(ngrid) * 1e-6, rng.standard_normal(ngrid) * 1e-2, rng.standard_normal(nw) * 0.1
    rad, jvp = engine.calculate_jvp(atmosphere, {"wf_extinction": t_k, "wf_ssa": t_w}, {"wf_albedo": t_a})
    want = np.einsum("qwls,q->wls", full["wf_extinction"], t_k) + np.einsum("qwls,q->wls", full["wf_ssa"], t_w) + full["wf_albedo"] * t_a[:, None, None]
    np.testing.assert_array_equal(rad, full["radiance"])
    np.testing.assert_allclose(jvp, want, rtol=1e-11, atol=1e-300)
    cot = rng.standard_normal(full["radiance"].shape)
    rad2, grads = engine.calculate_vjp(atmosphere, cot)
    np.testing.assert_array_equal(rad2, full["radiance"])
    np.testing.assert_allclose(grads["wf_extinction"], np.einsum("qwls,wls->q", full["wf_extinction"], cot), rtol=1e-11)
    np.testing.assert_allclose(grads["wf_albedo"], np.einsum("wls,wls->w", full["wf_albedo"], cot), rtol=1e-11)
    lhs = float(np.sum(cot * jvp))
    rhs = float(grads["wf_extinction"] @ t_k + grads["wf_ssa"] @ t_w + grads["wf_albedo"] @ t_a)
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))


@pytest.mark.parametrize("nstokes", [1, 3])
def test_device_resident_entry_point_with_weighting_functions_equals_the_host_entry_point(nstokes, monkeypatch):
    """sk_amd_so_engine_solve_device_wf (inputs and outputs in HBM, device.SoDeviceSolver) against sk_engine_calculate_radiance with host
    buffers: the same kernels -- radiances AND weighting functions agree bit for bit, also when the spectrum is solved in blocks of 128
    wavelengths and when the adjoint's chain splits the incoming rays differently over its wavefronts (every sum of the adjoint has one
    owner and a fixed order: so_adjoint.inc; the reference's threads are independent in the same way, cpp/lib/engine/engine.cpp:610-622)."""
    import torch
    from sasktran2_amd.device import SoDeviceSolver
    from tests.gpu_harness import build_so_engine
    nw = 200
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=nw)
    s["albedo"] = np.linspace(0.0, 0.6, nw)
    eng = build_so_engine(s)
    host = run_product_so(s, engine=eng, derivatives=True)
    again = run_product_so(s, engine=eng, derivatives=True)
    for key in ("radiance", "d_extinction", "d_ssa", "d_albedo"):
        assert np.array_equal(again[key], host[key]), key
    dev = torch.device("cuda", 0)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ext, ssa = to(s["extinction"].T), to(s["ssa"].T)
    leg, alb = to(np.transpose(s["leg"], (2, 1, 0))), to(s["albedo"])
    solver = SoDeviceSolver(eng[0])
    for chunk, rays_per_wave in ((None, None), ("128", None), ("128", "3"), (None, "50")):
        if chunk:
            monkeypatch.setenv("SASKTRAN2_AMD_SO_CHUNK", chunk)
        else:
            monkeypatch.delenv("SASKTRAN2_AMD_SO_CHUNK", raising=False)
        if rays_per_wave:
            monkeypatch.setenv("SASKTRAN2_AMD_SO_ADJ_RPW", rays_per_wave)
        else:
            monkeypatch.delenv("SASKTRAN2_AMD_SO_ADJ_RPW", raising=False)
        out = solver.allocate_outputs(nw, derivatives=True, device=dev)
        solver.solve(ext, ssa, leg, alb, out)
        torch.cuda.synchronize()
        assert np.array_equal(out["radiance"].cpu().numpy(), host["radiance"])
        for key, got in (("d_extinction", out["wf_extinction"].cpu().numpy().transpose(1, 2, 3, 0)), ("d_ssa", out["wf_ssa"].cpu().numpy().transpose(1, 2, 3, 0)),
                         ("d_albedo", out["wf_albedo"].cpu().numpy())):
            assert np.array_equal(got, host[key]), (key, chunk, rays_per_wave)


def test_weighting_functions_on_a_grid_of_81_levels():
    """81 grid levels (round 3 kept the adjoint's rows in LDS and needed the opt-in above 64 KB per workgroup here; the rows are sums over
    host-built lists now): device rows against the oracle's adjoint."""
    from tests.test_so_oracle import make_so_oracle
    s = sc.limb_c5(num_wavelengths=70, nlyr=80, num_iterations=2, num_directions=26)
    ref = make_so_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, nthreads=16)
    got = run_product_so(s, derivatives=True)
    assert_so_parity(got, ref)
    assert_so_wf_parity(got, ref, tol=1e-8)
