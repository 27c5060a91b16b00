"""CPU-only checks of the HIP kernel bodies: tests/emulate_kernels.cpp compiles sasktran2_amd/csrc/hip/disco_core.hpp
with g++ and runs one loop iteration per GPU thread in the driver's launch order; results must match the oracle."""
import numpy as np
import pytest

from tests import scenarios as sc
from tests.emulation import run_emulated
from tests.test_oracle_golden import run_oracle


def check(s, rtol_i=1e-9, rtol_wf=1e-8):
    o, e = run_oracle(s, derivatives=True), run_emulated(s)
    assert np.all(e["status"] == 0)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=rtol_i)
    for k in ("d_extinction", "d_ssa"):
        scale = np.abs(o[k]).max(axis=2, keepdims=True)
        assert np.max(np.abs(e[k] - o[k]) / scale) < rtol_wf, k
    np.testing.assert_allclose(e["d_albedo"], o["d_albedo"], rtol=rtol_wf, atol=1e-300)


def test_regression_scenario():
    s = sc.regression_1d()
    check(s)
    np.testing.assert_allclose(run_emulated(s)["radiance"], s["expected_radiance"], rtol=5e-7, atol=2e-13)


def test_c2_shape():
    check(sc.limb_c2(num_wavelengths=2))


@pytest.mark.parametrize("nstr", [2, 4, 16, 32])
def test_stream_counts(nstr):
    s = sc.regression_1d(1)
    s["config"]["nstr"] = nstr
    if nstr > s["leg"].shape[0]:
        s["leg"] = np.concatenate([s["leg"], np.zeros((nstr - 16, 25, 1))])
    check(s)


@pytest.mark.parametrize("scale", [50.0, 1e-3])
def test_optical_thickness_extremes(scale):
    s = sc.regression_1d(1)
    s["extinction"] = s["extinction"] * scale
    check(s, rtol_wf=1e-7)


def test_conservative_scattering_radiance():
    s = sc.regression_1d(1)
    s["ssa"][:] = 1.0
    o, e = run_oracle(s), run_emulated(s)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=2e-6)


# ---- plane-parallel post-processing (kpp_los) against the reference's own known answers and the oracle
import json
from pathlib import Path

GOLDEN = json.loads((Path(__file__).parent / "golden" / "do_scalar_planeparallel.json").read_text())


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_planeparallel_known_answers_emulated(case):
    # cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp: |radiance - expected| < 1e-8 (1e-6 for SSA=1)
    s = sc.planeparallel_case(case)
    e = run_emulated(s, derivatives=False)
    assert np.all(e["status"] == 0)
    assert np.max(np.abs(e["radiance"][0] - s["expected"])) < case["tolerance"]


@pytest.mark.parametrize("idx", [0, 3, 8, 14, 17])
def test_planeparallel_weighting_functions_emulated(idx):
    s = sc.planeparallel_case(GOLDEN["cases"][idx])
    o, e = run_oracle(s, derivatives=True), run_emulated(s)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=1e-8)
    for k in ("d_extinction", "d_ssa"):
        scale = np.abs(o[k]).max(axis=2, keepdims=True)
        assert np.max(np.abs(e[k] - o[k]) / scale) < 1e-7, k
    np.testing.assert_allclose(e["d_albedo"], o["d_albedo"], rtol=1e-7, atol=1e-14)


@pytest.mark.parametrize("nstr,observer", [(2, 200_000.0), (4, 200_000.0), (8, 37_300.0), (2, 50_000.0)])
def test_c1_nadir_emulated(nstr, observer):
    # BASELINE configs[0]; the observer inside the atmosphere exercises the partial-layer branch
    s = sc.nadir_c1(num_wavelengths=5, nstr=nstr, observer_altitude=observer)
    check(s, rtol_i=1e-9, rtol_wf=1e-7)


def test_planeparallel_without_single_scatter_emulated():
    s = sc.nadir_c1(num_wavelengths=3, nstr=4)
    s["config"]["ss_in_do"] = False
    from tests.test_oracle_golden import make_oracle
    o = make_oracle(s, ss_in_do=False).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True)
    e = run_emulated(s)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=1e-9)
    np.testing.assert_allclose(e["d_albedo"], o["d_albedo"], rtol=1e-7)
    for k in ("d_extinction", "d_ssa"):
        assert np.max(np.abs(e[k] - o[k]) / np.abs(o[k]).max(axis=2, keepdims=True)) < 1e-7


# ---- delta-M scaling (sk_atmosphere_apply_delta_m_scaling + the f subtraction in the layer optics)
def _peaked(nstr=8, nmom=32, g=0.85, nw=2):
    s = sc.regression_1d(nw)
    s["config"]["nstr"] = nstr
    l = np.arange(nmom)
    hg = (2 * l + 1) * g ** l
    ray = np.zeros(nmom)
    ray[0], ray[2] = 1.0, 0.5
    c = 0.5 * np.exp(-s["geometry"]["altitudes"] / 9_000.0)
    s["leg"] = ((1 - c)[None, :] * ray[:, None] + c[None, :] * hg[:, None])[:, :, None] * np.ones((1, 1, nw))
    return s


def test_delta_m_emulated_matches_oracle():
    d = sc.delta_m_scaled(_peaked())
    from tests.test_oracle_golden import make_oracle
    o = make_oracle(d).run(d["extinction"], d["ssa"], d["leg"], d["albedo"], derivatives=True, delta_f=d["delta_f"])
    e = run_emulated(d)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=1e-9)
    for k in ("d_extinction", "d_ssa"):
        assert np.max(np.abs(e[k] - o[k]) / np.abs(o[k]).max(axis=2, keepdims=True)) < 1e-7, k


def _pp_peaked(nstr, g=0.85, nmom=64):
    alt = np.linspace(0.0, 40_000.0, 9)
    rays = [(sc.GROUND_VIEWING_SOLAR, 0.6, a, 200_000.0, cz) for a, cz in ((0.3, 0.9), (1.5, 0.6), (2.8, 0.4))]
    k = (2e-5 * np.exp(-alt / 8_000.0))[:, None] * np.ones((1, 1))
    l = np.arange(nmom)
    leg = ((2 * l + 1) * g ** l)[:, None, None] * np.ones((1, alt.size, 1))
    return dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_372_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR,
                              geotype=sc.GEO_PLANEPARALLEL),
                config=dict(nstr=nstr, num_sza=1), rays=rays, extinction=k, ssa=np.full((alt.size, 1), 0.95), leg=leg,
                albedo=np.array([0.2]))


def test_delta_m_improves_low_stream_accuracy():
    """Forward-peaked aerosol (Henyey-Greenstein g = 0.85), plane parallel, single scatter inside DO: with 8 streams the
    delta-M scaled solution is far closer to the 32-stream solution than the unscaled one (Wiscombe 1977) -- measured
    1-5 % against 19-43 %."""
    from tests.test_oracle_golden import make_oracle
    t = _pp_peaked(32)
    truth = make_oracle(t).run(t["extinction"], t["ssa"], t["leg"][:32], t["albedo"])["radiance"]
    s8 = _pp_peaked(8)
    plain = make_oracle(s8).run(s8["extinction"], s8["ssa"], s8["leg"][:8], s8["albedo"])["radiance"]
    d = sc.delta_m_scaled(s8)
    scaled = make_oracle(d).run(d["extinction"], d["ssa"], d["leg"], d["albedo"], delta_f=d["delta_f"])["radiance"]
    assert np.abs(scaled / truth - 1).max() < 0.25 * np.abs(plain / truth - 1).max()
    e = run_emulated(d, derivatives=False)
    np.testing.assert_allclose(e["radiance"], scaled, rtol=1e-9)


# ---- observers inside the atmosphere looking up / sideways / down (ViewingUpSolar, sk_viewing_geometry_add_solar_angles_observer_location)
def _up_looking(nw=2):
    s = sc.regression_1d(nw)
    csz = s["geometry"]["cos_sza"]
    s["rays"] = [(sc.VIEWING_UP_SOLAR, csz, az, alt, cz)
                 for az, alt, cz in ((0.0, 1_000.0, 1.0), (0.7, 1_000.0, 0.3), (-2.0, 12_500.0, 0.6), (1.2, 31_000.0, 0.05),
                                     (0.4, 20_000.0, -0.4), (2.5, 59_000.0, 0.9))]
    return s


def test_up_looking_observers_emulated():
    s = _up_looking()
    check(s, rtol_wf=1e-7)
    o = run_oracle(s)
    # an observer just below the top of the atmosphere looking up sees almost nothing; one near the ground looking at the zenith does
    assert np.all(o["radiance"][:, 5] < 1e-3 * o["radiance"][:, 0])
    assert np.all(o["radiance"][:, 0] > 0)


def test_tangent_altitude_rays_equal_their_solar_form():
    """TangentAltitude (tangent point at an angle from the reference point in the solar plane, azimuth in the local horizontal
    plane: sk_viewing_geometry_add_tangent_altitude) traces the same line of sight as TangentAltitudeSolar with the tangent
    point's solar zenith angle and the same azimuth, so both must give the same radiance and weighting functions."""
    s = sc.regression_1d(2)
    s["geometry"]["saa"] = 0.0
    csz0 = s["geometry"]["cos_sza"]
    rays = []
    for theta, az in ((0.1, 0.0), (-0.2, 0.0), (0.05, np.pi)):
        sza_tp = np.arccos(csz0) - theta
        rays += [(sc.TANGENT_ALTITUDE, theta, az, 200_000.0, 20_000.0),
                 (sc.TANGENT_ALTITUDE_SOLAR, float(np.cos(sza_tp)), az, 200_000.0, 20_000.0)]
    s["rays"] = rays
    o = run_oracle(s, derivatives=True)
    # (the two constructions differ by the rounding of acos / cos, which moves a sza-table weight by ~1e-8)
    np.testing.assert_allclose(o["radiance"][:, 0::2], o["radiance"][:, 1::2], rtol=1e-7)
    np.testing.assert_allclose(o["d_extinction"][:, 0::2], o["d_extinction"][:, 1::2], rtol=1e-6, atol=1e-9)
    check(s, rtol_wf=1e-7)


# ---- pseudo-spherical geometry: plane-parallel diffuse field, solar beam along its true path through spherical shells
def _pseudo_spherical(csz, nw=3, nstr=8):
    s = sc.nadir_c1(num_wavelengths=nw, nstr=nstr)
    s["geometry"]["cos_sza"] = csz
    s["geometry"]["geotype"] = sc.GEO_PSEUDOSPHERICAL
    s["rays"] = [(r[0], csz) + tuple(r[2:]) for r in s["rays"]]
    return s


@pytest.mark.parametrize("csz", [0.9, 0.3, 0.08])
def test_pseudo_spherical_chapman_factors_and_parity(csz):
    s = _pseudo_spherical(csz)
    from tests.test_oracle_golden import make_oracle
    eng = make_oracle(s)
    ch = eng.chapman()[0]
    # straight-line Chapman factors of the reference (sktran_do_geometrylayerarray.cpp:73-113):
    # (sqrt(r_ceil^2 - r_p^2 sin^2) - sqrt(r_floor^2 - r_p^2 sin^2)) / (r_ceil - r_floor) for the layers q above the floor of layer p
    alt, R = s["geometry"]["altitudes"], s["geometry"]["earth_radius"]
    ceil, floor = alt[::-1][:-1], alt[::-1][1:]
    for p in range(len(ceil)):
        rp2 = (R + floor[p]) ** 2 * (1 - csz * csz)
        want = (np.sqrt((R + ceil[: p + 1]) ** 2 - rp2) - np.sqrt((R + floor[: p + 1]) ** 2 - rp2)) / (ceil[: p + 1] - floor[: p + 1])
        np.testing.assert_allclose(ch[p, : p + 1], want, rtol=1e-11)
        assert np.all(ch[p, p + 1:] == 0)
    check(s, rtol_wf=1e-7)
    # sun high: pseudo-spherical == plane parallel to a few 1e-4; sun lower: brighter than plane parallel (0.2 - 16 %)
    pp = dict(s, geometry=dict(s["geometry"], geotype=sc.GEO_PLANEPARALLEL))
    ratio = run_oracle(s)["radiance"] / run_oracle(pp)["radiance"]
    assert (np.abs(ratio - 1).max() < 5e-4) if csz > 0.8 else (ratio.min() > 1.001)


# ------------------------------------------------------------------------------------------------ thermal source inside the DO equations
@pytest.mark.parametrize("surface", [False, True], ids=["optical depth 100", "optical depth 1 with surface emission"])
def test_thermal_pass_emulated_hits_the_disort_pins(surface):
    """emission_source = discrete_ordinates: the thermal pass of the HIP driver (k0_thermal + the order-0 solve with the thermal "beam")
    on the CPU against the reference's DISORT known answers (tests/engine/thermal_emissions/test_disort.py:7-112)"""
    from tests.emulation import run_emulated
    s = sc.disort_thermal(surface)
    d = sc.delta_m_scaled(s)
    got = run_emulated(dict(d, emission=s["emission"], surface_emission=s["surface_emission"], solar_irradiance=s["solar_irradiance"]), derivatives=False)
    np.testing.assert_allclose(got["radiance"][0, 0], s["expected"], rtol=s["rtol"])


@pytest.mark.parametrize("nstr", [4, 16])
def test_thermal_weighting_functions_emulated_match_finite_differences(nstr):
    """weighting functions of the thermal source -- extinction, ssa, albedo, emission profile, surface emission -- from the solar adjoint run
    on the thermal layer parameters (k4a_thermal) against central differences of the emulated radiances; with the sun on as well"""
    from tests.emulation import run_emulated
    ng, nw = 9, 2
    alt = np.linspace(0.0, 16000.0, ng)
    geometry = dict(cos_sza=0.6, saa=0.0, earth_radius=6_372_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR, geotype=sc.GEO_PLANEPARALLEL)
    base = dict(extinction=(7e-5 * np.exp(-alt / 7000.0))[:, None] * np.array([0.7, 2.0])[None, :],
                ssa=np.linspace(0.4, 0.8, ng)[:, None] * np.ones((1, nw)),
                emission=(3.0 * np.exp(-alt / 9000.0) + 0.4)[:, None] * np.array([1.0, 1.4])[None, :],
                albedo=np.array([0.15, 0.3]), surface_emission=np.array([2.0, 1.2]))
    leg = ((2 * np.arange(16) + 1) * 0.5 ** np.arange(16))[:, None, None] * np.ones((1, ng, nw))
    rays = [(sc.GROUND_VIEWING_SOLAR, 0.6, 0.4, 200_000.0, 0.85), (sc.GROUND_VIEWING_SOLAR, 0.6, 1.9, 200_000.0, 0.45)]

    def solve(v, derivs):
        return run_emulated(dict(geometry=geometry, config=dict(nstr=nstr, num_sza=1), rays=rays, leg=leg, solar_irradiance=np.ones(nw), **v), derivatives=derivs)
    res = solve(base, True)

    def fd(key, q, h):
        plus, minus = {k: a.copy() for k, a in base.items()}, {k: a.copy() for k, a in base.items()}
        if q is None:
            plus[key] = plus[key] + h
            minus[key] = minus[key] - h
        else:
            plus[key][q] += h
            minus[key][q] -= h
        return (solve(plus, False)["radiance"] - solve(minus, False)["radiance"]) / (2 * h)
    scale = np.max(np.abs(res["radiance"]))
    for q in (0, 4, 8):
        np.testing.assert_allclose(res["d_emission"][:, :, q], fd("emission", q, 1e-4), rtol=2e-6, atol=1e-9 * scale)
        np.testing.assert_allclose(res["d_extinction"][:, :, q], fd("extinction", q, 1e-9), rtol=2e-5, atol=1e-6 * np.max(np.abs(res["d_extinction"])))
        np.testing.assert_allclose(res["d_ssa"][:, :, q], fd("ssa", q, 1e-6), rtol=2e-5, atol=1e-6 * np.max(np.abs(res["d_ssa"])))
    np.testing.assert_allclose(res["d_surface_emission"], fd("surface_emission", None, 1e-4), rtol=2e-6, atol=1e-9 * scale)
    np.testing.assert_allclose(res["d_albedo"], fd("albedo", None, 1e-6), rtol=2e-5, atol=1e-8 * scale)
