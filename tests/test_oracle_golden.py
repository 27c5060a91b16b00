"""Pins the CPU oracle against the reference's own known-answer tests (runs without a GPU)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle.oracle_py import OracleEngine
from tests import scenarios as sc

GOLDEN = json.loads((Path(__file__).parent / "golden" / "do_scalar_planeparallel.json").read_text())
GOLDEN_VEC = json.loads((Path(__file__).parent / "golden" / "do_vector_planeparallel.json").read_text())


def make_oracle(s, ss_in_do=True):
    g = s["geometry"]
    return OracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"],
                        s["config"]["nstr"], s["config"]["num_sza"], s["rays"], ss_in_do=ss_in_do,
                        nstokes=s["config"].get("nstokes", 1), stokes_basis=s["config"].get("stokes_basis", 0))


def run_oracle(s, derivatives=False, **over):
    eng = make_oracle(s)
    return eng.run(over.get("extinction", s["extinction"]), over.get("ssa", s["ssa"]), s["leg"],
                   over.get("albedo", s["albedo"]), derivatives=derivatives)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=[c["name"] for c in GOLDEN["cases"]])
def test_planeparallel_known_answers(case):
    # cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp: |radiance - expected| < SKDO_FPC_EPS (1e-8; 1e-6 for SSA=1)
    s = sc.planeparallel_case(case)
    out = run_oracle(s)
    assert np.max(np.abs(out["radiance"][0] - s["expected"])) < case["tolerance"]


def test_planeparallel_three_layers_equal_one():
    # "Scalar Boundary Conditions", test_scalar.cpp:847-880: 3 identical Rayleigh layers == 1 layer of 3x the depth
    base = dict(GOLDEN["cases"][0])

    def rayleigh(nl, od):
        s = sc.planeparallel_case({**base, "layers_od_ssa_g": [[od, 1.0, 0.0]] * nl})
        s["leg"][:] = 0
        s["leg"][0], s["leg"][2] = 1.0, 0.5
        return run_oracle(s)["radiance"][0]

    assert np.max(np.abs(rayleigh(1, 0.6) - rayleigh(3, 0.2))) < 1e-8


def test_spherical_regression_radiance_and_optical_depth():
    # tests/engine/test_1d_solver_regression.py:195-240: rtol 5e-7 (atol 2e-13 / 1e-13)
    s = sc.regression_1d()
    out = run_oracle(s, derivatives=True)
    np.testing.assert_allclose(out["radiance"], s["expected_radiance"], rtol=5e-7, atol=2e-13)
    np.testing.assert_allclose(out["los_optical_depth"], s["expected_od"], rtol=5e-7, atol=1e-13)


def _fd_check(s, out, rel=1e-4):
    """Analytic WFs == central finite differences, the reference's own criterion
    (src/sasktran2/test_util/wf.py:9-24: agreement relative to the largest WF of the column)."""
    ngrid = s["extinction"].shape[0]
    for name, key, idxs in (("extinction", "d_extinction", (0, 3, ngrid // 2, ngrid - 2)),
                            ("ssa", "d_ssa", (0, 4, ngrid // 2, ngrid - 1))):
        for q in idxs:
            base = s[name]
            if np.any(base[q, :] == 0):
                continue  # unused top grid point of the plane-parallel 'lower' grids
            h = 1e-4 * base[q, :]
            plus, minus = base.copy(), base.copy()
            plus[q, :] += h
            minus[q, :] -= h
            fd = (run_oracle(s, **{name: plus})["radiance"] - run_oracle(s, **{name: minus})["radiance"]) / (2 * h[:, None])
            an = out[key][:, :, q]
            scale = np.abs(out[key]).max(axis=2)
            assert np.max(np.abs(fd - an) / scale) < rel, (name, q)
    h = 1e-5
    fd = (run_oracle(s, albedo=s["albedo"] + h)["radiance"] - run_oracle(s, albedo=s["albedo"] - h)["radiance"]) / (2 * h)
    np.testing.assert_allclose(out["d_albedo"], fd, rtol=1e-5, atol=1e-12)


def test_spherical_weighting_functions_match_finite_differences():
    s = sc.regression_1d(2)
    _fd_check(s, run_oracle(s, derivatives=True))


def test_planeparallel_weighting_functions_match_finite_differences():
    case = dict(GOLDEN["cases"][0])
    case["los_coszen_az"] = case["los_coszen_az"][7:21:3]
    case["expected"] = case["expected"][7:21:3]
    s = sc.planeparallel_case(case)
    _fd_check(s, run_oracle(s, derivatives=True))


def test_limb_c2_small_runs_and_needs_many_azimuth_orders():
    s = sc.limb_c2(num_wavelengths=2)
    out = run_oracle(s, derivatives=True)
    assert np.all(np.isfinite(out["radiance"])) and np.all(out["radiance"] > 0)
    assert out["num_azimuth"].min() >= 6


# ---------------------------------------------------------------------------------------------- NSTOKES = 3
@pytest.mark.parametrize("case", GOLDEN_VEC["cases"], ids=[c["name"] for c in GOLDEN_VEC["cases"]])
def test_vector_planeparallel_known_answers(case):
    # cpp/lib/tests/sktran_disco/legacy/test_vector.cpp:81-277: |stokes * direct - expected| < tolerance for every (I, Q, U)
    s = sc.planeparallel_vector_case(case, GOLDEN_VEC["siewert_greek_coefficients"])
    out = run_oracle(s)
    assert out["radiance"].shape == (1, len(case["los_coszen_az"]), 3)
    assert np.max(np.abs(out["radiance"][0] - s["expected"])) * case["sun"]["direct"] < case["tolerance"]


def test_vector_three_layers_equal_one():
    # "Polarization Boundary Conditions", test_vector.cpp:279-314: 3 Rayleigh layers of 0.2 == 1 layer of 0.6, 16 streams, albedo 0.8
    base = dict(GOLDEN_VEC["cases"][0], layer_kind="Rayleigh", albedo=0.8)
    one = run_oracle(sc.planeparallel_vector_case({**base, "layers": [[0.6, 1.0, 0.0]]}))["radiance"]
    three = run_oracle(sc.planeparallel_vector_case({**base, "layers": [[0.2, 1.0, 0.0]] * 3}))["radiance"]
    assert np.abs(one[..., 1:]).max() > 1e-3   # the case is genuinely polarised
    assert np.max(np.abs(one - three)) < 1e-6


def test_vector_spherical_reduces_to_scalar_and_polarises():
    # with only a1 set the NSTOKES = 3 spherical solve is the scalar one (pinned above) with Q = U = 0
    s = sc.regression_1d(2)
    scalar = run_oracle(s, derivatives=True)
    vec = run_oracle(sc.to_vector(s), derivatives=True)
    np.testing.assert_allclose(vec["radiance"][..., 0], scalar["radiance"], rtol=1e-10)
    assert np.abs(vec["radiance"][..., 1:]).max() < 1e-14
    np.testing.assert_allclose(vec["d_extinction"][..., 0, :], scalar["d_extinction"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(vec["d_ssa"][..., 0, :], scalar["d_ssa"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(vec["los_optical_depth"], scalar["los_optical_depth"], rtol=1e-14)
    pol = run_oracle(sc.to_vector(s, rayleigh_fraction=0.9))
    assert np.abs(pol["radiance"][..., 1]).max() > 1e-4 and np.abs(pol["radiance"][..., 2]).max() > 1e-5
    # neglecting polarisation changes Rayleigh-like intensities by up to ~10 % (the classic scalar-approximation error)
    assert 0.01 < np.max(np.abs(pol["radiance"][..., 0] / scalar["radiance"] - 1)) < 0.12


def test_vector_spherical_weighting_functions_match_finite_differences():
    s = sc.to_vector(sc.regression_1d(1), rayleigh_fraction=0.9)
    out = run_oracle(s, derivatives=True)
    for name, key in (("extinction", "d_extinction"), ("ssa", "d_ssa")):
        for q in (0, 5, 12, 23):
            base = s[name]
            h = 1e-4 * base[q, :]
            plus, minus = base.copy(), base.copy()
            plus[q, :] += h
            minus[q, :] -= h
            fd = (run_oracle(s, **{name: plus})["radiance"] - run_oracle(s, **{name: minus})["radiance"]) / (2 * h[:, None, None])
            scale = np.abs(out[key]).max(axis=3)
            assert np.max(np.abs(fd - out[key][..., q]) / scale) < 1e-4, (name, q)
    h = 1e-5
    fd = (run_oracle(s, albedo=s["albedo"] + h)["radiance"] - run_oracle(s, albedo=s["albedo"] - h)["radiance"]) / (2 * h)
    np.testing.assert_allclose(out["d_albedo"], fd, rtol=1e-5, atol=1e-11)


@pytest.mark.parametrize("surface", [False, True], ids=["atmosphere only, optical depth 100", "with surface emission, optical depth 1"])
def test_disort_thermal_emission_pins(surface):
    """tests/engine/thermal_emissions/test_disort.py:7-112: emission_source = discrete_ordinates (thermal source inside the DO equations,
    sktran_do_rte.cpp:1335-1618), values generated with DISORT (test case 7a), rtol 1e-6."""
    s = sc.disort_thermal(surface)
    d = sc.delta_m_scaled(s)
    r = make_oracle(d).run(d["extinction"], d["ssa"], d["leg"], d["albedo"], solar_irradiance=s["solar_irradiance"], delta_f=d["delta_f"],
                           emission=s["emission"], surface_emission=s["surface_emission"])
    np.testing.assert_allclose(r["radiance"][0, 0], s["expected"], rtol=s["rtol"])
