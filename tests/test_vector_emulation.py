"""The task bodies of the polarised (NSTOKES = 3) device path, executed on the CPU, against the reference's vector known-answer
tests and the pinned oracle (runs without a GPU; the same bodies run as HIP kernels in tests/test_gpu_parity.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import scenarios as sc
from tests.emulation import run_emulated_vector
from tests.test_oracle_golden import run_oracle

GOLDEN_VEC = json.loads((Path(__file__).parent / "golden" / "do_vector_planeparallel.json").read_text())


@pytest.mark.parametrize("case", GOLDEN_VEC["cases"], ids=[c["name"] for c in GOLDEN_VEC["cases"]])
def test_vector_known_answers_emulated(case):
    # cpp/lib/tests/sktran_disco/legacy/test_vector.cpp:81-277 with the reference's own tolerances
    s = sc.planeparallel_vector_case(case, GOLDEN_VEC["siewert_greek_coefficients"])
    out = run_emulated_vector(s)
    assert not out["status"].any()
    assert np.max(np.abs(out["radiance"][0] - s["expected"])) * case["sun"]["direct"] < case["tolerance"]
    # and against the oracle at the north-star tolerance (1e-6 relative on Stokes I)
    ora = run_oracle(s)["radiance"]
    # conservative scattering (Coulson's tables, ssa dithered to 1 - 1e-9) makes the smallest eigenvalue ~1e-9 and the two
    # eigen-solvers agree to ~1e-5 only; the reference loosens its own tolerance a hundredfold for SSA = 1 (test_scalar.cpp)
    conservative = any(layer[1] >= 1.0 for layer in case["layers"])
    np.testing.assert_allclose(out["radiance"][..., 0], ora[..., 0], rtol=5e-5 if conservative else 1e-6)
    assert np.max(np.abs(out["radiance"] - ora)) < (5e-6 if conservative else 1e-9)


def test_vector_path_reproduces_scalar_path_for_one_stokes_component():
    # NSTOKES = 1 through the run-time-n task bodies == the pinned scalar oracle (plane parallel and spherical)
    case = dict(json.loads((Path(__file__).parent / "golden" / "do_scalar_planeparallel.json").read_text())["cases"][0])
    s = sc.planeparallel_case(case)
    np.testing.assert_allclose(run_emulated_vector(s)["radiance"], run_oracle(s)["radiance"], rtol=1e-10)
    s = sc.regression_1d(3)
    out = run_emulated_vector(s)
    np.testing.assert_allclose(out["radiance"], s["expected_radiance"], rtol=5e-7, atol=2e-13)   # the reference's regression values


def test_vector_spherical_matches_oracle():
    s = sc.to_vector(sc.regression_1d(2), rayleigh_fraction=0.9)
    out = run_emulated_vector(s)
    ora = run_oracle(s)
    assert not out["status"].any()
    np.testing.assert_allclose(out["radiance"][..., 0], ora["radiance"][..., 0], rtol=1e-6)
    assert np.max(np.abs(out["radiance"] - ora["radiance"])) < 1e-9 * np.abs(ora["radiance"]).max() + 1e-12
    np.testing.assert_allclose(out["los_optical_depth"], ora["los_optical_depth"], rtol=1e-12)


@pytest.mark.parametrize("basis", [1, 2], ids=["solar", "observer"])
def test_stokes_basis_rotation(basis):
    """cpp/lib/output/output.cpp:27-55, cpp/lib/geometry/geometry.cpp:232-306: (Q, U) rotated by (cos 2a, -/+ sin 2a) with a the angle
    between the projections of the reference z axis and of the sun (observer position) on the plane normal to the look vector."""
    base = sc.to_vector(sc.regression_1d(1), rayleigh_fraction=0.9)
    std = run_emulated_vector(base)["radiance"]
    s = dict(base, config=dict(base["config"], stokes_basis=basis))
    rot = run_emulated_vector(s)["radiance"]
    np.testing.assert_allclose(rot[..., 0], std[..., 0], rtol=1e-14)
    np.testing.assert_allclose(rot[..., 1] ** 2 + rot[..., 2] ** 2, std[..., 1] ** 2 + std[..., 2] ** 2, rtol=1e-12)
    assert np.abs(rot[..., 1:] - std[..., 1:]).max() > 1e-6   # the bases differ for these rays
    np.testing.assert_allclose(rot, run_oracle(s)["radiance"], rtol=1e-6, atol=1e-12)


def _wf_close(got, ref, tol=1e-9):
    for k in ("d_extinction", "d_ssa", "d_albedo"):
        scale = np.abs(ref[k]).max()
        assert np.max(np.abs(got[k] - ref[k])) <= tol * (scale if scale > 0 else 1.0), k


def test_vector_weighting_functions_match_oracle_spherical_and_planeparallel():
    """Adjoint weighting functions of I, Q and U (one sweep per line of sight and Stokes component) against the oracle's dense
    forward-mode linearization, which itself matches finite differences (tests/test_oracle_golden.py)."""
    s = sc.to_vector(sc.regression_1d(2), rayleigh_fraction=0.9)
    _wf_close(run_emulated_vector(s, derivatives=True), run_oracle(s, derivatives=True))
    case = dict(GOLDEN_VEC["cases"][1], nstr=8, layers=[[0.3, 0.95], [0.3, 0.9], [0.4, 0.973527]])
    s = sc.planeparallel_vector_case(case, GOLDEN_VEC["siewert_greek_coefficients"])
    s["albedo"][:] = 0.3
    _wf_close(run_emulated_vector(s, derivatives=True), run_oracle(s, derivatives=True))


def test_scalar_weighting_functions_through_the_runtime_n_bodies():
    s = sc.regression_1d(2)
    _wf_close(run_emulated_vector(s, derivatives=True), run_oracle(s, derivatives=True))


def _mixture_spherical(x, vector):
    """Background + a scattering constituent with profile x; returns the scenario and the constituent's mapping arrays (see
    tests/test_gpu_parity.py::_mixture).  vector: the constituent also carries a2 / b1 (polarising) coefficients."""
    s = sc.regression_1d(1)
    ka = s["extinction"]
    wa = np.full_like(ka, 0.93)
    nl = 16
    l = np.arange(nl)
    ba = np.zeros(nl)
    ba[0], ba[1], ba[2] = 1.0, 0.08, 0.5
    bc = (2 * l + 1) * 0.6 ** l
    e, wc = 0.3 * ka, 0.95
    k = ka + x[:, None] * e
    sca = ka * wa + x[:, None] * e * wc
    w = sca / k
    beta = ((ka * wa)[None] * ba[:, None, None] + (x[:, None] * e * wc)[None] * bc[:, None, None]) / sca[None]
    out = dict(s, extinction=k, ssa=w, leg=beta)
    dleg = bc[:, None, None] - beta
    if vector:
        out = sc.to_vector(out, rayleigh_fraction=None)
        frac = x[:, None] * e * wc / sca
        out["leg"][4 * 2 + 1] = 3.0 * frac
        out["leg"][4 * 2 + 3] = -np.sqrt(6.0) * 0.5 * frac
        bc4 = np.zeros(4 * nl)
        bc4[0::4] = bc
        bc4[4 * 2 + 1], bc4[4 * 2 + 3] = 3.0, -np.sqrt(6.0) * 0.5
        dleg = bc4[:, None, None] - out["leg"]
    return out, dict(d_extinction=e, d_ssa=e * (wc - w) / k, scat_factor=e * wc / sca, d_legendre=dleg)


@pytest.mark.parametrize("vector", [False, True], ids=["NSTOKES=1", "NSTOKES=3"])
def test_scattering_group_derivatives_match_finite_differences_emulated(vector):
    """Number-density weighting function of a scattering constituent = d_extinction dI/dk + d_ssa dI/dw + scat_factor dI/dscat
    (cpp/lib/output/outputc.cpp:60-90) against central finite differences of the radiances."""
    x0 = np.ones(25)
    s, mp = _mixture_spherical(x0, vector)
    out = run_emulated_vector(s, derivatives=True, d_legendre=[mp["d_legendre"]])
    for q in (0, 9, 24):
        h = 1e-5
        xp, xm = x0.copy(), x0.copy()
        xp[q] += h
        xm[q] -= h
        fd = (run_emulated_vector(_mixture_spherical(xp, vector)[0])["radiance"] - run_emulated_vector(_mixture_spherical(xm, vector)[0])["radiance"]) / (2 * h)
        scat = out["d_scat"][:, :, :, 0, q] if vector else out["d_scat"][:, :, 0, q]
        an = mp["d_extinction"][q, 0] * out["d_extinction"][..., q] + mp["d_ssa"][q, 0] * out["d_ssa"][..., q] + mp["scat_factor"][q, 0] * scat
        assert np.abs(mp["scat_factor"][q, 0] * scat).max() > 0.05 * np.abs(an).max()   # the scattering part matters
        assert np.abs(fd - an).max() < 1e-6 * np.abs(an).max(), q


@pytest.mark.parametrize("vector", [False, True], ids=["NSTOKES=1", "NSTOKES=3"])
def test_parked_eigen_stage_layout_is_bit_identical(vector, monkeypatch):
    """kv1_layer_solution with E+- outside the team's work space and four matrix slots shared by lifetime (the layout that puts two teams
    on a CU at 32 polarised streams) performs the same operations in the same order: radiances and weighting functions equal bit for bit."""
    s = sc.regression_1d(2)
    if vector:
        s = sc.to_vector(s, rayleigh_fraction=0.9)
    plain = run_emulated_vector(s, derivatives=True)
    monkeypatch.setenv("SKV_EMU_PARK", "1")
    parked = run_emulated_vector(s, derivatives=True)
    for k in ("radiance", "d_extinction", "d_ssa", "d_albedo"):
        assert np.array_equal(plain[k], parked[k]), k


@pytest.mark.parametrize("vector", [False, True], ids=["NSTOKES=1", "NSTOKES=3"])
@pytest.mark.parametrize("kind,args0", [("modis", [0.2, 0.05, 0.05]), ("kokhanovsky", [0.04])], ids=["modis", "kokhanovsky"])
def test_brdf_surface_bodies_match_the_oracle(vector, kind, args0):
    """Non-Lambertian ground in the run-time-n task bodies (reflection factors of surface_expand_pair, ground rows of kv2_bvp, ground operators
    of kv2c_layer_ops, reflected intensity and adjoint seeds of kvpp_los, the BRDF-argument rows of the gradient) against the oracle."""
    s = sc.nadir_c1(num_wavelengths=3, nstr=8)
    if vector:
        s = sc.to_vector(s, rayleigh_fraction=0.8)
    args = np.tile(np.asarray(args0, dtype=float), (3, 1)) * np.linspace(1.0, 1.3, 3)[:, None]
    from tests.test_oracle_golden import make_oracle
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=(kind, args))
    got = run_emulated_vector(s, derivatives=True, brdf=(kind, args))
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-12 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-10 * np.abs(ref[k]).max(), k


def test_brdf_surface_bodies_in_spherical_limb_geometry_match_the_oracle():
    s = sc.to_vector(sc.limb_c2(num_wavelengths=2, nlyr=20, nstr=8), rayleigh_fraction=0.8)
    s["rays"] = s["rays"][::6]
    args = np.tile([0.2, 0.05, 0.05], (2, 1)) * np.linspace(1.0, 1.3, 2)[:, None]
    from tests.test_oracle_golden import make_oracle
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=("modis", args))
    got = run_emulated_vector(s, derivatives=True, brdf=("modis", args))
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-12 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-10 * np.abs(ref[k]).max(), k


def _mixture_nadir(x):
    """As _mixture_spherical on the plane-parallel nadir scene (21 levels, 8 streams): a scattering constituent with profile x over the background."""
    s = sc.nadir_c1(num_wavelengths=1, nstr=8)
    ka, nl = s["extinction"], 8
    wa = np.full_like(ka, 0.9)
    l = np.arange(nl)
    ba = np.zeros(nl)
    ba[0], ba[2] = 1.0, 0.5
    bc = (2 * l + 1) * 0.6 ** l
    e, wc = 0.3 * ka, 0.95
    k = ka + x[:, None] * e
    sca = ka * wa + x[:, None] * e * wc
    beta = ((ka * wa)[None] * ba[:, None, None] + (x[:, None] * e * wc)[None] * bc[:, None, None]) / sca[None]
    out = dict(s, extinction=k, ssa=sca / k, leg=beta)
    return out, dict(d_extinction=e, d_ssa=e * (wc - sca / k) / k, scat_factor=e * wc / sca, d_legendre=bc[:, None, None] - beta)


def test_scattering_group_derivatives_over_a_brdf_surface_match_finite_differences_emulated():
    """The scattering constituent's weighting function when the ground is not Lambertian: the ground rows of the group's gradient weigh the
    incoming streams one by one (c_t = sum_i zB_i RS(i, t) + gw RL(los, t)) instead of through the rank-one Lambertian sums."""
    brdf = ("modis", np.array([[0.25, 0.08, 0.03]]))
    x0 = np.ones(21)
    s, mp = _mixture_nadir(x0)
    out = run_emulated_vector(s, derivatives=True, d_legendre=[mp["d_legendre"]], brdf=brdf)
    an = (mp["d_extinction"][:, 0] * out["d_extinction"] + mp["d_ssa"][:, 0] * out["d_ssa"] + mp["scat_factor"][:, 0] * out["d_scat"][:, :, 0, :])
    scale = np.abs(an).max()   # (the profile falls by three orders of magnitude with altitude: central differences resolve 1e-6 of its largest entry)
    for q in (0, 4, 10):
        h = 1e-5
        xp, xm = x0.copy(), x0.copy()
        xp[q] += h
        xm[q] -= h
        fd = (run_emulated_vector(_mixture_nadir(xp)[0], brdf=brdf)["radiance"] - run_emulated_vector(_mixture_nadir(xm)[0], brdf=brdf)["radiance"]) / (2 * h)
        assert np.abs(mp["scat_factor"][q, 0] * out["d_scat"][:, :, 0, q]).max() > 0.05 * np.abs(an[..., q]).max()
        assert np.abs(fd - an[..., q]).max() < 1e-6 * scale, q


def test_the_references_functional_brdf_known_answer_emulated():
    import json
    from pathlib import Path
    root = Path(__file__).parent / "golden"
    gold = json.loads((root / "do_scalar_brdf_legacy.json").read_text())
    cases = json.loads((root / "do_scalar_planeparallel.json").read_text())["cases"]
    s = sc.planeparallel_case(next(c for c in cases if c["name"] == gold["base_case"]))
    got = run_emulated_vector(s, brdf=(gold["brdf"], np.array([[1.0]])))["radiance"][0]
    diff = np.abs(got - np.asarray(gold["expected"]))
    assert diff.max() < 3e-4 and np.median(diff) < 3e-5


@pytest.mark.parametrize("num_sza", [1, 2])
def test_brdf_ground_source_bodies_in_spherical_geometry_match_the_oracle(num_sza):
    """Lines of sight that END on a non-Lambertian ground in spherical geometry: ground points per cos-angle grid point and order (kv3), their
    azimuth sum along the ray (kv4), the stream-by-stream ground seeds of the adjoint sweeps (kv2b), per solar zenith index."""
    s = sc.to_vector(sc.regression_1d(2), rayleigh_fraction=0.9)
    s["config"]["num_sza"] = num_sza
    args = np.tile([0.2, 0.05, 0.05], (2, 1)) * np.linspace(1.0, 1.3, 2)[:, None]
    from tests.test_oracle_golden import make_oracle
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=("modis", args))
    got = run_emulated_vector(s, derivatives=True, brdf=("modis", args))
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-12 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-9 * np.abs(ref[k]).max(), k



@pytest.mark.parametrize("vector", [False, True], ids=["NSTOKES=1", "NSTOKES=3"])
def test_team_form_of_the_table_values_is_bit_identical(vector, monkeypatch):
    """kv3_table_values_team (one wavefront per (layer, wavelength) on the device, unit-stride record reads) sums every output element in the
    order of the per-thread form: radiances, convergence bookkeeping and weighting functions equal bit for bit -- also over a BRDF ground."""
    s = sc.regression_1d(2)
    if vector:
        s = sc.to_vector(s, rayleigh_fraction=0.9)
    brdf = ("modis", np.tile([0.2, 0.05, 0.05], (2, 1)))
    plain = [run_emulated_vector(s, derivatives=True), run_emulated_vector(s, derivatives=True, brdf=brdf)]
    monkeypatch.setenv("SKV_EMU_K3_TEAM", "1")
    team = [run_emulated_vector(s, derivatives=True), run_emulated_vector(s, derivatives=True, brdf=brdf)]
    for a, b_ in zip(plain, team):
        for k in ("radiance", "d_extinction", "d_ssa"):
            assert np.array_equal(a[k], b_[k]), k
