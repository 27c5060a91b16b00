"""Non-Lambertian surfaces (SnowKokhanovsky, MODIS) on the device: the discrete-ordinates ground condition with the azimuth expansion of
the BRDF (kern_surface_expand, kv2_bvp, kv2c_layer_ops, kvpp_los of the run-time-n kernels) through the C ABI against the oracle
(oracle/disco_oracle.hpp, pinned by its Lambertian limit and by central differences, tests/test_brdf_oracle.py), and the reference's own
criterion for these surfaces: analytic weighting functions of the BRDF arguments = central differences (tests/constituent/test_modis.py)."""
import numpy as np
import pytest

import sasktran2_amd as sk
from tests import scenarios as sc
from tests.gpu_harness import build_engine, run_product
from tests.test_oracle_golden import make_oracle

pytestmark = pytest.mark.gpu

CASES = [("modis", [0.2, 0.05, 0.05]), ("kokhanovsky", [0.04])]


def _scene(nstokes, nstr, nw=5, pseudo_spherical=False):
    s = sc.nadir_c1(num_wavelengths=nw, nstr=nstr)
    if pseudo_spherical:
        s["geometry"]["geotype"] = sc.GEO_PSEUDOSPHERICAL
    if nstokes == 3:
        s = sc.to_vector(s, rayleigh_fraction=0.8)
    return s


def _args(args0, nw):
    return np.tile(np.asarray(args0, dtype=float), (nw, 1)) * np.linspace(1.0, 1.3, nw)[:, None]


@pytest.mark.parametrize("nstokes,nstr", [(1, 4), (1, 8), (1, 16), (3, 8)])
@pytest.mark.parametrize("kind,args0", CASES, ids=[c[0] for c in CASES])
def test_brdf_radiances_and_weighting_functions_match_the_oracle(nstokes, nstr, kind, args0):
    s = _scene(nstokes, nstr)
    nw = len(s["albedo"])
    s["brdf"] = (kind, _args(args0, nw))
    got = run_product(s, derivatives=True)
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    scale = np.abs(ref["radiance"]).max()
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * scale       # north star: 1e-6 on Stokes I
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-8 * np.abs(ref[k]).max(), k   # north star: 1e-4 on weighting functions


def test_brdf_in_pseudo_spherical_geometry_matches_the_oracle():
    s = _scene(1, 8, pseudo_spherical=True)
    nw = len(s["albedo"])
    s["brdf"] = ("modis", _args([0.2, 0.05, 0.05], nw))
    got = run_product(s, derivatives=True)
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-8 * np.abs(ref[k]).max(), k


@pytest.mark.parametrize("nstokes,nstr", [(1, 8), (1, 16), (3, 8)])
@pytest.mark.parametrize("kind,args0", CASES, ids=[c[0] for c in CASES])
def test_brdf_under_limb_lines_of_sight_in_spherical_geometry_matches_the_oracle(nstokes, nstr, kind, args0):
    """The bench geometry (BASELINE configs[1]: spherical shell, tangent rays from 600 km) over a non-Lambertian ground: the surface enters
    through the ground rows of the boundary-value problem of every azimuth order (and moves the limb radiance by tens of per cent)."""
    s = sc.limb_c2(num_wavelengths=3, nlyr=20, nstr=nstr)
    s["rays"] = s["rays"][::5]
    if nstokes == 3:
        s = sc.to_vector(s, rayleigh_fraction=0.8)
    nw = 3
    s["brdf"] = (kind, _args(args0, nw))
    got = run_product(s, derivatives=True)
    orc = make_oracle(s)
    ref = orc.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    lam = orc.run(s["extinction"], s["ssa"], s["leg"], s["albedo"])
    assert np.max(np.abs(ref["radiance"] - lam["radiance"])) > 0.05 * np.abs(lam["radiance"]).max()
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-8 * np.abs(ref[k]).max(), k


def test_modis_with_its_isotropic_kernel_alone_equals_the_lambertian_surface_on_the_device():
    s = _scene(1, 8)
    nw = len(s["albedo"])
    lam = run_product(s, derivatives=True)
    args = np.zeros((nw, 3))
    args[:, 0] = s["albedo"]
    s["brdf"] = ("modis", args)
    mod = run_product(s, derivatives=True)
    assert np.max(np.abs(lam["radiance"] - mod["radiance"])) < 1e-12 * np.abs(lam["radiance"]).max()
    assert np.max(np.abs(lam["d_albedo"] - mod["d_brdf"][..., 0])) < 1e-10 * np.abs(lam["d_albedo"]).max()
    assert np.max(np.abs(lam["d_extinction"] - mod["d_extinction"])) < 1e-10 * np.abs(lam["d_extinction"]).max()


@pytest.mark.parametrize("kind,args0,steps", [("modis", [0.1, 0.1, 0.1], [1e-3] * 3), ("kokhanovsky", [0.04], [2e-5])], ids=["modis", "kokhanovsky"])
def test_brdf_argument_weighting_functions_match_central_differences_of_the_device_radiances(kind, args0, steps):
    """tests/constituent/test_modis.py:103-160 / test_kokhanovsky.py: numeric_wf with a 0.001 step, decimal = 6."""
    s = _scene(1, 4)
    nw = len(s["albedo"])
    args = _args(args0, nw)
    eng = build_engine(s)
    s["brdf"] = (kind, args)
    base = run_product(s, derivatives=True, engine=eng)
    for a, h in enumerate(steps):
        ap, am = args.copy(), args.copy()
        ap[:, a] += h
        am[:, a] -= h
        rp = run_product({**s, "brdf": (kind, ap)}, derivatives=False, engine=eng)["radiance"]
        rm = run_product({**s, "brdf": (kind, am)}, derivatives=False, engine=eng)["radiance"]
        fd = (rp - rm) / (2 * h)
        assert np.max(np.abs(fd - base["d_brdf"][..., a])) < 1e-6 * np.abs(fd).max(), (kind, a)


def test_brdf_results_do_not_depend_on_the_wavelength_block(monkeypatch):
    s = _scene(1, 8, nw=7)
    s["brdf"] = ("kokhanovsky", _args([0.04], 7))
    whole = run_product(s, derivatives=True)
    monkeypatch.setenv("SASKTRAN2_AMD_CHUNK", "3")
    blocks = run_product(s, derivatives=True)
    for k in ("radiance", "d_extinction", "d_brdf"):
        assert np.array_equal(whole[k], blocks[k]), k


@pytest.mark.parametrize("nstokes,num_sza,nstr", [(1, 1, 8), (1, 2, 8), (3, 2, 8), (1, 1, 16)])
@pytest.mark.parametrize("kind,args0", CASES, ids=[c[0] for c in CASES])
def test_brdf_under_lines_of_sight_that_end_on_the_ground_in_spherical_geometry(nstokes, num_sza, nstr, kind, args0):
    """The reference's regression scene (two ground-viewing and two limb rays, spherical shell) over a non-Lambertian ground, diffuse field of
    the discrete-ordinates source alone (single_scatter_source = none): ground source per cos-angle grid point and order."""
    s = sc.regression_1d(3)
    s["config"].update(num_sza=num_sza, nstr=nstr)
    if nstokes == 3:
        s = sc.to_vector(s, rayleigh_fraction=0.9)
    s["brdf"] = (kind, _args(args0, 3))
    got = run_product(s, derivatives=True)
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-8 * np.abs(ref[k]).max(), k


@pytest.mark.parametrize("kind,args0", CASES, ids=[c[0] for c in CASES])
def test_discrete_ordinates_plus_exact_single_scatter_over_a_brdf_surface(kind, args0):
    """The reference's third BRDF scene (tests/constituent/test_modis.py:62-86: spherical, discrete ordinates + the default exact single scatter, a
    ground-viewing ray): radiance = DO diffuse part (oracle/disco_oracle.hpp) + single scatter with the direct bounce at pi x brdf
    (oracle/so_oracle.hpp); weighting functions of the BRDF arguments = the sum of the two sources'."""
    from tests.gpu_harness import build_so_engine, run_product_so
    from tests.test_so_oracle import make_so_oracle
    s = sc.regression_so(3, nstokes=1)
    s["config"].update(multiple_scatter="discrete_ordinates", num_streams=8)
    args = _args(args0, 3)
    eng = build_so_engine(s)
    s["brdf"] = (kind, args)
    got = run_product_so(s, engine=eng, derivatives=True)
    ss = make_so_oracle(s, multiple_scatter="none").run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    sd = dict(s, config=dict(nstr=8, num_sza=s["config"]["num_sza"], nstokes=1, ss_in_do=False))
    do = make_oracle(sd, ss_in_do=False).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    ref_rad = ss["radiance"][..., 0] + do["radiance"]
    assert np.max(np.abs(got["radiance"][..., 0] - ref_rad)) < 1e-9 * np.abs(ref_rad).max()
    ref_wf = ss["d_brdf"][:, :, 0, :] + do["d_brdf"]
    assert np.max(np.abs(got["d_brdf"] - ref_wf)) < 1e-8 * np.abs(ref_wf).max()
    assert np.abs(ss["d_brdf"]).max() > 0.05 * np.abs(ref_wf).max() and np.abs(do["d_brdf"]).max() > 0.05 * np.abs(ref_wf).max()   # both sources matter


def test_brdf_outside_the_served_configurations_is_refused_with_a_message():
    s = sc.regression_so(1, nstokes=1)   # the successive-orders source
    eng_ok = True
    from tests.gpu_harness import build_so_engine, run_product_so
    s["brdf"] = ("modis", _args([0.2, 0.05, 0.05], 1))
    with pytest.raises(RuntimeError, match="non-Lambertian|-2"):
        run_product_so(s, engine=build_so_engine(s))
    assert "non-Lambertian" in sk.last_error()


# ---- the constituent interface, as tests/constituent/test_modis.py / test_kokhanovsky.py of the reference drive it (their pseudo-spherical
#      discrete-ordinates scenario: 4 streams, one ground-viewing ray; their criterion: analytic = numeric weighting function, decimal = 6)
def _constituent_scene(sk, surface, derivs=True):
    from tests.test_gpu_frontend import _config
    c = sc.frontend_case(nwavel=5)
    cfg = _config(sk, streams=4, moments=4, delta_m_scaling=False)
    geo = sk.Geometry1D(0.6, 0.0, 6372000.0, c["altitudes"], sk.InterpolationMethod.LinearInterpolation, sk.GeometryType.PseudoSpherical)
    vg = sk.ViewingGeometry()
    vg.add_ray(sk.GroundViewingSolar(0.6, 0.3, 0.5, 200000.0))
    engine = sk.Engine(cfg, geo, vg)
    atmo = sc.frontend_product_atmosphere(sk, c, geo, cfg, calculate_derivatives=derivs)
    atmo["surface"] = surface
    return engine, atmo


def test_modis_constituent_weighting_functions_match_numeric_ones():
    surface = sk.constituent.MODIS(isotropic=0.1, volumetric=0.1, geometric=0.1)
    engine, atmo = _constituent_scene(sk, surface)
    base = engine.calculate_radiance(atmo, as_dataset=False)
    assert np.all(np.isfinite(base["radiance"]))
    for part in ("isotropic", "volumetric", "geometric"):
        wf = base[f"wf_surface_{part}"]
        assert wf.shape == (1, 5, 1, 1)   # one (wavelength-independent) parameter x wavelength x los x stokes
        h = 0.001
        vals = {}
        for sign in (+1, -1):
            setattr(surface, part, getattr(surface, part) + sign * h)
            vals[sign] = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
            setattr(surface, part, getattr(surface, part) - sign * h)
        fd = (vals[+1] - vals[-1]) / (2 * h)
        assert np.max(np.abs(wf[0] - fd)) < 1e-6 * np.abs(fd).max(), part


def test_kokhanovsky_constituent_weighting_functions_match_numeric_ones():
    ice = lambda wl: 1.31 - 1j * 2e-9 * (np.asarray(wl) / 500.0) ** 2    # a smooth stand-in for the ice data base the reference reads  # noqa: E731
    surface = sk.constituent.SnowKokhanovsky(L=3.6e6, M=5.5e-8, refractive_index_fn=ice)
    engine, atmo = _constituent_scene(sk, surface)
    base = engine.calculate_radiance(atmo, as_dataset=False)
    assert np.all(base["radiance"] > 0)
    for part, h in (("L", 3.6e3), ("M", 5.5e-11)):
        wf = base[f"wf_surface_{part}"]
        vals = {}
        for sign in (+1, -1):
            setattr(surface, part, getattr(surface, part) + sign * h)
            vals[sign] = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
            setattr(surface, part, getattr(surface, part) - sign * h)
        fd = (vals[+1] - vals[-1]) / (2 * h)
        assert np.max(np.abs(wf[0] - fd)) < 1e-6 * np.abs(fd).max(), part


def test_a_lambertian_constituent_after_a_brdf_constituent_restores_the_lambertian_surface():
    engine, atmo = _constituent_scene(sk, sk.constituent.MODIS(isotropic=0.3), derivs=False)
    modis = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
    atmo["surface"] = sk.constituent.LambertianSurface(0.3)
    lam = engine.calculate_radiance(atmo, as_dataset=False)["radiance"]
    assert np.max(np.abs(modis - lam)) < 1e-12 * np.abs(lam).max()   # MODIS with the isotropic kernel alone IS the Lambertian surface


def test_the_references_functional_brdf_known_answer_on_the_device():
    """cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp:806-845 ("BRDF surface": 32 streams, 35 lines of sight, a functional BRDF; disabled in the
    reference, its numbers are those of the legacy model): the device equals the oracle and both reproduce the 35 radiances to the truncation
    error of the azimuth expansion at the BRDF's kink (tests/test_brdf_oracle.py, tests/golden/do_scalar_brdf_legacy.json)."""
    import json
    from pathlib import Path
    root = Path(__file__).parent / "golden"
    gold = json.loads((root / "do_scalar_brdf_legacy.json").read_text())
    cases = json.loads((root / "do_scalar_planeparallel.json").read_text())["cases"]
    s = sc.planeparallel_case(next(c for c in cases if c["name"] == gold["base_case"]))
    s["brdf"] = (gold["brdf"], np.array([[1.0]]))
    got = run_product(s, derivatives=True)
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * np.abs(ref["radiance"]).max()
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-8 * np.abs(ref[k]).max(), k
    diff = np.abs(got["radiance"][0] - np.asarray(gold["expected"]))
    assert diff.max() < 3e-4 and np.median(diff) < 3e-5



@pytest.mark.parametrize("seed", range(14))
def test_random_brdf_scenes_match_the_oracle(seed):
    """Random surface arguments, stream counts, geometry types, Stokes dimensions and solar zenith profiles (nadir + limb rays in spherical
    geometry, nadir rays otherwise) against the oracle."""
    rng = np.random.default_rng(4100 + seed)
    nstokes = int(rng.choice([1, 3]))
    nstr = int(rng.choice([4, 6, 8, 10, 16] if nstokes == 1 else [4, 6, 8]))
    geotype = int(rng.choice([sc.GEO_PLANEPARALLEL, sc.GEO_PSEUDOSPHERICAL, sc.GEO_SPHERICAL]))
    nw = 3
    if geotype == sc.GEO_SPHERICAL:
        s = sc.regression_1d(nw)
        s["config"].update(num_sza=int(rng.choice([1, 2, 3])), nstr=nstr)
        s["leg"] = s["leg"][:max(nstr, 3)] if s["leg"].shape[0] > max(nstr, 3) else s["leg"]
    else:
        s = sc.nadir_c1(num_wavelengths=nw, nstr=nstr)
        s["geometry"]["geotype"] = geotype
    s["ssa"] = np.clip(s["ssa"] * rng.uniform(0.6, 1.05), 0.05, 0.999)
    s["extinction"] = s["extinction"] * rng.uniform(0.5, 2.0)
    if nstokes == 3:
        s = sc.to_vector(s, rayleigh_fraction=float(rng.uniform(0.3, 1.0)))
    if rng.random() < 0.5:
        kind, args = "modis", np.column_stack([rng.uniform(0.05, 0.5, nw), rng.uniform(0.0, 0.1, nw), rng.uniform(0.0, 0.05, nw)])
    else:
        kind, args = "kokhanovsky", rng.uniform(0.005, 0.2, (nw, 1))
    s["brdf"] = (kind, args)
    got = run_product(s, derivatives=True)
    ref = make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], derivatives=True, brdf=s["brdf"])
    assert np.all(np.isfinite(got["radiance"]))
    assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-9 * np.abs(ref["radiance"]).max(), (kind, nstr, geotype, nstokes)
    for k in ("d_extinction", "d_ssa", "d_brdf"):
        assert np.max(np.abs(got[k] - ref[k])) < 1e-7 * np.abs(ref[k]).max(), (k, kind, nstr, geotype, nstokes)


def test_vjp_and_jvp_over_a_brdf_surface_equal_the_contracted_jacobian():
    """sk_engine_calculate_vjp / _jvp (contractions of the Jacobian on this path) with the BRDF arguments among the parameters."""
    s = _scene(1, 8)
    nw, nlos = len(s["albedo"]), len(s["rays"])
    args = _args([0.2, 0.05, 0.05], nw)
    eng = build_engine(s)
    engine, config, geometry = eng
    s["brdf"] = ("modis", args)
    full = run_product(s, derivatives=True, engine=eng)
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.set_brdf("modis")
    atmosphere.surface.brdf_args[:] = args.T
    rng = np.random.default_rng(5)
    cot = rng.normal(size=(nw, nlos, 1))
    rad, grads = engine.calculate_vjp(atmosphere, cot)
    assert np.max(np.abs(rad[:, :, 0] - full["radiance"])) < 1e-13 * np.abs(full["radiance"]).max()
    for a in range(3):
        want = np.einsum("wl,wl->w", cot[:, :, 0], full["d_brdf"][:, :, a])
        assert np.max(np.abs(grads[f"wf_brdf_{a}"] - want)) < 1e-10 * np.abs(want).max(), a
    want = np.einsum("wl,wlq->q", cot[:, :, 0], full["d_extinction"])
    assert np.max(np.abs(grads["wf_extinction"] - want)) < 1e-10 * np.abs(want).max()
    t_surf = {f"wf_brdf_{a}": rng.normal(size=nw) for a in range(3)}
    t_ext = rng.normal(size=len(s["geometry"]["altitudes"])) * 1e-6
    rad2, jvp = engine.calculate_jvp(atmosphere, {"wf_extinction": t_ext}, surface_tangents=t_surf)
    want = np.einsum("wlq,q->wl", full["d_extinction"], t_ext) + sum(full["d_brdf"][:, :, a] * t_surf[f"wf_brdf_{a}"][:, None] for a in range(3))
    assert np.max(np.abs(jvp[:, :, 0] - want)) < 1e-10 * np.abs(want).max()
