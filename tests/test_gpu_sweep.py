"""A sweep of small configurations through the product against the oracle: stream counts 2 .. 16 (2 .. 8 polarised), plane-parallel /
pseudo-spherical / spherical geometry, one or two solar zenith angles in the table, ground-viewing and limb lines of sight, observers inside the
atmosphere, with weighting functions -- every stream count on the kernels that serve it by default and, for the matrix-core counts, on the run-time-n
kernels as well.  (Round 4 added it after a work-space overrun of the run-time-n kernels at n < 12 had gone unnoticed: no test had compared the small
stream counts of that path with the oracle in a scene with low single-scatter albedo.)"""
import numpy as np
import pytest

from tests import scenarios as sc
from tests.gpu_harness import run_product
from tests.test_oracle_golden import run_oracle

pytestmark = pytest.mark.gpu


def _scene(nstr, geotype, num_sza, nstokes, nw=2):
    ng = 13
    alt = np.linspace(0.0, 36_000.0, ng)
    k = (5e-5 * np.exp(-alt / 7000.0) + 2e-7)[:, None] * np.linspace(0.5, 2.2, nw)[None, :]
    ssa = np.clip(0.55 + 0.35 * np.sin(alt / 8000.0), 0.2, 0.97)[:, None] * np.ones((1, nw))
    l = np.arange(16)
    leg = ((2 * l + 1) * 0.6 ** l)[:, None, None] * np.ones((1, ng, nw))
    csz = 0.55
    rays = [(sc.GROUND_VIEWING_SOLAR, csz, 0.4, 200_000.0, 0.9), (sc.GROUND_VIEWING_SOLAR, csz, 2.1, 200_000.0, 0.45),
            (sc.GROUND_VIEWING_SOLAR, csz, 1.0, 14_300.0, 0.7)]   # (the last one from inside the atmosphere)
    if geotype == sc.GEO_SPHERICAL:
        rays += [(sc.TANGENT_ALTITUDE_SOLAR, csz, -0.3, 200_000.0, 9_500.0), (sc.TANGENT_ALTITUDE_SOLAR, csz, 0.8, 200_000.0, 22_250.0)]
    s = dict(geometry=dict(cos_sza=csz, saa=0.2, earth_radius=6_372_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR, geotype=geotype),
             config=dict(nstr=nstr, num_sza=num_sza), rays=rays, extinction=k, ssa=ssa, leg=leg, albedo=np.linspace(0.05, 0.6, nw))
    return sc.to_vector(s, rayleigh_fraction=0.7) if nstokes == 3 else s


def _check(s):
    got, ref = run_product(s), run_oracle(s, derivatives=True)
    S = s["config"].get("nstokes", 1)
    if S == 1:
        np.testing.assert_allclose(got["radiance"], ref["radiance"], rtol=1e-9)
        axis = 2
    else:
        np.testing.assert_allclose(got["radiance"][..., 0], ref["radiance"][..., 0], rtol=1e-9)
        assert np.max(np.abs(got["radiance"] - ref["radiance"])) < 1e-10 * np.abs(ref["radiance"]).max()
        axis = 3
    for key in ("d_extinction", "d_ssa"):
        scale = np.abs(ref[key]).max(axis=axis, keepdims=True)
        assert np.max(np.abs(got[key] - ref[key]) / np.where(scale > 0, scale, 1.0)) < 1e-7, key
    np.testing.assert_allclose(got["d_albedo"], ref["d_albedo"], rtol=1e-7, atol=1e-13 * np.abs(ref["radiance"]).max())


GEOS = [(sc.GEO_PLANEPARALLEL, "plane parallel"), (sc.GEO_PSEUDOSPHERICAL, "pseudo-spherical"), (sc.GEO_SPHERICAL, "spherical")]


@pytest.mark.parametrize("geotype", [g for g, _ in GEOS], ids=[n for _, n in GEOS])
@pytest.mark.parametrize("nstr", [2, 4, 6, 8, 10, 12, 16])
def test_scalar_stream_counts_match_the_oracle(nstr, geotype):
    _check(_scene(nstr, geotype, 2 if geotype == sc.GEO_SPHERICAL else 1, 1))


@pytest.mark.parametrize("geotype", [g for g, _ in GEOS], ids=[n for _, n in GEOS])
@pytest.mark.parametrize("nstr", [2, 4, 8, 16])
def test_matrix_core_stream_counts_on_the_runtime_n_kernels_match_the_oracle(nstr, geotype, monkeypatch):
    monkeypatch.setenv("SASKTRAN2_AMD_RUNTIME_N", "1")
    _check(_scene(nstr, geotype, 1, 1))


@pytest.mark.parametrize("geotype", [g for g, _ in GEOS], ids=[n for _, n in GEOS])
@pytest.mark.parametrize("nstr", [2, 4, 6, 8])
def test_polarised_stream_counts_match_the_oracle(nstr, geotype):
    _check(_scene(nstr, geotype, 2 if geotype == sc.GEO_SPHERICAL else 1, 3))


def _check_delta_m(s):
    """delta-M scaled solve (the product scales storage and mappings, the oracle gets the scaled inputs and the truncation fraction):
    user-level weighting functions are the native ones mapped back to the unscaled inputs (scenarios.delta_m_scaled)."""
    from tests.gpu_harness import build_engine
    from tests.test_oracle_golden import make_oracle
    d = sc.delta_m_scaled(s)
    o = make_oracle(d).run(d["extinction"], d["ssa"], d["leg"], d["albedo"], derivatives=True, delta_f=d["delta_f"])
    engine, config, geometry = build_engine(s)
    config.delta_m_scaling = True
    got = run_product(s, engine=(engine, config, geometry))
    u = d["unscale"]
    if s["config"].get("nstokes", 1) == 1:
        np.testing.assert_allclose(got["radiance"], o["radiance"], rtol=1e-9)
        want_k = o["d_extinction"] * u["dk_dk"].T[:, None, :]
        want_w = o["d_ssa"] * u["dw_dw"].T[:, None, :] + o["d_extinction"] * u["dk_dw"].T[:, None, :]
        axis = 2
    else:
        np.testing.assert_allclose(got["radiance"][..., 0], o["radiance"][..., 0], rtol=1e-9)
        want_k = o["d_extinction"] * u["dk_dk"].T[:, None, None, :]
        want_w = o["d_ssa"] * u["dw_dw"].T[:, None, None, :] + o["d_extinction"] * u["dk_dw"].T[:, None, None, :]
        axis = 3
    for name, want in (("d_extinction", want_k), ("d_ssa", want_w)):
        scale = np.abs(want).max(axis=axis, keepdims=True)
        assert np.max(np.abs(got[name] - want) / np.where(scale > 0, scale, 1.0)) < 1e-7, name


@pytest.mark.parametrize("runtime_n", [False, True], ids=["default kernels", "run-time-n kernels"])
@pytest.mark.parametrize("geotype", [sc.GEO_PLANEPARALLEL, sc.GEO_SPHERICAL], ids=["plane parallel", "spherical"])
@pytest.mark.parametrize("nstr", [4, 6, 8, 12])
def test_delta_m_scaled_stream_counts_match_the_oracle(nstr, geotype, runtime_n, monkeypatch):
    if runtime_n:
        monkeypatch.setenv("SASKTRAN2_AMD_RUNTIME_N", "1")
    _check_delta_m(_scene(nstr, geotype, 1, 1))


@pytest.mark.parametrize("nstr", [2, 4, 8])
def test_delta_m_scaled_polarised_stream_counts_match_the_oracle(nstr):
    _check_delta_m(_scene(nstr, sc.GEO_SPHERICAL, 1, 3))
