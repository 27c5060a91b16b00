"""Flux observers of the plane-parallel discrete-ordinates source (upwelling, downwelling and actinic flux through a horizontal
surface at an altitude; DOSourcePlaneParallelPostProcessing::compute_flux, cpp/lib/sktran_disco/source_term/do_source_planeparallel.cpp:221-566).

The criterion of the reference's own tests (tests/engine/test_flux.py:9-66, 69-160): with one azimuth order (`num_forced_azimuth = 1`)
the upwelling flux equals the double-Gauss quadrature  sum_l w_l mu_l 2 pi I(-mu_l)  of the radiances an observer at the same altitude
sees looking down along the stream directions, to rtol 2e-8.  The oracle's restatement is held to that and to the identities of the
boundary conditions (top of the atmosphere: the downwelling flux is the direct beam; Lambertian surface: upwelling = albedo x downwelling);
the device is held to the oracle and to the same criterion on its own radiances.
"""
import numpy as np
import pytest
from scipy.special import roots_legendre

from tests import scenarios as sc
from tests.test_oracle_golden import make_oracle

ALTITUDES = [39_000.0, 20_000.0, 40_000.0, 999.99, 0.0, 12_345.6]   # inside the top layer, on a level, the top, near the ground, the ground


def flux_scene(nstr=8, thermal=False, nw=4):
    """A sunlit (and, with `thermal`, emitting) plane-parallel atmosphere; one group of `nstr / 2` lines of sight per flux observer,
    looking down along the stream directions from the observer's altitude at azimuth 0."""
    alt = np.linspace(0.0, 40_000.0, 11)
    k = (1.5e-4 * np.exp(-alt / 8_000.0)[:, None] + 2e-7) * np.linspace(0.5, 2.0, nw)[None, :]
    ssa = np.clip(0.6 + 0.3 * np.sin(alt / 9_000.0)[:, None] * np.ones((1, nw)), 0.05, 0.97)
    l = np.arange(nstr)
    leg = ((2 * l + 1) * 0.55 ** l)[:, None, None] * np.ones((1, alt.size, nw))
    nodes, weights = roots_legendre(nstr // 2)
    nodes, weights = 0.5 * nodes + 0.5, 0.5 * weights
    rays = [(sc.GROUND_VIEWING_SOLAR, 0.6, 0.0, a, n) for a in ALTITUDES for n in nodes]
    s = dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_372_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR, geotype=sc.GEO_PLANEPARALLEL),
             config=dict(nstr=nstr, num_sza=1), rays=rays, extinction=k, ssa=ssa, leg=leg, albedo=np.linspace(0.0, 0.8, nw),
             solar_irradiance=np.linspace(1.0, 3.0e-1, nw), integrand=weights * nodes * 2.0 * np.pi)
    if thermal:
        em = (2.0e-2 * (1 + 0.6 * np.sin(alt / 6_000.0)))[:, None] * np.linspace(0.8, 1.3, nw)[None, :]
        em[3], em[4] = em[2], em[2]   # two layers of constant emission (b1 = 0)
        s.update(emission=em, surface_emission=np.linspace(4.0e-2, 0.0, nw))
    return s


def oracle_fluxes(s, forced_azimuth=0):
    extra = dict(emission=s["emission"], surface_emission=s["surface_emission"]) if "emission" in s else {}
    return make_oracle(s).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], solar_irradiance=s["solar_irradiance"],
                              forced_azimuth=forced_azimuth, flux_altitudes=ALTITUDES, **extra)


def quadrature_of_radiances(s, radiance):
    """[nwavel, observer]: sum over the observer's group of lines of sight of w mu 2 pi I."""
    return radiance.reshape(radiance.shape[0], len(ALTITUDES), -1) @ s["integrand"]


@pytest.mark.parametrize("thermal", [False, True], ids=["sunlit", "sunlit and emitting"])
@pytest.mark.parametrize("nstr", [8, 10])
def test_oracle_upwelling_flux_is_the_quadrature_of_its_radiances(nstr, thermal):
    """tests/engine/test_flux.py:9-66 (and :69-160 with the thermal source) on the oracle: one azimuth order, rtol 2e-8."""
    s = flux_scene(nstr, thermal)
    out = oracle_fluxes(s, forced_azimuth=1)
    assert np.all(out["num_azimuth"] == 1)
    np.testing.assert_allclose(out["upwelling_flux"], quadrature_of_radiances(s, out["radiance"]), rtol=2e-8, atol=1e-16)   # (atol: black surface, observer on the ground)


def test_oracle_fluxes_obey_the_boundary_conditions():
    s = flux_scene(8)
    out = oracle_fluxes(s)
    top, ground = ALTITUDES.index(40_000.0), ALTITUDES.index(0.0)
    # no diffuse light enters at the top: the downwelling flux there is the direct beam, the actinic flux the beam plus the upwelling intensities
    np.testing.assert_allclose(out["downwelling_flux"][:, top], s["solar_irradiance"] * 0.6, rtol=1e-12)
    assert np.all(out["actinic_flux"][:, top] >= s["solar_irradiance"])
    # Lambertian surface: what goes up at the ground is the albedo times what comes down there, direct beam included.  The reference adds
    # the direct beam as it is at the CEILING of the observer's layer (do_source_planeparallel.cpp:544-566), so the beam is moved to the
    # ground by hand here: layer depths of the linearly interpolated extinction, slant factor 1 / cos(sza) in plane-parallel geometry
    alt, k = s["geometry"]["altitudes"], s["extinction"]
    tau = 0.5 * (k[1:] + k[:-1]) * np.diff(alt)[:, None]
    beam_ceiling = s["solar_irradiance"] * np.exp(-tau[1:].sum(axis=0) / 0.6) * 0.6
    beam_ground = s["solar_irradiance"] * np.exp(-tau.sum(axis=0) / 0.6) * 0.6
    np.testing.assert_allclose(out["upwelling_flux"][:, ground], s["albedo"] * (out["downwelling_flux"][:, ground] - beam_ceiling + beam_ground),
                               rtol=1e-9, atol=1e-15)
    # the fluxes do not depend on the number of azimuth orders that are solved afterwards
    one = oracle_fluxes(s, forced_azimuth=1)
    for name in ("upwelling_flux", "downwelling_flux", "actinic_flux"):
        np.testing.assert_array_equal(one[name], out[name])


# ------------------------------------------------------------------------------------------------ device
def run_device(s, forced_azimuth=0, flux_types=None, constituents_free=True):
    import sasktran2_amd as sk
    g = s["geometry"]
    config = sk.Config()
    config.num_streams = s["config"]["nstr"]
    config.num_singlescatter_moments = s["leg"].shape[0]
    config.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
    config.single_scatter_source = sk.SingleScatterSource.DiscreteOrdinates
    if "emission" in s:
        config.emission_source = sk.EmissionSource.DiscreteOrdinates
    if forced_azimuth:
        config.num_forced_azimuth = forced_azimuth
    if flux_types is not None:
        config.flux_types = flux_types
    geometry = sk.Geometry1D(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], sk.InterpolationMethod(g["interp_method"]), sk.GeometryType(g["geotype"]))
    viewing = sk.ViewingGeometry()
    for a in ALTITUDES:
        viewing.add_flux_observer(sk.FluxObserverSolar(g["cos_sza"], a))
    for kind, csz, az, obs, x in s["rays"]:
        viewing.add_ray(sk.GroundViewingSolar(csz, az, x, obs))
    engine = sk.Engine(config, geometry, viewing)
    nw = s["extinction"].shape[1]
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=False)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.storage.solar_irradiance[:] = s["solar_irradiance"]
    if "emission" in s:
        atmosphere.storage.emission_source[:] = s["emission"]
        atmosphere.surface.emission[:] = s["surface_emission"]
    atmosphere.surface.albedo[:] = s["albedo"]
    return engine.calculate_radiance(atmosphere, as_dataset=False)


@pytest.mark.gpu
@pytest.mark.parametrize("thermal", [False, True], ids=["sunlit", "sunlit and emitting"])
@pytest.mark.parametrize("nstr", [4, 8, 16])
def test_device_fluxes_match_the_oracle(nstr, thermal, monkeypatch):
    s = flux_scene(nstr, thermal, nw=7)
    ref = oracle_fluxes(s)
    import sasktran2_amd as sk
    got = run_device(s, flux_types=[sk.FluxType.Actinic, sk.FluxType.Upwelling, sk.FluxType.Downwelling])
    for name in ("upwelling_flux", "downwelling_flux", "actinic_flux"):
        assert got[name].shape == (7, len(ALTITUDES))
        np.testing.assert_allclose(got[name], ref[name], rtol=1e-8, atol=1e-14, err_msg=name)
    np.testing.assert_allclose(got["radiance"][:, :, 0], ref["radiance"], rtol=1e-8)
    # the default flux types are upwelling and downwelling (Config::flux_types); blocks of three wavelengths give the same numbers
    monkeypatch.setenv("SASKTRAN2_AMD_CHUNK", "3")
    blocks = run_device(s)
    assert "actinic_flux" not in blocks
    np.testing.assert_array_equal(blocks["upwelling_flux"], got["upwelling_flux"])
    np.testing.assert_array_equal(blocks["downwelling_flux"], got["downwelling_flux"])


@pytest.mark.gpu
@pytest.mark.parametrize("thermal", [False, True], ids=["sunlit", "sunlit and emitting"])
def test_device_upwelling_flux_is_the_quadrature_of_its_radiances(thermal):
    """tests/engine/test_flux.py:9-66, 69-160 through the C ABI: num_forced_azimuth = 1, rtol 2e-8."""
    s = flux_scene(8, thermal)
    got = run_device(s, forced_azimuth=1)
    np.testing.assert_allclose(got["upwelling_flux"], quadrature_of_radiances(s, got["radiance"][:, :, 0]), rtol=2e-8, atol=1e-16)
    ref = oracle_fluxes(s, forced_azimuth=1)
    np.testing.assert_allclose(got["radiance"][:, :, 0], ref["radiance"], rtol=1e-8)


@pytest.mark.gpu
def test_flux_observers_are_refused_where_the_reference_has_no_flux():
    """Spherical geometry (the reference computes fluxes in its plane-parallel post-processing only), and flux weighting functions."""
    import sasktran2_amd as sk
    s = flux_scene(8)
    s["geometry"] = dict(s["geometry"], geotype=sc.GEO_SPHERICAL)
    with pytest.raises((RuntimeError, ValueError), match="flux observers"):
        run_device(s)
