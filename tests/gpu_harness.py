"""Runs a tests/scenarios.py scenario through the product (Python mirror -> C ABI -> HIP) and returns arrays in the
oracle's output convention ([wavelength, los] radiance, [wavelength, los, altitude] derivatives)."""
from __future__ import annotations

import numpy as np

import sasktran2_amd as sk
from tests import scenarios as sc


def build_engine(s):
    g = s["geometry"]
    config = sk.Config()
    config.num_streams = s["config"]["nstr"]
    config.num_sza = s["config"]["num_sza"]
    nstokes = s["config"].get("nstokes", 1)
    config.num_stokes = nstokes
    config.stokes_basis = s["config"].get("stokes_basis", 0)
    config.num_singlescatter_moments = s["leg"].shape[0] // (4 if nstokes == 3 else 1)
    config.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
    config.single_scatter_source = sk.SingleScatterSource.DiscreteOrdinates
    geometry = sk.Geometry1D(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"],
                             sk.InterpolationMethod(g["interp_method"]), sk.GeometryType(g["geotype"]))
    viewing = sk.ViewingGeometry()
    for kind, csz, az, obs, x in s["rays"]:
        if kind == sc.TANGENT_ALTITUDE_SOLAR:
            viewing.add_ray(sk.TangentAltitudeSolar(x, az, obs, csz))
        elif kind == sc.TANGENT_ALTITUDE:
            viewing.add_ray(sk.TangentAltitude(x, obs, csz, az))
        elif kind == sc.VIEWING_UP_SOLAR:
            viewing.add_ray(sk.SolarAnglesObserverLocation(csz, az, x, obs))
        else:
            viewing.add_ray(sk.GroundViewingSolar(csz, az, x, obs))
    return sk.Engine(config, geometry, viewing), config, geometry


def run_product(s, derivatives=True, engine=None):
    if engine is None:
        engine, config, geometry = build_engine(s)
    else:
        engine, config, geometry = engine
    nw = s["extinction"].shape[1]
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=derivatives)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    if s["config"].get("nstokes", 1) == 3:
        atmosphere.storage.leg_coeff[:] = s["leg"]   # (a1, a2, a3, b1) per order, as the reference stores them
    else:
        atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
    if s["config"].get("nstokes", 1) == 3:
        out = {"radiance": res["radiance"], "los_optical_depth": res["los_optical_depth"]}
        if derivatives:   # [wavel, los, stokes, altitude] as the oracle front end returns them
            out["d_extinction"] = np.transpose(res["wf_extinction"], (1, 2, 3, 0))
            out["d_ssa"] = np.transpose(res["wf_ssa"], (1, 2, 3, 0))
            out["d_albedo"] = res["wf_albedo"]
        return out
    out = {"radiance": res["radiance"][:, :, 0], "los_optical_depth": res["los_optical_depth"]}
    if derivatives:
        out["d_extinction"] = np.transpose(res["wf_extinction"][:, :, :, 0], (1, 2, 0))
        out["d_ssa"] = np.transpose(res["wf_ssa"][:, :, :, 0], (1, 2, 0))
        out["d_albedo"] = res["wf_albedo"][:, :, 0]
    return out


def build_so_engine(s, **override):
    """Engine for a tests/scenarios.py successive-orders scenario (config keys of scenarios.regression_so)."""
    g, c = s["geometry"], dict(s["config"], **override)
    config = sk.Config()
    config.num_stokes = c["nstokes"]
    config.stokes_basis = c.get("stokes_basis", 0)
    config.num_streams = c["num_streams"]
    config.num_singlescatter_moments = c["num_singlescatter_moments"]
    config.num_sza = c["num_sza"]
    config.single_scatter_source = {"exact": sk.SingleScatterSource.Exact, "none": sk.SingleScatterSource.NoSource}[c["single_scatter"]]
    config.multiple_scatter_source = {"successive_orders_legacy": sk.MultipleScatterSource.SuccessiveOrdersLegacy,
                                      "discrete_ordinates": sk.MultipleScatterSource.DiscreteOrdinates,
                                      "none": sk.MultipleScatterSource.NoSource}[c["multiple_scatter"]]
    config.num_successive_orders_iterations = c["num_iterations"]
    config.num_successive_orders_incoming = c["num_incoming"]
    config.num_successive_orders_outgoing = c["num_outgoing"]
    config.init_successive_orders_with_discrete_ordinates = False
    config.emission_source = {"none": sk.EmissionSource.NoSource, "standard": sk.EmissionSource.Standard,
                              "volume_emission_rate": sk.EmissionSource.VolumeEmissionRate}[c.get("emission", "none")]
    config.occultation_source = sk.OccultationSource.Standard if c.get("occultation", False) else sk.OccultationSource.NoSource
    geometry = sk.Geometry1D(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"],
                             sk.InterpolationMethod(g["interp_method"]), sk.GeometryType(g["geotype"]))
    viewing = sk.ViewingGeometry()
    for kind, csz, az, obs, x in s["rays"]:
        if kind == sc.TANGENT_ALTITUDE_SOLAR:
            viewing.add_ray(sk.TangentAltitudeSolar(x, az, obs, csz))
        elif kind == sc.TANGENT_ALTITUDE:
            viewing.add_ray(sk.TangentAltitude(x, obs, csz, az))
        elif kind == sc.VIEWING_UP_SOLAR:
            viewing.add_ray(sk.SolarAnglesObserverLocation(csz, az, x, obs))
        else:
            viewing.add_ray(sk.GroundViewingSolar(csz, az, x, obs))
    return sk.Engine(config, geometry, viewing), config, geometry


def run_product_so(s, engine=None, solar_irradiance=None, derivatives=False, emission_source=None, surface_emission=None, **override):
    """Radiances [wavelength, los, stokes] and LOS optical depths of the exact single-scatter / successive-orders path; with
    `derivatives` the weighting functions in the oracle's convention ([wavelength, los, stokes, altitude])."""
    engine, config, geometry = engine if engine is not None else build_so_engine(s, **override)
    nw = s["extinction"].shape[1]
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=derivatives)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.storage.leg_coeff[:] = s["leg"]
    if solar_irradiance is not None:
        atmosphere.storage.solar_irradiance[:] = solar_irradiance
    if emission_source is not None:
        atmosphere.storage.emission_source[:] = emission_source
    if surface_emission is not None:
        atmosphere.surface.emission[:] = surface_emission
    atmosphere.surface.albedo[:] = s["albedo"]
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
    out = {"radiance": res["radiance"], "los_optical_depth": res["los_optical_depth"]}
    if derivatives:
        out["d_extinction"] = np.transpose(res["wf_extinction"], (1, 2, 3, 0))
        out["d_ssa"] = np.transpose(res["wf_ssa"], (1, 2, 3, 0))
        out["d_albedo"] = res["wf_albedo"]
    return out
