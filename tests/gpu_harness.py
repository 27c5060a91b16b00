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
    brdf = s.get("brdf")   # (kind, args [nwavel, nargs]): a non-Lambertian surface; its weighting functions come back as d_brdf [wavel, los, (stokes,) argument]
    if brdf is not None:
        atmosphere.surface.set_brdf(brdf[0])
        atmosphere.surface.brdf_args[:] = np.asarray(brdf[1], dtype=float).T
    res = engine.calculate_radiance(atmosphere, as_dataset=False)
    if s["config"].get("nstokes", 1) == 3:
        out = {"radiance": res["radiance"], "los_optical_depth": res["los_optical_depth"]}
        if derivatives:   # [wavel, los, stokes, altitude] as the oracle front end returns them
            out["d_extinction"] = np.transpose(res["wf_extinction"], (1, 2, 3, 0))
            out["d_ssa"] = np.transpose(res["wf_ssa"], (1, 2, 3, 0))
            if brdf is None:
                out["d_albedo"] = res["wf_albedo"]
            else:
                out["d_brdf"] = np.transpose(res["wf_brdf"], (1, 2, 3, 0))
        return out
    out = {"radiance": res["radiance"][:, :, 0], "los_optical_depth": res["los_optical_depth"]}
    if derivatives:
        out["d_extinction"] = np.transpose(res["wf_extinction"][:, :, :, 0], (1, 2, 0))
        out["d_ssa"] = np.transpose(res["wf_ssa"][:, :, :, 0], (1, 2, 0))
        if brdf is None:
            out["d_albedo"] = res["wf_albedo"][:, :, 0]
        else:
            out["d_brdf"] = np.transpose(res["wf_brdf"][:, :, :, 0], (1, 2, 0))
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
    config.single_scatter_source = {"exact": sk.SingleScatterSource.Exact, "solartable": sk.SingleScatterSource.SolarTable, "none": sk.SingleScatterSource.NoSource}[c["single_scatter"]]
    config.multiple_scatter_source = {"successive_orders_legacy": sk.MultipleScatterSource.SuccessiveOrdersLegacy,
                                      "successive_orders": sk.MultipleScatterSource.SuccessiveOrders,
                                      "discrete_ordinates": sk.MultipleScatterSource.DiscreteOrdinates,
                                      "none": sk.MultipleScatterSource.NoSource}[c["multiple_scatter"]]
    config.num_successive_orders_iterations = c["num_iterations"]
    config.num_successive_orders_incoming = c["num_incoming"]
    config.num_successive_orders_outgoing = c["num_outgoing"]
    config.init_successive_orders_with_discrete_ordinates = False
    if c["multiple_scatter"] == "successive_orders":
        config.successive_orders_relative_tolerance = c.get("relative_tolerance", 1e-6)
        config.successive_orders_absolute_tolerance = c.get("absolute_tolerance", 1e-12)
        config.successive_orders_damping = c.get("damping", 1.0)
        config.successive_orders_anderson_depth = c.get("anderson_depth", 3)
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
    `derivatives` the weighting functions in the oracle's convention ([wavelength, los, stokes, altitude]); with an emission source also
    d_emission / d_surface_emission through "wf_emission" / "wf_surface_emission" mappings with d_emission = 1."""
    import ctypes
    engine, config, geometry = engine if engine is not None else build_so_engine(s, **override)
    ng, nw = s["extinction"].shape
    nlos, S = len(s["rays"]), int(config.num_stokes)
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
    brdf = s.get("brdf")   # (kind, args [nwavel, nargs]): non-Lambertian surface; d_brdf [wavelength, los, argument] (intensity component)
    if brdf is not None:
        atmosphere.surface.set_brdf(brdf[0])
        atmosphere.surface.brdf_args[:] = np.asarray(brdf[1], dtype=float).T
    dp, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
    with_emission = derivatives and int(config.emission_source) in (int(sk.EmissionSource.Standard), int(sk.EmissionSource.VolumeEmissionRate))

    def hook(L, h_st, h_out, out):
        def fill(getter, mapping, values):
            f = getattr(L, getter)
            f.argtypes = [vp, ctypes.POINTER(dp)]
            ptr = dp()
            assert f(mapping, ctypes.byref(ptr)) == 0
            np.ctypeslib.as_array(ptr, shape=(values.size,))[:] = values
        m = vp()
        L.sk_atmosphere_storage_get_derivative_mapping(h_st, b"wf_emission", ctypes.byref(m))
        fill("sk_deriv_mapping_get_d_emission", m, np.ones(ng * nw))
        out["wf_emission"] = np.zeros((ng, nw, nlos, S))
        L.sk_output_assign_derivative_memory(h_out, b"wf_emission", out["wf_emission"].ctypes.data_as(dp), nw * nlos, S, ng)
        L.sk_surface_get_derivative_mapping(out["_surface_handle"], b"wf_surface_emission", ctypes.byref(m))
        fill("sk_surface_deriv_mapping_get_d_emission", m, np.ones(nw))
        out["wf_surface_emission"] = np.zeros((nw, nlos, S))
        L.sk_output_assign_surface_derivative_memory(h_out, b"wf_surface_emission", out["wf_surface_emission"].ctypes.data_as(dp), nw * nlos, S)

    res = engine.calculate_radiance(atmosphere, as_dataset=False, _mapping_hook=hook if with_emission else None)
    out = {"radiance": res["radiance"], "los_optical_depth": res["los_optical_depth"]}
    if derivatives:
        out["d_extinction"] = np.transpose(res["wf_extinction"], (1, 2, 3, 0))
        out["d_ssa"] = np.transpose(res["wf_ssa"], (1, 2, 3, 0))
        if brdf is None:
            out["d_albedo"] = res["wf_albedo"]
        else:
            out["d_brdf"] = np.transpose(res["wf_brdf"][:, :, :, 0], (1, 2, 0))
        if with_emission:
            out["d_emission"] = np.transpose(res["wf_emission"], (1, 2, 3, 0))
            out["d_surface_emission"] = res["wf_surface_emission"]
    return out


def build_ts_engine(s, solar=True, thermal=False, single_scatter="none", delta_m=False):
    """Engine with the two-stream source for a tests/scenarios.py scenario (multiple_scatter_source = twostream and / or emission_source =
    twostream, as the reference's tests/engine/test_twostream.py:8-31 configures it)."""
    g = s["geometry"]
    config = sk.Config()
    config.num_streams = 2
    config.num_sza = s["config"]["num_sza"]
    config.do_backprop = True
    config.num_singlescatter_moments = s["leg"].shape[0]
    config.delta_m_scaling = delta_m
    config.single_scatter_source = {"exact": sk.SingleScatterSource.Exact, "solartable": sk.SingleScatterSource.SolarTable, "none": sk.SingleScatterSource.NoSource}[single_scatter]
    config.multiple_scatter_source = sk.MultipleScatterSource.TwoStream if solar else sk.MultipleScatterSource.NoSource
    config.emission_source = sk.EmissionSource.TwoStream if thermal else sk.EmissionSource.NoSource
    geometry = sk.Geometry1D(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"],
                             sk.InterpolationMethod(g["interp_method"]), sk.GeometryType(g["geotype"]))
    viewing = sk.ViewingGeometry()
    for kind, csz, az, obs, x in s["rays"]:
        if kind == sc.TANGENT_ALTITUDE_SOLAR:
            viewing.add_ray(sk.TangentAltitudeSolar(x, az, obs, csz))
        elif kind == sc.VIEWING_UP_SOLAR:
            viewing.add_ray(sk.SolarAnglesObserverLocation(csz, az, x, obs))
        else:
            viewing.add_ray(sk.GroundViewingSolar(csz, az, x, obs))
    return sk.Engine(config, geometry, viewing), config, geometry


def run_product_ts(s, engine=None, derivatives=True, solar=True, thermal=False, b1_mapping=True, **kw):
    """The two-stream source through the C ABI, in the two-stream oracle's output convention: radiance [wavelength, los], d_extinction /
    d_ssa / d_b1 / d_emission [wavelength, los, altitude], d_albedo / d_surface_emission [wavelength, los].  d_b1 and the emission
    derivatives come out of derivative mappings registered the way the reference's tests do (rust/sasktran2-rs/tests/twostream.rs:38-83):
    "wf_b1" with d_legendre(1) = 1, scat_factor = 1; "wf_emission" / "wf_surface_emission" with d_emission = 1."""
    import ctypes
    engine, config, geometry = engine if engine is not None else build_ts_engine(s, solar=solar, thermal=thermal, **kw)
    ng, nw = s["extinction"].shape
    nlos, nleg = len(s["rays"]), s["leg"].shape[0]
    dp, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
    atmosphere = sk.Atmosphere(geometry, config, numwavel=nw, calculate_derivatives=derivatives)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    if s.get("solar_irradiance") is not None:
        atmosphere.storage.solar_irradiance[:] = s["solar_irradiance"]
    if thermal:
        atmosphere.storage.emission_source[:] = s["emission"]
        atmosphere.surface.emission[:] = s["surface_emission"]

    def fill(L, getter, mapping, values):
        f = getattr(L, getter)
        f.argtypes = [vp, ctypes.POINTER(dp)]
        ptr = dp()
        assert f(mapping, ctypes.byref(ptr)) == 0
        np.ctypeslib.as_array(ptr, shape=(values.size,))[:] = values

    def hook(L, h_st, h_out, out):
        m = vp()
        if b1_mapping:
            L.sk_atmosphere_storage_get_derivative_mapping(h_st, b"wf_b1", ctypes.byref(m))
            dleg = np.zeros((nleg, ng, nw))
            dleg[1] = 1.0
            fill(L, "sk_deriv_mapping_get_d_legendre", m, np.asfortranarray(dleg).ravel(order="F"))
            fill(L, "sk_deriv_mapping_get_scat_factor", m, np.ones(ng * nw))
            out["wf_b1"] = np.zeros((ng, nw, nlos, 1))
            L.sk_output_assign_derivative_memory(h_out, b"wf_b1", out["wf_b1"].ctypes.data_as(dp), nw * nlos, 1, ng)
        if thermal:
            L.sk_atmosphere_storage_get_derivative_mapping(h_st, b"wf_emission", ctypes.byref(m))
            fill(L, "sk_deriv_mapping_get_d_emission", m, np.ones(ng * nw))
            out["wf_emission"] = np.zeros((ng, nw, nlos, 1))
            L.sk_output_assign_derivative_memory(h_out, b"wf_emission", out["wf_emission"].ctypes.data_as(dp), nw * nlos, 1, ng)
            L.sk_surface_get_derivative_mapping(out["_surface_handle"], b"wf_surface_emission", ctypes.byref(m))
            fill(L, "sk_surface_deriv_mapping_get_d_emission", m, np.ones(nw))
            out["wf_surface_emission"] = np.zeros((nw, nlos, 1))
            L.sk_output_assign_surface_derivative_memory(h_out, b"wf_surface_emission", out["wf_surface_emission"].ctypes.data_as(dp), nw * nlos, 1)

    res = engine.calculate_radiance(atmosphere, as_dataset=False, _mapping_hook=hook if derivatives else None)
    out = {"radiance": res["radiance"][:, :, 0], "los_optical_depth": res["los_optical_depth"]}
    if derivatives:
        for name, key in (("d_extinction", "wf_extinction"), ("d_ssa", "wf_ssa"), ("d_b1", "wf_b1"), ("d_emission", "wf_emission")):
            out[name] = np.transpose(res[key][..., 0], (1, 2, 0)) if key in res else np.zeros((nw, nlos, ng))
        out["d_albedo"] = res["wf_albedo"][:, :, 0]
        out["d_surface_emission"] = res["wf_surface_emission"][:, :, 0] if "wf_surface_emission" in res else np.zeros((nw, nlos))
    return out
