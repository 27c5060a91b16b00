"""CPU checks of the two-stream oracle (oracle/ts_oracle.hpp) with the criteria of the reference's own tests for this source
(/root/reference/tests/engine/test_twostream.py, rust/sasktran2-rs/tests/twostream.rs): agreement with the two-stream discrete-ordinates
oracle (itself pinned by the reference's stored vectors), central finite differences of every weighting function, finite values at the
removable singularities."""
from __future__ import annotations

import numpy as np
import pytest

from tests import scenarios as sc
from oracle.oracle_py import OracleEngine, TsOracleEngine


def do_oracle(s, num_sza=1, ss_in_do=False):
    g = s["geometry"]
    return OracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"], 2, num_sza, s["rays"],
                        ss_in_do=ss_in_do)


def ts_oracle(s, solar=True, thermal=False, num_sza=None):
    g = s["geometry"]
    return TsOracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"], s["rays"],
                          num_sza=s["config"]["num_sza"] if num_sza is None else num_sza, solar=solar, thermal=thermal)


def run_ts(s, eng, derivatives=False, **over):
    return eng.run(over.get("extinction", s["extinction"]), over.get("ssa", s["ssa"]), over.get("leg", s["leg"]),
                   over.get("albedo", s["albedo"]), solar_irradiance=s.get("solar_irradiance"), derivatives=derivatives,
                   delta_f=s.get("delta_f"), emission=over.get("emission", s.get("emission")),
                   surface_emission=over.get("surface_emission", s.get("surface_emission")))


def test_delta_m_matches_two_stream_discrete_ordinates():
    # tests/engine/test_twostream.py:100-156 (rtol 2e-8, atol 2e-12 on every output variable)
    d = sc.delta_m_scaled(sc.twostream_delta_m_case(), 2)
    ref = do_oracle(d).run(d["extinction"], d["ssa"], d["leg"], d["albedo"], solar_irradiance=d["solar_irradiance"], derivatives=True,
                           delta_f=d["delta_f"])
    out = run_ts(d, ts_oracle(d), derivatives=True)
    np.testing.assert_allclose(out["radiance"], ref["radiance"], rtol=2e-8, atol=2e-12)
    for name in ("d_extinction", "d_ssa", "d_albedo"):
        np.testing.assert_allclose(out[name], ref[name], rtol=2e-8, atol=2e-12, err_msg=name)


def test_thermal_matches_two_stream_discrete_ordinates():
    # rust/sasktran2-rs/tests/twostream.rs:495-523 (2e-8 relative to max(|expected|, 1)); the discrete-ordinates thermal source is pinned by
    # the DISORT values of tests/test_oracle_golden.py
    alt = np.arange(6) * 5_000.0
    nw = 17
    lev, wav = np.arange(6)[:, None], np.arange(nw)[None, :]
    leg = np.zeros((4, 6, nw))
    leg[0] = 1.0
    leg[1] = 0.32 + 0.002 * lev
    s = dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_371_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR,
                           geotype=sc.GEO_PLANEPARALLEL),
             config=dict(nstr=2, num_sza=1),
             rays=[(sc.GROUND_VIEWING_SOLAR, 0.6, 0.2, 200_000.0, 0.72), (sc.GROUND_VIEWING_SOLAR, 0.6, 1.1, 200_000.0, 0.43)],
             extinction=8.0e-6 + 2.0e-7 * lev + 1.0e-8 * wav, ssa=0.72 + 0.001 * lev + 0 * wav, leg=leg, albedo=np.full(nw, 0.23),
             emission=1.5 + 0.03 * lev + 0.001 * wav, surface_emission=np.full(nw, 1.4))
    ref = do_oracle(s, ss_in_do=True).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], solar_irradiance=np.zeros(nw),
                                          emission=s["emission"], surface_emission=s["surface_emission"])
    out = run_ts(s, ts_oracle(s, solar=False, thermal=True))
    assert np.max(np.abs(out["radiance"] - ref["radiance"]) / np.maximum(np.abs(ref["radiance"]), 1.0)) < 2e-8


def test_spherical_solar_is_close_to_two_stream_interpolated_do():
    # tests/engine/test_twostream.py:76-97 (rtol 2e-2)
    s = sc.twostream_reference_case(spherical=True)
    ref = do_oracle(s, num_sza=3).run(s["extinction"], s["ssa"], s["leg"], s["albedo"], solar_irradiance=s["solar_irradiance"])
    eng = ts_oracle(s)
    assert eng.sizes()["columns"] == 3
    out = run_ts(s, eng)
    np.testing.assert_allclose(out["radiance"], ref["radiance"], rtol=2e-2, atol=1e-12)


def test_combined_sources_add():
    # tests/engine/test_twostream.py:159-175: solar + thermal engine == solar engine + thermal engine
    s = sc.twostream_reference_case(spherical=True, thermal=True)
    both = run_ts(s, ts_oracle(s, solar=True, thermal=True), derivatives=True)
    solar = run_ts(s, ts_oracle(s, solar=True, thermal=False), derivatives=True)
    thermal = run_ts(s, ts_oracle(s, solar=False, thermal=True), derivatives=True)
    assert np.all(np.isfinite(both["radiance"])) and np.all(thermal["radiance"] > 0)
    for name in both:
        np.testing.assert_allclose(both[name], solar[name] + thermal[name], rtol=1e-12, atol=1e-300, err_msg=name)


def finite_difference_check(s, eng, wavelength, los_list, inputs, tol=2e-6):
    """Central differences with the reference's step and tolerance (twostream.rs:185-199: step 1e-5 max(|x|, 1e-3),
    |analytic - numeric| < 2e-6 max(|numeric|, 1))."""
    base = run_ts(s, eng, derivatives=True)
    worst = 0.0
    for name, key, levels in inputs:
        for level in levels:
            arr = np.array(s[key], dtype=float, copy=True)
            idx = (1, level, wavelength) if key == "leg" else ((wavelength,) if arr.ndim == 1 else (level, wavelength))
            x0 = arr[idx]
            # (extinctions are far below the 1e-3 floor of the reference's step: keep the step small against the value)
            step = 1e-5 * (abs(x0) if key == "extinction" else max(abs(x0), 1e-3))
            arr[idx] = x0 + step
            plus = run_ts(s, eng, **{key: arr})["radiance"]
            arr[idx] = x0 - step
            minus = run_ts(s, eng, **{key: arr})["radiance"]
            for los in los_list:
                numeric = (plus[wavelength, los] - minus[wavelength, los]) / (2 * step)
                analytic = base[name][wavelength, los] if arr.ndim == 1 else base[name][wavelength, los, level]
                err = abs(analytic - numeric) / max(abs(numeric), 1.0)
                worst = max(worst, err)
                assert err < tol, (name, level, los, analytic, numeric)
    return worst


@pytest.mark.parametrize("thermal", [False, True])
def test_plane_parallel_jacobians_match_finite_differences(thermal):
    # rust/sasktran2-rs/tests/twostream.rs:164-239
    s = sc.twostream_reference_case(spherical=False, thermal=thermal)
    s["ssa"] = s["ssa"] + 0.002 * np.arange(13)[:, None]
    s["leg"][1] += 0.01 * np.arange(13)[:, None]
    eng = ts_oracle(s, solar=not thermal, thermal=thermal)
    inputs = [("d_extinction", "extinction", range(13)), ("d_ssa", "ssa", range(13)), ("d_b1", "leg", range(13)), ("d_albedo", "albedo", [0])]
    if thermal:
        inputs += [("d_emission", "emission", range(13)), ("d_surface_emission", "surface_emission", [0])]
    finite_difference_check(s, eng, 4, [0, 1], inputs)


@pytest.mark.parametrize("thermal", [False, True])
def test_spherical_jacobians_match_finite_differences(thermal):
    # rust/sasktran2-rs/tests/twostream.rs:525-610
    s = sc.twostream_reference_case(spherical=True, thermal=thermal)
    s["ssa"] = s["ssa"] + 0.002 * np.arange(13)[:, None]
    eng = ts_oracle(s, solar=not thermal, thermal=thermal)
    inputs = [("d_extinction", "extinction", range(13)), ("d_ssa", "ssa", range(13)), ("d_b1", "leg", range(13)), ("d_albedo", "albedo", [0])]
    if thermal:
        inputs += [("d_emission", "emission", range(13)), ("d_surface_emission", "surface_emission", [0])]
    finite_difference_check(s, eng, 2, [0, 1, 2], inputs)


def resonance_cases():
    """rust/sasktran2-rs/tests/twostream.rs:415-492: one layer whose eigenvalue equals the slant secant (solar) or the emission decay rate
    (thermal), so that the particular solutions sit on their removable singularities."""
    mu, b1 = 0.5, 0.4
    alt = np.array([0.0, 1.0])
    leg = np.zeros((2, 2, 1))
    leg[0], leg[1] = 1.0, b1
    w = 0.2
    k = np.sqrt((w * b1 * mu - 1 / mu) * ((w - 1) / mu))
    solar = dict(geometry=dict(cos_sza=1 / k, saa=0.0, earth_radius=6_371_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR,
                               geotype=sc.GEO_PLANEPARALLEL),
                 config=dict(nstr=2, num_sza=1), rays=[(sc.GROUND_VIEWING_SOLAR, 1 / k, 0.0, 2.0, 0.7)], extinction=np.ones((2, 1)),
                 ssa=np.full((2, 1), w), leg=leg, albedo=np.full(1, 0.2), solar_irradiance=np.ones(1))
    w = 0.7
    k = np.sqrt((w * b1 * mu - 1 / mu) * ((w - 1) / mu))
    leg2 = np.zeros((2, 2, 2))
    leg2[0], leg2[1] = 1.0, b1
    em = np.ones((2, 2))
    em[1] = np.exp(np.array([1.0, -1.0]) * k)
    thermal = dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_371_000.0, altitudes=alt, interp_method=sc.INTERP_LINEAR,
                                 geotype=sc.GEO_PLANEPARALLEL),
                   config=dict(nstr=2, num_sza=1), rays=[(sc.GROUND_VIEWING_SOLAR, 0.6, 0.0, 2.0, 0.5)], extinction=np.ones((2, 2)),
                   ssa=np.full((2, 2), w), leg=leg2, albedo=np.full(2, 0.2), emission=em, surface_emission=np.ones(2))
    return solar, thermal


def test_resonances_are_finite_and_continuous():
    solar, thermal = resonance_cases()
    for s, is_thermal in ((solar, False), (thermal, True)):
        eng = ts_oracle(s, solar=not is_thermal, thermal=is_thermal)
        out = run_ts(s, eng, derivatives=True)
        for v in out.values():
            assert np.all(np.isfinite(v))
        # the value on the singularity continues the values next to it
        near = run_ts(s, eng, ssa=s["ssa"] * (1 + 1e-7))
        np.testing.assert_allclose(near["radiance"], out["radiance"], rtol=1e-5)


def test_thin_upper_layers():
    # rust/sasktran2-rs/tests/twostream.rs:617-659: vanishing extinction in the two highest levels stays finite; a tiny one differentiates
    s = sc.twostream_reference_case(spherical=True)
    eng = ts_oracle(s)
    for upper in (0.0, 1e-8):
        k = s["extinction"].copy()
        k[-2:] = upper
        out = run_ts(s, eng, derivatives=True, extinction=k)
        for v in out.values():
            assert np.all(np.isfinite(v))
        if upper > 0:
            kp, km = k.copy(), k.copy()
            kp[-1, 3] += 1e-10
            km[-1, 3] -= 1e-10
            numeric = (run_ts(s, eng, extinction=kp)["radiance"][3, 0] - run_ts(s, eng, extinction=km)["radiance"][3, 0]) / 2e-10
            assert abs(out["d_extinction"][3, 0, -1] - numeric) < 2e-6 * max(abs(numeric), 1.0)
