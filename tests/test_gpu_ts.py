"""GPU parity of the two-stream source (multiple_scatter_source = twostream, emission_source = twostream) through the C ABI against the
two-stream oracle (oracle/ts_oracle.hpp, pinned in tests/test_ts_oracle.py): radiances to 1e-6 and weighting functions to 1e-4 of the
profile maximum are the bar of BASELINE.json's north star; the assertions below are tighter."""
from __future__ import annotations

import numpy as np
import pytest

from tests import scenarios as sc
from tests.test_ts_oracle import resonance_cases, run_ts, ts_oracle

pytestmark = pytest.mark.gpu

NAMES = ("d_extinction", "d_ssa", "d_b1", "d_emission", "d_albedo", "d_surface_emission")
RTOL_I, RTOL_WF = 1e-9, 1e-8


def compare(out, ref, rtol_i=RTOL_I, rtol_wf=RTOL_WF, names=NAMES):
    np.testing.assert_allclose(out["radiance"], ref["radiance"], rtol=rtol_i, atol=1e-300)
    floor = 1e-10 * np.max(np.abs(ref["radiance"]))
    for name in names:
        scale = np.max(np.abs(ref[name]), axis=-1, keepdims=True) if ref[name].ndim == 3 else np.abs(ref[name])
        err = np.abs(out[name] - ref[name]) / np.maximum(scale, floor)
        assert np.max(err) < rtol_wf, (name, float(np.max(err)))


def varied(s):
    s["ssa"] = s["ssa"] + 0.002 * np.arange(s["ssa"].shape[0])[:, None]
    s["leg"][1] += 0.01 * np.arange(s["ssa"].shape[0])[:, None]
    return s


@pytest.mark.parametrize("spherical", [False, True], ids=["plane parallel", "spherical"])
@pytest.mark.parametrize("sources", ["solar", "thermal", "both"])
def test_reference_case(spherical, sources):
    # the atmosphere and lines of sight of tests/engine/test_twostream.py:34-73
    from tests.gpu_harness import run_product_ts
    s = varied(sc.twostream_reference_case(spherical=spherical, thermal=sources != "solar"))
    solar, thermal = sources != "thermal", sources != "solar"
    ref = run_ts(s, ts_oracle(s, solar=solar, thermal=thermal), derivatives=True)
    out = run_product_ts(s, solar=solar, thermal=thermal)
    compare(out, ref)
    fwd = run_product_ts(s, solar=solar, thermal=thermal, derivatives=False)
    np.testing.assert_allclose(fwd["radiance"], ref["radiance"], rtol=RTOL_I)


def test_delta_m_pin():
    # tests/engine/test_twostream.py:100-156: delta-M scaled plane-parallel case; the product scales the optics itself
    # (sk_atmosphere_apply_delta_m_scaling) and maps the weighting functions back to the unscaled inputs
    from tests.gpu_harness import run_product_ts
    s = sc.twostream_delta_m_case()
    d = sc.delta_m_scaled(s, 2)
    ref = run_ts(d, ts_oracle(d), derivatives=True)
    out = run_product_ts(s, delta_m=True)
    np.testing.assert_allclose(out["radiance"], ref["radiance"], rtol=RTOL_I)
    u = d["unscale"]
    want_k = ref["d_extinction"] * u["dk_dk"].T[:, None, :]
    want_w = ref["d_ssa"] * u["dw_dw"].T[:, None, :] + ref["d_extinction"] * u["dk_dw"].T[:, None, :]
    for got, want in ((out["d_extinction"], want_k), (out["d_ssa"], want_w)):
        assert np.max(np.abs(got - want) / np.max(np.abs(want), axis=-1, keepdims=True)) < RTOL_WF
    np.testing.assert_allclose(out["d_albedo"], ref["d_albedo"], rtol=RTOL_WF)


def test_delta_m_phase_derivative_matches_finite_differences():
    # the scattering-group derivative through d_legendre(1) with the delta-M correction -3 d_f / (1 - f)^2 (cpp_twostream_source.cpp:3583-3604):
    # with two streams the truncation order is 2, so a change of the first moment leaves f alone and the oracle's d_b1 / (1 - f) applies
    from tests.gpu_harness import run_product_ts
    s = sc.twostream_delta_m_case(3)
    d = sc.delta_m_scaled(s, 2)
    ref = run_ts(d, ts_oracle(d), derivatives=True)
    out = run_product_ts(s, delta_m=True)
    want = ref["d_b1"] / (1 - d["delta_f"]).T[:, None, :]
    assert np.max(np.abs(out["d_b1"] - want) / np.max(np.abs(want), axis=-1, keepdims=True)) < RTOL_WF


def test_limb_scan_50_layers():
    # the limb geometry of BASELINE configs[1] (25 tangent heights, 51 levels) with three solar-zenith columns
    from tests.gpu_harness import run_product_ts
    c2 = sc.limb_c2(num_wavelengths=70, nlyr=50, nstr=16)
    s = dict(c2, leg=c2["leg"][:2].copy(), config=dict(nstr=2, num_sza=3), solar_irradiance=np.linspace(0.9, 1.2, 70))
    ref = run_ts(s, ts_oracle(s), derivatives=True)
    out = run_product_ts(s)
    compare(out, ref, names=NAMES[:3] + ("d_albedo",))
    np.testing.assert_allclose(out["los_optical_depth"], _los_od(s), rtol=1e-12)


def _los_od(s):
    from oracle.oracle_py import OracleEngine
    g = s["geometry"]
    do = OracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"], 2, 1, s["rays"], ss_in_do=False)
    return do.run(s["extinction"], s["ssa"], s["leg"], s["albedo"])["los_optical_depth"]


def twostream_benchmark(num_wavelengths):
    """The reference's two-stream benchmark (cpp/lib/tests/twostream/test_twostream.cpp:6-66): plane parallel, 20 layers of 1 m, extinction
    0.01, conservative scattering, a1 = [1, 0, 0.5], albedo 0.5, one nadir-like line of sight; here with a spectral ramp on the extinction."""
    alt = np.arange(21.0)
    nw = num_wavelengths
    leg = np.zeros((16, 21, nw))
    leg[0], leg[2] = 1.0, 0.5
    return dict(geometry=dict(cos_sza=0.6, saa=0.0, earth_radius=6_371_000.0, altitudes=alt, interp_method=sc.INTERP_LOWER, geotype=sc.GEO_PLANEPARALLEL),
                config=dict(nstr=2, num_sza=1), rays=[(sc.GROUND_VIEWING_SOLAR, 0.6, np.pi / 3, 200_000.0, 0.6)],
                extinction=0.01 * np.linspace(0.5, 20.0, nw)[None, :] * np.ones((21, 1)), ssa=np.ones((21, nw)), leg=leg, albedo=np.full(nw, 0.5))


def test_reference_benchmark_configuration():
    from tests.gpu_harness import run_product_ts
    s = twostream_benchmark(300)
    ref = run_ts(s, ts_oracle(s), derivatives=True)
    out = run_product_ts(s)
    compare(out, ref, names=("d_extinction", "d_b1", "d_albedo"))
    assert np.max(np.abs(out["d_ssa"])) <= 1e-12 * np.max(np.abs(out["d_extinction"]))   # conservative scattering: capped, no ssa derivative


@pytest.mark.parametrize("sources", ["solar", "thermal", "both"])
def test_single_line_of_sight_fused_passes(sources, monkeypatch):
    # plane-parallel geometry with one line of sight takes two fused passes instead of three kernels: the oracle's numbers either way
    from tests.gpu_harness import run_product_ts
    s = varied(sc.twostream_reference_case(spherical=False, thermal=sources != "solar"))
    s["rays"] = s["rays"][:1]
    solar, thermal = sources != "thermal", sources != "solar"
    ref = run_ts(s, ts_oracle(s, solar=solar, thermal=thermal), derivatives=True)
    fused = run_product_ts(s, solar=solar, thermal=thermal)
    compare(fused, ref)
    fwd = run_product_ts(s, solar=solar, thermal=thermal, derivatives=False)
    np.testing.assert_allclose(fwd["radiance"], ref["radiance"], rtol=RTOL_I)
    monkeypatch.setenv("SASKTRAN2_AMD_TS_UNFUSED", "1")
    compare(run_product_ts(s, solar=solar, thermal=thermal), ref)
    monkeypatch.delenv("SASKTRAN2_AMD_TS_UNFUSED")
    # several launches with a ragged last one: bit-identical to the single launch
    t = varied(sc.twostream_reference_case(spherical=False, thermal=sources != "solar", num_wavelengths=150))
    t["rays"] = t["rays"][:1]
    one = run_product_ts(t, solar=solar, thermal=thermal)
    monkeypatch.setenv("SASKTRAN2_AMD_TS_CHUNK", "64")
    blocks = run_product_ts(t, solar=solar, thermal=thermal)
    for k in one:
        assert np.array_equal(one[k], blocks[k]), k


def test_resonances_on_device():
    from tests.gpu_harness import run_product_ts
    solar, thermal = resonance_cases()
    compare(run_product_ts(solar), run_ts(solar, ts_oracle(solar), derivatives=True), rtol_i=1e-8, rtol_wf=1e-6)
    compare(run_product_ts(thermal, solar=False, thermal=True), run_ts(thermal, ts_oracle(thermal, solar=False, thermal=True), derivatives=True),
            rtol_i=1e-8, rtol_wf=1e-6)


def test_blocks_of_the_spectrum_are_independent(monkeypatch):
    from tests.gpu_harness import run_product_ts
    s = varied(sc.twostream_reference_case(spherical=True, thermal=True, num_wavelengths=200))
    full = run_product_ts(s, solar=True, thermal=True)
    monkeypatch.setenv("SASKTRAN2_AMD_TS_CHUNK", "64")
    blocks = run_product_ts(s, solar=True, thermal=True)
    for k in full:
        assert np.array_equal(full[k], blocks[k]), k


def test_exact_single_scatter_on_top():
    # single_scatter_source = exact + multiple_scatter_source = twostream: the two sources add (radiances and weighting functions)
    from tests.gpu_harness import build_so_engine, run_product_so, run_product_ts
    s = varied(sc.twostream_reference_case(spherical=True))
    both = run_product_ts(s, single_scatter="exact", b1_mapping=False)   # (without the group mapping: the sum below adds the two engines' own extinction / albedo results)
    ms = run_product_ts(s)
    so_s = dict(s, config=dict(nstokes=1, num_streams=2, num_singlescatter_moments=2, num_sza=3, single_scatter="exact", multiple_scatter="none",
                               num_iterations=0, num_incoming=26, num_outgoing=26))
    ss = run_product_so(so_s, solar_irradiance=s["solar_irradiance"], derivatives=True)
    np.testing.assert_allclose(both["radiance"], ms["radiance"] + ss["radiance"][..., 0], rtol=1e-12)
    np.testing.assert_allclose(both["d_extinction"], ms["d_extinction"] + ss["d_extinction"][:, :, 0, :], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(both["d_albedo"], ms["d_albedo"] + ss["d_albedo"][..., 0], rtol=1e-10, atol=1e-18)


def test_device_resident_entry_point():
    # torch and the product library each bring a HIP runtime; torch has to initialise first (as in bench.py), so this runs in a fresh process
    import subprocess
    import sys
    from pathlib import Path
    code = "import torch; torch.cuda.init(); from tests.test_gpu_ts import device_resident_check; device_resident_check()"
    r = subprocess.run([sys.executable, "-c", code], cwd=Path(__file__).resolve().parents[1], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def device_resident_check():
    import torch
    from sasktran2_amd.device import TsDeviceSolver
    from tests.gpu_harness import build_ts_engine, run_product_ts
    s = varied(sc.twostream_reference_case(spherical=True, thermal=True, num_wavelengths=130))
    engine, config, geometry = build_ts_engine(s, solar=True, thermal=True)
    host = run_product_ts(s, engine=(engine, config, geometry), solar=True, thermal=True)
    solver = TsDeviceSolver(engine)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")  # noqa: E731
    out = solver.allocate_outputs(130)
    solver.solve(t(s["extinction"].T), t(s["ssa"].T), t(np.transpose(s["leg"], (2, 1, 0))), t(s["albedo"]), out, irradiance=t(s["solar_irradiance"]),
                 emission=t(s["emission"].T), surface_emission=t(s["surface_emission"]))
    torch.cuda.synchronize()
    assert np.array_equal(out["radiance"].cpu().numpy(), host["radiance"])
    assert np.array_equal(np.transpose(out["wf_extinction"].cpu().numpy(), (1, 2, 0)), host["d_extinction"])
    assert np.array_equal(np.transpose(out["wf_emission"].cpu().numpy(), (1, 2, 0)), host["d_emission"])
    assert np.array_equal(out["wf_surface_emission"].cpu().numpy(), host["d_surface_emission"])
    assert np.array_equal(out["los_optical_depth"].cpu().numpy(), host["los_optical_depth"])


def test_refusals():
    import sasktran2_amd as sk
    from tests.gpu_harness import build_ts_engine
    s = sc.twostream_reference_case(spherical=True)
    g = s["geometry"]
    config = sk.Config()
    config.num_stokes = 3
    config.multiple_scatter_source = sk.MultipleScatterSource.TwoStream
    geometry = sk.Geometry1D(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"])
    viewing = sk.ViewingGeometry()
    viewing.add_ray(sk.TangentAltitudeSolar(15_000.0, 0.4, 200_000.0, 0.6))
    with pytest.raises(Exception, match="NSTOKES = 1"):
        sk.Engine(config, geometry, viewing)
    # downward-looking plane-parallel rays only (cpp_twostream_source.cpp:3419-3424)
    pp = sc.twostream_reference_case(spherical=False)
    pp["rays"] = [(sc.VIEWING_UP_SOLAR, 0.6, 0.0, 0.0, 0.5)]
    with pytest.raises(Exception):
        build_ts_engine(pp)


def test_jvp_and_vjp_with_the_two_stream_source():
    # sk_engine_calculate_jvp / _vjp (contractions of the Jacobian) with multiple_scatter_source = twostream
    import sasktran2_amd as sk
    from tests.gpu_harness import build_ts_engine
    s = varied(sc.twostream_reference_case(spherical=True, num_wavelengths=3))
    engine, config, geometry = build_ts_engine(s)
    atmosphere = sk.Atmosphere(geometry, config, numwavel=3, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.leg_coeff.a1[:] = s["leg"]
    atmosphere.surface.albedo[:] = s["albedo"]
    atmosphere.storage.solar_irradiance[:] = s["solar_irradiance"]
    full = engine.calculate_radiance(atmosphere, as_dataset=False)
    rng = np.random.default_rng(11)
    ngrid = s["extinction"].shape[0]
    t_k, t_w, t_a = rng.standard_normal(ngrid) * 1e-6, rng.standard_normal(ngrid) * 1e-2, rng.standard_normal(3) * 0.1
    rad, jvp = engine.calculate_jvp(atmosphere, {"wf_extinction": t_k, "wf_ssa": t_w}, {"wf_albedo": t_a})
    want = np.einsum("qwls,q->wls", full["wf_extinction"], t_k) + np.einsum("qwls,q->wls", full["wf_ssa"], t_w) + full["wf_albedo"] * t_a[:, None, None]
    np.testing.assert_array_equal(rad, full["radiance"])
    np.testing.assert_allclose(jvp, want, rtol=1e-12, atol=1e-300)
    cot = rng.standard_normal(full["radiance"].shape)
    rad2, grads = engine.calculate_vjp(atmosphere, cot)
    np.testing.assert_array_equal(rad2, full["radiance"])
    np.testing.assert_allclose(grads["wf_extinction"], np.einsum("qwls,wls->q", full["wf_extinction"], cot), rtol=1e-12)
    np.testing.assert_allclose(grads["wf_albedo"], np.einsum("wls,wls->w", full["wf_albedo"], cot), rtol=1e-12)
    lhs = float(np.sum(cot * jvp))
    rhs = float(grads["wf_extinction"] @ t_k + grads["wf_ssa"] @ t_w + grads["wf_albedo"] @ t_a)
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))


def test_pseudo_spherical_matches_two_stream_discrete_ordinates():
    # pseudo-spherical geometry: plane-parallel diffuse field, beam attenuated along its true path (ray-traced Chapman factors): the same
    # equality with two-stream discrete ordinates as in plane-parallel geometry (oracle and device)
    from oracle.oracle_py import OracleEngine
    from tests.gpu_harness import run_product_ts
    # (phase moments constant with altitude as in the reference's comparison: the discrete-ordinates extinction / ssa weighting functions hold the
    # layer-averaged phase function fixed -- sktran_do_layerarray.cpp:730-756 -- while the two-stream source differentiates that average too)
    s = sc.twostream_reference_case(spherical=False)
    s["ssa"] = s["ssa"] + 0.002 * np.arange(13)[:, None]
    s["geometry"] = dict(s["geometry"], geotype=sc.GEO_PSEUDOSPHERICAL)
    g = s["geometry"]
    do = OracleEngine(g["cos_sza"], g["saa"], g["earth_radius"], g["altitudes"], g["interp_method"], g["geotype"], 2, 1, s["rays"], ss_in_do=False)
    ref_do = do.run(s["extinction"], s["ssa"], s["leg"], s["albedo"], solar_irradiance=s["solar_irradiance"], derivatives=True)
    ref = run_ts(s, ts_oracle(s), derivatives=True)
    np.testing.assert_allclose(ref["radiance"], ref_do["radiance"], rtol=2e-8)
    for k in ("d_extinction", "d_ssa"):
        np.testing.assert_allclose(ref[k], ref_do[k], rtol=2e-8, atol=2e-12, err_msg=k)
    compare(run_product_ts(s), ref)


def test_device_entry_point_refuses_bad_arguments():
    import ctypes
    from sasktran2_amd._lib_loader import last_error, lib
    from tests.gpu_harness import build_ts_engine
    s = sc.twostream_reference_case(spherical=False, thermal=True)
    engine, _, _ = build_ts_engine(s, solar=True, thermal=True)
    n = [None] * 17   # 16 device arrays and the stream
    rc = lib().sk_amd_ts_engine_solve_device(engine._h, 9, 2, *n, 0)
    assert rc == -3 and "null device pointer" in last_error()
    rc = lib().sk_amd_ts_engine_solve_device(engine._h, 0, 2, *n, 0)
    assert rc == -3 and "nwavel" in last_error()
    # a discrete-ordinates engine has no two-stream source
    from tests.gpu_harness import build_engine
    do_engine, _, _ = build_engine(sc.regression_1d())
    rc = lib().sk_amd_ts_engine_solve_device(do_engine._h, 9, 2, *n, 0)
    assert rc == -3 and "no two-stream source" in last_error()


def test_full_size_twostream_properties():
    """The benchmark workload at its full size (220 000 wavelengths, bench_ts.scenario) through properties that do not need the oracle at
    that size: a sub-range of the spectrum solved alone is bit-identical to the same wavelengths inside the full run (blocks are independent
    whatever the launch shape), radiance and weighting functions are exactly linear in the solar irradiance, every output is finite, and an
    evenly spaced sample agrees with the oracle."""
    import subprocess
    import sys
    from pathlib import Path
    code = "import torch; torch.cuda.init(); from tests.test_gpu_ts import full_size_check; full_size_check()"
    r = subprocess.run([sys.executable, "-c", code], cwd=Path(__file__).resolve().parents[1], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def full_size_check():
    import torch
    from bench_ts import scenario
    from sasktran2_amd.device import TsDeviceSolver
    from tests.gpu_harness import build_ts_engine
    NW = 220_000
    s = scenario(NW)
    engine, _, _ = build_ts_engine(s)
    solver = TsDeviceSolver(engine)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    ext, ssa = to(s["extinction"].T), to(s["ssa"].T)
    leg, alb = to(np.transpose(s["leg"], (2, 1, 0))), to(s["albedo"])
    irr = torch.ones(NW, dtype=torch.float64, device="cuda")
    full = solver.solve(ext, ssa, leg, alb, solver.allocate_outputs(NW), irradiance=irr)
    torch.cuda.synchronize()
    for v in full.values():
        assert bool(torch.isfinite(v).all())
    lo, hi = 70_001, 83_456   # not aligned to a wavefront
    part = solver.solve(ext[lo:hi].contiguous(), ssa[lo:hi].contiguous(), leg[lo:hi].contiguous(), alb[lo:hi].contiguous(),
                        solver.allocate_outputs(hi - lo), irradiance=irr[lo:hi].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(part["radiance"], full["radiance"][lo:hi])
    for k in ("wf_extinction", "wf_ssa", "wf_b1"):
        assert torch.equal(part[k], full[k][:, lo:hi]), k
    assert torch.equal(part["wf_albedo"], full["wf_albedo"][lo:hi])
    double = solver.solve(ext, ssa, leg, alb, solver.allocate_outputs(NW), irradiance=2.0 * irr)
    torch.cuda.synchronize()
    assert torch.equal(double["radiance"], 2.0 * full["radiance"]) and torch.equal(double["wf_extinction"], 2.0 * full["wf_extinction"])
    idx = np.linspace(0, NW - 1, 257).astype(int)
    ref = ts_oracle(s).run(s["extinction"][:, idx], s["ssa"][:, idx], s["leg"][:, :, idx], s["albedo"][idx], derivatives=True)
    np.testing.assert_allclose(full["radiance"].cpu().numpy()[idx], ref["radiance"], rtol=1e-9)
    got = np.transpose(full["wf_extinction"].cpu().numpy()[:, idx], (1, 2, 0))
    assert np.max(np.abs(got - ref["d_extinction"]) / np.max(np.abs(ref["d_extinction"]), axis=-1, keepdims=True)) < 1e-8
