"""The two-stream HIP kernel bodies (sasktran2_amd/csrc/hip/ts_core.hpp) executed on the CPU against the two-stream oracle
(oracle/ts_oracle.hpp): radiances and every weighting function, plane parallel and spherical, solar and thermal."""
from __future__ import annotations

import numpy as np
import pytest

from tests import scenarios as sc
from tests.emulation import run_emulated_twostream
from tests.test_ts_oracle import resonance_cases, run_ts, ts_oracle

NAMES = ("d_extinction", "d_ssa", "d_b1", "d_emission", "d_albedo", "d_surface_emission")


def compare(s, solar=True, thermal=False, rtol=1e-9, dual_adjoint=False):
    ref = run_ts(s, ts_oracle(s, solar=solar, thermal=thermal), derivatives=True)
    out = run_emulated_twostream(s, derivatives=True, solar=solar, thermal=thermal, dual_adjoint=dual_adjoint)
    fwd = run_emulated_twostream(s, derivatives=False, solar=solar, thermal=thermal)
    np.testing.assert_allclose(out["radiance"], ref["radiance"], rtol=rtol, atol=1e-300)
    np.testing.assert_allclose(fwd["radiance"], ref["radiance"], rtol=rtol, atol=1e-300)
    for name in NAMES:
        scale = np.max(np.abs(ref[name]), axis=-1, keepdims=True) if ref[name].ndim == 3 else np.abs(ref[name])
        # (a profile that is rounding noise -- the ssa derivative under the conservative-scattering cap -- is compared to the radiance scale)
        err = np.abs(out[name] - ref[name]) / np.maximum(scale, 1e-10 * np.max(np.abs(ref["radiance"])))
        assert np.max(err) < rtol, (name, float(np.max(err)))
    return out


def test_plane_parallel_delta_m():
    compare(sc.delta_m_scaled(sc.twostream_delta_m_case(), 2))


@pytest.mark.parametrize("spherical", [False, True])
@pytest.mark.parametrize("sources", ["solar", "thermal", "both"])
@pytest.mark.parametrize("dual_adjoint", [False, True], ids=["reverse sweep (the product)", "dual numbers (cross-check)"])
def test_reference_case(spherical, sources, dual_adjoint):
    s = sc.twostream_reference_case(spherical=spherical, thermal=sources != "solar")
    s["ssa"] = s["ssa"] + 0.002 * np.arange(13)[:, None]
    s["leg"][1] += 0.01 * np.arange(13)[:, None]
    compare(s, solar=sources != "thermal", thermal=sources != "solar", dual_adjoint=dual_adjoint)


def test_single_column_spherical():
    s = sc.twostream_reference_case(spherical=True)
    s["config"] = dict(s["config"], num_sza=1)
    compare(s)


@pytest.mark.parametrize("dual_adjoint", [False, True], ids=["reverse sweep (the product)", "dual numbers (cross-check)"])
def test_resonances(dual_adjoint):
    solar, thermal = resonance_cases()
    compare(solar, rtol=1e-7, dual_adjoint=dual_adjoint)
    compare(thermal, solar=False, thermal=True, rtol=1e-7, dual_adjoint=dual_adjoint)


def test_view_resonances():
    # a viewing cosine equal to 1 / eigenvalue of either azimuth order, in thin and in thick layers (the quotients of the layer-integrated source)
    s = sc.twostream_reference_case(spherical=False)
    w, b1, mu = 0.85, 0.2, 0.5
    k0 = np.sqrt((w * b1 * mu - 1 / mu) * ((w - 1) / mu))
    k1 = np.sqrt((-1 / mu) * ((w * b1 * (1 - mu * mu) - 2) / (2 * mu)))
    s["rays"] = [(sc.GROUND_VIEWING_SOLAR, 0.6, 0.3, 200_000.0, 1 / k1), (sc.GROUND_VIEWING_SOLAR, 0.6, -0.4, 200_000.0, min(1 / k0, 0.999))]
    for scale in (1.0, 30.0):
        t = dict(s, extinction=s["extinction"] * scale)
        compare(t, rtol=1e-7)
        compare(t, rtol=1e-7, dual_adjoint=True)


def test_conservative_scattering_and_empty_upper_levels():
    s = sc.twostream_reference_case(spherical=True)
    s["ssa"] = np.ones_like(s["ssa"])       # capped at 1 - 1e-9 inside, no ssa derivative
    k = s["extinction"].copy()
    k[-2:] = 0.0
    s["extinction"] = k
    out = compare(s, rtol=1e-7)
    assert np.max(np.abs(out["d_ssa"])) < 1e-15


@pytest.mark.parametrize("sources", ["solar", "thermal", "both"])
def test_single_line_of_sight_fused_passes(sources):
    # plane-parallel geometry with one line of sight takes the two fused passes (LU sweep + dF/dc rows down, back substitution + sources +
    # adjoint up): same numbers as the three-kernel path and as the oracle
    s = sc.twostream_reference_case(spherical=False, thermal=sources != "solar")
    s["rays"] = s["rays"][:1]
    s["ssa"] = s["ssa"] + 0.002 * np.arange(13)[:, None]
    s["leg"][1] += 0.01 * np.arange(13)[:, None]
    solar, thermal = sources != "thermal", sources != "solar"
    fused = compare(s, solar=solar, thermal=thermal)
    plain = run_emulated_twostream(s, derivatives=True, solar=solar, thermal=thermal, allow_fused=False)
    for name in fused:
        np.testing.assert_allclose(fused[name], plain[name], rtol=1e-9, atol=1e-12 * np.max(np.abs(plain[name])), err_msg=name)
    fwd = run_emulated_twostream(s, derivatives=False, solar=solar, thermal=thermal)
    np.testing.assert_allclose(fwd["radiance"], plain["radiance"], rtol=1e-12)
