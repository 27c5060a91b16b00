"""CPU-only checks of the HIP kernel bodies: tests/emulate_kernels.cpp compiles sasktran2_amd/csrc/hip/disco_core.hpp
with g++ and runs one loop iteration per GPU thread in the driver's launch order; results must match the oracle."""
import numpy as np
import pytest

from tests import scenarios as sc
from tests.emulation import run_emulated
from tests.test_oracle_golden import run_oracle


def check(s, rtol_i=1e-9, rtol_wf=1e-8):
    o, e = run_oracle(s, derivatives=True), run_emulated(s)
    assert np.all(e["status"] == 0)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=rtol_i)
    for k in ("d_extinction", "d_ssa"):
        scale = np.abs(o[k]).max(axis=2, keepdims=True)
        assert np.max(np.abs(e[k] - o[k]) / scale) < rtol_wf, k
    np.testing.assert_allclose(e["d_albedo"], o["d_albedo"], rtol=rtol_wf, atol=1e-300)


def test_regression_scenario():
    s = sc.regression_1d()
    check(s)
    np.testing.assert_allclose(run_emulated(s)["radiance"], s["expected_radiance"], rtol=5e-7, atol=2e-13)


def test_c2_shape():
    check(sc.limb_c2(num_wavelengths=2))


@pytest.mark.parametrize("nstr", [2, 4, 16, 32])
def test_stream_counts(nstr):
    s = sc.regression_1d(1)
    s["config"]["nstr"] = nstr
    if nstr > s["leg"].shape[0]:
        s["leg"] = np.concatenate([s["leg"], np.zeros((nstr - 16, 25, 1))])
    check(s)


@pytest.mark.parametrize("scale", [50.0, 1e-3])
def test_optical_thickness_extremes(scale):
    s = sc.regression_1d(1)
    s["extinction"] = s["extinction"] * scale
    check(s, rtol_wf=1e-7)


def test_conservative_scattering_radiance():
    s = sc.regression_1d(1)
    s["ssa"][:] = 1.0
    o, e = run_oracle(s), run_emulated(s)
    np.testing.assert_allclose(e["radiance"], o["radiance"], rtol=2e-6)
