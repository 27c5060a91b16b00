"""GPU parity of the exact single-scatter + successive-orders path (Python mirror -> C ABI -> HIP kernels on cuda:0) against the
reference's pinned Stokes vectors and against the CPU oracle (oracle/so_oracle.hpp).  Tolerance: BASELINE.json's 1e-6 relative on
Stokes I; Q and U to 1e-6 of the intensity."""
import numpy as np
import pytest

from tests import scenarios as sc
from tests.test_so_oracle import run_so_oracle

pytestmark = pytest.mark.gpu


def run_product_so(*a, **k):
    from tests.gpu_harness import run_product_so as rp
    return rp(*a, **k)


def assert_so_parity(got, ref, rtol=1e-6):
    np.testing.assert_allclose(got["los_optical_depth"], ref["los_optical_depth"], rtol=1e-12)
    np.testing.assert_allclose(got["radiance"][..., 0], ref["radiance"][..., 0], rtol=rtol, atol=1e-300)
    if got["radiance"].shape[-1] == 3:
        scale = np.abs(ref["radiance"][..., :1])
        assert np.max(np.abs(got["radiance"][..., 1:] - ref["radiance"][..., 1:]) / scale) < rtol


def test_reference_regression_pin_on_the_device():
    """tests/engine/test_1d_solver_regression.py:140-240 (NSTOKES = 3, exact single scatter + legacy successive orders) through the C ABI."""
    s = sc.regression_so()
    got = run_product_so(s)
    np.testing.assert_allclose(got["radiance"], s["expected_radiance"], rtol=s["rtol"], atol=s["atol"])
    np.testing.assert_allclose(got["los_optical_depth"], s["expected_od"], rtol=5e-7, atol=1e-13)
    assert_so_parity(got, run_so_oracle(s))


@pytest.mark.parametrize("nstokes", [1, 3])
@pytest.mark.parametrize("sources", [("exact", "none"), ("none", "successive_orders_legacy"), ("exact", "successive_orders_legacy")],
                         ids=["single scatter only", "multiple scatter only", "both"])
def test_source_combinations_match_oracle(nstokes, sources):
    s = sc.regression_so(nstokes=nstokes, num_wavelengths=5)
    o = dict(single_scatter=sources[0], multiple_scatter=sources[1])
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


@pytest.mark.parametrize("num_sza,ndir,niter", [(1, 26, 1), (1, 50, 4), (3, 14, 2), (1, 110, 6)])
def test_discretisations_match_oracle(num_sza, ndir, niter):
    s = sc.regression_so(nstokes=1, num_wavelengths=3)
    o = dict(num_sza=num_sza, num_incoming=ndir, num_outgoing=ndir, num_iterations=niter)
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


def test_different_incoming_and_outgoing_spheres_polarised():
    s = sc.regression_so(nstokes=3, num_wavelengths=2)
    o = dict(num_incoming=38, num_outgoing=26, num_iterations=2, num_sza=1)
    assert_so_parity(run_product_so(s, **o), run_so_oracle(s, **o))


def test_c5_shape_limb_scene_matches_oracle_and_is_block_invariant(monkeypatch):
    """BASELINE configs[4] geometry at a reduced wavelength count: 50 layers, 25 limb rays, 110 / 110 directions, 10 orders, aerosol
    phase functions with all 16 moments; a spectrum that does not fill the last wavefront; blocks of wavelengths give identical bits."""
    s = sc.limb_c5(num_wavelengths=200)
    ref = run_so_oracle(s, nthreads=8)
    got = run_product_so(s)
    assert_so_parity(got, ref)
    monkeypatch.setenv("SASKTRAN2_AMD_SO_CHUNK", "128")
    blocks = run_product_so(s)
    assert np.array_equal(blocks["radiance"], got["radiance"])


def test_solar_irradiance_scales_the_radiance_and_derivatives_are_refused():
    import sasktran2_amd as sk
    from tests.gpu_harness import build_so_engine
    s = sc.regression_so(nstokes=1)
    eng = build_so_engine(s)
    one = run_product_so(s, engine=eng)["radiance"]
    two = run_product_so(s, engine=eng, solar_irradiance=np.array([2.0, 3.0]))["radiance"]
    np.testing.assert_allclose(two, one * np.array([2.0, 3.0])[:, None, None], rtol=1e-13)
    atmosphere = sk.Atmosphere(eng[2], eng[1], numwavel=2, calculate_derivatives=True)
    atmosphere.storage.total_extinction[:] = s["extinction"]
    atmosphere.storage.ssa[:] = s["ssa"]
    atmosphere.storage.leg_coeff[:] = s["leg"]
    with pytest.raises(RuntimeError, match="weighting functions are not available"):
        eng[0].calculate_radiance(atmosphere, as_dataset=False)


def test_exact_single_scatter_on_top_of_discrete_ordinates():
    """single_scatter_source = exact with multiple_scatter_source = discrete_ordinates: in spherical geometry the DO source gives the
    multiple-scatter term, the exact source the single scatter (engine.cpp:217-230, 280-369)."""
    from tests.test_oracle_golden import run_oracle
    s = sc.regression_so(nstokes=1)
    got = run_product_so(s, multiple_scatter="discrete_ordinates")["radiance"][..., 0]
    ss = run_so_oracle(s, multiple_scatter="none")["radiance"][..., 0]
    d = sc.regression_1d(2)
    do = run_oracle(d, derivatives=False)["radiance"]
    np.testing.assert_allclose(got, ss + do, rtol=1e-6)
