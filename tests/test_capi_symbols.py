"""The C-ABI library loads without a GPU and exports every symbol include/sasktran2_amd.h declares; compute entry
points fail loudly (no CPU fallback) when no HIP device is present."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from sasktran2_amd import _lib_loader
    return _lib_loader


def test_header_symbols_are_exported(built):
    header = (ROOT / "include" / "sasktran2_amd.h").read_text()
    declared = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) > 60
    L = built.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert {n for n, _, _ in built.SIGNATURES} <= declared


def test_every_reference_capi_symbol_is_exported(built):
    """Link-time drop-in: every function the reference's C API headers declare (tests/golden/reference_capi_symbols.json,
    205 names) is declared in include/sasktran2_amd.h and exported by the library."""
    import json
    names = json.loads((ROOT / "tests" / "golden" / "reference_capi_symbols.json").read_text())["symbols"]
    assert len(names) == 205
    header = (ROOT / "include" / "sasktran2_amd.h").read_text()
    declared = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", header))
    L = built.lib()
    assert not [n for n in names if n not in declared]
    assert not [n for n in names if not hasattr(L, n)]


def test_stored_settings_and_unsupported_entry_points(built):
    import ctypes
    L = built.lib()
    vp, ip, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    L.sk_config_create.restype = vp
    c = L.sk_config_create()
    # defaults of the reference's Config (cpp/lib/config/config.cpp:5-32) and round trips
    for key, default in (("num_hr_incoming", 110), ("num_do_forced_azimuth", -1), ("emission_source", 1), ("wf_precision", 0)):
        get, set_ = getattr(L, f"sk_config_get_{key}"), getattr(L, f"sk_config_set_{key}")
        get.argtypes, set_.argtypes = [vp, ip], [vp, ctypes.c_int]
        v = ctypes.c_int(-99)
        assert get(c, ctypes.byref(v)) == 0 and v.value == default
        assert set_(c, 7) == 0 and get(c, ctypes.byref(v)) == 0 and v.value == 7
        assert set_(c, default) == 0
    L.sk_config_get_successive_orders_damping.argtypes = [vp, dp]
    d = ctypes.c_double()
    assert L.sk_config_get_successive_orders_damping(c, ctypes.byref(d)) == 0 and d.value == 1.0
    L.sk_config_get_num_flux_types.argtypes = [vp, ip]
    n = ctypes.c_int()
    assert L.sk_config_get_num_flux_types(c, ctypes.byref(n)) == 0 and n.value == 2
    # engines that are not part of this build
    L.sk_geometry2d_create.restype = vp
    L.sk_geometry2d_create.argtypes = [ctypes.c_double] * 3 + [dp, ctypes.c_int, dp, ctypes.c_int, ctypes.c_int]
    assert not L.sk_geometry2d_create(0.5, 0.0, 6.37e6, None, 0, None, 0, 1)
    L.sk_engine_calculate_jvp.argtypes = [vp, vp, vp]
    assert L.sk_engine_calculate_jvp(None, None, None) == -1
    L.sk_viewing_geometry_create.restype = vp
    vg = L.sk_viewing_geometry_create()
    L.sk_viewing_geometry_add_flux_observer_solar.argtypes = [vp, ctypes.c_double, ctypes.c_double]
    # flux observers are stored (the plane-parallel discrete-ordinates engine serves them: tests/test_flux.py)
    assert L.sk_viewing_geometry_add_flux_observer_solar(vg, 0.5, 1000.0) == 0
    L.sk_viewing_geometry_num_flux_observers.argtypes = [vp, ip]
    assert L.sk_viewing_geometry_num_flux_observers(vg, ctypes.byref(n)) == 0 and n.value == 1
    L.sk_viewing_geometry_destroy.argtypes = [vp]
    L.sk_viewing_geometry_destroy(vg)
    L.sk_config_destroy.argtypes = [vp]
    L.sk_config_destroy(c)


def test_lapack_dgesv_entry_point(built):
    # sk_lapack.h:18 -- column-major dgesv with partial pivoting
    import ctypes
    import numpy as np
    L = built.lib()
    i64, dp = ctypes.c_int64, ctypes.POINTER(ctypes.c_double)
    L.sk_lapack_dgesv.restype = i64
    L.sk_lapack_dgesv.argtypes = [i64, i64, dp, i64, ctypes.POINTER(i64), dp, i64]
    rng = np.random.default_rng(3)
    n, nrhs = 7, 3
    a = rng.standard_normal((n, n))
    a[0, 0] = 0.0  # forces a row interchange
    b = rng.standard_normal((n, nrhs))
    af, bf = np.asfortranarray(a.copy()), np.asfortranarray(b.copy())
    ipiv = (i64 * n)()
    info = L.sk_lapack_dgesv(n, nrhs, af.ctypes.data_as(dp), n, ipiv, bf.ctypes.data_as(dp), n)
    assert info == 0
    np.testing.assert_allclose(a @ bf, b, atol=1e-12)
    sing = np.asfortranarray(np.zeros((2, 2)))
    rhs = np.asfortranarray(np.ones((2, 1)))
    assert L.sk_lapack_dgesv(2, 1, sing.ctypes.data_as(dp), 2, (i64 * 2)(), rhs.ctypes.data_as(dp), 2) == 1


def test_handles_without_gpu(built):
    import torch
    import sasktran2_amd as sk
    c = sk.Config()
    c.num_streams = 8
    assert c.num_streams == 8 and c.num_stokes == 1
    if not torch.cuda.is_available():
        from tests import scenarios as sc
        from tests.gpu_harness import build_engine
        with pytest.raises(RuntimeError, match="no HIP device"):
            build_engine(sc.regression_1d())


def _storage_with_mapping(L, nloc, nw, nleg, ssa, ext, em, leg, irr):
    import ctypes
    dp = ctypes.POINTER(ctypes.c_double)
    p = lambda a: a.ctypes.data_as(dp)
    h = L.sk_atmosphere_storage_create(nloc, nw, nleg, 1, p(ssa), p(ext), p(em), p(leg), p(irr))
    m = ctypes.c_void_p()
    assert L.sk_atmosphere_storage_get_derivative_mapping(h, b"wf_x", ctypes.byref(m)) == 0
    views = {}
    for which, n in (("d_ssa", nloc * nw), ("d_extinction", nloc * nw), ("scat_factor", nloc * nw), ("d_legendre", nleg * nloc * nw)):
        ptr = dp()
        assert getattr(L, f"sk_deriv_mapping_get_{which}")(m, ctypes.byref(ptr)) == 0
        import numpy as np
        views[which] = np.ctypeslib.as_array(ptr, shape=(n,))
    return h, m, views


def test_storage_handle_reuse_resets_delta_m_state(built):
    """The reference's Python layer runs set_zero -> refill -> apply_delta_m_scaling on ONE storage handle before every calculation
    (src/sasktran2/atmosphere.py:719-858; set_zero zeroes f and every mapping, grid_storage.h:319-329).  A second pass over the same
    handle with different optics must leave exactly the state a fresh handle gets."""
    import ctypes
    import numpy as np
    L = built.lib()
    nloc, nw, nleg, order = 5, 2, 12, 8
    rng = np.random.default_rng(11)
    l = np.arange(nleg)

    def optics(g):
        ext = np.asfortranarray(rng.uniform(1e-6, 1e-4, (nloc, nw)))
        ssa = np.asfortranarray(rng.uniform(0.5, 0.99, (nloc, nw)))
        leg = np.asfortranarray(np.broadcast_to(((2 * l + 1) * g ** l)[:, None, None], (nleg, nloc, nw)).copy())
        return ext, ssa, leg

    def fill_and_scale(h_atm, views, arrays, src):
        for a, b in zip(arrays, src):
            a[...] = b
        views["d_ssa"][:] = 1.0
        views["d_extinction"][:] = 1.0
        assert L.sk_atmosphere_apply_delta_m_scaling(h_atm, order) == 0

    em, irr = np.zeros((nloc, nw), order="F"), np.ones(nw)
    surface_emission = np.zeros(nw)
    h_su = L.sk_surface_create(nw, 1, surface_emission.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    first, second = optics(0.8), optics(0.6)
    # handle A: first optics, then set_zero and the second optics
    ext, ssa, leg = (np.zeros_like(x) for x in first)
    hA, mA, vA = _storage_with_mapping(L, nloc, nw, nleg, ssa, ext, em, leg, irr)
    atmA = L.sk_atmosphere_create(hA, h_su, 1, 0)
    fill_and_scale(atmA, vA, (ext, ssa, leg), first)
    assert L.sk_atmosphere_storage_set_zero(hA) == 0
    assert not ext.any() and not ssa.any() and not leg.any()
    assert not any(v.any() for v in vA.values())
    fill_and_scale(atmA, vA, (ext, ssa, leg), second)
    # handle B: second optics on a fresh handle
    ext2, ssa2, leg2 = (np.zeros_like(x) for x in second)
    hB, mB, vB = _storage_with_mapping(L, nloc, nw, nleg, ssa2, ext2, em, leg2, irr)
    atmB = L.sk_atmosphere_create(hB, h_su, 1, 0)
    fill_and_scale(atmB, vB, (ext2, ssa2, leg2), second)
    for a, b in ((ext, ext2), (ssa, ssa2), (leg, leg2), (vA["d_ssa"], vB["d_ssa"]), (vA["d_extinction"], vB["d_extinction"])):
        np.testing.assert_array_equal(a, b)
    f = second[2][order] / (2 * order + 1)
    np.testing.assert_allclose(ext, second[0] * (1 - second[1] * f), rtol=1e-15)
    # sk_deriv_mapping_set_zero clears every native array of the mapping
    vA["scat_factor"][:] = 2.0
    vA["d_legendre"][:] = 3.0
    assert L.sk_deriv_mapping_set_zero(mA) == 0
    assert not any(v.any() for v in vA.values())
    for h in (atmA, atmB):
        L.sk_atmosphere_destroy(h)
    for h in (hA, hB):
        L.sk_atmosphere_storage_destroy(h)
    L.sk_surface_destroy(h_su)


def test_two_stream_configuration_is_routed(built):
    # multiple_scatter_source = twostream / emission_source = twostream are accepted by sk_engine_create (configuration errors are reported
    # before any device is touched; without a GPU the creation then fails for the lack of a HIP device, never for the source)
    import torch
    import sasktran2_amd as sk
    from tests import scenarios as sc
    from tests.gpu_harness import build_ts_engine
    config = sk.Config()
    config.num_stokes = 3
    config.multiple_scatter_source = sk.MultipleScatterSource.TwoStream
    geometry = sk.Geometry1D(0.6, 0.0, 6_371_000.0, [0.0, 1000.0, 2000.0])
    viewing = sk.ViewingGeometry()
    viewing.add_ray(sk.GroundViewingSolar(0.6, 0.0, 0.8, 200_000.0))
    with pytest.raises(Exception, match="NSTOKES = 1"):
        sk.Engine(config, geometry, viewing)
    config = sk.Config()
    config.emission_source = sk.EmissionSource.TwoStream
    config.multiple_scatter_source = sk.MultipleScatterSource.DiscreteOrdinates
    with pytest.raises(Exception, match="not part of this build"):
        sk.Engine(config, geometry, viewing)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            build_ts_engine(sc.twostream_reference_case(spherical=True), solar=True, thermal=True)
