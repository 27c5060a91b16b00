"""Loads the HIP shared library (sasktran2_amd/_lib/libsasktran2_amd.so) and declares the C ABI of include/sasktran2_amd.h.

There is deliberately no CPU fallback: if the library is missing or no MI355X is visible, engine construction fails."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# SASKTRAN2_AMD_LIB selects another build of the same HIP library (kernel tuning variants); never a different backend
LIB_PATH = Path(os.environ.get("SASKTRAN2_AMD_LIB") or (_HERE / "_lib" / "libsasktran2_amd.so"))
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p
_LIB = None

# every symbol include/sasktran2_amd.h declares: (name, restype, argtypes)
_I, _D, _S = ctypes.c_int, ctypes.c_double, ctypes.c_char_p
SIGNATURES = [
    ("sk_config_create", _vp, []), ("sk_config_destroy", None, [_vp]),
    *[(f"sk_config_get_{k}", _I, [_vp, _ip]) for k in ("num_stokes", "multiple_scatter_source", "single_scatter_source",
      "num_streams", "num_threads", "wavelength_batch_size", "num_singlescatter_moments", "num_do_sza", "wf_enabled",
      "output_los_optical_depth", "apply_delta_scaling", "stokes_basis", "num_hr_spherical_iterations", "num_hr_incoming",
      "num_hr_outgoing", "initialize_hr_with_do", "emission_source", "occultation_source", "do_backprop", "num_do_forced_azimuth")],
    *[(f"sk_config_set_{k}", _I, [_vp, _I]) for k in ("num_stokes", "multiple_scatter_source", "single_scatter_source",
      "num_streams", "num_threads", "wavelength_batch_size", "num_singlescatter_moments", "num_do_sza", "wf_enabled",
      "output_los_optical_depth", "apply_delta_scaling", "stokes_basis", "num_hr_spherical_iterations", "num_hr_incoming",
      "num_hr_outgoing", "initialize_hr_with_do", "emission_source", "occultation_source", "do_backprop", "num_do_forced_azimuth")],
    *[(f"sk_config_get_{k}", _I, [_vp, _dp]) for k in ("successive_orders_relative_tolerance", "successive_orders_absolute_tolerance", "successive_orders_damping")],
    *[(f"sk_config_set_{k}", _I, [_vp, _D]) for k in ("successive_orders_relative_tolerance", "successive_orders_absolute_tolerance", "successive_orders_damping")],
    ("sk_config_get_successive_orders_anderson_depth", _I, [_vp, _ip]), ("sk_config_set_successive_orders_anderson_depth", _I, [_vp, _I]),
    ("sk_geometry1d_create", _vp, [_D, _D, _D, _dp, _I, _I, _I]), ("sk_geometry1d_destroy", None, [_vp]),
    ("sk_geometry1d_get_num_altitudes", _I, [_vp]), ("sk_geometry1d_get_altitudes", _I, [_vp, _dp]),
    ("sk_viewing_geometry_create", _vp, []), ("sk_viewing_geometry_destroy", None, [_vp]),
    ("sk_viewing_geometry_add_ground_viewing_solar", None, [_vp, _D, _D, _D, _D]),
    ("sk_viewing_geometry_add_tangent_altitude_solar", _I, [_vp, _D, _D, _D, _D]),
    ("sk_viewing_geometry_add_solar_angles_observer_location", _I, [_vp, _D, _D, _D, _D]),
    ("sk_viewing_geometry_add_tangent_altitude", _I, [_vp, _D, _D, _D, _D]),
    ("sk_viewing_geometry_num_rays", _I, [_vp, _ip]),
    ("sk_viewing_geometry_add_flux_observer_solar", _I, [_vp, _D, _D]), ("sk_viewing_geometry_num_flux_observers", _I, [_vp, _ip]),
    ("sk_config_get_num_flux_types", _I, [_vp, _ip]), ("sk_config_get_flux_types", _I, [_vp, _ip]), ("sk_config_set_flux_types", _I, [_vp, _ip, _I]),
    ("sk_atmosphere_storage_create", _vp, [_I, _I, _I, _I, _dp, _dp, _dp, _dp, _dp]),
    ("sk_atmosphere_storage_destroy", None, [_vp]),
    ("sk_atmosphere_storage_get_derivative_mapping", _I, [_vp, _S, ctypes.POINTER(_vp)]),
    ("sk_atmosphere_storage_get_num_derivative_mappings", _I, [_vp, _ip]),
    ("sk_atmosphere_storage_get_derivative_mapping_name", _I, [_vp, _I, ctypes.POINTER(_S)]),
    ("sk_deriv_mapping_get_d_ssa", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_deriv_mapping_get_d_extinction", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_deriv_mapping_get_scat_factor", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_deriv_mapping_get_d_legendre", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_atmosphere_storage_set_zero", _I, [_vp]), ("sk_deriv_mapping_set_log_radiance_space", _I, [_vp, _I]),
    ("sk_deriv_mapping_get_num_location", _I, [_vp, _ip]), ("sk_deriv_mapping_get_num_wavel", _I, [_vp, _ip]),
    ("sk_deriv_mapping_get_num_output", _I, [_vp, _ip]), ("sk_deriv_mapping_set_zero", _I, [_vp]),
    ("sk_brdf_create_lambertian", _vp, [_I]), ("sk_brdf_destroy", None, [_vp]),
    ("sk_brdf_create_kokhanovsky", _vp, [_I]), ("sk_brdf_create_modis", _vp, [_I]), ("sk_amd_brdf_create_legacy_test_function", _vp, [_I]),
    ("sk_surface_create", _vp, [_I, _I, _dp]), ("sk_surface_destroy", None, [_vp]),
    ("sk_surface_set_brdf", _I, [_vp, _vp, _dp]),
    ("sk_surface_get_derivative_mapping", _I, [_vp, _S, ctypes.POINTER(_vp)]),
    ("sk_surface_deriv_mapping_get_d_brdf", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_surface_deriv_mapping_get_d_emission", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_atmosphere_create", _vp, [_vp, _vp, _I, _I]), ("sk_atmosphere_destroy", None, [_vp]),
    ("sk_output_create", _vp, [_dp, _I, _I, _dp, _I]), ("sk_output_destroy", None, [_vp]),
    ("sk_output_assign_derivative_memory", _I, [_vp, _S, _dp, _I, _I, _I]),
    ("sk_output_assign_surface_derivative_memory", _I, [_vp, _S, _dp, _I, _I]),
    ("sk_output_get_los_optical_depth", _I, [_vp, ctypes.POINTER(_dp)]),
    ("sk_output_jvp_create", _vp, [_dp, _dp, _I, _I]), ("sk_output_jvp_destroy", None, [_vp]),
    ("sk_output_jvp_assign_derivative_tangent", _I, [_vp, _S, _dp, _I]), ("sk_output_jvp_assign_surface_tangent", _I, [_vp, _S, _dp, _I]),
    ("sk_output_vjp_create", _vp, [_dp, _dp, _I, _I]), ("sk_output_vjp_destroy", None, [_vp]),
    ("sk_output_vjp_assign_derivative_gradient", _I, [_vp, _S, _dp, _I]), ("sk_output_vjp_assign_surface_gradient", _I, [_vp, _S, _dp, _I]),
    ("sk_output_vjp_finalize", _I, [_vp]),
    ("sk_engine_calculate_jvp", _I, [_vp, _vp, _vp]), ("sk_engine_calculate_vjp", _I, [_vp, _vp, _vp]),
    ("sk_engine_create", _vp, [_vp, _vp, _vp]),
    ("sk_engine_calculate_radiance", _I, [_vp, _vp, _vp, _I]),
    ("sk_engine_effective_wavelength_batch_size", _I, [_vp, _I]),
    ("sk_engine_calculate_radiance_block_thread", _I, [_vp, _vp, _I, _I, _I]),
    ("sk_engine_destroy", None, [_vp]), ("sk_openmp_support_enabled", _I, []),
    ("sk_amd_engine_solve_device", _I, [_vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _I]),
    ("sk_amd_engine_set_delta_m_fraction", _I, [_vp, _vp]),
    ("sk_amd_slab_row_doubles", _I, [_vp, _I, ctypes.POINTER(ctypes.c_ulonglong)]),
    ("sk_amd_engine_solve_device_slab", _I, [_vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _I, _vp, _vp, _vp, _I]),
    ("sk_amd_unpack_slab", _I, [_vp, _vp, _I, _I, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("sk_amd_comm_unique_id", _I, [ctypes.c_char_p]), ("sk_amd_comm_create", _vp, [ctypes.c_char_p, _I, _I]), ("sk_amd_comm_destroy", None, [_vp]),
    ("sk_amd_engine_calculate_radiance_sharded", _I, [_vp, _vp, _vp, _vp, _I, _I]),
    ("sk_amd_engine_calculate_radiance_sharded_emulated", _I, [_vp, _vp, _vp, _I, _I]),
    ("sk_amd_so_engine_solve_device", _I, [_vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("sk_amd_so_engine_solve_device_wf", _I, [_vp, _I, _I, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("sk_amd_so_engine_sizes", _I, [_vp, ctypes.POINTER(ctypes.c_ulonglong)]),
    ("sk_amd_ts_engine_solve_device", _I, [_vp, _I, _I] + [_vp] * 17 + [_I]),
    ("sk_atmosphere_apply_delta_m_scaling", _I, [_vp, _I]),
    ("sk_amd_frontend_create", _vp, [_I, _I, _I, _I]), ("sk_amd_frontend_destroy", None, [_vp]), ("sk_amd_frontend_reset", _I, [_vp]),
    ("sk_amd_frontend_add_term", _I, [_vp, _vp, _vp, _vp, _vp]), ("sk_amd_frontend_readd_term", _I, [_vp, _vp]), ("sk_amd_frontend_set_profile", _I, [_vp, _I, _vp]),
    ("sk_amd_frontend_set_planck_emission", _I, [_vp, _vp, _vp]), ("sk_amd_frontend_set_solar_irradiance", _I, [_vp, _vp]),
    ("sk_amd_frontend_assemble", _I, [_vp]),
    ("sk_amd_frontend_add_mapping", _I, [_vp, _S, _I, _vp, _vp, _I, _I, _vp, _I, _S, _S]),
    ("sk_amd_frontend_finalize", _I, [_vp]),
    ("sk_amd_frontend_storage", _vp, [_vp]),
    ("sk_amd_frontend_device_pointer", _I, [_vp, _S, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_ulonglong)]),
    ("sk_amd_frontend_download", _I, [_vp, _S, _vp]),
    ("sk_amd_frontend_set_profiling", _I, [_vp, _I]),
    ("sk_amd_frontend_kernel_time", _I, [_vp, _S, _dp, ctypes.POINTER(ctypes.c_long)]),
    ("sk_amd_frontend_kernel_names", _I, [_vp, ctypes.c_char_p, _I]),
    ("sk_amd_engine_num_los", _I, [_vp, _ip]),
    ("sk_amd_engine_scratch_bytes", _I, [_vp, _I, ctypes.POINTER(ctypes.c_ulonglong)]),
    ("sk_amd_engine_launch_shape", _I, [_vp, _ip, _ip, ctypes.POINTER(ctypes.c_ulonglong)]),
    ("sk_amd_host_alloc", _vp, [ctypes.c_ulonglong]),
    ("sk_amd_host_free", None, [_vp]),
    ("sk_amd_engine_set_profiling", _I, [_vp, _I]), ("sk_amd_engine_reset_profile", _I, [_vp]),
    ("sk_amd_engine_kernel_time", _I, [_vp, _S, _dp, ctypes.POINTER(ctypes.c_long)]),
    ("sk_amd_engine_kernel_names", _I, [_vp, ctypes.c_char_p, _I]),
    ("sk_amd_last_error", _S, []),
]


def lib():
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        L = ctypes.CDLL(str(LIB_PATH))
        for name, res, args in SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error() -> str:
    return lib().sk_amd_last_error().decode()
