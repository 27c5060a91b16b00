/* sasktran2_amd -- C ABI of the MI355X-native DISCO hot path.
 *
 * Drop-in boundary: the `sk_*` entry points below have exactly the names, argument order and return-code
 * convention of the reference's C API for this path (libcsasktran2), so the reference's Rust binding
 * (rust/sasktran2-sys, bindgen of cpp/include/c_api/sasktran2.h) or any other FFI can bind them unchanged.
 * Each declaration cites the reference header line it replaces (paths relative to /root/reference).
 *
 * Return codes (cpp/c_api/engine.cpp:52-85,116-138): 0 success, -1 null/invalid handle, -2 unsupported mode
 * (NSTOKES != 1, non-DISCO sources, non-spherical geometry), -3 runtime failure (message on stderr).
 * No exception crosses the ABI.  All numeric arrays are caller-owned; the library only reads/writes them
 * during calls (cpp/include/sasktran2/atmosphere/grid_storage.h:77-102).
 *
 * Scope of this build: NSTOKES = 1, multiple_scatter_source = discrete_ordinates (0), single_scatter_source in
 * {discrete_ordinates (2), none (3)}, spherical Geometry1D with linear interpolation, Lambertian surface,
 * derivative mappings with d_extinction / d_ssa (no interpolator, no scattering groups) and surface d_brdf.
 * Everything is computed on the GPU; there is no CPU fallback.
 */
#ifndef SASKTRAN2_AMD_H
#define SASKTRAN2_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Config Config;
typedef struct Geometry1D Geometry1D;
typedef struct ViewingGeometry ViewingGeometry;
typedef struct AtmosphereStorage AtmosphereStorage;
typedef struct DerivativeMapping DerivativeMapping;
typedef struct SurfaceDerivativeMapping SurfaceDerivativeMapping;
typedef struct Atmosphere Atmosphere;
typedef struct Surface Surface;
typedef struct BRDF BRDF;
typedef struct OutputC OutputC;
typedef struct Engine Engine;

/* ---- config: cpp/include/c_api/config.h:9-134 (keys used by this path) ---- */
Config* sk_config_create(void);                                                  /* config.h:9  */
void sk_config_destroy(Config* config);                                          /* config.h:11 */
int sk_config_get_num_stokes(Config* config, int* num_stokes);                   /* config.h:13 */
int sk_config_set_num_stokes(Config* config, int num_stokes);                    /* config.h:14 */
int sk_config_get_multiple_scatter_source(Config* config, int* source);          /* config.h:16 */
int sk_config_set_multiple_scatter_source(Config* config, int source);           /* config.h:18 */
int sk_config_get_single_scatter_source(Config* config, int* source);            /* config.h:21 */
int sk_config_set_single_scatter_source(Config* config, int source);             /* config.h:23 */
int sk_config_get_num_streams(Config* config, int* num_streams);                 /* config.h:26 */
int sk_config_set_num_streams(Config* config, int num_streams);                  /* config.h:27 */
int sk_config_get_num_threads(Config* config, int* num_threads);                 /* config.h:30 */
int sk_config_set_num_threads(Config* config, int num_threads);                  /* config.h:31 */
int sk_config_get_wavelength_batch_size(Config* config, int* batch_size);        /* config.h:34 */
int sk_config_set_wavelength_batch_size(Config* config, int batch_size);         /* config.h:35 */
int sk_config_get_num_singlescatter_moments(Config* config, int* num_moments);   /* config.h:38 */
int sk_config_set_num_singlescatter_moments(Config* config, int num_moments);    /* config.h:39 */
int sk_config_get_num_do_sza(Config* config, int* num_sza);                      /* config.h:48 */
int sk_config_set_num_do_sza(Config* config, int num_sza);                       /* config.h:49 */
int sk_config_get_wf_enabled(Config* config, int* enabled);                      /* config.h:108 */
int sk_config_set_wf_enabled(Config* config, int enabled);                       /* config.h:109 */
int sk_config_get_output_los_optical_depth(Config* config, int* output);         /* config.h:122 */
int sk_config_set_output_los_optical_depth(Config* config, int output);          /* config.h:123 */

/* ---- geometry: cpp/include/c_api/geometry.h:10-20 ---- */
Geometry1D* sk_geometry1d_create(double cos_sza, double saa, double earth_radius, double* grid_values,
                                 int ngrid_values, int interp_method, int geotype);   /* geometry.h:10 */
void sk_geometry1d_destroy(Geometry1D* geometry);                                     /* geometry.h:15 */
int sk_geometry1d_get_num_altitudes(const Geometry1D* geometry);                      /* geometry.h:17 */
int sk_geometry1d_get_altitudes(const Geometry1D* geometry, double* altitudes);       /* geometry.h:19 */

/* ---- viewing geometry: cpp/include/c_api/viewing_geometry.h:9-39 ---- */
ViewingGeometry* sk_viewing_geometry_create(void);                                    /* viewing_geometry.h:9  */
void sk_viewing_geometry_destroy(ViewingGeometry* geometry);                          /* viewing_geometry.h:11 */
void sk_viewing_geometry_add_ground_viewing_solar(ViewingGeometry* geometry, double cos_sza,
                                                  double relative_azimuth_angle, double observeraltitude,
                                                  double cos_viewing_zenith);         /* viewing_geometry.h:13 */
int sk_viewing_geometry_add_tangent_altitude_solar(ViewingGeometry* geometry, double tangent_altitude_m,
                                                   double relative_azimuth_angle, double observeraltitude,
                                                   double cos_sza);                   /* viewing_geometry.h:19 */
int sk_viewing_geometry_num_rays(ViewingGeometry* geometry, int* num_rays);           /* viewing_geometry.h:36 */

/* ---- atmosphere: cpp/include/c_api/atmosphere.h:86-178, deriv_mapping.h:10-86, brdf.h:9-16 ---- */
AtmosphereStorage* sk_atmosphere_storage_create(int nlocation, int nwavel, int nphase_moments, int nstokes, double* ssa,
                                                double* total_extinction, double* emission_source, double* leg_coeff,
                                                double* solar_irradiance);            /* atmosphere.h:86 */
void sk_atmosphere_storage_destroy(AtmosphereStorage* storage);                       /* atmosphere.h:97 */
int sk_atmosphere_storage_get_derivative_mapping(AtmosphereStorage* storage, const char* name,
                                                 DerivativeMapping** mapping);        /* atmosphere.h:107 */
int sk_atmosphere_storage_get_num_derivative_mappings(AtmosphereStorage* storage, int* num_mappings); /* atmosphere.h:114 */
int sk_atmosphere_storage_get_derivative_mapping_name(AtmosphereStorage* storage, int index, const char** name); /* atmosphere.h:117 */
int sk_deriv_mapping_get_d_ssa(DerivativeMapping* mapping, double** ssa);             /* deriv_mapping.h:12 */
int sk_deriv_mapping_get_d_extinction(DerivativeMapping* mapping, double** extinction); /* deriv_mapping.h:13 */
int sk_deriv_mapping_get_num_location(DerivativeMapping* mapping, int* num_location); /* deriv_mapping.h:26 */
int sk_deriv_mapping_get_num_wavel(DerivativeMapping* mapping, int* num_wavel);       /* deriv_mapping.h:28 */
int sk_deriv_mapping_get_num_output(DerivativeMapping* mapping, int* num_output);     /* deriv_mapping.h:44 */
int sk_deriv_mapping_set_zero(DerivativeMapping* mapping);                            /* deriv_mapping.h:11 */
BRDF* sk_brdf_create_lambertian(int nstokes);                                         /* brdf.h:9 */
void sk_brdf_destroy(BRDF* brdf);                                                     /* brdf.h:16 */
Surface* sk_surface_create(int nwavel, int nstokes, double* emission);                /* atmosphere.h:169 */
void sk_surface_destroy(Surface* surface);                                            /* atmosphere.h:176 */
int sk_surface_set_brdf(Surface* surface, BRDF* brdf, double* brdf_args);             /* atmosphere.h:178 */
int sk_surface_get_derivative_mapping(Surface* surface, const char* name, SurfaceDerivativeMapping** mapping); /* atmosphere.h:180 */
int sk_surface_deriv_mapping_get_d_brdf(SurfaceDerivativeMapping* mapping, double** brdf); /* deriv_mapping.h:64 */
Atmosphere* sk_atmosphere_create(AtmosphereStorage* storage, Surface* surface, int calculate_derivatives,
                                 int calculate_emission_derivatives);                 /* atmosphere.h:139 */
void sk_atmosphere_destroy(Atmosphere* atmosphere);                                   /* atmosphere.h:148 */

/* ---- output: cpp/include/c_api/output.h:11-33 ---- */
OutputC* sk_output_create(double* radiance, int nrad, int nstokes, double* flux, int nflux);   /* output.h:11 */
void sk_output_destroy(OutputC* output);                                              /* output.h:13 */
int sk_output_assign_derivative_memory(OutputC* output, const char* name, double* derivative_mapping, int nrad,
                                       int nstokes, int nderiv);                      /* output.h:15 */
int sk_output_assign_surface_derivative_memory(OutputC* output, const char* name, double* derivative_mapping, int nrad,
                                               int nstokes);                          /* output.h:19 */
int sk_output_get_los_optical_depth(OutputC* output, double** od);                    /* output.h:33 */

/* ---- engine: cpp/include/c_api/engine.h:17-49 ---- */
Engine* sk_engine_create(Config* config, Geometry1D* geometry, ViewingGeometry* viewing_geometry);  /* engine.h:17 */
int sk_engine_calculate_radiance(Engine* engine, Atmosphere* atmosphere, OutputC* output, int only_initialize); /* engine.h:23 */
int sk_engine_effective_wavelength_batch_size(Engine* engine, int num_wavelengths);   /* engine.h:25 */
int sk_engine_calculate_radiance_block_thread(Engine* engine, OutputC* output, int wavelength_start,
                                              int wavelength_count, int thread_idx);  /* engine.h:42 */
void sk_engine_destroy(Engine* engine);                                               /* engine.h:47 */
int sk_openmp_support_enabled(void);                                                  /* engine.h:49 */

/* ---- additive MI355X entry points (no reference counterpart) ----
 * Device-resident solve: all pointers are HIP device pointers (see DeviceEngine::solve_device for layouts); `stream`
 * is a hipStream_t (NULL = default stream).  Used by bench.py so the timed region starts with inputs in HBM. */
int sk_amd_engine_solve_device(Engine* engine, int nwavel, int nleg, const double* d_ext, const double* d_ssa,
                               const double* d_leg, const double* d_albedo, const double* d_irradiance, int derivatives,
                               double* d_radiance, double* d_wf_ext, double* d_wf_ssa, double* d_wf_albedo,
                               double* d_los_od, int* d_status, void* stream, int max_chunk);
int sk_amd_engine_num_los(Engine* engine, int* nlos);
int sk_amd_engine_scratch_bytes(Engine* engine, int nwavel, unsigned long long* bytes);
/* shape of the last solve: wavelengths per kernel launch, blocks in flight (streams), scratch bytes allocated */
int sk_amd_engine_launch_shape(Engine* engine, int* chunk, int* slots, unsigned long long* scratch_bytes);
int sk_amd_engine_set_profiling(Engine* engine, int enabled);   /* per-kernel HIP-event timing (serialises launches) */
int sk_amd_engine_reset_profile(Engine* engine);
int sk_amd_engine_kernel_time(Engine* engine, const char* kernel_name, double* total_ms, long* launches);
int sk_amd_engine_kernel_names(Engine* engine, char* buffer, int buffer_len);
const char* sk_amd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
