/* sasktran2_amd -- C ABI of the MI355X-native DISCO hot path.
 *
 * Drop-in boundary: the `sk_*` entry points below have exactly the names, argument order and return-code
 * convention of the reference's C API for this path (libcsasktran2), so the reference's Rust binding
 * (rust/sasktran2-sys, bindgen of cpp/include/c_api/sasktran2.h) or any other FFI can bind them unchanged.
 * Each declaration cites the reference header line it replaces (paths relative to /root/reference).
 *
 * Return codes (cpp/c_api/engine.cpp:52-85,116-138): 0 success, -1 null/invalid handle, -2 unsupported mode
 * (NSTOKES other than 1 / 3, non-DISCO sources, refraction, emission derivatives), -3 runtime failure (message on stderr).
 * No exception crosses the ABI.  All numeric arrays are caller-owned; the library only reads/writes them
 * during calls (cpp/include/sasktran2/atmosphere/grid_storage.h:77-102).
 *
 * Scope of this build (README.md holds the table of accepted source combinations): NSTOKES = 1 and NSTOKES = 3 (radiances + weighting
 * functions; standard / solar / observer Stokes basis); multiple_scatter_source discrete_ordinates (0), successive_orders_legacy (1),
 * twostream (2), none (3); single_scatter_source exact (0), solartable (1), discrete_ordinates (2), none (3); emission_source standard (0),
 * discrete_ordinates (2), volume_emission_rate (3), twostream (4); occultation_source standard; Geometry1D in spherical,
 * pseudo-spherical or plane-parallel form with linear interpolation, Lambertian surface, delta-M scaling, derivative mappings with
 * d_extinction / d_ssa / d_emission (optionally with an interpolator and in log-radiance space, and with d_legendre / scat_factor =
 * scattering-group derivatives) and surface d_brdf / d_emission; the optical-property front end on the device (sk_amd_frontend_*).
 * Everything is computed on the GPU; there is no CPU fallback.
 */
#ifndef SASKTRAN2_AMD_H
#define SASKTRAN2_AMD_H

#include <stdint.h>

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
/* delta-M truncation fraction of already scaled optical properties, DEVICE pointer [nwavel][nlocation] or NULL, for the
 * following sk_amd_engine_solve_device calls (the host entry point does this itself after sk_atmosphere_apply_delta_m_scaling) */
int sk_amd_engine_set_delta_m_fraction(Engine* engine, const double* d_f);
/* Wavelength-sharded runs, one process per GPU (the reference threads over wavelengths, cpp/lib/engine/engine.cpp:610-622):
 * a rank solves its contiguous wavelength range into ONE wavelength-major slab -- row w = [radiance | los_od | wf_extinction(q) |
 * wf_ssa(q) | wf_albedo] x nlos doubles, sk_amd_slab_row_doubles per row -- the ranks exchange slabs with ONE all-gather
 * (ncclAllGather; rows_per_rank = the largest shard) and sk_amd_unpack_slab writes the reference layouts from the gathered block. */
int sk_amd_slab_row_doubles(Engine* engine, int derivatives, unsigned long long* doubles_per_row);
int sk_amd_engine_solve_device_slab(Engine* engine, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                    const double* d_albedo, const double* d_irradiance, int derivatives, double* d_slab, int* d_status,
                                    void* stream, int max_chunk);
int sk_amd_unpack_slab(Engine* engine, const double* d_slab_all, int nwavel, int world, int rows_per_rank, int derivatives,
                       double* d_radiance, double* d_los_od, double* d_wf_ext, double* d_wf_ssa, double* d_wf_albedo, void* stream);
/* The same for drop-in callers of the HOST entry point that run one process per GPU: sk_engine_calculate_radiance's arguments plus an
 * RCCL communicator.  Every rank passes the full atmosphere / output buffers, solves its wavelength range and receives the full result
 * through one ncclAllGather of the packed slab.  The communicator: rank 0 obtains a 128-byte id (sk_amd_comm_unique_id), the caller
 * distributes it (MPI_Bcast, a file, a torch store ...), every rank calls sk_amd_comm_create(id, rank, world) after selecting its GPU. */
int sk_amd_comm_unique_id(char* id128);
void* sk_amd_comm_create(const char* id128, int rank, int world);
void sk_amd_comm_destroy(void* comm);
int sk_amd_engine_calculate_radiance_sharded(Engine* engine, Atmosphere* atmosphere, OutputC* output, void* comm, int rank, int world);
/* Errors: more ranks than wavelengths, flux derivative outputs and device-resident storage are refused alike on every rank (-3) before
 * anything is exchanged.  A failure of ONE rank's local solve does not leave the others waiting: the rank still takes part in the
 * all-gather with a zeroed slab and a status word, and every rank returns that rank's code with an error that names it.
 * RCCL (librccl.so) is loaded on the first sk_amd_comm_* call; the library itself does not depend on it.
 * ..._emulated: the same partition, local solves and slab arithmetic with the `world` ranks played in turn by the calling process and no
 * collective -- what one GPU can check of the entry point above (fail_rank >= 0 marks that rank as failed; -1: none). */
int sk_amd_engine_calculate_radiance_sharded_emulated(Engine* engine, Atmosphere* atmosphere, OutputC* output, int world, int fail_rank);
/* The same for an engine with single_scatter_source = exact and / or multiple_scatter_source = successive_orders_legacy
 * (replaces, per wavelength, SingleScatterSource<SolarTransmissionExact>::calculate / integrated_source,
 * cpp/lib/solar/singlescattersource.cpp:103-160,950-1167, and hr::DiffuseTable::calculate, cpp/lib/hr/diffuse_table.cpp:885-962):
 * d_leg [nwavel][nlocation][nleg (x 4 for NSTOKES = 3)], d_radiance [nwavel][nlos][nstokes], d_los_od [nwavel][nlos] or NULL. */
int sk_amd_so_engine_solve_device(Engine* engine, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                  const double* d_albedo, const double* d_irradiance, double* d_radiance, double* d_los_od, void* stream);
/* ... with the native weighting functions of every source of the solve -- the exact single scatter in reverse mode along the line of sight,
 * the successive-orders diffuse field by the discrete adjoint of its passes (the reference's new source serves these through
 * Sasktran2::calculate_radiance's native VJP, cpp/lib/engine/engine.cpp:625-866; its legacy source has none):
 * d_wf_ext / d_wf_ssa [nlocation][nwavel][nlos][nstokes], d_wf_albedo [nwavel][nlos][nstokes]. */
int sk_amd_so_engine_solve_device_wf(Engine* engine, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                     const double* d_albedo, const double* d_irradiance, double* d_radiance, double* d_los_od,
                                     double* d_wf_ext, double* d_wf_ssa, double* d_wf_albedo, void* stream);
/* The same for an engine whose source is the two-stream one (multiple_scatter_source = twostream and / or emission_source = twostream;
 * replaces, per wavelength block, CppTwoStreamSourceAdapter::calculate + start_of_ray_source,
 * cpp/lib/sktran_disco/cpp_twostream_source.cpp:3532-3630).  Inputs in the reference storage layout: d_ext, d_ssa, d_delta_f, d_emission
 * [nwavel][nlocation], d_leg [nwavel][nlocation][nleg], d_albedo / d_irradiance / d_surface_emission [nwavel]; d_delta_f (truncation
 * fraction of already delta-M scaled optics), d_irradiance, d_emission and d_surface_emission may be NULL as the configured sources allow.
 * Outputs: d_radiance [nwavel][nlos], d_los_od [nwavel][nlos] or NULL, and -- all of d_wf_ext / d_wf_ssa / d_wf_b1 / d_wf_albedo or none --
 * d_wf_ext, d_wf_ssa, d_wf_b1 (first Legendre coefficient), d_wf_emission [nlocation][nwavel][nlos], d_wf_albedo,
 * d_wf_surface_emission [nwavel][nlos] (the two emission outputs may be NULL).  max_chunk: wavelengths per kernel launch, 0 = automatic. */
int sk_amd_ts_engine_solve_device(Engine* engine, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                  const double* d_delta_f, const double* d_albedo, const double* d_irradiance, const double* d_emission,
                                  const double* d_surface_emission, double* d_radiance, double* d_los_od, double* d_wf_ext, double* d_wf_ssa,
                                  double* d_wf_b1, double* d_wf_albedo, double* d_wf_emission, double* d_wf_surface_emission, void* stream,
                                  int max_chunk);
/* ---- optical-property front end on the device ----
 * Replaces, for the constituents whose contribution is (a sum of) profile x spectrum, what the reference's Rust front end does on the host:
 * Constituent::add_to_atmosphere / register_derivatives (rust/sasktran2-rs/src/constituent/types/rayleigh.rs:110-258,
 * vmr_alt_absorber.rs:64-260, number_density_scatterer.rs:218-440, emission.rs:52-116) and AtmosphereStorage::normalize_by_extinctions
 * (rust/sasktran2-rs/src/bindings/atmosphere_storage.rs:113-153).  The caller hands over what is small -- amount profiles [nlocation],
 * spectra [nwavel] -- and the storage arrays [nwavel][nlocation] and every derivative mapping are formed in HBM; the storage handle goes
 * to sk_atmosphere_create / sk_engine_calculate_radiance like one built by sk_atmosphere_storage_create, without an upload.  Host-array
 * accessors of such a mapping (sk_deriv_mapping_get_d_ssa ...) and sk_atmosphere_storage_set_zero return -2; sk_amd_frontend_download
 * copies the arrays for inspection.  sk_atmosphere_apply_delta_m_scaling scales storage and mappings in place on the device. */
typedef struct FrontEnd sk_amd_frontend;
sk_amd_frontend* sk_amd_frontend_create(int nlocation, int nwavel, int nleg /* first dimension of leg_coeff */, int nstokes);
void sk_amd_frontend_destroy(sk_amd_frontend* f);
int sk_amd_frontend_reset(sk_amd_frontend* f);   /* forget terms, mappings, emission */
/* one rank-one term: amount profile [nlocation] x extinction cross section [nwavel]; scat (scattering cross section [nwavel]) and
 * leg (phase moments per scattering event [nwavel][nleg]) may be NULL.  Returns the term's index (>= 0) or an error (< 0). */
int sk_amd_frontend_add_term(sk_amd_frontend* f, const double* profile, const double* ext, const double* scat, const double* leg);
/* the next term again after a reset, keeping the spectra the term at this position held (still in HBM) and taking a new profile; returns
   the term index, -2 when no term was added at this position before the last reset */
int sk_amd_frontend_readd_term(sk_amd_frontend* f, const double* profile);
int sk_amd_frontend_set_profile(sk_amd_frontend* f, int term, const double* profile);
/* emission[q][w] += Planck radiance B(temperature_k[q], wavelengths_nm[w]) in W / (m^2 sr nm)  (emission.rs:9-16) */
int sk_amd_frontend_set_planck_emission(sk_amd_frontend* f, const double* temperature_k, const double* wavelengths_nm);
int sk_amd_frontend_set_solar_irradiance(sk_amd_frontend* f, const double* irradiance /* [nwavel] or NULL */);
int sk_amd_frontend_assemble(sk_amd_frontend* f);
/* a derivative mapping from (term index, per-location weight) pairs: e = sum_r weight_r[q] ext_r[w], s = sum_r weight_r[q] scat_r[w] ->
 * d_extinction = e, d_ssa = e (s / e - ssa) / k, and for scatterer != 0 scat_factor = s / (ssa k), d_legendre = P / s - leg_coeff.
 * weights: host [nterms][nlocation] or NULL (1); planck_temperature != 0: d_emission = dB/dT; interpolator: host, column major
 * [nlocation x nout], or NULL; interp_dim / assign_name as deriv_mapping.h:32,40. */
int sk_amd_frontend_add_mapping(sk_amd_frontend* f, const char* name, int nterms, const int* terms, const double* weights, int scatterer,
                                int planck_temperature, const double* interpolator, int nout, const char* interp_dim, const char* assign_name);
/* add_mapping only records a mapping; its arrays are formed, together with those of every other pending mapping (two kernel launches for
 * all of them), by sk_amd_frontend_finalize -- or by the first call that needs them (_storage, _device_pointer, _download, delta-M scaling) */
int sk_amd_frontend_finalize(sk_amd_frontend* f);
AtmosphereStorage* sk_amd_frontend_storage(sk_amd_frontend* f);   /* borrowed: valid until the front end is destroyed, reset or assembled again */
/* what = "extinction" | "ssa" | "leg_coeff" | "emission" | "delta_f" | "<mapping>:d_extinction" | ":d_ssa" | ":scat_factor" |
 * ":d_legendre" | ":d_emission" | ":d_f" */
int sk_amd_frontend_device_pointer(sk_amd_frontend* f, const char* what, const double** pointer, unsigned long long* count);
int sk_amd_frontend_download(sk_amd_frontend* f, const char* what, double* host_destination);
/* per-kernel HIP-event timing of the front end's launches (serialises them), as sk_amd_engine_set_profiling / _kernel_time / _kernel_names */
int sk_amd_frontend_set_profiling(sk_amd_frontend* f, int enabled);
int sk_amd_frontend_kernel_time(sk_amd_frontend* f, const char* kernel_name, double* total_ms, long* launches);
int sk_amd_frontend_kernel_names(sk_amd_frontend* f, char* buffer, int buffer_len);
/* sizes[0..5]: incoming rays, their layers, accumulation-matrix entries, outgoing directions, scratch bytes per wavelength,
 * wavelengths per block of the last solve */
int sk_amd_so_engine_sizes(Engine* engine, unsigned long long* sizes);
int sk_amd_engine_num_los(Engine* engine, int* nlos);
int sk_amd_engine_scratch_bytes(Engine* engine, int nwavel, unsigned long long* bytes);
/* shape of the last solve: wavelengths per kernel launch, blocks in flight (streams), scratch bytes allocated */
int sk_amd_engine_launch_shape(Engine* engine, int* chunk, int* slots, unsigned long long* scratch_bytes);
int sk_amd_engine_set_profiling(Engine* engine, int enabled);   /* 1: per-kernel HIP-event timing, launches serialised; 2 (discrete-ordinates engine):
                                                                    event pairs on each launch's own stream, nothing serialised -- the kernels'
                                                                    durations inside the pipelined step; 0: off */
int sk_amd_engine_reset_profile(Engine* engine);
int sk_amd_engine_kernel_time(Engine* engine, const char* kernel_name, double* total_ms, long* launches);
int sk_amd_engine_kernel_names(Engine* engine, char* buffer, int buffer_len);
const char* sk_amd_last_error(void);
/* Page-locked host memory for the arrays a caller hands to sk_atmosphere_storage_create / sk_output_assign_derivative_memory (additive; the
 * reference maps caller memory, cpp/include/sasktran2/atmosphere/grid_storage.h:77-102, and so does this library): the DMA engines read
 * and write such arrays without a staging copy -- the host entry point moves 74 MB in and 208 MB out per 10 000 wavelengths of the 16-stream
 * limb case.  sk_amd_host_alloc returns NULL when the memory cannot be pinned (the caller then uses ordinary memory: every entry point takes both). */
void* sk_amd_host_alloc(unsigned long long bytes);
void sk_amd_host_free(void* ptr);

/* =====================================================================================================================
 * Rest of the reference's C API (the headers under cpp/include/c_api): every symbol its FFI links against is exported, so replacing
 * libcsasktran2 by this library is a pure link-time choice.  Entry points outside the DISCO hot path keep the
 * reference's signatures; they store / return their settings where that is all they do, and report "unsupported"
 * (-2, or a null handle) where they would need an engine this build does not contain (2D geometry, flux weighting
 * functions, geodetic utilities).  Flux observers are served by the plane-parallel discrete-ordinates source
 * (do_source_planeparallel.cpp:221-566): sk_output_create's flux block is [flux type][wavelength][observer] in the order of
 * sk_config_set_flux_types (0 upwelling, 1 downwelling, 2 actinic).  Reference line numbers as above.
 * ===================================================================================================================== */
typedef struct Geometry2D Geometry2D;
typedef struct Geodetic Geodetic;
typedef struct OutputJVP OutputJVP;
typedef struct OutputVJP OutputVJP;

/* config.h:32-133 -- stored and returned; values other than the defaults are rejected by sk_engine_create when they would
 * change the DISCO result (refraction, delta scaling, forced azimuth, emission / occultation sources) */
int sk_config_get_threading_model(Config* config, int* threading_model);                       /* config.h:32 */
int sk_config_set_threading_model(Config* config, int threading_model);                        /* config.h:33 */
int sk_config_get_singlescatter_phasemode(Config* config, int* phasemode);                     /* config.h:40 */
int sk_config_set_singlescatter_phasemode(Config* config, int phasemode);                      /* config.h:41 */
int sk_config_get_apply_delta_scaling(Config* config, int* apply_delta_scaling);               /* config.h:44 */
int sk_config_set_apply_delta_scaling(Config* config, int apply_delta_scaling);                /* config.h:45 */
int sk_config_get_num_do_forced_azimuth(Config* config, int* num_forced_azimuth);              /* config.h:50 */
int sk_config_set_num_do_forced_azimuth(Config* config, int num_forced_azimuth);               /* config.h:52 */
int sk_config_get_do_backprop(Config* config, int* do_backprop);                               /* config.h:53 */
int sk_config_set_do_backprop(Config* config, int do_backprop);                                /* config.h:54 */
int sk_config_get_num_do_spherical_iterations(Config* config, int* num_iterations);            /* config.h:55 */
int sk_config_set_num_do_spherical_iterations(Config* config, int num_iterations);             /* config.h:57 */
int sk_config_get_num_hr_spherical_iterations(Config* config, int* num_iterations);            /* config.h:61 */
int sk_config_set_num_hr_spherical_iterations(Config* config, int num_iterations);             /* config.h:63 */
int sk_config_get_num_hr_incoming(Config* config, int* num_incoming);                          /* config.h:65 */
int sk_config_set_num_hr_incoming(Config* config, int num_incoming);                           /* config.h:66 */
int sk_config_get_num_hr_outgoing(Config* config, int* num_outgoing);                          /* config.h:67 */
int sk_config_set_num_hr_outgoing(Config* config, int num_outgoing);                           /* config.h:68 */
int sk_config_get_num_hr_full_incoming_points(Config* config, int* num_points);                /* config.h:69 */
int sk_config_set_num_hr_full_incoming_points(Config* config, int num_points);                 /* config.h:70 */
int sk_config_get_initialize_hr_with_do(Config* config, int* initialize);                      /* config.h:71 */
int sk_config_set_initialize_hr_with_do(Config* config, int initialize);                       /* config.h:72 */
int sk_config_get_successive_orders_relative_tolerance(Config* config, double* tolerance);     /* config.h:75 */
int sk_config_set_successive_orders_relative_tolerance(Config* config, double tolerance);      /* config.h:77 */
int sk_config_get_successive_orders_absolute_tolerance(Config* config, double* tolerance);     /* config.h:79 */
int sk_config_set_successive_orders_absolute_tolerance(Config* config, double tolerance);      /* config.h:81 */
int sk_config_get_successive_orders_anderson_depth(Config* config, int* depth);                /* config.h:83 */
int sk_config_set_successive_orders_anderson_depth(Config* config, int depth);                 /* config.h:84 */
int sk_config_get_successive_orders_damping(Config* config, double* damping);                  /* config.h:85 */
int sk_config_set_successive_orders_damping(Config* config, double damping);                   /* config.h:86 */
int sk_config_get_num_successive_orders_altitudes(Config* config, int* num_altitudes);         /* config.h:87 */
int sk_config_get_successive_orders_altitude_grid_m(Config* config, double* altitude_grid_m);  /* config.h:89 */
int sk_config_set_successive_orders_altitude_grid_m(Config* config, const double* altitude_grid_m, int num_altitudes); /* config.h:91 */
int sk_config_get_occultation_source(Config* config, int* occultation_source);                 /* config.h:95 */
int sk_config_set_occultation_source(Config* config, int occultation_source);                  /* config.h:96 */
int sk_config_get_emission_source(Config* config, int* emission_source);                       /* config.h:97 */
int sk_config_set_emission_source(Config* config, int emission_source);                        /* config.h:98 */
int sk_config_get_los_refraction(Config* config, int* refraction);                             /* config.h:101 */
int sk_config_set_los_refraction(Config* config, int refraction);                              /* config.h:102 */
int sk_config_get_solar_refraction(Config* config, int* refraction);                           /* config.h:103 */
int sk_config_set_solar_refraction(Config* config, int refraction);                            /* config.h:104 */
int sk_config_get_multiple_scatter_refraction(Config* config, int* refraction);                /* config.h:105 */
int sk_config_set_multiple_scatter_refraction(Config* config, int refraction);                 /* config.h:106 */
int sk_config_get_wf_precision(Config* config, int* precision);                                /* config.h:111 */
int sk_config_set_wf_precision(Config* config, int precision);                                 /* config.h:112 */
int sk_config_get_stokes_basis(Config* config, int* basis);                                    /* config.h:115 */
int sk_config_set_stokes_basis(Config* config, int basis);                                     /* config.h:116 */
int sk_config_get_input_validation_mode(Config* config, int* mode);                            /* config.h:119 */
int sk_config_set_input_validation_mode(Config* config, int mode);                             /* config.h:120 */
int sk_config_get_log_level(Config* config, int* log_level);                                   /* config.h:127 */
int sk_config_set_log_level(Config* config, int log_level);                                    /* config.h:128 */
int sk_config_get_num_flux_types(Config* config, int* num_flux_types);                         /* config.h:131 */
int sk_config_get_flux_types(Config* config, int* flux_types);                                 /* config.h:132 */
int sk_config_set_flux_types(Config* config, const int* flux_types, int num_flux_types);       /* config.h:133 */

/* atmosphere.h:111-188 */
int sk_atmosphere_storage_get_derivative_mapping_by_index(AtmosphereStorage* storage, int index, DerivativeMapping** mapping); /* :111 */
int sk_atmosphere_storage_finalize_scattering_derivatives(AtmosphereStorage* storage);         /* atmosphere.h:120 */
int sk_atmosphere_storage_set_zero(AtmosphereStorage* storage);                                /* atmosphere.h:123 */
int sk_atmosphere_apply_delta_m_scaling(Atmosphere* atmosphere, int order);                    /* atmosphere.h:150 (NSTOKES = 1, no scattering-group derivatives) */
int sk_atmosphere_mark_changed(Atmosphere* atmosphere);                                        /* atmosphere.h:152 */
int sk_atmosphere_get_revision(Atmosphere* atmosphere, unsigned long long* revision);          /* atmosphere.h:154 */
int sk_surface_get_num_derivative_mappings(Surface* storage, int* num_mappings);               /* atmosphere.h:183 */
int sk_surface_get_derivative_mapping_name(Surface* storage, int index, const char** name);    /* atmosphere.h:185 */
int sk_surface_set_zero(Surface* storage);                                                     /* atmosphere.h:188 */

/* brdf.h:10-14 -- SnowKokhanovsky (1 argument) and MODIS (isotropic, volumetric, geometric) surfaces: served by the discrete-ordinates source in
 * every geometry type, alone or next to the exact / solar-table single-scatter source; brdf_args of
 * sk_surface_set_brdf is (num_args x num_wavel) column-major, a surface derivative mapping's d_brdf (num_wavel x num_args) column-major
 * (create the mapping after sk_surface_set_brdf).  Other sources / geometries answer -2 with a message in sk_amd_last_error */
BRDF* sk_brdf_create_kokhanovsky(int nstokes);                                                 /* brdf.h:10 */
BRDF* sk_brdf_create_modis(int nstokes);                                                       /* brdf.h:11 */
/* additive: the functional BRDF of the reference's "BRDF surface" known answer (legacy/test_scalar.cpp:806-845), one scale argument; for the parity suite */
BRDF* sk_amd_brdf_create_legacy_test_function(int nstokes);
int sk_brdf_get_num_deriv(BRDF* config, int* num_deriv);                                       /* brdf.h:13 */
int sk_brdf_get_num_args(BRDF* config, int* num_args);                                         /* brdf.h:14 */

/* deriv_mapping.h:10-86 -- interpolators ([ngrid x nout], column-major), log-radiance space and scattering-group mappings
 * (d_legendre [nleg, nloc, nwavel], scat_factor) are applied on the device (cpp/lib/output/outputc.cpp:60-132); emission derivative
 * settings are stored, and a mapping that uses them makes sk_engine_calculate_radiance return -2 */
int sk_deriv_mapping_destroy(DerivativeMapping* mapping);                                      /* deriv_mapping.h:10 */
int sk_deriv_mapping_get_scat_factor(DerivativeMapping* mapping, double** scat_factor);        /* deriv_mapping.h:15 */
int sk_deriv_mapping_get_d_legendre(DerivativeMapping* mapping, double** d_legendre);          /* deriv_mapping.h:17 */
int sk_deriv_mapping_get_d_emission(DerivativeMapping* mapping, double** d_emission);          /* deriv_mapping.h:19 */
int sk_deriv_mapping_get_scat_deriv_index(DerivativeMapping* mapping, int* scat_deriv_index);  /* deriv_mapping.h:21 */
int sk_deriv_mapping_set_scat_deriv_index(DerivativeMapping* mapping, int scat_deriv_index);   /* deriv_mapping.h:23 */
int sk_deriv_mapping_get_num_legendre(DerivativeMapping* mapping, int* num_legendre);          /* deriv_mapping.h:29 */
int sk_deriv_mapping_set_interp_dim(DerivativeMapping* mapping, const char* name);             /* deriv_mapping.h:32 */
int sk_deriv_mapping_set_assign_name(DerivativeMapping* mapping, const char* name);            /* deriv_mapping.h:34 */
int sk_deriv_mapping_set_log_radiance_space(DerivativeMapping* mapping, int log_radiance_space); /* deriv_mapping.h:36 */
int sk_deriv_mapping_get_log_radiance_space(DerivativeMapping* mapping, int* log_radiance_space); /* deriv_mapping.h:38 */
int sk_deriv_mapping_is_scattering_derivative(DerivativeMapping* mapping, int* is_scattering_derivative); /* deriv_mapping.h:40 */
int sk_deriv_mapping_get_assign_name(DerivativeMapping* mapping, const char** name);           /* deriv_mapping.h:46 */
int sk_deriv_mapping_get_interp_dim(DerivativeMapping* mapping, const char** name);            /* deriv_mapping.h:48 */
int sk_deriv_mapping_set_interpolator(DerivativeMapping* mapping, double* interpolator, int dim1, int dim2); /* deriv_mapping.h:50 */
int sk_deriv_mapping_clear_interpolator(DerivativeMapping* mapping);                           /* deriv_mapping.h:52 */
int sk_deriv_mapping_get_interpolator(DerivativeMapping* mapping, double** interpolator, int* dim1, int* dim2); /* deriv_mapping.h:53 */
int sk_surface_deriv_mapping_get_d_emission(SurfaceDerivativeMapping* mapping, double** emission); /* deriv_mapping.h:58 */
int sk_surface_deriv_mapping_get_num_wavel(SurfaceDerivativeMapping* mapping, int* num_wavel); /* deriv_mapping.h:61 */
int sk_surface_deriv_mapping_get_num_brdf_args(SurfaceDerivativeMapping* mapping, int* num_brdf_args); /* deriv_mapping.h:64 */
int sk_surface_deriv_mapping_get_interpolator(SurfaceDerivativeMapping* mapping, double** interpolator, int* dim1, int* dim2); /* :70 */
int sk_surface_deriv_mapping_set_interpolator(SurfaceDerivativeMapping* mapping, double* interpolator, int dim1, int dim2); /* :74 */
int sk_surface_deriv_mapping_get_interp_dim(SurfaceDerivativeMapping* mapping, const char** name); /* deriv_mapping.h:78 */
int sk_surface_deriv_mapping_set_interp_dim(SurfaceDerivativeMapping* mapping, const char* name);  /* deriv_mapping.h:81 */
int sk_surface_deriv_mapping_set_zero(SurfaceDerivativeMapping* mapping);                      /* deriv_mapping.h:84 */
int sk_surface_deriv_mapping_destroy(SurfaceDerivativeMapping* mapping);                       /* deriv_mapping.h:86 */

/* engine.h:20-39 -- the weighting functions of this build are the Jacobian mode; JVP / VJP entry points return -2 */
Engine* sk_engine_create_2d(Config* engine, Geometry2D* geometry, ViewingGeometry* viewing_geometry); /* engine.h:20: NULL */
int sk_engine_supports_linearization(Engine* engine, int mode, int* supported);                /* engine.h:27 */
int sk_engine_linearization_backend(Engine* engine, int mode, int* backend);                   /* engine.h:28 */
int sk_engine_calculate_jvp(Engine* engine, Atmosphere* atmosphere, OutputJVP* output);        /* engine.h:29 */
int sk_engine_initialize_jvp(Engine* engine, Atmosphere* atmosphere, OutputJVP* output);       /* engine.h:31 */
int sk_engine_calculate_jvp_wavelength_thread(Engine* engine, OutputJVP* output, int wavelength, int thread_idx); /* engine.h:33 */
int sk_engine_calculate_vjp(Engine* engine, Atmosphere* atmosphere, OutputVJP* output);        /* engine.h:35 */
int sk_engine_initialize_vjp(Engine* engine, Atmosphere* atmosphere, OutputVJP* output);       /* engine.h:37 */
int sk_engine_calculate_vjp_block_thread(Engine* engine, OutputVJP* output, int wavelength_start, int wavelength_count, int thread_idx); /* :39 */

/* geodetic.h:9-55 -- utility class outside the solver: a handle can be created, every query returns -2 */
Geodetic* sk_geodetic_create(double equatorial_radius, double flattening_factor);             /* geodetic.h:9 */
void sk_geodetic_destroy(Geodetic* geodetic);                                                  /* geodetic.h:12 */
int sk_geodetic_get_altitude(const Geodetic* geodetic, double* altitude);                      /* geodetic.h:14 */
int sk_geodetic_get_latitude(const Geodetic* geodetic, double* latitude);                      /* geodetic.h:15 */
int sk_geodetic_get_longitude(const Geodetic* geodetic, double* longitude);                    /* geodetic.h:16 */
int sk_geodetic_get_location(const Geodetic* geodetic, double* x, double* y, double* z);       /* geodetic.h:17 */
int sk_geodetic_get_local_south(const Geodetic* geodetic, double* x, double* y, double* z);    /* geodetic.h:20 */
int sk_geodetic_get_local_up(const Geodetic* geodetic, double* x, double* y, double* z);       /* geodetic.h:23 */
int sk_geodetic_get_local_west(const Geodetic* geodetic, double* x, double* y, double* z);     /* geodetic.h:26 */
int sk_geodetic_get_altitude_intercepts(const Geodetic* geodetic, double altitude, double observer_x, double observer_y,
                                        double observer_z, double look_vector_x, double look_vector_y, double look_vector_z,
                                        double* x1, double* y1, double* z1, double* x2, double* y2, double* z2); /* geodetic.h:29 */
int sk_geodetic_from_lat_lon_altitude(const Geodetic* geodetic, double latitude, double longitude, double altitude); /* geodetic.h:34 */
int sk_geodetic_from_tangent_altitude(const Geodetic* geodetic, double altitude, double observer_x, double observer_y,
                                      double observer_z, double boresight_x, double boresight_y, double boresight_z,
                                      double* look_vector_x, double* look_vector_y, double* look_vector_z); /* geodetic.h:37 */
int sk_geodetic_from_tangent_point(const Geodetic* geodetic, double observer_x, double observer_y, double observer_z,
                                   double look_vector_x, double look_vector_y, double look_vector_z); /* geodetic.h:45 */
int sk_geodetic_from_xyz(const Geodetic* geodetic, double x, double y, double z);              /* geodetic.h:50 */
int sk_geodetic_is_valid(const Geodetic* geodetic, int* is_valid);                             /* geodetic.h:53 */
int sk_geodetic_get_osculating_spheroid(const Geodetic* geodetic, double* radius, double* offset_x, double* offset_y, double* offset_z); /* :55 */

/* geometry.h:21-47 -- no refraction, no 2D geometry in this build */
int sk_geometry1d_get_refractive_index_ptr(const Geometry1D* geometry, double** refractive_index); /* geometry.h:21 */
Geometry2D* sk_geometry2d_create(double cos_sza, double saa, double earth_radius, const double* altitude_grid_values, int num_altitudes,
                                 const double* horizontal_angle_grid_values, int num_horizontal_locations,
                                 int altitude_interp_method);                                   /* geometry.h:25: NULL */
void sk_geometry2d_destroy(Geometry2D* geometry);                                              /* geometry.h:30 */
int sk_geometry2d_get_location_shape(const Geometry2D* geometry, int* num_horizontal_locations, int* num_altitudes); /* geometry.h:32 */
int sk_geometry2d_get_altitudes(const Geometry2D* geometry, double* altitudes);                /* geometry.h:36 */
int sk_geometry2d_get_horizontal_angles(const Geometry2D* geometry, double* horizontal_angles); /* geometry.h:38 */
int sk_geometry2d_get_refractive_index_ptr(const Geometry2D* geometry, const double** refractive_index); /* geometry.h:41 */
int sk_geometry2d_get_refractive_index_mut_ptr(Geometry2D* geometry, double** refractive_index); /* geometry.h:44 */
int sk_geometry2d_get_location_index(const Geometry2D* geometry, int altitude_index, int horizontal_index, int* location_index); /* :47 */

/* output.h:24-51 */
int sk_output_assign_flux_derivative_memory(OutputC* output, const char* name, double* derivative_mapping, int nrad, int nderiv); /* :24 */
int sk_output_assign_surface_flux_derivative_memory(OutputC* output, const char* name, double* derivative_mapping, int nrad); /* :28 */
OutputJVP* sk_output_jvp_create(double* radiance, double* jvp, int nrad, int nstokes);         /* output.h:35 */
void sk_output_jvp_destroy(OutputJVP* output);                                                 /* output.h:37 */
int sk_output_jvp_assign_derivative_tangent(OutputJVP* output, const char* name, const double* tangent, int nparam); /* output.h:38 */
int sk_output_jvp_assign_surface_tangent(OutputJVP* output, const char* name, const double* tangent, int nparam); /* output.h:40 */
OutputVJP* sk_output_vjp_create(double* radiance, const double* cotangent, int nrad, int nstokes); /* output.h:43 */
void sk_output_vjp_destroy(OutputVJP* output);                                                 /* output.h:45 */
int sk_output_vjp_assign_derivative_gradient(OutputVJP* output, const char* name, double* gradient, int nparam); /* output.h:46 */
int sk_output_vjp_assign_surface_gradient(OutputVJP* output, const char* name, double* gradient, int nparam); /* output.h:49 */
int sk_output_vjp_finalize(OutputVJP* output);                                                 /* output.h:51 */

/* sk_lapack.h:18 -- dense A X = B with partial pivoting, LAPACK dgesv conventions (column major, 1-based ipiv, info) */
int64_t sk_lapack_dgesv(int64_t n, int64_t nrhs, double* a, int64_t lda, int64_t* ipiv, double* b, int64_t ldb);

/* viewing_geometry.h:23-38 -- ray kinds this build cannot trace return -2 */
int sk_viewing_geometry_add_tangent_altitude(ViewingGeometry* geometry, double tangent_altitude_m, double observer_altitude_m,
                                             double horizontal_angle_radians, double viewing_azimuth_radians); /* viewing_geometry.h:23 */
int sk_viewing_geometry_add_solar_angles_observer_location(ViewingGeometry* geometry, double cos_sza, double relative_azimuth_angle,
                                                           double cos_viewing_zenith, double observeraltitude); /* viewing_geometry.h:29 */
int sk_viewing_geometry_add_flux_observer_solar(ViewingGeometry* geometry, double cos_sza, double observeraltitude); /* viewing_geometry.h:33 */
int sk_viewing_geometry_num_flux_observers(ViewingGeometry* geometry, int* num_observers);     /* viewing_geometry.h:38 */


#ifdef __cplusplus
}
#endif
#endif
