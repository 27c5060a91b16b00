// Host-facing interface of the device driver (see disco_engine.hip).
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "../host/geo_tables.hpp"

namespace skdev {

class DeviceEngine {
  public:
    struct Impl;
    struct VecImpl;
    // ss_in_do: the DO source carries the solar single-scatter term (plane-parallel post-processing only)
    explicit DeviceEngine(const skgeo::DiscoGeometry& dg, bool ss_in_do = true);
    ~DeviceEngine();
    DeviceEngine(const DeviceEngine&) = delete;
    DeviceEngine& operator=(const DeviceEngine&) = delete;

    int nlos() const;
    int ngrid() const;
    int nstr() const;

    // All pointers are DEVICE pointers.  Inputs use the reference storage layout (location fastest):
    //   ext, ssa [nwavel][ngrid], leg [nwavel][ngrid][nleg], albedo / irradiance [nwavel] (irradiance may be null).
    // Outputs: radiance [nwavel][nlos]; wf_ext, wf_ssa [ngrid][nwavel][nlos]; wf_alb, los_od [nwavel][nlos];
    // status [nwavel] (error bits, see skd::ERR_*).  Derivative outputs may be null.
    void solve_device(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                      const double* d_albedo, const double* d_irradiance, int derivs, double* d_radiance, double* d_wf_ext,
                      double* d_wf_ssa, double* d_wf_alb, double* d_los_od, int* d_status, void* stream, int max_chunk);

    // delta-M truncation fraction of the (already scaled) optical properties: DEVICE pointer [nwavel][ngrid] used by the following
    // solve_device calls, or null (sk_atmosphere_apply_delta_m_scaling, cpp/lib/atmosphere/atmosphere.cpp:73-200)
    void set_delta_m_fraction(const double* d_f);
    // Wavelength-major output slab for the following solve_device calls (or null): row w of `d_slab` holds, for wavelength w of the call,
    // [radiance | los_od | wf_extinction(q = 0 .. ngrid-1) | wf_ssa(q) | wf_albedo] x nlos doubles -- what one rank of a wavelength-sharded
    // run contributes to the single all-gather; the separate output pointers of solve_device are then ignored (status excepted)
    void set_slab_output(double* d_slab);
    static int slab_fields(int ngrid, int derivs) { return derivs ? 2 * ngrid + 3 : 2; }
    // EmissionSource::discrete_ordinates (thermal source inside the DO equations; plane-parallel post-processing, NSTOKES = 1, radiances only):
    // DEVICE pointers emission_source [nwavel][ngrid] and surface emission [nwavel] (or null) for the following solve_device calls
    void set_thermal_emission(const double* d_emission, const double* d_surface_emission);
    // weighting functions of the thermal pass with respect to the emission profile [ngrid][nwavel][nlos] and the surface emission
    // [nwavel][nlos] (device pointers or null), written by the next solve_device with derivatives
    void set_thermal_wf_outputs(double* d_wf_emission, double* d_wf_surface_emission);
    // Flux observers (plane-parallel post-processing, reference: DOSourcePlaneParallelPostProcessing::compute_flux,
    // source_term/do_source_planeparallel.cpp:221-566): per observer the layer (from the top) that holds it and its position inside it
    // (0 = ceiling, 1 = floor); per solve the output [flux type][wavelength][observer] (device) and the requested types
    // (0 upwelling, 1 downwelling, 2 actinic).  Radiance-type fluxes only (no flux weighting functions).
    void set_flux_observers(const std::vector<int>& layer, const std::vector<double>& fraction);
    void set_flux_output(double* d_flux, const std::vector<int>& types);
    int num_flux_observers() const;
    // Scattering-group weighting functions on the 16-stream matrix-core kernels (d_legendre / scat_factor mappings): the groups of the NEXT
    // solve_device calls -- d_dleg [group][nwavel][ngrid][nleg], d_dfg [group][nwavel][ngrid] (delta-M scaled storage) or null, output
    // d_wf_scat [group][ngrid][nwavel][nlos].  groups_on_matrix_path says whether this engine's geometry and the group count are served here
    // (otherwise solve_device_vector's run-time-n kernels are).
    void set_scattering_groups(int ngroups, const double* d_dleg, double* d_wf_scat, const double* d_dfg);
    bool groups_on_matrix_path(int ngroups) const;
    // non-Lambertian surface of the next solve_device_vector calls (plane-parallel / pseudo-spherical post-processing, run-time-n kernels):
    // kind 0 Lambertian (the albedo argument), 1 SnowKokhanovsky, 2 MODIS; d_args [nwavel][nargs] on the device.  The weighting functions of
    // the arguments go to the albedo output, [nargs][nwavel][los][stokes]
    void set_surface_brdf(int kind, int nargs, const double* d_args);
    void set_forced_azimuth(int n);   // plane-parallel post-processing: solve exactly n azimuth orders (n <= streams)

    // Polarised path (NSTOKES = 3; also NSTOKES = 1 with a stream count the matrix-core path does not cover).
    // d_leg: [nwavel][ngrid][nleg][4] (a1, a2, a3, b1) for NSTOKES = 3; d_radiance: [nwavel][nlos][nstokes].
    bool vector_mode() const { return vector_mode_; }
    int nstokes() const;
    bool plane_parallel() const;
    // With derivs (spherical post-processing only): wf_ext / wf_ssa [ngrid][nwavel][nlos][nstokes], wf_alb [nwavel][nlos][nstokes].
    void solve_device_vector(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg, const double* d_albedo,
                             const double* d_irradiance, int derivs, double* d_radiance, double* d_wf_ext, double* d_wf_ssa, double* d_wf_alb,
                             double* d_los_od, int* d_status, void* stream, int ngroups = 0, const double* d_dleg = nullptr,
                             double* d_wf_scat = nullptr, const double* d_dfg = nullptr);   // scattering groups: d_dleg [G][nwavel][ngrid][nleg (*4)], wf_scat [G][ngrid][nwavel][nlos][S]
    size_t bytes_per_wavelength() const;
    size_t scratch_bytes() const;   // scratch currently allocated (all slots)
    int last_chunk() const;         // wavelengths per launch of the last solve
    int last_slots() const;         // blocks of wavelengths in flight (streams) of the last solve
    void set_profiling(bool on);
    // event pairs around every launch on its own stream, nothing serialised: kernel durations inside the pipelined step
    void set_in_situ_timing(bool on);
    void reset_profile();
    bool kernel_time(const char* name, double* ms, long* launches) const;
    std::string kernel_names() const;

  private:
    void ensure_scratch(int nw, int nslots);
    void ensure_group_scratch(int ngroups);
    void init_vector(const skgeo::DiscoGeometry& dg);
    void ensure_vector_scratch(int nw, bool derivs, int ngroups);
    void destroy_vector();
    Impl* impl_;
    VecImpl* vec_ = nullptr;
    bool vector_mode_ = false;
};

}  // namespace skdev
