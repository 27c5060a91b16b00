// Host-facing interface of the exact single-scatter + successive-orders device driver (see so_engine.hip).
#pragma once
#include <cstddef>
#include <string>

#include "../host/so_geometry.hpp"

namespace skdev {

class SoDeviceEngine {
  public:
    struct Impl;
    SoDeviceEngine(const skso::SoGeometry& g, int num_iterations);
    ~SoDeviceEngine();
    SoDeviceEngine(const SoDeviceEngine&) = delete;
    SoDeviceEngine& operator=(const SoDeviceEngine&) = delete;

    int nlos() const;
    int ngrid() const;
    int nstokes() const;

    // All pointers are DEVICE pointers in the reference storage layout (location fastest):
    //   ext, ssa [nwavel][ngrid]; leg [nwavel][ngrid][nleg (x 4: a1, a2, a3, b1 for NSTOKES = 3)]; albedo / irradiance [nwavel] (irradiance may be null).
    // Outputs: radiance [nwavel][nlos][nstokes] (accumulate != 0: added to what is there), los_od [nwavel][nlos] (may be null).
    // Native weighting functions of the sources without a diffuse field -- exact single scatter, line-of-sight emission, occultation -- (engines
    // without the successive-orders source only): wf_ext / wf_ssa [ngrid][nwavel][nlos][nstokes], wf_alb [nwavel][nlos][nstokes], all three or
    // none; with an emission source also wf_emission [ngrid][nwavel][nlos][nstokes] and wf_surface_emission [nwavel][nlos][nstokes] (either may
    // be null); `accumulate` applies to them too.
    void solve_device(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg, const double* d_albedo,
                      const double* d_irradiance, double* d_radiance, double* d_los_od, void* stream, int accumulate = 0,
                      double* d_wf_ext = nullptr, double* d_wf_ssa = nullptr, double* d_wf_alb = nullptr, double* d_wf_emission = nullptr,
                      double* d_wf_surface_emission = nullptr,
                      // scattering groups of the single-scatter source: ngroups directions d_dleg [group][nwavel][ngrid][nleg (x 4)] of the
                      // Legendre / Greek coefficients -> d_wf_groups [group][ngrid][nwavel][nlos][nstokes] (accumulated like the others)
                      int ngroups = 0, const double* d_dleg = nullptr, double* d_wf_groups = nullptr);
    // MultipleScatterSource::SuccessiveOrders (cpp/lib/successive_orders/fixed_point.h): instead of `num_iterations` legacy passes, solve the fixed
    // point x = S (first + T x) of the diffuse field with damped steps, Type-II Anderson mixing of depth <= 3 and the infinity-norm stop rule
    // |map(x) - x| <= absolute + relative max(|x|, |map(x)|) per wavelength; `num_iterations` is the iteration limit.  Depth 0, damping 1 and
    // zero tolerances reproduce the legacy passes bit for bit.  Call before the first solve_device.
    void set_fixed_point(bool enabled, double relative_tolerance, double absolute_tolerance, double damping, int anderson_depth);
    int last_fixed_point_iterations() const;   // iterations the last block ran (the limit unless every wavelength met the tolerance earlier)
    bool has_successive_orders() const;
    bool has_unlinearised_sources() const;   // successive orders: no weighting functions
    bool has_emission() const;               // line-of-sight emission source (standard / volume_emission_rate)
    // non-Lambertian surface under the exact single scatter of the next solves (kind 0 Lambertian, 1 SnowKokhanovsky, 2 MODIS; d_args [nwavel][nargs]
    // on the device): the direct bounce of a line of sight that ends on the ground uses pi x brdf at its ground geometry.  The albedo weighting
    // function output is then d / d(that value); los_albedo_derivative() [argument][los][nwavel] holds d(value) / d(argument) of the last solve
    void set_surface_brdf(int kind, int nargs, const double* d_args);
    const double* los_albedo_derivative() const;
    // emission_source [nwavel][ngrid] and surface emission [nwavel] (DEVICE pointers, or null) used by the following solve_device calls of an
    // engine built with an emission source
    void set_emission(const double* d_emission, const double* d_surface_emission);

    size_t bytes_per_wavelength() const;
    size_t scratch_bytes() const;
    int last_chunk() const;
    void set_profiling(bool on);
    void reset_profile();
    bool kernel_time(const char* name, double* ms, long* launches) const;
    std::string kernel_names() const;
    // sizes of the diffuse discretisation (for the roofline bookkeeping of bench.py)
    size_t num_incoming_rays() const;
    size_t num_incoming_layers() const;
    size_t num_accumulation_entries() const;
    size_t num_outgoing() const;

  private:
    Impl* impl_;
};

}  // namespace skdev
