// Host-facing interface of the two-stream device driver (see ts_engine.hip).
#pragma once
#include <cstddef>
#include <string>

#include "../host/ts_geometry.hpp"

namespace skdev {

class TsDeviceEngine {
  public:
    struct Impl;
    explicit TsDeviceEngine(const skts::TsGeometry& g);
    ~TsDeviceEngine();
    TsDeviceEngine(const TsDeviceEngine&) = delete;
    TsDeviceEngine& operator=(const TsDeviceEngine&) = delete;

    int nlos() const;
    int ngrid() const;
    bool solar() const;     // multiple_scatter_source = twostream
    bool thermal() const;   // emission_source = twostream

    // All pointers are DEVICE pointers in the reference storage layout (location fastest):
    //   ext, ssa, delta_f, emission [nwavel][ngrid]; leg [nwavel][ngrid][nleg]; albedo / irradiance / surface_emission [nwavel].
    //   delta_f (truncation fraction of already delta-M scaled optics) and irradiance may be null; emission is required by the thermal source.
    struct Inputs {
        int nleg = 0;
        const double *ext = nullptr, *ssa = nullptr, *leg = nullptr, *delta_f = nullptr, *albedo = nullptr, *irradiance = nullptr,
                     *emission = nullptr, *surface_emission = nullptr;
    };
    // radiance [nwavel][nlos] (accumulate != 0: added to what is there, weighting functions too).  Weighting functions, all or none of
    // wf_ext / wf_ssa / wf_b1 / wf_albedo: wf_ext, wf_ssa, wf_b1, wf_emission [ngrid][nwavel][nlos] (b1: the first Legendre coefficient
    // after the delta-M correction), wf_albedo, wf_surface_emission [nwavel][nlos]; the two emission outputs may be null.
    struct Outputs {
        double* radiance = nullptr;
        int accumulate = 0;
        double* los_od = nullptr;   // [nwavel][nlos] optical depth along every line of sight, or null
        double *wf_ext = nullptr, *wf_ssa = nullptr, *wf_b1 = nullptr, *wf_albedo = nullptr, *wf_emission = nullptr, *wf_surface_emission = nullptr;
    };
    void solve_device(int nwavel, const Inputs& in, const Outputs& out, void* stream, int max_chunk = 0);

    size_t bytes_per_wavelength(bool derivs) const;
    size_t scratch_bytes() const;
    int last_chunk() const;
    void set_profiling(bool on);
    void reset_profile();
    bool kernel_time(const char* name, double* ms, long* launches) const;
    std::string kernel_names() const;

  private:
    Impl* impl_;
};

}  // namespace skdev
