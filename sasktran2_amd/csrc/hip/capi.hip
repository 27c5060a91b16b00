// C ABI of sasktran2_amd (see include/sasktran2_amd.h).  Host side only: object model mirroring the reference's
// c_api handles, H2D/D2H staging around DeviceEngine::solve_device and the user derivative mappings
// (reference: OutputC::assign_lane, cpp/lib/output/outputc.cpp:38-236).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the four RCCL calls of the sharded entry point are resolved with dlopen (capi_sharded.inc)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/sasktran2_amd.h"
#include "disco_engine.hpp"
#include "so_engine.hpp"
#include "ts_engine.hpp"
#include "frontend_kernels.hpp"

namespace {
// SASKTRAN2_AMD_HOST_TIMING=1: wall time of the phases of the host entry point (upload, solve, download ...) on stderr
struct HostPhases {
    const bool on = [] { const char* e = std::getenv("SASKTRAN2_AMD_HOST_TIMING"); return e && atoi(e) != 0; }();
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::string log;
    void mark(const char* name) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        log += std::string(name) + " " + std::to_string(std::chrono::duration<double, std::milli>(t1 - t0).count()) + " ms; ";
        t0 = t1;
    }
    ~HostPhases() { if (on) std::fprintf(stderr, "sasktran2_amd host phases: %s\n", log.c_str()); }
};
thread_local std::string g_last_error;
void set_error(const std::string& e) {
    g_last_error = e;
    std::fprintf(stderr, "sasktran2_amd: %s\n", e.c_str());
}
#define HIPC(expr)                                                                               \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e)); \
    } while (0)
}  // namespace

struct Config {
    int num_stokes = 1, ms_source = 3 /* none */, ss_source = 0 /* exact */, num_streams = 16, num_threads = 1,
        batch = 1, nmoments = 16, num_sza = 1, wf_enabled = 1, los_od = 0;
    // settings of the reference's Config that this path only stores (defaults: cpp/lib/config/config.cpp:5-32)
    int threading_model = 0, singlescatter_phasemode = 0, apply_delta_scaling = 0, num_do_forced_azimuth = -1, do_backprop = 0,
        num_do_spherical_iterations = 0, num_hr_spherical_iterations = 50, num_hr_incoming = 110, num_hr_outgoing = 110,
        num_hr_full_incoming_points = -1, initialize_hr_with_do = 0, successive_orders_anderson_depth = 3,
        occultation_source = 1 /* none */, emission_source = 1 /* none */, los_refraction = 0, solar_refraction = 0,
        multiple_scatter_refraction = 0, wf_precision = 0, stokes_basis = 0, input_validation_mode = 0, log_level = 3 /* warn */;
    double successive_orders_relative_tolerance = 1.0e-6, successive_orders_absolute_tolerance = 1.0e-12,
           successive_orders_damping = 1.0;
    std::vector<double> successive_orders_altitude_grid_m;
    std::vector<int> flux_types{0, 1};  // upwelling, downwelling
};
struct Geometry1D {
    double cos_sza, saa, earth_radius;
    std::vector<double> grid;
    int interp, geotype;
};
struct ViewingGeometry {
    std::vector<skgeo::RaySpec> rays;
    std::vector<std::pair<double, double>> flux_observers;   // FluxObserverSolar (viewing_geometry.h:33): (cos_sza, altitude in m)
};
struct DerivativeMapping {
    int nloc = 0, nwavel = 0, nleg = 0;
    std::vector<double> d_ssa, d_extinction;
    // scattering-group / emission arrays (deriv_mapping.h:15-53): allocated when the caller first asks for one of them (ensure_optional) --
    // the 16 Legendre rows of a 10 000-wavelength mapping are 65 MB that a plain extinction / ssa mapping never touches, and every call
    // scans them for a scattering group.  Empty = all zero.
    std::vector<double> scat_factor, d_legendre, d_emission;
    void ensure_optional() {
        const size_t n = size_t(nloc) * nwavel;
        if (scat_factor.size() != n) scat_factor.assign(n, 0.0);
        if (d_emission.size() != n) d_emission.assign(n, 0.0);
        if (d_legendre.size() != n * nleg) d_legendre.assign(n * nleg, 0.0);
    }
    std::vector<double> d_f;   // [nloc * nwavel] derivative of the delta-M truncation fraction along this scattering group (set by the scaling)
    int scat_deriv_index = -1, log_radiance_space = 0;
    std::string interp_dim = "altitude", assign_name;
    double* interpolator = nullptr;
    int interp_dim1 = 0, interp_dim2 = 0;
    // a mapping formed in HBM by the optical-property front end (frontend.inc): the arrays below replace the host vectors above
    bool on_device = false, scatterer = false, has_emission = false;
    const double *dev_d_ssa = nullptr, *dev_d_extinction = nullptr, *dev_scat_factor = nullptr, *dev_d_legendre = nullptr, *dev_d_emission = nullptr,
                 *dev_d_f = nullptr;
    std::vector<double> owned_interpolator;
    const double* src_d_ssa() const { return on_device ? dev_d_ssa : d_ssa.data(); }
    const double* src_d_extinction() const { return on_device ? dev_d_extinction : d_extinction.data(); }
    const double* src_scat_factor() const { return on_device ? dev_scat_factor : scat_factor.data(); }
    const double* src_d_legendre() const { return on_device ? dev_d_legendre : d_legendre.data(); }
    const double* src_d_emission() const { return on_device ? dev_d_emission : d_emission.data(); }
    const double* src_d_f() const { return on_device ? dev_d_f : d_f.data(); }
    bool has_d_f(size_t n) const { return on_device ? dev_d_f != nullptr : d_f.size() == n; }
    bool uses_emission() const {
        if (on_device) return has_emission;
        return std::any_of(d_emission.begin(), d_emission.end(), [](double v) { return v != 0.0; });
    }
};
struct SurfaceDerivativeMapping {
    int nwavel = 0, num_brdf_args = 1;
    std::vector<double> d_brdf, d_emission;
    std::string interp_dim = "dummy";
    double* interpolator = nullptr;
    int interp_dim1 = 0, interp_dim2 = 0;
};
struct AtmosphereStorage {
    int nloc, nwavel, nleg, nstokes;
    double *ssa, *ext, *emission, *leg, *irradiance;
    std::map<std::string, std::unique_ptr<DerivativeMapping>> maps;
    std::vector<std::string> names;
    std::vector<double> delta_f;  // [nloc, nwavel] (location fastest) once sk_atmosphere_apply_delta_m_scaling has run
    int delta_order = -1;
    // storage assembled in HBM by the optical-property front end: ssa / ext / emission / leg are DEVICE pointers (irradiance stays a host
    // array) and the truncation fraction of the delta-M scaling lives in dev_delta_f
    bool on_device = false;
    const double* dev_delta_f = nullptr;
    struct FrontEnd* owner = nullptr;
    bool has_delta_f() const { return on_device ? dev_delta_f != nullptr : !delta_f.empty(); }
    const double* src_delta_f() const { return on_device ? dev_delta_f : delta_f.data(); }
};
struct BRDF {
    int kind = 0;
};
struct Surface {
    int nwavel, nstokes;
    double* emission;
    BRDF* brdf = nullptr;
    double* brdf_args = nullptr;
    std::map<std::string, std::unique_ptr<SurfaceDerivativeMapping>> maps;
};
struct Atmosphere {
    AtmosphereStorage* storage;
    Surface* surface;
    int derivs;
    unsigned long long revision = 0;
};
struct Geometry2D { int unused; };
struct Geodetic { double equatorial_radius, flattening; };
struct OutputJVP { double* radiance; double* jvp; int nrad, nstokes; std::map<std::string, std::pair<const double*, int>> tangents, surface_tangents; };
struct OutputVJP { double* radiance; const double* cotangent; int nrad, nstokes; std::map<std::string, std::pair<double*, int>> gradients, surface_gradients; };
// JVP / VJP requests (sk_engine_calculate_jvp / _vjp): the native derivatives are contracted with the tangents / the cotangent ON THE
// DEVICE by map_user_derivatives -- no user weighting function [nderiv][wavel][los][stokes] is formed anywhere, nothing of that size
// crosses PCIe
struct Contraction {
    const double* cotangent = nullptr;   // host [wavel][los][stokes], VJP
    double* jvp = nullptr;               // host [wavel][los][stokes], JVP (overwritten)
    const std::map<std::string, std::pair<const double*, int>>* tangents = nullptr;
    const std::map<std::string, std::pair<const double*, int>>* surface_tangents = nullptr;
    const std::map<std::string, std::pair<double*, int>>* gradients = nullptr;
    const std::map<std::string, std::pair<double*, int>>* surface_gradients = nullptr;
};
struct OutputC {
    double* radiance;
    int nrad, nstokes;
    struct D { double* mem; int nrad, nstokes, nderiv; };
    std::map<std::string, D> derivs, surface_derivs, flux_derivs, surface_flux_derivs;
    std::vector<double> los_od;
    const Contraction* contract = nullptr;
    double* flux = nullptr;   // [flux type][wavelength][flux observer] (output.h:11), nflux doubles
    int nflux = 0;
};
// grow-only device staging buffers of the host entry point (kept between calls: no hipMalloc / hipFree per spectrum)
struct DeviceCache {
    struct Buf { void* p = nullptr; size_t cap = 0; };
    std::map<std::string, Buf> bufs;
    void* get(const char* name, size_t bytes) {
        Buf& b = bufs[name];
        if (bytes > b.cap) {
            if (b.p) (void)hipFree(b.p);
            b.p = nullptr;
            b.cap = 0;
            if (hipMalloc(&b.p, std::max<size_t>(bytes, 8)) != hipSuccess) throw std::runtime_error(std::string("hipMalloc failed for ") + name);
            b.cap = bytes;
        }
        return b.p;
    }
    ~DeviceCache() { for (auto& kv : bufs) if (kv.second.p) (void)hipFree(kv.second.p); }
};
struct Engine {
    Config config;
    std::unique_ptr<skdev::DeviceEngine> dev;      // discrete-ordinates source (multiple_scatter_source = discrete_ordinates)
    std::unique_ptr<skdev::SoDeviceEngine> so;     // exact single scatter and / or legacy successive orders
    std::unique_ptr<skdev::TsDeviceEngine> ts;     // two-stream source (multiple_scatter_source = twostream, emission_source = twostream)
    int nlos = 0, ngrid = 0, nflux_obs = 0;
    DeviceCache cache;
};

// user weighting function = d_ssa (x) dI/d(ssa) + d_extinction (x) dI/dk  (OutputC::assign_lane, cpp/lib/output/outputc.cpp:38-236),
// evaluated on the device in the output layout [nderiv][wavel][los]; the mapping factors are [wavel][location]
__global__ void __launch_bounds__(256) kern_map_wf(const double* we, const double* ws, const double* f_ssa, const double* f_ext,
                                                   double* dst, int ngrid, int nw, int nlos, const double* wg = nullptr,
                                                   const double* f_scat = nullptr, const double* wem = nullptr, const double* f_em = nullptr) {
    // wg / f_scat: native scattering-group derivatives and the mapping's scat_factor (outputc.cpp:76-90), or null
    const size_t n = size_t(ngrid) * nw * nlos;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t qw = i / nlos;
        const int q = int(qw / nw), w = int(qw % nw);
        double acc = f_ssa[size_t(w) * ngrid + q] * ws[i] + f_ext[size_t(w) * ngrid + q] * we[i];
        if (wg) acc += f_scat[size_t(w) * ngrid + q] * wg[i];
        if (wem) acc += f_em[size_t(w) * ngrid + q] * wem[i];   // d_emission x dI/d(emission) (two-stream thermal source)
        dst[i] = acc;
    }
}
// the same followed by the mapping's interpolator (column major [ngrid x nout], OutputC::assign_lane: row of native derivatives
// times the interpolator) and / or the division by the radiance of a log-radiance-space mapping; dst is [nout][wavel][los]
__global__ void __launch_bounds__(256) kern_map_wf_interp(const double* we, const double* ws, const double* f_ssa, const double* f_ext,
                                                          const double* interp, const double* radiance, int log_space, double* dst,
                                                          int ngrid, int nout, int nw, int nlos, int S = 1, const double* wg = nullptr,
                                                          const double* f_scat = nullptr, const double* wem = nullptr, const double* f_em = nullptr) {
    // nlos counts (line of sight, Stokes component) pairs; a log-radiance mapping divides every component by the intensity
    const size_t n = size_t(nout) * nw * nlos;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t jw = i / nlos;
        const int j = int(jw / nw), w = int(jw % nw), l = int(i % nlos);
        double acc = 0;
        if (interp) {
            for (int q = 0; q < ngrid; ++q) {
                const size_t o = (size_t(q) * nw + w) * nlos + l;
                double v = f_ssa[size_t(w) * ngrid + q] * ws[o] + f_ext[size_t(w) * ngrid + q] * we[o];
                if (wg) v += f_scat[size_t(w) * ngrid + q] * wg[o];
                if (wem) v += f_em[size_t(w) * ngrid + q] * wem[o];
                acc += interp[size_t(j) * ngrid + q] * v;
            }
        } else {
            acc = f_ssa[size_t(w) * ngrid + j] * ws[i] + f_ext[size_t(w) * ngrid + j] * we[i];
            if (wg) acc += f_scat[size_t(w) * ngrid + j] * wg[i];
            if (wem) acc += f_em[size_t(w) * ngrid + j] * wem[i];
        }
        if (log_space) acc /= radiance[size_t(w) * nlos + (l / S) * S];
        dst[i] = acc;
    }
}
// VJP of one mapping at the native level: g[q] = sum over (wavel, los) of cotangent x (d_ssa dI/d(ssa) + d_extinction dI/dk + ...) at grid
// point q (a log-radiance mapping divides the cotangent by the radiance); one block per q, the mapping's interpolator is applied to g
// on the host afterwards (gradient = interpolator^T g)
__global__ void __launch_bounds__(256) kern_vjp_native(const double* we, const double* ws, const double* f_ssa, const double* f_ext, const double* cot,
                                                       const double* radiance, int log_space, int S, double* g, int ngrid, int nw, int nlos,
                                                       const double* wg, const double* f_scat, const double* wem, const double* f_em) {
    __shared__ double red[256];
    const int q = blockIdx.x;
    const size_t n = size_t(nw) * nlos;
    double acc = 0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int w = int(i / nlos), l = int(i % nlos);
        const size_t o = size_t(q) * n + i, fo = size_t(w) * ngrid + q;
        double v = f_ssa[fo] * ws[o] + f_ext[fo] * we[o];
        if (wg) v += f_scat[fo] * wg[o];
        if (wem) v += f_em[fo] * wem[o];
        double c = cot[i];
        if (log_space) c /= radiance[size_t(w) * nlos + (l / S) * S];
        acc += c * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (int(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) g[q] = red[0];
}
// JVP of one mapping: jvp[wavel][los] += sum_q tn[q] x (d_ssa dI/d(ssa) + d_extinction dI/dk + ...)[q][wavel][los], tn = interpolator x tangent
__global__ void __launch_bounds__(256) kern_jvp_native(const double* we, const double* ws, const double* f_ssa, const double* f_ext, const double* tn,
                                                       const double* radiance, int log_space, int S, double* jvp, int ngrid, int nw, int nlos,
                                                       const double* wg, const double* f_scat, const double* wem, const double* f_em) {
    const size_t n = size_t(nw) * nlos;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / nlos), l = int(i % nlos);
        double acc = 0;
        for (int q = 0; q < ngrid; ++q) {
            const double t = tn[q];
            if (t == 0.0) continue;
            const size_t o = size_t(q) * n + i, fo = size_t(w) * ngrid + q;
            double v = f_ssa[fo] * ws[o] + f_ext[fo] * we[o];
            if (wg) v += f_scat[fo] * wg[o];
            if (wem) v += f_em[fo] * wem[o];
            acc += t * v;
        }
        if (log_space) acc /= radiance[size_t(w) * nlos + (l / S) * S];
        jvp[i] += acc;
    }
}
// two-stream source: scattering-group derivative = dI/d(b1) x (d_legendre(1) - 3 d_f / (1 - f)^2)
// (CppTwoStreamSourceAdapter::start_of_ray_source, cpp/lib/sktran_disco/cpp_twostream_source.cpp:3583-3604); wb1 / dst [ngrid][wavel][los]
__global__ void __launch_bounds__(256) kern_ts_scat_group(const double* wb1, const double* dleg, const double* f, const double* df, double* dst,
                                                          int ngrid, int nw, int nlos, int nleg) {
    const size_t n = size_t(ngrid) * nw * nlos;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t qw = i / nlos;
        const int q = int(qw / nw), w = int(qw % nw);
        const size_t o = size_t(w) * ngrid + q;
        double d = nleg > 1 ? dleg[o * nleg + 1] : 0.0;
        if (f && df) d -= 3.0 * df[o] / ((1.0 - f[o]) * (1.0 - f[o]));
        dst[i] = wb1[i] * d;
    }
}
// Host -> device copy of the caller's input arrays.  Page-locked host memory (sk_amd_host_alloc, hipHostMalloc, hipHostRegister) is visible
// to the device: a streaming kernel reads it over the host link and writes HBM -- no copy-engine round trip (measured under PyTorch's HIP
// runtime: the FIRST hipMemcpy of a call waited 14-24 ms for the copy engine to wake up, whatever its size; tools/probes/host_path_phases.py).
// Ordinary (pageable) memory goes through hipMemcpy as before.
__global__ void __launch_bounds__(256) kern_copy_in(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = __builtin_nontemporal_load(&src[i]);
}
static void upload_doubles(double* dst, const double* src, size_t n) {
    if (n == 0) return;
    hipPointerAttribute_t at{};
    if (n >= 4096 && hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
        hipLaunchKernelGGL(kern_copy_in, dim3(unsigned(std::min<size_t>(2048, (n + 255) / 256))), dim3(256), 0, nullptr,
                           static_cast<const double*>(at.devicePointer), dst, n);
        HIPC(hipGetLastError());
        return;   // (stream order: the kernels of the solve follow on the same stream)
    }
    (void)hipGetLastError();   // (an unregistered pointer makes hipPointerGetAttributes fail: that is the pageable case)
    HIPC(hipMemcpy(dst, src, n * 8, hipMemcpyDefault));
}
__global__ void __launch_bounds__(256) kern_transpose_od(const double* od, double* out, int nw, int nlos) {
    const size_t n = size_t(nw) * nlos;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int l = int(i / nw), w = int(i % nw);   // reference layout: column-major [wavel, los]
        out[i] = od[size_t(w) * nlos + l];
    }
}

extern "C" {

const char* sk_amd_last_error(void) { return g_last_error.c_str(); }

#define CFG_RW(name, field)                                             \
    int sk_config_get_##name(Config* c, int* v) {                       \
        if (!c || !v) return -1;                                        \
        *v = c->field;                                                  \
        return 0;                                                       \
    }                                                                   \
    int sk_config_set_##name(Config* c, int v) {                        \
        if (!c) return -1;                                              \
        c->field = v;                                                   \
        return 0;                                                       \
    }
Config* sk_config_create(void) { return new Config(); }
void sk_config_destroy(Config* c) { delete c; }
CFG_RW(num_stokes, num_stokes)
CFG_RW(multiple_scatter_source, ms_source)
CFG_RW(single_scatter_source, ss_source)
CFG_RW(num_streams, num_streams)
CFG_RW(num_threads, num_threads)
CFG_RW(wavelength_batch_size, batch)
CFG_RW(num_singlescatter_moments, nmoments)
CFG_RW(num_do_sza, num_sza)
CFG_RW(wf_enabled, wf_enabled)
CFG_RW(output_los_optical_depth, los_od)

Geometry1D* sk_geometry1d_create(double cos_sza, double saa, double earth_radius, double* grid_values, int n, int interp,
                                 int geotype) {
    if (!grid_values || n < 2) return nullptr;
    auto* g = new Geometry1D{cos_sza, saa, earth_radius, std::vector<double>(grid_values, grid_values + n), interp, geotype};
    return g;
}
void sk_geometry1d_destroy(Geometry1D* g) { delete g; }
int sk_geometry1d_get_num_altitudes(const Geometry1D* g) { return g ? (int)g->grid.size() : -1; }
int sk_geometry1d_get_altitudes(const Geometry1D* g, double* a) {
    if (!g || !a) return -1;
    std::copy(g->grid.begin(), g->grid.end(), a);
    return 0;
}

ViewingGeometry* sk_viewing_geometry_create(void) { return new ViewingGeometry(); }
void sk_viewing_geometry_destroy(ViewingGeometry* v) { delete v; }
void sk_viewing_geometry_add_ground_viewing_solar(ViewingGeometry* v, double cos_sza, double rel_az, double obs_alt,
                                                  double cos_viewing) {
    if (!v) return;
    skgeo::RaySpec r{};
    r.kind = skgeo::RaySpec::ground_viewing_solar;
    r.cos_sza = cos_sza; r.relative_azimuth = rel_az; r.observer_altitude = obs_alt; r.cos_viewing_zenith = cos_viewing;
    v->rays.push_back(r);
}
int sk_viewing_geometry_add_tangent_altitude_solar(ViewingGeometry* v, double tangent_alt, double rel_az, double obs_alt,
                                                   double cos_sza) {
    if (!v) return -1;
    skgeo::RaySpec r{};
    r.kind = skgeo::RaySpec::tangent_altitude_solar;
    r.cos_sza = cos_sza; r.relative_azimuth = rel_az; r.observer_altitude = obs_alt; r.tangent_altitude = tangent_alt;
    v->rays.push_back(r);
    return 0;
}
int sk_viewing_geometry_num_rays(ViewingGeometry* v, int* n) {
    if (!v || !n) return -1;
    *n = (int)v->rays.size();
    return 0;
}

AtmosphereStorage* sk_atmosphere_storage_create(int nloc, int nwavel, int nleg, int nstokes, double* ssa, double* ext,
                                                double* emission, double* leg, double* irradiance) {
    if (!ssa || !ext || !leg || nloc < 2 || nwavel < 1) return nullptr;
    auto* s = new AtmosphereStorage{nloc, nwavel, nleg, nstokes, ssa, ext, emission, leg, irradiance, {}, {}};
    return s;
}
void sk_atmosphere_storage_destroy(AtmosphereStorage* s) {
    if (s && s->owner) return;   // the handle of a device front end (sk_amd_frontend_storage) belongs to it: sk_amd_frontend_destroy frees it
    delete s;
}
int sk_atmosphere_storage_get_derivative_mapping(AtmosphereStorage* s, const char* name, DerivativeMapping** m) {
    if (!s || !name || !m) return -1;
    auto it = s->maps.find(name);
    if (it == s->maps.end()) {
        auto dm = std::make_unique<DerivativeMapping>();
        dm->nloc = s->nloc;
        dm->nwavel = s->nwavel;
        dm->d_ssa.assign(size_t(s->nloc) * s->nwavel, 0.0);
        dm->d_extinction.assign(size_t(s->nloc) * s->nwavel, 0.0);
        dm->nleg = s->nleg;
        dm->assign_name = name;
        it = s->maps.emplace(name, std::move(dm)).first;
        s->names.push_back(name);
    }
    *m = it->second.get();
    return 0;
}
int sk_atmosphere_storage_get_num_derivative_mappings(AtmosphereStorage* s, int* n) {
    if (!s || !n) return -1;
    *n = (int)s->names.size();
    return 0;
}
int sk_atmosphere_storage_get_derivative_mapping_name(AtmosphereStorage* s, int i, const char** name) {
    if (!s || !name || i < 0 || i >= (int)s->names.size()) return -1;
    *name = s->names[i].c_str();
    return 0;
}
// a mapping the front end formed in HBM has no host arrays to hand out (sk_amd_frontend_download copies them for inspection)
#define SK_HOST_MAPPING(m) do { if ((m)->on_device) { set_error("this derivative mapping lives on the device (sk_amd_frontend_download reads it)"); return -2; } } while (0)
int sk_deriv_mapping_get_d_ssa(DerivativeMapping* m, double** p) { if (!m || !p) return -1; SK_HOST_MAPPING(m); *p = m->d_ssa.data(); return 0; }
int sk_deriv_mapping_get_d_extinction(DerivativeMapping* m, double** p) { if (!m || !p) return -1; SK_HOST_MAPPING(m); *p = m->d_extinction.data(); return 0; }
int sk_deriv_mapping_get_num_location(DerivativeMapping* m, int* n) { if (!m || !n) return -1; *n = m->nloc; return 0; }
int sk_deriv_mapping_get_num_wavel(DerivativeMapping* m, int* n) { if (!m || !n) return -1; *n = m->nwavel; return 0; }
int sk_deriv_mapping_get_num_output(DerivativeMapping* m, int* n) { if (!m || !n) return -1; *n = m->nloc; return 0; }
int sk_deriv_mapping_set_zero(DerivativeMapping* m) {
    if (!m) return -1;
    SK_HOST_MAPPING(m);
    // every native array of the mapping, as DerivativeMapping::set_zero (cpp/lib/derivative_mapping/derivative_mapping.cpp:61-77),
    // and the truncation-fraction derivative the delta-M scaling left behind
    for (std::vector<double>* v : {&m->d_ssa, &m->d_extinction, &m->scat_factor, &m->d_legendre, &m->d_emission, &m->d_f})
        std::fill(v->begin(), v->end(), 0.0);
    return 0;
}
BRDF* sk_brdf_create_lambertian(int) { return new BRDF(); }
void sk_brdf_destroy(BRDF* b) { delete b; }
Surface* sk_surface_create(int nwavel, int nstokes, double* emission) { return new Surface{nwavel, nstokes, emission}; }
void sk_surface_destroy(Surface* s) { delete s; }
int sk_surface_set_brdf(Surface* s, BRDF* b, double* args) {
    if (!s || !b || !args) return -1;
    s->brdf = b;
    s->brdf_args = args;
    return 0;
}
int sk_surface_get_derivative_mapping(Surface* s, const char* name, SurfaceDerivativeMapping** m) {
    if (!s || !name || !m) return -1;
    auto it = s->maps.find(name);
    if (it == s->maps.end()) {
        auto dm = std::make_unique<SurfaceDerivativeMapping>();
        dm->nwavel = s->nwavel;
        // (num_wavel x num_brdf_args, column-major: cpp/include/sasktran2/derivative_mapping.h, surface.h:556-561 -- sized by the BRDF set NOW)
        dm->num_brdf_args = (s->brdf && s->brdf->kind == 2) ? 3 : 1;
        dm->d_brdf.assign(size_t(s->nwavel) * dm->num_brdf_args, 0.0);
        dm->d_emission.assign(s->nwavel, 0.0);
        it = s->maps.emplace(name, std::move(dm)).first;
    }
    *m = it->second.get();
    return 0;
}
int sk_surface_deriv_mapping_get_d_brdf(SurfaceDerivativeMapping* m, double** p) { if (!m || !p) return -1; *p = m->d_brdf.data(); return 0; }
Atmosphere* sk_atmosphere_create(AtmosphereStorage* st, Surface* su, int derivs, int) {
    if (!st || !su) return nullptr;
    return new Atmosphere{st, su, derivs};
}
void sk_atmosphere_destroy(Atmosphere* a) { delete a; }

OutputC* sk_output_create(double* radiance, int nrad, int nstokes, double* flux, int nflux) {
    if (!radiance) return nullptr;
    OutputC* o = new OutputC{radiance, nrad, nstokes, {}, {}, {}};
    o->flux = nflux > 0 ? flux : nullptr;
    o->nflux = flux ? std::max(nflux, 0) : 0;
    return o;
}
void sk_output_destroy(OutputC* o) { delete o; }
int sk_output_assign_derivative_memory(OutputC* o, const char* name, double* mem, int nrad, int nstokes, int nderiv) {
    if (!o || !name || !mem) return -1;
    o->derivs[name] = {mem, nrad, nstokes, nderiv};
    return 0;
}
int sk_output_assign_surface_derivative_memory(OutputC* o, const char* name, double* mem, int nrad, int nstokes) {
    if (!o || !name || !mem) return -1;
    o->surface_derivs[name] = {mem, nrad, nstokes, 1};
    return 0;
}
int sk_output_get_los_optical_depth(OutputC* o, double** od) {
    if (!o || !od) return -1;
    *od = o->los_od.data();
    return 0;
}

Engine* sk_engine_create(Config* c, Geometry1D* g, ViewingGeometry* v) {
    if (!c || !g || !v) return nullptr;
    try {
        if (c->num_stokes != 1 && c->num_stokes != 3) throw std::runtime_error("NSTOKES must be 1 or 3");
        if (c->stokes_basis < 0 || c->stokes_basis > 2) throw std::runtime_error("stokes_basis must be standard (0), solar (1) or observer (2)");
        // sources of this build (cpp/include/sasktran2/config.h:41-71): multiple scatter discrete_ordinates (0), successive_orders_legacy (1),
        // twostream (2) or none (3); single scatter exact (0), solartable (1), discrete_ordinates (2) or none (3)
        if (c->ms_source < 0 || c->ms_source > 4)
            throw std::runtime_error("multiple_scatter_source must be discrete_ordinates (0), successive_orders_legacy (1), twostream (2), none (3) or "
                                     "successive_orders (4)");
        // successive_orders (4, cpp/include/sasktran2/config.h:70): the reference's fixed-point / Anderson driver (successive_orders_* keys) on the
        // diffuse-point discretisation of the legacy source (Lebedev spheres per layer mid point); an explicit successive_orders_altitude_grid_m
        // (its own node grid) is not part of this build
        const bool so_new = c->ms_source == 4;
        if (so_new && !c->successive_orders_altitude_grid_m.empty())
            throw std::runtime_error("successive_orders_altitude_grid_m is not part of this build (the diffuse points sit at the layer mid points)");
        const bool two_stream = c->ms_source == 2 || c->emission_source == 4;
        if (c->ss_source < 0 || c->ss_source > 3)
            throw std::runtime_error("single_scatter_source must be exact (0), solartable (1), discrete_ordinates (2) or none (3)");
        const bool ss_los = c->ss_source == 0 || c->ss_source == 1;   // single scatter integrated along the lines of sight (exact or tabulated solar transmission)
        if (c->ss_source == 2 && c->ms_source != 0) throw std::runtime_error("single_scatter_source = discrete_ordinates needs multiple_scatter_source = discrete_ordinates");
        // do_backprop selects the reverse-mode linearisation of the reference's discrete-ordinates source; the two-stream source is
        // linearised in reverse mode whatever the flag says (the reference's two-stream tests all set it)
        // num_do_forced_azimuth: the plane-parallel post-processing of the discrete-ordinates source honours it (do_source_planeparallel.cpp:55-60)
        const bool forced_azimuth_ok = c->num_do_forced_azimuth <= 0 || (c->ms_source == 0 && g->geotype != 2);
        if (c->los_refraction || c->solar_refraction || c->multiple_scatter_refraction ||
            !forced_azimuth_ok || c->num_do_spherical_iterations > 0 || (c->do_backprop && !two_stream))
            throw std::runtime_error("refraction, a forced azimuth count outside plane-parallel discrete ordinates, DO spherical iterations and backprop are not part of this build");
        // EmissionSource { standard = 0, none = 1, discrete_ordinates = 2, volume_emission_rate = 3, twostream = 4 }, OccultationSource { standard = 0, none = 1 }
        if (c->emission_source < 0 || c->emission_source > 4)
            throw std::runtime_error("emission_source must be standard (0), none (1), discrete_ordinates (2), volume_emission_rate (3) or twostream (4)");
        if (two_stream) {
            // the reference builds the two-stream sources for NSTOKES = 1 only (cpp/lib/engine/engine.cpp:264-276, 384-397)
            if (c->num_stokes != 1) throw std::runtime_error("the two-stream source is only implemented for NSTOKES = 1");
            if (c->ms_source == 0 || c->ms_source == 1 || c->ms_source == 4)
                throw std::runtime_error("emission_source = twostream next to a discrete-ordinates or successive-orders multiple-scatter source is not part of this build");
        }
        if (c->emission_source == 2 && (c->ms_source != 0 || c->num_stokes != 1 || g->geotype == 2))
            throw std::runtime_error("emission_source = discrete_ordinates needs multiple_scatter_source = discrete_ordinates, NSTOKES = 1 and plane-parallel / "
                                     "pseudo-spherical geometry (the reference's thermal post-processing exists for the plane-parallel source only)");
        if (c->occultation_source != 0 && c->occultation_source != 1) throw std::runtime_error("occultation_source must be standard (0) or none (1)");
        const int emission = c->emission_source == 0 ? 1 : (c->emission_source == 3 ? 2 : 0);
        const bool occultation = c->occultation_source == 0;
        skgeo::Geometry1D geo(g->cos_sza, g->saa, g->earth_radius, g->grid, (skgeo::Interpolation)g->interp,
                              (skgeo::GeometryType)g->geotype);
        auto e = std::make_unique<Engine>();
        e->config = *c;
        e->nlos = (int)v->rays.size();
        e->ngrid = (int)g->grid.size();
        if (c->ms_source == 0) {
            skgeo::DiscoGeometry dg;
            skgeo::build_disco_geometry(geo, v->rays, c->num_streams, c->num_sza, dg, c->num_stokes, c->stokes_basis);
            e->dev = std::make_unique<skdev::DeviceEngine>(dg, c->ss_source == 2);
            e->nlos = dg.nlos;
            e->ngrid = dg.ngrid;
            // (any even stream count: the run-time-n kernels carry the thermal pass too; NSTOKES = 1 is checked above)
            if (ss_los && !dg.spherical_pp)
                throw std::runtime_error("single_scatter_source = exact with discrete ordinates needs spherical geometry (in plane-parallel "
                                         "geometry the discrete-ordinates source carries the single scatter itself)");
            if (c->num_do_forced_azimuth > 0) {
                if (e->dev->vector_mode()) throw std::runtime_error("num_do_forced_azimuth needs a stream count of the matrix-core path (2, 4, 8, 16 or 32) and NSTOKES = 1");
                e->dev->set_forced_azimuth(c->num_do_forced_azimuth);
            }
            if (!v->flux_observers.empty()) {
                // DOSourcePlaneParallelPostProcessing::compute_flux (do_source_planeparallel.cpp:221-566): the layer that holds the observer
                // (OpticalLayerArray::layerAtAltitude: the first layer from the top whose floor is not above it) and the fraction of the
                // layer above it
                if (dg.spherical_pp) throw std::runtime_error("flux observers need plane-parallel or pseudo-spherical geometry (the reference computes fluxes in its plane-parallel post-processing only)");
                if (e->dev->vector_mode()) throw std::runtime_error("flux observers need a stream count of the matrix-core path (2, 4, 8, 16 or 32) and NSTOKES = 1");
                std::vector<int> layer;
                std::vector<double> frac;
                for (const auto& fo : v->flux_observers) {
                    const double alt = fo.second;
                    int p = dg.nlyr - 1;
                    for (int r = 0; r < dg.nlyr; ++r)
                        if (dg.floor_h[r] <= alt) { p = r; break; }
                    layer.push_back(p);
                    frac.push_back((dg.ceil_h[p] - alt) / (dg.ceil_h[p] - dg.floor_h[p]));
                }
                e->dev->set_flux_observers(layer, frac);
                e->nflux_obs = (int)layer.size();
            }
        } else if (!v->flux_observers.empty()) {
            throw std::runtime_error("flux observers need multiple_scatter_source = discrete_ordinates");
        }
        if (two_stream) {
            skts::Settings cfg;
            cfg.num_sza = c->num_sza;
            cfg.solar = c->ms_source == 2;
            cfg.thermal = c->emission_source == 4;
            skts::TsGeometry tg;
            skts::build_ts_geometry(geo, v->rays, cfg, tg);
            e->ts = std::make_unique<skdev::TsDeviceEngine>(tg);
        }
        if (ss_los || c->ms_source == 1 || so_new || emission != 0 || occultation) {
            if (g->geotype != 2) throw std::runtime_error("the exact single-scatter, successive-orders, emission and occultation sources need spherical geometry");
            if ((c->ms_source == 1 || so_new) && (c->initialize_hr_with_do || c->num_hr_full_incoming_points > 0))
                throw std::runtime_error("initialize_hr_with_do and num_hr_full_incoming_points are not part of this build");
            skso::Settings cfg;
            cfg.nstokes = c->num_stokes;
            cfg.num_ss_moments = c->nmoments;
            cfg.num_ms_moments = c->num_streams;
            cfg.num_incoming = c->num_hr_incoming;
            cfg.num_outgoing = c->num_hr_outgoing;
            cfg.num_sza = c->num_sza;
            cfg.single_scatter = ss_los;
            cfg.solar_table = c->ss_source == 1;
            cfg.multiple_scatter = c->ms_source == 1 || so_new;
            cfg.emission = emission;
            cfg.occultation = occultation;
            skso::SoGeometry sg;
            skso::build_so_geometry(geo, v->rays, cfg, sg, c->stokes_basis);
            e->so = std::make_unique<skdev::SoDeviceEngine>(sg, c->num_hr_spherical_iterations);
            if (so_new)
                e->so->set_fixed_point(true, c->successive_orders_relative_tolerance, c->successive_orders_absolute_tolerance, c->successive_orders_damping,
                                       c->successive_orders_anderson_depth);
        }
        if (!e->dev && !e->so && !e->ts) throw std::runtime_error("no source term configured (single_scatter_source and multiple_scatter_source are both none)");
        return e.release();
    } catch (const std::exception& ex) {
        set_error(std::string("sk_engine_create: ") + ex.what());
        return nullptr;
    }
}
void sk_engine_destroy(Engine* e) { delete e; }
int sk_openmp_support_enabled(void) { return 0; }
int sk_engine_effective_wavelength_batch_size(Engine* e, int) { return e ? 1 : -1; }
int sk_engine_calculate_radiance_block_thread(Engine* e, OutputC* o, int, int, int) {
    // the whole spectrum is solved on the GPU inside sk_engine_calculate_radiance (also with only_initialize=1),
    // so the per-block entry point of the Rayon scheduler has nothing left to do.
    return (e && o) ? 0 : -1;
}

// native weighting functions (device, [ngrid][wavel][los][S] / [wavel][los][S]) -> the caller's derivative mappings
static bool is_scattering_mapping(const DerivativeMapping& dm) {
    if (dm.on_device) return dm.scatterer;
    if (dm.scat_deriv_index >= 0) return true;
    for (double v : dm.d_legendre) if (v != 0.0) return true;
    for (double v : dm.scat_factor) if (v != 0.0) return true;
    return false;
}

// d_wg: native scattering-group derivatives [group][ngrid][wavel][los][S] with groups in the order of `group_of` (mapping name -> group)
// d_wem / d_wsem: native derivatives with respect to the emission profile [ngrid][wavel][los] and the surface emission [wavel][los]
// (two-stream thermal source), applied through the mappings' d_emission
static void map_user_derivatives(Engine* e, const AtmosphereStorage& st, const Surface& su, OutputC* out, const double* d_we, const double* d_ws,
                                 const double* d_wa, const double* d_rad, int S, const double* d_wg = nullptr,
                                 const std::map<std::string, int>* group_of = nullptr, const double* d_wem = nullptr, const double* d_wsem = nullptr) {
    DeviceCache& C = e->cache;
    const int nw = st.nwavel, ngrid = e->ngrid, nlos = e->nlos * S;   // (line of sight, Stokes component) pairs
    const size_t nloc_w = size_t(ngrid) * nw, nt = size_t(nw) * nlos;
    {
            double* d_fs = (double*)C.get("map_d_ssa", nloc_w * 8);
            double* d_fk = (double*)C.get("map_d_extinction", nloc_w * 8);
            const Contraction* contract = out->contract;
            double *d_cot = nullptr, *d_jvp = nullptr;
            if (contract && contract->gradients) {
                d_cot = (double*)C.get("vjp_cotangent", nt * 8);
                HIPC(hipMemcpy(d_cot, contract->cotangent, nt * 8, hipMemcpyHostToDevice));
            }
            if (contract && contract->tangents) {
                d_jvp = (double*)C.get("jvp_product", nt * 8);
                HIPC(hipMemset(d_jvp, 0, nt * 8));
            }
            for (auto& kv : out->derivs) {
                auto it = st.maps.find(kv.first);
                if (it == st.maps.end()) throw std::runtime_error("no derivative mapping named " + kv.first);
                const DerivativeMapping& m = *it->second;
                const int nout = m.interpolator ? m.interp_dim2 : ngrid;
                if (m.interpolator && m.interp_dim1 != ngrid) throw std::runtime_error("interpolator of " + kv.first + " does not have one row per grid location");
                if (kv.second.nderiv != nout || kv.second.nrad * kv.second.nstokes != nw * nlos) throw std::runtime_error("derivative memory size mismatch for " + kv.first);
                HIPC(hipMemcpy(d_fs, m.src_d_ssa(), nloc_w * 8, hipMemcpyDefault));
                HIPC(hipMemcpy(d_fk, m.src_d_extinction(), nloc_w * 8, hipMemcpyDefault));
                double* d_dst = contract ? nullptr : (double*)C.get("wf_user", size_t(nout) * nt * 8);
                const double* d_wgm = nullptr;
                double* d_fg = nullptr;
                if (group_of && group_of->count(kv.first)) {
                    d_wgm = d_wg + size_t(group_of->at(kv.first)) * ngrid * nt;
                    d_fg = (double*)C.get("map_scat_factor", nloc_w * 8);
                    HIPC(hipMemcpy(d_fg, m.src_scat_factor(), nloc_w * 8, hipMemcpyDefault));
                }
                const double* d_wemm = nullptr;
                double* d_fem = nullptr;
                if (d_wem && m.uses_emission()) {
                    d_wemm = d_wem;
                    d_fem = (double*)C.get("map_d_emission", nloc_w * 8);
                    HIPC(hipMemcpy(d_fem, m.src_d_emission(), nloc_w * 8, hipMemcpyDefault));
                }
                if (contract && contract->gradients) {
                    // VJP: native gradient on the device, the interpolator on the host
                    double* d_g = (double*)C.get("vjp_native", size_t(ngrid) * 8);
                    hipLaunchKernelGGL(kern_vjp_native, dim3(ngrid), dim3(256), 0, nullptr, d_we, d_ws, d_fs, d_fk, d_cot, d_rad, m.log_radiance_space, S, d_g,
                                       ngrid, nw, nlos, d_wgm, d_fg, d_wemm, d_fem);
                    HIPC(hipGetLastError());
                    std::vector<double> gn(ngrid);
                    HIPC(hipMemcpy(gn.data(), d_g, size_t(ngrid) * 8, hipMemcpyDeviceToHost));
                    double* dst = contract->gradients->at(kv.first).first;
                    for (int j = 0; j < nout; ++j) {
                        if (!m.interpolator) { dst[j] = gn[j]; continue; }
                        double acc = 0;
                        for (int q = 0; q < ngrid; ++q) acc += m.interpolator[size_t(j) * ngrid + q] * gn[q];
                        dst[j] = acc;
                    }
                    continue;
                }
                if (contract && contract->tangents) {
                    const double* t = contract->tangents->at(kv.first).first;
                    std::vector<double> tn(ngrid, 0.0);
                    for (int j = 0; j < nout; ++j) {
                        if (!m.interpolator) { tn[j] = t[j]; continue; }
                        for (int q = 0; q < ngrid; ++q) tn[q] += m.interpolator[size_t(j) * ngrid + q] * t[j];
                    }
                    double* d_tn = (double*)C.get("jvp_tangent", size_t(ngrid) * 8);
                    HIPC(hipMemcpy(d_tn, tn.data(), size_t(ngrid) * 8, hipMemcpyHostToDevice));
                    hipLaunchKernelGGL(kern_jvp_native, dim3(1024), dim3(256), 0, nullptr, d_we, d_ws, d_fs, d_fk, d_tn, d_rad, m.log_radiance_space, S, d_jvp,
                                       ngrid, nw, nlos, d_wgm, d_fg, d_wemm, d_fem);
                    HIPC(hipGetLastError());
                    continue;
                }
                if (!m.interpolator && !m.log_radiance_space) {
                    hipLaunchKernelGGL(kern_map_wf, dim3(4096), dim3(256), 0, nullptr, d_we, d_ws, d_fs, d_fk, d_dst, ngrid, nw, nlos, d_wgm, d_fg, d_wemm, d_fem);
                } else {
                    double* d_interp = nullptr;
                    if (m.interpolator) {
                        d_interp = (double*)C.get("interpolator", size_t(ngrid) * nout * 8);
                        HIPC(hipMemcpy(d_interp, m.interpolator, size_t(ngrid) * nout * 8, hipMemcpyDefault));
                    }
                    hipLaunchKernelGGL(kern_map_wf_interp, dim3(4096), dim3(256), 0, nullptr, d_we, d_ws, d_fs, d_fk, d_interp, d_rad,
                                       m.log_radiance_space, d_dst, ngrid, nout, nw, nlos, S, d_wgm, d_fg, d_wemm, d_fem);
                }
                HIPC(hipGetLastError());
                HIPC(hipMemcpy(kv.second.mem, d_dst, size_t(nout) * nt * 8, hipMemcpyDeviceToHost));  // C order [nderiv][wavel][los]
            }
            if (contract && contract->tangents) HIPC(hipMemcpy(contract->jvp, d_jvp, nt * 8, hipMemcpyDeviceToHost));
            if (!out->surface_derivs.empty()) {
                // weighting functions of the BRDF arguments: [argument][wavel][los (, stokes)]; user derivative = sum_a d_brdf(wavel, a) dI / d arg_a
                // (OutputC::assign_lane, cpp/lib/output/outputc.cpp:165-185)
                const int nargs = (su.brdf && su.brdf->kind == 2) ? 3 : 1;
                std::vector<double> wa(nt * nargs), wsem;
                HIPC(hipMemcpy(wa.data(), d_wa, nt * nargs * 8, hipMemcpyDeviceToHost));
                if (d_wsem) {
                    wsem.resize(nt);
                    HIPC(hipMemcpy(wsem.data(), d_wsem, nt * 8, hipMemcpyDeviceToHost));
                }
                for (auto& kv : out->surface_derivs) {
                    auto it = su.maps.find(kv.first);
                    if (it == su.maps.end()) throw std::runtime_error("no surface derivative mapping named " + kv.first);
                    const SurfaceDerivativeMapping& m = *it->second;
                    if (m.d_brdf.size() != size_t(nw) * nargs) throw std::runtime_error("surface derivative mapping " + kv.first + " was created for another BRDF (argument count)");
                    for (int w = 0; w < nw; ++w) {
                        double gsum = 0;
                        for (int l = 0; l < nlos; ++l) {
                            double v = 0;
                            for (int a = 0; a < nargs; ++a) v += wa[size_t(a) * nt + size_t(w) * nlos + l] * m.d_brdf[size_t(a) * nw + w];
                            if (d_wsem) v += wsem[size_t(w) * nlos + l] * m.d_emission[w];
                            if (!contract) kv.second.mem[size_t(w) * nlos + l] = v;
                            else if (contract->surface_tangents) contract->jvp[size_t(w) * nlos + l] += v * contract->surface_tangents->at(kv.first).first[w];
                            else gsum += v * contract->cotangent[size_t(w) * nlos + l];
                        }
                        if (contract && contract->surface_gradients) contract->surface_gradients->at(kv.first).first[w] = gsum;
                    }
                }
            }
    }
}

// scattering groups of a request: one per requested mapping that carries d_legendre / scat_factor, in the order of the map
static std::map<std::string, int> scattering_groups_of(const AtmosphereStorage& st, const OutputC* out) {
    std::map<std::string, int> group_of;
    for (auto& kv : out->derivs) {
        auto it = st.maps.find(kv.first);
        if (it != st.maps.end() && is_scattering_mapping(*it->second)) {
            const int gidx = (int)group_of.size();
            group_of[kv.first] = gidx;
        }
    }
    return group_of;
}
// their d_legendre arrays side by side on the device: [group][nwavel][ngrid][nleg] (what the single-scatter source differentiates along)
static double* upload_group_directions(Engine* e, const AtmosphereStorage& st, const std::map<std::string, int>& group_of, const char* slot) {
    const size_t per_g = size_t(st.nloc) * st.nwavel * st.nleg;
    double* d = (double*)e->cache.get(slot, std::max<size_t>(group_of.size(), 1) * per_g * 8);
    for (auto& kv : group_of) HIPC(hipMemcpy(d + size_t(kv.second) * per_g, st.maps.at(kv.first)->src_d_legendre(), per_g * 8, hipMemcpyDefault));
    return d;
}

// NSTOKES = 3 (and NSTOKES = 1 stream counts outside the matrix-core path): radiances [wavel][los][stokes], LOS optical depths and
// weighting functions [nderiv][wavel][los][stokes].
static void upload_emission(Engine* e, const AtmosphereStorage& st, const Surface& su);
// wf of BRDF argument a [a][wavel][los][S] += (single scatter's d I / d(pi x brdf of the ray) [wavel][los][S]) x d(pi x brdf) / d(argument) [a][los][wavel]
__global__ void kern_brdf_ss_wf(const double* wa_ss, const double* dalb, double* wa, int nw, int nlos, int S, int nargs) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = size_t(nw) * nlos * S;
    if (tid >= nt) return;
    const int w = int(tid / (size_t(nlos) * S)), l = int((tid / S) % nlos);
    for (int a = 0; a < nargs; ++a) wa[size_t(a) * nt + tid] += wa_ss[tid] * dalb[(size_t(a) * nlos + l) * nw + w];
}
static int calculate_radiance_vector(Engine* e, Atmosphere* atm, OutputC* out) {
    const AtmosphereStorage& st = *atm->storage;
    const Surface& su = *atm->surface;
    const int S = e->config.num_stokes;
    const int derivs = atm->derivs && e->config.wf_enabled && (!out->derivs.empty() || !out->surface_derivs.empty());
    if (!out->flux_derivs.empty() || !out->surface_flux_derivs.empty()) return -2;
    const int nw = st.nwavel, nlos = e->nlos, ngrid = e->ngrid;
    if (out->nrad != nw * nlos) throw std::runtime_error("output radiance size mismatch");
    // st.nleg is the first dimension of leg_coeff as the caller passes it (nphase_moments, cpp/c_api/atmosphere.cpp:18-19): for
    // NSTOKES = 3 that is 4 numbers (a1, a2, a3, b1) per Legendre order
    const size_t nloc_w = size_t(ngrid) * nw, nt = size_t(nw) * nlos, per = (S == 3) ? 4 : 1;
    if (st.nleg % int(per) != 0) throw std::runtime_error("leg_coeff must hold four coefficients per Legendre order for NSTOKES = 3");
    const int norder = st.nleg / int(per);
    DeviceCache& C = e->cache;
    double* d_ext = (double*)C.get("ext", nloc_w * 8);
    double* d_ssa = (double*)C.get("ssa", nloc_w * 8);
    double* d_leg = (double*)C.get("leg", nloc_w * st.nleg * 8);
    double* d_alb = (double*)C.get("albedo", size_t(nw) * 8);
    double* d_rad = (double*)C.get("radiance", nt * S * 8);
    double* d_od = (double*)C.get("los_od", nt * 8);
    double* d_odt = (double*)C.get("los_od_t", nt * 8);
    int* d_status = (int*)C.get("status", size_t(nw) * 4);
    double *d_irr = nullptr, *d_we = nullptr, *d_ws = nullptr, *d_wa = nullptr, *d_wem = nullptr, *d_wsem = nullptr;
    HIPC(hipMemcpy(d_ext, st.ext, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_ssa, st.ssa, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_leg, st.leg, nloc_w * st.nleg * 8, hipMemcpyDefault));
    // surface: the Lambertian albedo, or the arguments of a non-Lambertian BRDF [nwavel][nargs] (brdf_args is (nargs x nwavel) column-major,
    // cpp/include/sasktran2/atmosphere/surface.h:366-372)
    const int brdf_kind = su.brdf ? su.brdf->kind : 0, brdf_nargs = brdf_kind == 2 ? 3 : 1;
    if (brdf_kind == 0) {
        HIPC(hipMemcpy(d_alb, su.brdf_args, size_t(nw) * 8, hipMemcpyDefault));
    } else {
        HIPC(hipMemset(d_alb, 0, size_t(nw) * 8));
        double* d_bargs = (double*)C.get("brdf_args", size_t(nw) * brdf_nargs * 8);
        HIPC(hipMemcpy(d_bargs, su.brdf_args, size_t(nw) * brdf_nargs * 8, hipMemcpyDefault));
        e->dev->set_surface_brdf(brdf_kind, brdf_nargs, d_bargs);
    }
    struct BrdfReset { skdev::DeviceEngine* d; ~BrdfReset() { d->set_surface_brdf(0, 1, nullptr); } } brdf_reset{e->dev.get()};
    if (st.irradiance) {
        d_irr = (double*)C.get("irradiance", size_t(nw) * 8);
        HIPC(hipMemcpy(d_irr, st.irradiance, size_t(nw) * 8, hipMemcpyDefault));
    }
    // scattering derivative groups: one per requested mapping that carries d_legendre / scat_factor (finalize_scattering_derivatives,
    // cpp/include/sasktran2/atmosphere/grid_storage.h:190-214)
    std::map<std::string, int> group_of;
    double *d_dleg = nullptr, *d_wg = nullptr, *d_dfg = nullptr;
    if (derivs) {
        d_we = (double*)C.get("wf_ext", size_t(ngrid) * nt * S * 8);
        d_ws = (double*)C.get("wf_ssa", size_t(ngrid) * nt * S * 8);
        d_wa = (double*)C.get("wf_alb", nt * S * brdf_nargs * 8);
        for (auto& kv : out->derivs) {
            auto it = st.maps.find(kv.first);
            if (it != st.maps.end() && is_scattering_mapping(*it->second)) {
                const int gidx = (int)group_of.size();
                group_of[kv.first] = gidx;
            }
        }
        if (!group_of.empty()) {
            const size_t per_g = nloc_w * st.nleg;
            d_dleg = (double*)C.get("d_legendre", group_of.size() * per_g * 8);
            for (auto& kv : group_of)
                HIPC(hipMemcpy(d_dleg + size_t(kv.second) * per_g, st.maps.at(kv.first)->src_d_legendre(), per_g * 8, hipMemcpyDefault));
            d_wg = (double*)C.get("wf_scat", group_of.size() * size_t(ngrid) * nt * S * 8);
            if (st.has_delta_f()) {
                d_dfg = (double*)C.get("d_f_groups", group_of.size() * nloc_w * 8);
                for (auto& kv : group_of) {
                    const DerivativeMapping& gm = *st.maps.at(kv.first);
                    if (!gm.has_d_f(nloc_w)) throw std::runtime_error("scattering mapping " + kv.first + " was created after the delta-M scaling");
                    HIPC(hipMemcpy(d_dfg + size_t(kv.second) * nloc_w, gm.src_d_f(), nloc_w * 8, hipMemcpyDefault));
                }
            }
        }
    }
    double* d_f = nullptr;
    if (st.has_delta_f()) {
        d_f = (double*)C.get("delta_f", nloc_w * 8);
        HIPC(hipMemcpy(d_f, st.src_delta_f(), nloc_w * 8, hipMemcpyDefault));
    }
    e->dev->set_delta_m_fraction(d_f);
    if (e->config.emission_source == 2) {   // thermal source inside the DO equations (NSTOKES = 1, plane-parallel post-processing)
        if (derivs) {
            d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * 8);
            d_wsem = (double*)C.get("wf_surface_emission", nt * 8);
            e->dev->set_thermal_wf_outputs(d_wem, d_wsem);
        }
        if (!st.emission) throw std::runtime_error("emission_source = discrete_ordinates but the atmosphere storage has no emission_source array");
        double* d_em = (double*)C.get("do_emission", nloc_w * 8);
        HIPC(hipMemcpy(d_em, st.emission, nloc_w * 8, hipMemcpyDefault));
        double* d_sem = nullptr;
        if (su.emission) {
            d_sem = (double*)C.get("do_surface_emission", size_t(nw) * 8);
            HIPC(hipMemcpy(d_sem, su.emission, size_t(nw) * 8, hipMemcpyDefault));
        }
        e->dev->set_thermal_emission(d_em, d_sem);
    }
    e->dev->solve_device_vector(nw, norder, d_ext, d_ssa, d_leg, d_alb, d_irr, derivs, d_rad, d_we, d_ws, d_wa, d_od, d_status, nullptr,
                                (int)group_of.size(), d_dleg, d_wg, d_dfg);
    e->dev->set_delta_m_fraction(nullptr);
    e->dev->set_thermal_emission(nullptr, nullptr);
    e->dev->set_thermal_wf_outputs(nullptr, nullptr);
    if (e->so) {
        upload_emission(e, st, su);
        // the exact single scatter differentiates along the same group directions (d_dleg holds them group by group) into the same d_wg
        if (derivs && e->so->has_emission()) {
            d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * S * 8);
            d_wsem = (double*)C.get("wf_surface_emission", nt * S * 8);
            HIPC(hipMemset(d_wem, 0, size_t(ngrid) * nt * S * 8));
            HIPC(hipMemset(d_wsem, 0, nt * S * 8));
        }
        // non-Lambertian surface: the direct bounce of a line of sight that ends on the ground sees pi x brdf at its own ground geometry; the
        // single scatter is linear in that value, so its albedo weighting function times d(value) / d(argument) joins the argument's plane
        double* d_wa_ss = d_wa;
        if (brdf_kind != 0) {
            e->so->set_surface_brdf(brdf_kind, brdf_nargs, (const double*)C.get("brdf_args", size_t(nw) * brdf_nargs * 8));
            if (derivs) {
                d_wa_ss = (double*)C.get("wf_alb_ss", nt * S * 8);
                HIPC(hipMemset(d_wa_ss, 0, nt * S * 8));
            }
        }
        struct SoBrdfReset { skdev::SoDeviceEngine* d; ~SoBrdfReset() { d->set_surface_brdf(0, 1, nullptr); } } so_brdf_reset{e->so.get()};
        e->so->solve_device(nw, norder, d_ext, d_ssa, d_leg, d_alb, d_irr, d_rad, nullptr, nullptr, 1, derivs ? d_we : nullptr, d_ws, d_wa_ss, d_wem,
                            d_wsem, derivs ? (int)group_of.size() : 0, d_dleg, d_wg);
        if (brdf_kind != 0 && derivs) {
            hipLaunchKernelGGL(kern_brdf_ss_wf, dim3(unsigned((nt * S + 255) / 256)), dim3(256), 0, nullptr, d_wa_ss, e->so->los_albedo_derivative(), d_wa, nw, nlos, S,
                               brdf_nargs);
            HIPC(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(kern_transpose_od, dim3(1024), dim3(256), 0, nullptr, d_od, d_odt, nw, nlos);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    std::vector<int> status(nw);
    HIPC(hipMemcpy(status.data(), d_status, size_t(nw) * 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < nw; ++w)
        if (status[w] != 0) throw std::runtime_error("solver failure (status bits " + std::to_string(status[w]) + ") at wavelength " + std::to_string(w));
    HIPC(hipMemcpy(out->radiance, d_rad, nt * S * 8, hipMemcpyDeviceToHost));
    out->los_od.resize(nt);
    HIPC(hipMemcpy(out->los_od.data(), d_odt, nt * 8, hipMemcpyDeviceToHost));
    if (derivs) map_user_derivatives(e, st, su, out, d_we, d_ws, d_wa, d_rad, S, d_wg, group_of.empty() ? nullptr : &group_of, d_wem, d_wsem);
    return 0;
}

// emission_source / surface emission of the storage -> device, for an engine with an emission source (emission_source.cpp:37-173)
static void upload_emission(Engine* e, const AtmosphereStorage& st, const Surface& su) {
    if (!e->so || (e->config.emission_source != 0 && e->config.emission_source != 3)) return;
    if (!st.emission) throw std::runtime_error("the engine has an emission source but the atmosphere storage has no emission_source array");
    DeviceCache& C = e->cache;
    const size_t nloc_w = size_t(e->ngrid) * st.nwavel;
    double* d_em = (double*)C.get("so_emission", nloc_w * 8);
    HIPC(hipMemcpy(d_em, st.emission, nloc_w * 8, hipMemcpyDefault));
    double* d_sem = nullptr;
    if (su.emission) {
        d_sem = (double*)C.get("so_surface_emission", size_t(st.nwavel) * 8);
        HIPC(hipMemcpy(d_sem, su.emission, size_t(st.nwavel) * 8, hipMemcpyDefault));
    }
    e->so->set_emission(d_em, d_sem);
}

// exact single scatter and / or legacy successive orders alone (no discrete-ordinates source): radiances [wavel][los][stokes], LOS optical
// depths and -- exact single scatter without the diffuse source only -- weighting functions through the caller's derivative mappings.
static int calculate_radiance_so(Engine* e, Atmosphere* atm, OutputC* out) {
    const AtmosphereStorage& st = *atm->storage;
    const Surface& su = *atm->surface;
    const int S = e->config.num_stokes, per = (S == 3) ? 4 : 1;
    const int nw = st.nwavel, nlos = e->nlos, ngrid = e->ngrid;
    const int derivs = atm->derivs && e->config.wf_enabled && (!out->derivs.empty() || !out->surface_derivs.empty());
    if (out->nrad != nw * nlos) throw std::runtime_error("output radiance size mismatch");
    if (st.nleg % per != 0) throw std::runtime_error("leg_coeff must hold four coefficients per Legendre order for NSTOKES = 3");
    const size_t nloc_w = size_t(ngrid) * nw, nt = size_t(nw) * nlos;
    DeviceCache& C = e->cache;
    double* d_ext = (double*)C.get("so_ext", nloc_w * 8);
    double* d_ssa = (double*)C.get("so_ssa", nloc_w * 8);
    double* d_leg = (double*)C.get("so_leg", nloc_w * st.nleg * 8);
    double* d_alb = (double*)C.get("so_albedo", size_t(nw) * 8);
    double* d_rad = (double*)C.get("so_radiance", nt * S * 8);
    double* d_od = (double*)C.get("so_los_od", nt * 8);
    double *d_irr = nullptr, *d_we = nullptr, *d_ws = nullptr, *d_wa = nullptr;
    HIPC(hipMemcpy(d_ext, st.ext, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_ssa, st.ssa, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_leg, st.leg, nloc_w * st.nleg * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_alb, su.brdf_args, size_t(nw) * 8, hipMemcpyDefault));
    if (st.irradiance) {
        d_irr = (double*)C.get("so_irradiance", size_t(nw) * 8);
        HIPC(hipMemcpy(d_irr, st.irradiance, size_t(nw) * 8, hipMemcpyDefault));
    }
    if (derivs) {
        d_we = (double*)C.get("wf_ext", size_t(ngrid) * nt * S * 8);
        d_ws = (double*)C.get("wf_ssa", size_t(ngrid) * nt * S * 8);
        d_wa = (double*)C.get("wf_alb", nt * S * 8);
    }
    upload_emission(e, st, su);
    double *d_wem = nullptr, *d_wsem = nullptr;   // emission-profile / surface-emission weighting functions of a line-of-sight emission source
    if (derivs && e->so->has_emission()) {
        d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * S * 8);
        d_wsem = (double*)C.get("wf_surface_emission", nt * S * 8);
    }
    std::map<std::string, int> group_of;
    double *d_dleg = nullptr, *d_wg = nullptr;
    if (derivs) group_of = scattering_groups_of(st, out);
    if (!group_of.empty()) {
        d_dleg = upload_group_directions(e, st, group_of, "so_d_legendre");
        d_wg = (double*)C.get("wf_scat", group_of.size() * size_t(ngrid) * nt * S * 8);
    }
    e->so->solve_device(nw, st.nleg / per, d_ext, d_ssa, d_leg, d_alb, d_irr, d_rad, d_od, nullptr, 0, d_we, d_ws, d_wa, d_wem, d_wsem,
                        (int)group_of.size(), d_dleg, d_wg);
    HIPC(hipDeviceSynchronize());
    std::vector<double> od(nt);
    HIPC(hipMemcpy(out->radiance, d_rad, nt * S * 8, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(od.data(), d_od, nt * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nt * S; ++i)
        if (!std::isfinite(out->radiance[i])) throw std::runtime_error("solver failure (non-finite radiance) at wavelength " + std::to_string(i / (size_t(nlos) * S)));
    out->los_od.resize(nt);
    for (int w = 0; w < nw; ++w)
        for (int r = 0; r < nlos; ++r) out->los_od[size_t(r) * nw + w] = od[size_t(w) * nlos + r];
    if (derivs) map_user_derivatives(e, st, su, out, d_we, d_ws, d_wa, d_rad, S, d_wg, group_of.empty() ? nullptr : &group_of, d_wem, d_wsem);
    return 0;
}

// two-stream source (multiple_scatter_source = twostream and / or emission_source = twostream), optionally with the exact single scatter
// and the line-of-sight sources of the successive-orders engine on top: radiances [wavel][los], LOS optical depths and weighting functions
// through the caller's derivative mappings (extinction, ssa, scattering groups through the first Legendre coefficient, albedo, emission)
static int calculate_radiance_ts(Engine* e, Atmosphere* atm, OutputC* out) {
    const AtmosphereStorage& st = *atm->storage;
    const Surface& su = *atm->surface;
    const int nw = st.nwavel, nlos = e->nlos, ngrid = e->ngrid;
    const int derivs = atm->derivs && e->config.wf_enabled && (!out->derivs.empty() || !out->surface_derivs.empty());
    if (!out->flux_derivs.empty() || !out->surface_flux_derivs.empty()) return -2;
    if (out->nrad != nw * nlos) throw std::runtime_error("output radiance size mismatch");
    const size_t nloc_w = size_t(ngrid) * nw, nt = size_t(nw) * nlos;
    DeviceCache& C = e->cache;
    skdev::TsDeviceEngine::Inputs in;
    skdev::TsDeviceEngine::Outputs o;
    in.nleg = st.nleg;
    double* d_ext = (double*)C.get("ext", nloc_w * 8);
    double* d_ssa = (double*)C.get("ssa", nloc_w * 8);
    double* d_leg = (double*)C.get("leg", nloc_w * st.nleg * 8);
    double* d_alb = (double*)C.get("albedo", size_t(nw) * 8);
    double* d_rad = (double*)C.get("radiance", nt * 8);
    double* d_od = (double*)C.get("los_od", nt * 8);
    HIPC(hipMemcpy(d_ext, st.ext, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_ssa, st.ssa, nloc_w * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_leg, st.leg, nloc_w * st.nleg * 8, hipMemcpyDefault));
    HIPC(hipMemcpy(d_alb, su.brdf_args, size_t(nw) * 8, hipMemcpyDefault));
    in.ext = d_ext; in.ssa = d_ssa; in.leg = d_leg; in.albedo = d_alb;
    double* d_irr = nullptr;
    if (st.irradiance) {
        d_irr = (double*)C.get("irradiance", size_t(nw) * 8);
        HIPC(hipMemcpy(d_irr, st.irradiance, size_t(nw) * 8, hipMemcpyDefault));
    }
    in.irradiance = d_irr;
    double* d_f = nullptr;
    if (st.has_delta_f()) {
        d_f = (double*)C.get("delta_f", nloc_w * 8);
        HIPC(hipMemcpy(d_f, st.src_delta_f(), nloc_w * 8, hipMemcpyDefault));
    }
    in.delta_f = d_f;
    if (e->ts->thermal()) {
        if (!st.emission) throw std::runtime_error("emission_source = twostream but the atmosphere storage has no emission_source array");
        double* d_em = (double*)C.get("ts_emission", nloc_w * 8);
        HIPC(hipMemcpy(d_em, st.emission, nloc_w * 8, hipMemcpyDefault));
        in.emission = d_em;
        if (su.emission) {
            double* d_sem = (double*)C.get("ts_surface_emission", size_t(nw) * 8);
            HIPC(hipMemcpy(d_sem, su.emission, size_t(nw) * 8, hipMemcpyDefault));
            in.surface_emission = d_sem;
        }
    }
    o.radiance = d_rad;
    o.los_od = d_od;
    double *d_we = nullptr, *d_ws = nullptr, *d_wb = nullptr, *d_wa = nullptr, *d_wem = nullptr, *d_wsem = nullptr, *d_wg = nullptr;
    std::map<std::string, int> group_of;
    if (derivs) {
        d_we = (double*)C.get("wf_ext", size_t(ngrid) * nt * 8);
        d_ws = (double*)C.get("wf_ssa", size_t(ngrid) * nt * 8);
        d_wb = (double*)C.get("wf_b1", size_t(ngrid) * nt * 8);
        d_wa = (double*)C.get("wf_alb", nt * 8);
        o.wf_ext = d_we; o.wf_ssa = d_ws; o.wf_b1 = d_wb; o.wf_albedo = d_wa;
        if (e->ts->thermal() || (e->so && e->so->has_emission())) {
            d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * 8);
            d_wsem = (double*)C.get("wf_surface_emission", nt * 8);
            o.wf_emission = d_wem; o.wf_surface_emission = d_wsem;
        }
        for (auto& kv : out->derivs) {
            auto it = st.maps.find(kv.first);
            if (it != st.maps.end() && is_scattering_mapping(*it->second)) {
                const int gidx = (int)group_of.size();
                group_of[kv.first] = gidx;
            }
        }
    }
    e->ts->solve_device(nw, in, o, nullptr);
    if (!group_of.empty()) {
        // native scattering-group derivatives from dI/d(b1)
        const size_t per_g = nloc_w * st.nleg;
        double* d_dleg = (double*)C.get("d_legendre", per_g * 8);
        double* d_df = (double*)C.get("d_f_groups", nloc_w * 8);
        d_wg = (double*)C.get("wf_scat", group_of.size() * size_t(ngrid) * nt * 8);
        for (auto& kv : group_of) {
            const DerivativeMapping& m = *st.maps.at(kv.first);
            HIPC(hipMemcpy(d_dleg, m.src_d_legendre(), per_g * 8, hipMemcpyDefault));
            const bool with_f = d_f && m.has_d_f(nloc_w);
            if (d_f && !with_f) throw std::runtime_error("scattering mapping " + kv.first + " was created after the delta-M scaling");
            if (with_f) HIPC(hipMemcpy(d_df, m.src_d_f(), nloc_w * 8, hipMemcpyDefault));
            hipLaunchKernelGGL(kern_ts_scat_group, dim3(2048), dim3(256), 0, nullptr, d_wb, d_dleg, with_f ? d_f : nullptr, with_f ? d_df : nullptr,
                               d_wg + size_t(kv.second) * ngrid * nt, ngrid, nw, nlos, st.nleg);
            HIPC(hipGetLastError());
            HIPC(hipDeviceSynchronize());   // d_dleg / d_df are reused by the next group
        }
    }
    if (e->so) {
        // exact single scatter (and the emission / occultation line-of-sight sources) accumulated on top; its optical depths are the same rays'
        upload_emission(e, st, su);
        double* d_dleg_all = group_of.empty() ? nullptr : upload_group_directions(e, st, group_of, "so_d_legendre");
        e->so->solve_device(nw, st.nleg, d_ext, d_ssa, d_leg, d_alb, d_irr, d_rad, nullptr, nullptr, 1, derivs ? d_we : nullptr, d_ws, d_wa,
                            e->so->has_emission() ? d_wem : nullptr, e->so->has_emission() ? d_wsem : nullptr, (int)group_of.size(), d_dleg_all, d_wg);
    }
    HIPC(hipDeviceSynchronize());
    std::vector<double> od(nt);
    HIPC(hipMemcpy(out->radiance, d_rad, nt * 8, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(od.data(), d_od, nt * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nt; ++i)
        if (!std::isfinite(out->radiance[i])) throw std::runtime_error("solver failure (non-finite radiance) at wavelength " + std::to_string(i / size_t(nlos)));
    out->los_od.resize(nt);
    for (int w = 0; w < nw; ++w)
        for (int r = 0; r < nlos; ++r) out->los_od[size_t(r) * nw + w] = od[size_t(w) * nlos + r];
    if (derivs) map_user_derivatives(e, st, su, out, d_we, d_ws, d_wa, d_rad, 1, d_wg, group_of.empty() ? nullptr : &group_of, d_wem, d_wsem);
    return 0;
}

// Sasktran2::validate_input_atmosphere (cpp/lib/engine/engine.cpp:481-540, cpp/include/sasktran2/validation/validation.h:12-64): unless
// input_validation_mode is `disabled` (2), extinction must be finite and >= 0 and the single scatter albedo finite and inside [0, 1];
// the first offending (location, wavelength) is reported in the error text the way the reference logs it.
static void validate_input_atmosphere(const Engine* e, const AtmosphereStorage& st) {
    if (e->config.input_validation_mode == 2) return;
    if (!st.ext || !st.ssa || !st.leg) throw std::runtime_error("Invalid input: atmosphere storage has null arrays");
    auto scan = [&](const double* a, const char* name, double lo, double hi) {
        const size_t n = size_t(st.nloc) * st.nwavel;
        if (st.on_device) {
            // the arrays live in HBM (optical-property front end): the same scan as a kernel, the first offender comes back in one word
            unsigned long long* d_first = (unsigned long long*)const_cast<Engine*>(e)->cache.get("validate_first", 8);
            const unsigned long long none = ~0ull;
            HIPC(hipMemcpy(d_first, &none, 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(skfe::k_fe_validate, dim3(skfe::stream_grid(n)), dim3(256), 0, nullptr, a, n, lo, hi, d_first);
            HIPC(hipGetLastError());
            unsigned long long first = none;
            HIPC(hipMemcpy(&first, d_first, 8, hipMemcpyDeviceToHost));
            if (first == none) return;
            const size_t i = size_t(first >> 2);
            const std::string where = " at location: " + std::to_string(i % st.nloc) + " and wavelength: " + std::to_string(i / st.nloc);
            const int kind = int(first & 3);
            if (kind == 1) throw std::runtime_error(std::string("Invalid input. ") + name + " contains non-finite values" + where);
            if (kind == 2) throw std::runtime_error(std::string("Invalid input. ") + name + " contains values less than " + std::to_string(lo) + where);
            throw std::runtime_error(std::string("Invalid input. ") + name + " contains values greater than " + std::to_string(hi) + where);
        }
        for (size_t i = 0; i < n; ++i) {
            const double v = a[i];
            if (std::isfinite(v) && v >= lo && v <= hi) continue;
            const std::string where = " at location: " + std::to_string(i % st.nloc) + " and wavelength: " + std::to_string(i / st.nloc);
            if (!std::isfinite(v)) throw std::runtime_error(std::string("Invalid input. ") + name + " contains non-finite values" + where);
            if (v < lo) throw std::runtime_error(std::string("Invalid input. ") + name + " contains values less than " + std::to_string(lo) + where);
            throw std::runtime_error(std::string("Invalid input. ") + name + " contains values greater than " + std::to_string(hi) + where);
        }
    };
    scan(st.ext, "Atmosphere total extinction", 0.0, std::numeric_limits<double>::infinity());
    scan(st.ssa, "Atmosphere single scatter albedo", 0.0, 1.0);
}

int sk_engine_calculate_radiance(Engine* e, Atmosphere* atm, OutputC* out, int /*only_initialize*/) {
    if (!e || !atm || !out || !atm->storage || !atm->surface) return -1;
    try {
        const AtmosphereStorage& st = *atm->storage;
        const Surface& su = *atm->surface;
        const int S = e->config.num_stokes;
        if (st.nstokes != S || out->nstokes != S) return -2;
        if (st.nloc != e->ngrid) throw std::runtime_error("atmosphere grid does not match the engine geometry");
        if (!su.brdf_args) throw std::runtime_error("surface has no BRDF set");
        // non-Lambertian surfaces (SnowKokhanovsky, MODIS): the discrete-ordinates source in plane-parallel / pseudo-spherical geometry, on the
        // run-time-n kernels; other sources and the spherical post-processing (whose ground points would need a source per outgoing angle,
        // do_source_diffuse_storage.cpp:435-695) answer -2
        const bool brdf_surface = su.brdf && su.brdf->kind != 0;
        const bool so_touches_ground = e->so && (e->so->has_successive_orders() || e->so->has_emission());
        if (brdf_surface && (e->ts || !e->dev || so_touches_ground || (e->so && e->dev->plane_parallel()) || e->config.emission_source == 2 || e->nflux_obs > 0)) {
            set_error("sk_engine_calculate_radiance: a non-Lambertian BRDF is served for multiple_scatter_source = discrete_ordinates -- in plane-parallel or "
                      "pseudo-spherical geometry with the single scatter inside it, in spherical geometry for lines of sight that do not end on the ground -- "
                      "without thermal source, successive orders and flux observers");
            return -2;
        }
        HostPhases ph;
        validate_input_atmosphere(e, st);
        ph.mark("validate");
        const bool wants_derivs = atm->derivs && e->config.wf_enabled && (!out->derivs.empty() || !out->surface_derivs.empty());
        if (e->so && wants_derivs) {
            // a source without a linearisation is an error rather than a silently incomplete Jacobian (none today: the successive-orders
            // diffuse field is differentiated by its discrete adjoint, so_adjoint.inc)
            if (e->so->has_unlinearised_sources()) {
                set_error("sk_engine_calculate_radiance: weighting functions are not available with this combination of sources");
                return -2;
            }
        }
        if (e->ts) return calculate_radiance_ts(e, atm, out);
        if (e->so && !e->dev) return calculate_radiance_so(e, atm, out);
        if (brdf_surface) return calculate_radiance_vector(e, atm, out);
        bool scattering_groups = false;
        for (auto& kv : out->derivs) {
            auto it = st.maps.find(kv.first);
            if (it == st.maps.end()) continue;
            // emission derivatives: only the thermal source inside the discrete-ordinates equations carries them on this path
            if (it->second->uses_emission() && e->config.emission_source != 2 && !(e->so && e->so->has_emission())) return -2;
            scattering_groups = scattering_groups || is_scattering_mapping(*it->second);
        }
        // scattering-group derivatives are served by the run-time-n kernels for every stream count
        // SASKTRAN2_AMD_RUNTIME_N=1 sends every request through the run-time-n kernels (cross-check of the two device implementations)
        const char* force_env = std::getenv("SASKTRAN2_AMD_RUNTIME_N");
        const bool force = force_env && atoi(force_env) != 0;
        // 16 streams in spherical geometry with at most 32 lines of sight: up to two scattering groups ride the matrix-core kernels (one more
        // row per group through kern_k2c_mfma8 / kern_k2b_mfma8, DeviceEngine::set_scattering_groups); SASKTRAN2_AMD_GROUPS=vector keeps the
        // run-time-n kernels for them (the parity suite compares the two)
        int ngroups_requested = 0;
        if (atm->derivs && e->config.wf_enabled)
            for (auto& kv : out->derivs) {
                auto it = st.maps.find(kv.first);
                if (it != st.maps.end() && is_scattering_mapping(*it->second)) ++ngroups_requested;
            }
        const char* groups_env = std::getenv("SASKTRAN2_AMD_GROUPS");
        const bool groups_here = scattering_groups && ngroups_requested > 0 && !force && !e->dev->vector_mode() && e->config.emission_source != 2 &&
                                 !(groups_env && std::string(groups_env) == "vector") && e->dev->groups_on_matrix_path(ngroups_requested);
        if (force || e->dev->vector_mode() || (scattering_groups && atm->derivs && e->config.wf_enabled && !groups_here)) {
            if (e->nflux_obs > 0) {
                set_error("sk_engine_calculate_radiance: flux observers are not available together with scattering-group weighting functions or "
                          "SASKTRAN2_AMD_RUNTIME_N in this build");
                return -2;
            }
            return calculate_radiance_vector(e, atm, out);
        }
        if (!out->flux_derivs.empty() || !out->surface_flux_derivs.empty()) return -2;
        const int nw = st.nwavel, nlos = e->nlos, ngrid = e->ngrid;
        if (out->nrad != nw * nlos) throw std::runtime_error("output radiance size mismatch");
        // fluxes at the flux observers: [flux type][wavelength][observer] in the order of Config::flux_types (0 upwelling, 1 downwelling, 2 actinic)
        double* d_flux = nullptr;
        const size_t nflux = size_t(e->nflux_obs) * e->config.flux_types.size() * nw;
        if (e->nflux_obs > 0) {
            for (int ft : e->config.flux_types)
                if (ft < 0 || ft > 2) throw std::runtime_error("flux type " + std::to_string(ft) + " is not part of this build (upwelling 0, downwelling 1, actinic 2)");
            if (!out->flux || size_t(out->nflux) != nflux)
                throw std::runtime_error("output flux size mismatch: " + std::to_string(out->nflux) + " doubles given, flux types x wavelengths x observers = " + std::to_string(nflux));
            if (nflux > 0) d_flux = (double*)e->cache.get("flux", nflux * 8);
            e->dev->set_flux_output(d_flux, e->config.flux_types);
        }
        const int derivs = atm->derivs && e->config.wf_enabled && (!out->derivs.empty() || !out->surface_derivs.empty());
        const size_t nloc_w = size_t(ngrid) * nw, nt = size_t(nw) * nlos;
        DeviceCache& C = e->cache;
        double* d_ext = (double*)C.get("ext", nloc_w * 8);
        double* d_ssa = (double*)C.get("ssa", nloc_w * 8);
        double* d_leg = (double*)C.get("leg", nloc_w * st.nleg * 8);
        double* d_alb = (double*)C.get("albedo", size_t(nw) * 8);
        double* d_rad = (double*)C.get("radiance", nt * 8);
        double* d_od = (double*)C.get("los_od", nt * 8);
        double* d_odt = (double*)C.get("los_od_t", nt * 8);
        int* d_status = (int*)C.get("status", size_t(nw) * 4);
        double *d_irr = nullptr, *d_we = nullptr, *d_ws = nullptr, *d_wa = nullptr;
        ph.mark("buffers");
        upload_doubles(d_ext, st.ext, nloc_w);
        upload_doubles(d_ssa, st.ssa, nloc_w);
        ph.mark("ext ssa up");
        upload_doubles(d_leg, st.leg, nloc_w * st.nleg);
        ph.mark("leg up");
        HIPC(hipMemcpy(d_alb, su.brdf_args, size_t(nw) * 8, hipMemcpyDefault));
        if (st.irradiance) {
            d_irr = (double*)C.get("irradiance", size_t(nw) * 8);
            HIPC(hipMemcpy(d_irr, st.irradiance, size_t(nw) * 8, hipMemcpyDefault));
        }
        ph.mark("albedo irradiance up");
        if (derivs) {
            d_we = (double*)C.get("wf_ext", size_t(ngrid) * nt * 8);
            d_ws = (double*)C.get("wf_ssa", size_t(ngrid) * nt * 8);
            d_wa = (double*)C.get("wf_alb", nt * 8);
        }
        double* d_f = nullptr;
        if (st.has_delta_f()) {
            d_f = (double*)C.get("delta_f", nloc_w * 8);
            HIPC(hipMemcpy(d_f, st.src_delta_f(), nloc_w * 8, hipMemcpyDefault));
        }
        e->dev->set_delta_m_fraction(d_f);
        // scattering derivative groups on the matrix-core path (as calculate_radiance_vector prepares them for the run-time-n kernels)
        std::map<std::string, int> group_of;
        double *d_dleg = nullptr, *d_wg = nullptr, *d_dfg = nullptr;
        if (derivs && groups_here) {
            for (auto& kv : out->derivs) {
                auto it = st.maps.find(kv.first);
                if (it != st.maps.end() && is_scattering_mapping(*it->second)) {
                    const int gidx = (int)group_of.size();
                    group_of[kv.first] = gidx;
                }
            }
            const size_t per_g = nloc_w * st.nleg;
            d_dleg = (double*)C.get("d_legendre", group_of.size() * per_g * 8);
            for (auto& kv : group_of)
                HIPC(hipMemcpy(d_dleg + size_t(kv.second) * per_g, st.maps.at(kv.first)->src_d_legendre(), per_g * 8, hipMemcpyDefault));
            d_wg = (double*)C.get("wf_scat", group_of.size() * size_t(ngrid) * nt * 8);
            if (st.has_delta_f()) {
                d_dfg = (double*)C.get("d_f_groups", group_of.size() * nloc_w * 8);
                for (auto& kv : group_of) {
                    const DerivativeMapping& gm = *st.maps.at(kv.first);
                    if (!gm.has_d_f(nloc_w)) throw std::runtime_error("scattering mapping " + kv.first + " was created after the delta-M scaling");
                    HIPC(hipMemcpy(d_dfg + size_t(kv.second) * nloc_w, gm.src_d_f(), nloc_w * 8, hipMemcpyDefault));
                }
            }
        }
        e->dev->set_scattering_groups((int)group_of.size(), d_dleg, d_wg, d_dfg);
        ph.mark("upload");
        double *d_wem = nullptr, *d_wsem = nullptr;
        if (e->config.emission_source == 2) {   // thermal source inside the DO equations
            if (derivs) {
                d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * 8);
                d_wsem = (double*)C.get("wf_surface_emission", nt * 8);
                e->dev->set_thermal_wf_outputs(d_wem, d_wsem);
            }
            if (!st.emission) throw std::runtime_error("emission_source = discrete_ordinates but the atmosphere storage has no emission_source array");
            double* d_em = (double*)C.get("do_emission", nloc_w * 8);
            HIPC(hipMemcpy(d_em, st.emission, nloc_w * 8, hipMemcpyDefault));
            double* d_sem = nullptr;
            if (su.emission) {
                d_sem = (double*)C.get("do_surface_emission", size_t(nw) * 8);
                HIPC(hipMemcpy(d_sem, su.emission, size_t(nw) * 8, hipMemcpyDefault));
            }
            e->dev->set_thermal_emission(d_em, d_sem);
        }
        e->dev->solve_device(nw, st.nleg, d_ext, d_ssa, d_leg, d_alb, d_irr, derivs, d_rad, d_we, d_ws, d_wa, d_od, d_status,
                             nullptr, 0);
        e->dev->set_delta_m_fraction(nullptr);
        e->dev->set_thermal_emission(nullptr, nullptr);
        e->dev->set_thermal_wf_outputs(nullptr, nullptr);
        e->dev->set_flux_output(nullptr, {});
        e->dev->set_scattering_groups(0, nullptr, nullptr, nullptr);
        // single_scatter_source = exact on top of the discrete-ordinates multiple scatter: radiance and native weighting functions are
        // accumulated on the device before the user mappings are applied
        if (e->so) {
            upload_emission(e, st, su);
            if (derivs && e->so->has_emission()) {   // line-of-sight emission next to the discrete-ordinates source: its emission derivatives
                d_wem = (double*)C.get("wf_emission", size_t(ngrid) * nt * 8);
                d_wsem = (double*)C.get("wf_surface_emission", nt * 8);
                HIPC(hipMemset(d_wem, 0, size_t(ngrid) * nt * 8));
                HIPC(hipMemset(d_wsem, 0, nt * 8));
            }
            // (the exact single scatter differentiates along the same group directions into the same d_wg)
            e->so->solve_device(nw, st.nleg, d_ext, d_ssa, d_leg, d_alb, d_irr, d_rad, nullptr, nullptr, 1, derivs ? d_we : nullptr, d_ws, d_wa, d_wem, d_wsem,
                                derivs ? (int)group_of.size() : 0, d_dleg, d_wg);
        }
        hipLaunchKernelGGL(kern_transpose_od, dim3(1024), dim3(256), 0, nullptr, d_od, d_odt, nw, nlos);
        HIPC(hipGetLastError());
        HIPC(hipDeviceSynchronize());
        ph.mark("solve");
        std::vector<int> status(nw);
        HIPC(hipMemcpy(status.data(), d_status, size_t(nw) * 4, hipMemcpyDeviceToHost));
        for (int w = 0; w < nw; ++w)
            if (status[w] != 0) throw std::runtime_error("solver failure (status bits " + std::to_string(status[w]) + ") at wavelength " + std::to_string(w));
        HIPC(hipMemcpy(out->radiance, d_rad, nt * 8, hipMemcpyDeviceToHost));
        out->los_od.resize(nt);
        HIPC(hipMemcpy(out->los_od.data(), d_odt, nt * 8, hipMemcpyDeviceToHost));
        if (d_flux) HIPC(hipMemcpy(out->flux, d_flux, nflux * 8, hipMemcpyDeviceToHost));
        ph.mark("radiance out");
        if (derivs) {
            map_user_derivatives(e, st, su, out, d_we, d_ws, d_wa, d_rad, 1, d_wg, group_of.empty() ? nullptr : &group_of, d_wem, d_wsem);
        }
        ph.mark("weighting functions out");
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_engine_calculate_radiance: ") + ex.what());
        return -3;
    }
}

int sk_amd_engine_solve_device(Engine* e, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                               const double* d_albedo, const double* d_irradiance, int derivatives, double* d_radiance,
                               double* d_wf_ext, double* d_wf_ssa, double* d_wf_albedo, double* d_los_od, int* d_status,
                               void* stream, int max_chunk) {
    if (!e) return -1;
    try {
        // the device-resident entry point cannot look at the values (they live in HBM; non-physical optics surface as status bits),
        // but shapes and pointers are checked before any kernel sees them
        if (!e->dev || e->so) throw std::runtime_error("this entry point serves discrete-ordinates engines: use sk_amd_so_engine_solve_device or sk_engine_calculate_radiance");
        if (e->dev->vector_mode()) throw std::runtime_error("this engine runs the run-time-n / polarised kernels: use sk_engine_calculate_radiance");
        if (nwavel <= 0 || nleg <= 0) throw std::runtime_error("nwavel and nleg must be positive");
        if (!d_ext || !d_ssa || !d_leg || !d_albedo || !d_radiance || !d_los_od || !d_status)
            throw std::runtime_error("null device pointer (extinction, ssa, leg_coeff, albedo, radiance, los_od and status are required)");
        if (derivatives && (!d_wf_ext || !d_wf_ssa || !d_wf_albedo)) throw std::runtime_error("derivatives requested without weighting-function buffers");
        if (max_chunk < 0) throw std::runtime_error("max_chunk must be >= 0");
        e->dev->solve_device(nwavel, nleg, d_ext, d_ssa, d_leg, d_albedo, d_irradiance, derivatives, d_radiance, d_wf_ext,
                             d_wf_ssa, d_wf_albedo, d_los_od, d_status, stream, max_chunk);
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_amd_engine_solve_device: ") + ex.what());
        return -3;
    }
}
// ---- wavelength-sharded runs (one process per GPU): every rank solves its contiguous wavelength range into ONE wavelength-major slab,
// the ranks exchange the slabs with ONE all-gather (RCCL over xGMI; the caller's collective or sk_amd_engine_calculate_radiance_sharded),
// and one kernel unpacks the gathered block into the reference layouts.
int sk_amd_slab_row_doubles(Engine* e, int derivatives, unsigned long long* n) {
    if (!e || !n) return -1;
    *n = (unsigned long long)skdev::DeviceEngine::slab_fields(e->ngrid, derivatives) * (unsigned long long)e->nlos;
    return 0;
}
int sk_amd_engine_solve_device_slab(Engine* e, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                    const double* d_albedo, const double* d_irradiance, int derivatives, double* d_slab, int* d_status,
                                    void* stream, int max_chunk) {
    if (!e) return -1;
    try {
        if (!e->dev || e->so || e->dev->vector_mode()) throw std::runtime_error("the slab entry point serves the scalar discrete-ordinates engines");
        if (nwavel <= 0 || nleg <= 0) throw std::runtime_error("nwavel and nleg must be positive");
        if (!d_ext || !d_ssa || !d_leg || !d_albedo || !d_slab || !d_status) throw std::runtime_error("null device pointer");
        if (max_chunk < 0) throw std::runtime_error("max_chunk must be >= 0");
        e->dev->set_slab_output(d_slab);
        try {
            e->dev->solve_device(nwavel, nleg, d_ext, d_ssa, d_leg, d_albedo, d_irradiance, derivatives, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 d_status, stream, max_chunk);
        } catch (...) {
            e->dev->set_slab_output(nullptr);
            throw;
        }
        e->dev->set_slab_output(nullptr);
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_amd_engine_solve_device_slab: ") + ex.what());
        return -3;
    }
}
// gathered slabs [world][rows_per_rank][row] (rank r holds wavelengths shard(r) in its first rows, the rest is padding) -> radiance, los_od
// [nwavel][nlos], wf_extinction, wf_ssa [ngrid][nwavel][nlos], wf_albedo [nwavel][nlos]; shards as sasktran2_amd.distributed.shard_bounds
__global__ void __launch_bounds__(256) kern_unpack_slab(const double* slab, int nwavel, int world, int rows_per_rank, int ngrid, int nlos, int derivs,
                                                        double* radiance, double* los_od, double* wf_ext, double* wf_ssa, double* wf_alb) {
    const int nf = derivs ? 2 * ngrid + 3 : 2;
    const size_t n = size_t(nwavel) * nf * nlos;
    const int base = nwavel / world, extra = nwavel % world;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int l = int(i % nlos), f = int((i / nlos) % nf), w = int(i / (size_t(nlos) * nf));
        // rank and local row of wavelength w: the first `extra` ranks hold base + 1 wavelengths
        int r, loc;
        if (w < extra * (base + 1)) { r = w / (base + 1); loc = w - r * (base + 1); }
        else { r = extra + (base > 0 ? (w - extra * (base + 1)) / base : 0); loc = w - extra * (base + 1) - (r - extra) * base; }
        const double v = slab[((size_t(r) * rows_per_rank + loc) * nf + f) * nlos + l];
        const size_t o = size_t(w) * nlos + l, plane = size_t(nwavel) * nlos;
        if (f == 0) radiance[o] = v;
        else if (f == 1) { if (los_od) los_od[o] = v; }
        else if (f < 2 + ngrid) wf_ext[size_t(f - 2) * plane + o] = v;
        else if (f < 2 + 2 * ngrid) wf_ssa[size_t(f - 2 - ngrid) * plane + o] = v;
        else wf_alb[o] = v;
    }
}
int sk_amd_unpack_slab(Engine* e, const double* d_slab_all, int nwavel, int world, int rows_per_rank, int derivatives, double* d_radiance,
                       double* d_los_od, double* d_wf_ext, double* d_wf_ssa, double* d_wf_albedo, void* stream) {
    if (!e || !d_slab_all || !d_radiance) return -1;
    if (nwavel <= 0 || world <= 0 || rows_per_rank * world < nwavel) return -2;
    if (derivatives && (!d_wf_ext || !d_wf_ssa || !d_wf_albedo)) return -1;
    hipLaunchKernelGGL(kern_unpack_slab, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), d_slab_all, nwavel, world, rows_per_rank, e->ngrid,
                       e->nlos, derivatives, d_radiance, d_los_od, d_wf_ext, d_wf_ssa, d_wf_albedo);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int sk_amd_engine_set_delta_m_fraction(Engine* e, const double* d_f) {
    if (!e || !e->dev) return -1;
    e->dev->set_delta_m_fraction(d_f);
    return 0;
}
int sk_amd_engine_num_los(Engine* e, int* n) { if (!e || !n) return -1; *n = e->nlos; return 0; }
int sk_amd_engine_scratch_bytes(Engine* e, int nwavel, unsigned long long* bytes) {
    if (!e || !bytes) return -1;
    *bytes = (unsigned long long)((e->dev ? e->dev->bytes_per_wavelength() : 0) + (e->so ? e->so->bytes_per_wavelength() : 0) +
                                  (e->ts ? e->ts->bytes_per_wavelength(true) : 0)) * (unsigned long long)nwavel;
    return 0;
}
void* sk_amd_host_alloc(unsigned long long bytes) {
    void* p = nullptr;
    unsigned flags = hipHostMallocDefault;
    if (const char* e = std::getenv("SASKTRAN2_AMD_HOST_FLAGS")) flags = unsigned(strtoul(e, nullptr, 0));   // (experiments: coherent / non-coherent / write-combined)
    if (bytes == 0 || hipHostMalloc(&p, size_t(bytes), flags) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void sk_amd_host_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}
int sk_amd_engine_launch_shape(Engine* e, int* chunk, int* slots, unsigned long long* scratch_bytes) {
    if (!e) return -1;
    if (chunk) *chunk = e->dev ? e->dev->last_chunk() : (e->ts ? e->ts->last_chunk() : e->so->last_chunk());
    if (slots) *slots = e->dev ? e->dev->last_slots() : 1;
    if (scratch_bytes) *scratch_bytes = (unsigned long long)((e->dev ? e->dev->scratch_bytes() : 0) + (e->so ? e->so->scratch_bytes() : 0) +
                                                             (e->ts ? e->ts->scratch_bytes() : 0));
    return 0;
}
// per-kernel timing (HIP events around every launch, serialised): both device engines of the handle
int sk_amd_engine_set_profiling(Engine* e, int on) {
    if (!e) return -1;
    if (e->dev) { if (on == 2) e->dev->set_in_situ_timing(true); else e->dev->set_profiling(on != 0); }
    if (e->so) e->so->set_profiling(on != 0);
    if (e->ts) e->ts->set_profiling(on != 0);
    return 0;
}
int sk_amd_engine_reset_profile(Engine* e) {
    if (!e) return -1;
    if (e->dev) e->dev->reset_profile();
    if (e->so) e->so->reset_profile();
    if (e->ts) e->ts->reset_profile();
    return 0;
}
int sk_amd_engine_kernel_time(Engine* e, const char* name, double* ms, long* launches) {
    if (!e || !name || !ms || !launches) return -1;
    if (e->dev && e->dev->kernel_time(name, ms, launches)) return 0;
    if (e->so && e->so->kernel_time(name, ms, launches)) return 0;
    return (e->ts && e->ts->kernel_time(name, ms, launches)) ? 0 : -1;
}
int sk_amd_engine_kernel_names(Engine* e, char* buf, int len) {
    if (!e || !buf || len < 1) return -1;
    std::string s = e->dev ? e->dev->kernel_names() : std::string();
    if (e->so) {
        const std::string t = e->so->kernel_names();
        if (!t.empty()) s += (s.empty() ? "" : ",") + t;
    }
    if (e->ts) {
        const std::string t = e->ts->kernel_names();
        if (!t.empty()) s += (s.empty() ? "" : ",") + t;
    }
    std::snprintf(buf, len, "%s", s.c_str());
    return 0;
}

// device-resident entry point of the two-stream engine (bench.py, callers that keep spectra in HBM).  Inputs in the reference storage
// layout: ext, ssa, delta_f, emission [nwavel][ngrid], leg [nwavel][ngrid][nleg], albedo / irradiance / surface_emission [nwavel]
// (delta_f, irradiance, emission, surface_emission may be null as the configured sources allow).  Outputs: radiance [nwavel][nlos], los_od
// [nwavel][nlos] or null, and -- all of wf_ext / wf_ssa / wf_b1 / wf_albedo or none -- wf_ext, wf_ssa, wf_b1, wf_emission
// [ngrid][nwavel][nlos], wf_albedo, wf_surface_emission [nwavel][nlos] (the two emission outputs may be null).
int sk_amd_ts_engine_solve_device(Engine* e, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                  const double* d_delta_f, const double* d_albedo, const double* d_irradiance, const double* d_emission,
                                  const double* d_surface_emission, double* d_radiance, double* d_los_od, double* d_wf_ext, double* d_wf_ssa,
                                  double* d_wf_b1, double* d_wf_albedo, double* d_wf_emission, double* d_wf_surface_emission, void* stream,
                                  int max_chunk) {
    if (!e) return -1;
    try {
        if (!e->ts) throw std::runtime_error("the engine has no two-stream source");
        if (e->so) throw std::runtime_error("this entry point serves engines whose only source is the two-stream one: use sk_engine_calculate_radiance");
        if (max_chunk < 0) throw std::runtime_error("max_chunk must be >= 0");
        skdev::TsDeviceEngine::Inputs in;
        in.nleg = nleg;
        in.ext = d_ext; in.ssa = d_ssa; in.leg = d_leg; in.delta_f = d_delta_f; in.albedo = d_albedo; in.irradiance = d_irradiance;
        in.emission = d_emission; in.surface_emission = d_surface_emission;
        skdev::TsDeviceEngine::Outputs o;
        o.radiance = d_radiance; o.los_od = d_los_od;
        o.wf_ext = d_wf_ext; o.wf_ssa = d_wf_ssa; o.wf_b1 = d_wf_b1; o.wf_albedo = d_wf_albedo; o.wf_emission = d_wf_emission;
        o.wf_surface_emission = d_wf_surface_emission;
        e->ts->solve_device(nwavel, in, o, stream, max_chunk);
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_amd_ts_engine_solve_device: ") + ex.what());
        return -3;
    }
}

// device-resident entry point of the exact single-scatter / successive-orders engine (bench.py, callers that keep spectra in HBM)
int sk_amd_so_engine_solve_device(Engine* e, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                  const double* d_albedo, const double* d_irradiance, double* d_radiance, double* d_los_od, void* stream) {
    if (!e) return -1;
    try {
        if (!e->so) throw std::runtime_error("the engine has no exact single-scatter / successive-orders source");
        if (!d_ext || !d_ssa || !d_leg || !d_albedo || !d_radiance) throw std::runtime_error("null device pointer");
        e->so->solve_device(nwavel, nleg, d_ext, d_ssa, d_leg, d_albedo, d_irradiance, d_radiance, d_los_od, stream, 0);
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_amd_so_engine_solve_device: ") + ex.what());
        return -3;
    }
}
// the same with the native weighting functions: d_wf_ext / d_wf_ssa [nlocation][nwavel][nlos][nstokes], d_wf_albedo [nwavel][nlos][nstokes]
int sk_amd_so_engine_solve_device_wf(Engine* e, int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                     const double* d_albedo, const double* d_irradiance, double* d_radiance, double* d_los_od, double* d_wf_ext,
                                     double* d_wf_ssa, double* d_wf_albedo, void* stream) {
    if (!e) return -1;
    try {
        if (!e->so) throw std::runtime_error("the engine has no exact single-scatter / successive-orders source");
        if (!d_ext || !d_ssa || !d_leg || !d_albedo || !d_radiance || !d_wf_ext || !d_wf_ssa || !d_wf_albedo) throw std::runtime_error("null device pointer");
        if (e->so->has_emission()) throw std::runtime_error("emission sources take the host entry point (their derivative outputs are mappings)");
        e->so->solve_device(nwavel, nleg, d_ext, d_ssa, d_leg, d_albedo, d_irradiance, d_radiance, d_los_od, stream, 0, d_wf_ext, d_wf_ssa, d_wf_albedo);
        return 0;
    } catch (const std::exception& ex) {
        set_error(std::string("sk_amd_so_engine_solve_device_wf: ") + ex.what());
        return -3;
    }
}
// sizes[0..5]: incoming rays, their layers, accumulation-matrix entries, outgoing directions, scratch bytes per wavelength, wavelengths per block of the last solve
int sk_amd_so_engine_sizes(Engine* e, unsigned long long* sizes) {
    if (!e || !sizes || !e->so) return -1;
    sizes[0] = e->so->num_incoming_rays();
    sizes[1] = e->so->num_incoming_layers();
    sizes[2] = e->so->num_accumulation_entries();
    sizes[3] = e->so->num_outgoing();
    sizes[4] = e->so->bytes_per_wavelength();
    sizes[5] = (unsigned long long)e->so->last_chunk();
    return 0;
}

static int frontend_apply_delta_m(AtmosphereStorage& st, int order, int stack);
#include "capi_rest.inc"
#include "frontend.inc"
#include "capi_sharded.inc"

}  // extern "C"
