// Two-stream source on the device (multiple_scatter_source = twostream, emission_source = twostream).
//
// One thread per wavelength (kernel k_ts_solve) or per (wavelength, line of sight) (k_ts_view, k_ts_adjoint); the arithmetic is in ts_core.hpp, shared with
// the CPU emulation of the tests.  Every piece of state that outlives a kernel sits in scratch rows of W lanes ([item][lane]: a
// wavefront's access to one item is one contiguous 512-byte row); the wave-uniform geometry (Chapman factors, segments, stencils) is
// read through the scalar cache.  A launch takes as many wavelengths as the HBM budget allows.
//
// Reference replaced: CppTwoStreamSourceAdapter<ONLY_SOLAR | ONLY_THERMAL> (cpp/lib/sktran_disco/cpp_twostream_source.cpp), which packs 8
// wavelengths into one SIMD register the same way (LANES = 8, :19); here a wavefront carries 64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "ts_core.hpp"
#include "ts_engine.hpp"

namespace skdev {

#define TS_HIP_CHECK(expr)                                                                                     \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + \
                                     ":" + std::to_string(__LINE__));                                          \
    } while (0)

namespace {
constexpr int kWave = 64;
#ifndef TS_ADJOINT_WAVES
#define TS_ADJOINT_WAVES 2
#endif

template <bool SOLAR, bool FUSED> __global__ void __launch_bounds__(kWave) k_ts_solve(const skt::Geo g, const skt::Bufs b) {
    extern __shared__ double lds[];
    const int w = blockIdx.x * kWave + threadIdx.x;
    if (w >= b.nw) return;
    skt::solve_thread<SOLAR, FUSED>(g, b, w, lds + threadIdx.x, kWave);
}
// plane-parallel geometry with one line of sight: the upward pass after k_ts_solve<.., true> (back substitution, layer sources, radiance
// and -- DERIVS -- the adjoint sweep)
template <bool SOLAR, bool DERIVS> __global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(TS_ADJOINT_WAVES)))
k_ts_up(const skt::Geo g, const skt::Bufs b) {
    extern __shared__ double lds[];
    const int w = blockIdx.x * kWave + threadIdx.x;
    if (w >= b.nw) return;
    if (DERIVS) skt::view_adjoint_thread<SOLAR, false, false, true>(g, b, w, 0, lds + threadIdx.x, kWave);
    else skt::fused_radiance_thread<SOLAR>(g, b, w);
}
// (wavelength block, line of sight) of a workgroup of the adjoint sweep.  The lines of sight of one wavelength block read the same layer
// parameters and LU rows: their workgroups get consecutive ids ON THE SAME XCD (workgroup i runs on XCD i mod 8), so that those rows are
// fetched into one L2 once instead of eight times (6.5 -> 6.2 ms at 25 lines of sight x 10 000 wavelengths).  The forward sweep is better off
// with the plain (wavelength block fastest) order: neighbours then share a line of sight and with it the scalar-cache lines of its segments.
__device__ inline bool view_block(int nwb, int nlos, int& wblock, int& view) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    view = slot % nlos;
    wblock = (slot / nlos) * 8 + xcd;
    return wblock < nwb;
}
inline unsigned view_grid(int nwb, int nlos) { return unsigned((nwb + 7) / 8 * 8 * nlos); }

template <bool SOLAR, bool DERIVS, bool SPHERICAL> __global__ void __launch_bounds__(kWave) k_ts_view(const skt::Geo g, const skt::Bufs b) {
    const int w = blockIdx.x * kWave + threadIdx.x;
    if (w >= b.nw) return;
    skt::view_forward_thread<SOLAR, DERIVS, SPHERICAL>(g, b, w, blockIdx.y);
}
// two wavefronts per SIMD: the plane-parallel solar instance needs a few registers more than the 256 that allows, a handful of spills is the better trade
template <bool SOLAR, bool SPHERICAL> __global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(TS_ADJOINT_WAVES))) k_ts_adjoint(const skt::Geo g, const skt::Bufs b) {
    extern __shared__ double lds[];
    int wblock, view;
    if (!view_block((b.nw + kWave - 1) / kWave, g.nlos, wblock, view)) return;
    const int w = wblock * kWave + threadIdx.x;
    if (w >= b.nw) return;
    skt::view_adjoint_thread<SOLAR, SPHERICAL>(g, b, w, view, lds + threadIdx.x, kWave);
}
template <bool SOLAR, bool DERIVS> void launch_view(bool spherical, dim3 grid, hipStream_t st, const skt::Geo& g, const skt::Bufs& b) {
    if (spherical) hipLaunchKernelGGL((k_ts_view<SOLAR, DERIVS, true>), grid, dim3(kWave), 0, st, g, b);
    else hipLaunchKernelGGL((k_ts_view<SOLAR, DERIVS, false>), grid, dim3(kWave), 0, st, g, b);
}
template <bool SOLAR> void launch_adjoint(bool spherical, dim3 grid, size_t lds_bytes, hipStream_t st, const skt::Geo& g, const skt::Bufs& b) {
    if (spherical) hipLaunchKernelGGL((k_ts_adjoint<SOLAR, true>), grid, dim3(kWave), lds_bytes, st, g, b);
    else hipLaunchKernelGGL((k_ts_adjoint<SOLAR, false>), grid, dim3(kWave), lds_bytes, st, g, b);
}

struct DevMem {
    std::vector<void*> ptrs;
    template <typename T> T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("sasktran2_amd: hipMalloc failed (two-stream scratch)");
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
    template <typename T> const T* upload(const std::vector<T>& v) {
        T* p = alloc<T>(v.size());
        if (!v.empty()) TS_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
    ~DevMem() { release(); }
};
}  // namespace

struct TsDeviceEngine::Impl {
    skt::Geo g{};
    skts::Settings cfg;
    DevMem tables, scratch;
    skt::Bufs b{};
    int capacity = 0;
    bool cap_derivs = false;
    size_t scratch_bytes = 0;
    int last_chunk = 0;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, std::pair<double, long>> times;
};

namespace {
struct TsTimer {
    TsDeviceEngine::Impl& I;
    hipStream_t st;
    const char* name;
    TsTimer(TsDeviceEngine::Impl& i, hipStream_t s, const char* n) : I(i), st(s), name(n) {
        if (I.profiling) (void)hipEventRecord(I.ev0, st);
    }
    ~TsTimer() {
        if (!I.profiling) return;
        (void)hipEventRecord(I.ev1, st);
        (void)hipEventSynchronize(I.ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, I.ev0, I.ev1);
        auto& e = I.times[name];
        e.first += ms;
        e.second += 1;
    }
};
}  // namespace

TsDeviceEngine::TsDeviceEngine(const skts::TsGeometry& tg) : impl_(new Impl()) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        delete impl_;
        throw std::runtime_error("sasktran2_amd: no HIP device available (there is no CPU fallback)");
    }
    try {
        Impl& I = *impl_;
        I.cfg = tg.cfg;
        skt::Geo& g = I.g;
        g.nlyr = tg.nlyr;
        g.ngrid = tg.ngrid;
        g.nlos = tg.nlos;
        g.ncol = tg.ncol;
        g.spherical = tg.spherical ? 1 : 0;
        g.mu = tg.mu;
        g.max_seg = 0;
        for (size_t i = 0; i + 1 < tg.ray_start.size(); ++i) g.max_seg = std::max(g.max_seg, tg.ray_start[i + 1] - tg.ray_start[i]);
        if (g.nlyr < 1 || g.nlos < 1) throw std::runtime_error("two-stream source: empty geometry");
        if (tg.chapman.size() != size_t(g.ncol) * g.nlyr * g.nlyr) throw std::runtime_error("two-stream source: Chapman table size mismatch");
        if (g.spherical) {
            if ((int)tg.ray_start.size() != g.nlos + 1 || (int)tg.od_start.size() != tg.nseg() + 1) throw std::runtime_error("two-stream source: segment tables inconsistent");
            for (int v : tg.seg_layer) if (v < 0 || v >= g.nlyr) throw std::runtime_error("two-stream source: segment layer out of range");
            for (int v : tg.od_level) if (v < 0 || v >= g.ngrid) throw std::runtime_error("two-stream source: stencil level out of range");
            for (int v : tg.seg_col) if (v < 0 || v >= g.ncol) throw std::runtime_error("two-stream source: column out of range");
        }
        g.thickness = I.tables.upload(tg.thickness);
        g.csz = I.tables.upload(tg.csz);
        g.chapman = I.tables.upload(tg.chapman);
        g.chapman_t = I.tables.upload(tg.chapman_t);
        g.chap_uniform = I.tables.upload(tg.chap_uniform);
        g.view_cos = I.tables.upload(tg.view_cos);
        g.view_azimuth = I.tables.upload(tg.view_azimuth);
        g.ray_start = I.tables.upload(tg.ray_start);
        g.ground_hit = I.tables.upload(tg.ground_hit);
        g.view_cols = I.tables.upload(tg.view_cols);
        g.ground_col = I.tables.upload(tg.ground_col);
        g.ground_colw = I.tables.upload(tg.ground_colw);
        g.seg_layer = I.tables.upload(tg.seg_layer);
        g.seg_frac = I.tables.upload(tg.seg_frac);
        g.seg_cos = I.tables.upload(tg.seg_cos);
        g.seg_azimuth = I.tables.upload(tg.seg_azimuth);
        g.seg_col = I.tables.upload(tg.seg_col);
        g.seg_colw = I.tables.upload(tg.seg_colw);
        g.od_start = I.tables.upload(tg.od_start);
        g.od_level = I.tables.upload(tg.od_level);
        g.od_weight = I.tables.upload(tg.od_weight);
        TS_HIP_CHECK(hipEventCreate(&I.ev0));
        TS_HIP_CHECK(hipEventCreate(&I.ev1));
    } catch (...) {
        delete impl_;
        throw;
    }
}

TsDeviceEngine::~TsDeviceEngine() {
    if (impl_->ev0) (void)hipEventDestroy(impl_->ev0);
    if (impl_->ev1) (void)hipEventDestroy(impl_->ev1);
    delete impl_;
}

int TsDeviceEngine::nlos() const { return impl_->g.nlos; }
int TsDeviceEngine::ngrid() const { return impl_->g.ngrid; }
bool TsDeviceEngine::solar() const { return impl_->cfg.solar; }
bool TsDeviceEngine::thermal() const { return impl_->cfg.thermal; }
size_t TsDeviceEngine::bytes_per_wavelength(bool derivs) const {
    const skt::Geo& g = impl_->g;
    return skt::scratch_sizes(g.nlyr, g.ngrid, g.nlos, g.ncol, g.max_seg, g.spherical != 0, derivs, 1).total() * sizeof(double);
}
size_t TsDeviceEngine::scratch_bytes() const { return impl_->scratch_bytes; }
int TsDeviceEngine::last_chunk() const { return impl_->last_chunk; }
void TsDeviceEngine::set_profiling(bool on) { impl_->profiling = on; }
void TsDeviceEngine::reset_profile() { impl_->times.clear(); }
bool TsDeviceEngine::kernel_time(const char* name, double* ms, long* launches) const {
    auto it = impl_->times.find(name);
    if (it == impl_->times.end()) return false;
    if (ms) *ms = it->second.first;
    if (launches) *launches = it->second.second;
    return true;
}
std::string TsDeviceEngine::kernel_names() const {
    std::string s;
    for (auto& kv : impl_->times) s += (s.empty() ? "" : ",") + kv.first;
    return s;
}

void TsDeviceEngine::solve_device(int nwavel, const Inputs& in, const Outputs& out, void* stream, int max_chunk) {
    Impl& I = *impl_;
    const skt::Geo& g = I.g;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nwavel <= 0) throw std::runtime_error("two-stream source: nwavel must be positive");
    if (!in.ext || !in.ssa || !in.leg || !in.albedo || !out.radiance) throw std::runtime_error("two-stream source: null device pointer (extinction, ssa, leg_coeff, albedo and radiance are required)");
    if (in.nleg < 1) throw std::runtime_error("two-stream source: nleg must be positive");
    if (I.cfg.thermal && !in.emission) throw std::runtime_error("two-stream source: emission_source = twostream needs the atmosphere's emission_source array");
    const bool derivs = out.wf_ext != nullptr;
    if (derivs && (!out.wf_ssa || !out.wf_b1 || !out.wf_albedo)) throw std::runtime_error("two-stream source: weighting functions need wf_ext, wf_ssa, wf_b1 and wf_albedo together");

    // wavelengths per launch: as many as the HBM budget allows
    const size_t per_w = bytes_per_wavelength(derivs);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = size_t(0.5 * double(free_b + I.scratch_bytes));
    if (const char* e = std::getenv("SASKTRAN2_AMD_HBM_BUDGET_MB")) budget = size_t(std::max(1, atoi(e))) << 20;
    size_t fit = std::max<size_t>(kWave, budget / per_w / kWave * kWave);
    size_t want = size_t(1) << 22;   // no cache to stay inside: the more wavefronts per launch the better the latencies overlap
    if (const char* e = std::getenv("SASKTRAN2_AMD_TS_CHUNK")) want = std::max<size_t>(kWave, size_t(atoi(e)));
    if (max_chunk > 0) want = std::min<size_t>(want, size_t(max_chunk));
    fit = std::min(fit, std::max<size_t>(kWave, want / kWave * kWave));
    const size_t nblocks = (size_t(nwavel) + fit - 1) / fit;
    int chunk = int((size_t(nwavel) + nblocks - 1) / nblocks);
    chunk = (chunk + kWave - 1) / kWave * kWave;
    I.last_chunk = chunk;
    if (chunk > I.capacity || (derivs && !I.cap_derivs)) {
        I.scratch.release();
        I.cap_derivs = I.cap_derivs || derivs;
        I.capacity = std::max(chunk, I.capacity);
        const skt::ScratchSizes z = skt::scratch_sizes(g.nlyr, g.ngrid, g.nlos, g.ncol, g.max_seg, g.spherical != 0, I.cap_derivs, size_t(I.capacity));
        I.b.lay = I.scratch.alloc<double>(z.lay);
        I.b.lev = I.scratch.alloc<double>(z.lev);
        I.b.colv = I.scratch.alloc<double>(z.colv);
        I.b.lu = I.scratch.alloc<double>(z.lu);
        I.b.lam = I.scratch.alloc<double>(z.lam);
        I.b.g5 = I.scratch.alloc<double>(z.g5);
        I.b.gsum = I.scratch.alloc<double>(z.gsum);
        I.b.vstate = I.scratch.alloc<double>(z.vstate);
        I.b.path = I.scratch.alloc<double>(z.path);
        I.b.glev = I.scratch.alloc<double>(z.glev);
        I.scratch_bytes = z.total() * sizeof(double);
    }
    skt::Bufs b = I.b;
    b.W = I.capacity;
    b.nw_total = nwavel;
    b.nleg = in.nleg;
    b.derivs = derivs ? 1 : 0;
    b.ext = in.ext; b.ssa = in.ssa; b.leg = in.leg; b.delta_f = in.delta_f; b.albedo = in.albedo; b.irradiance = in.irradiance;
    b.emission = in.emission; b.surface_emission = in.surface_emission;
    b.radiance = out.radiance; b.wf_ext = out.wf_ext; b.wf_ssa = out.wf_ssa; b.wf_b1 = out.wf_b1; b.wf_alb = out.wf_albedo;
    b.wf_em = out.wf_emission; b.wf_sem = out.wf_surface_emission;
    b.los_od = out.los_od;
    const size_t lds_bytes = size_t(g.nlyr + 1) * kWave * sizeof(double);
    if (lds_bytes > 64 * 1024) throw std::runtime_error("two-stream source: too many layers for the per-wavefront LDS rows (" + std::to_string(g.nlyr) + ")");

    for (int w0 = 0; w0 < nwavel; w0 += chunk) {
        b.w0 = w0;
        b.nw = std::min(chunk, nwavel - w0);
        const dim3 grid_w((b.nw + kWave - 1) / kWave), grid_v((b.nw + kWave - 1) / kWave, g.nlos), grid_a(view_grid((b.nw + kWave - 1) / kWave, g.nlos));
        int pass = 0;
        for (int solar = 1; solar >= 0; --solar) {
            if (solar ? !I.cfg.solar : !I.cfg.thermal) continue;
            b.accumulate = (out.accumulate || pass > 0) ? 1 : 0;
            // plane-parallel geometry with one line of sight: two fused passes instead of three kernels (ts_core.hpp, solve_thread FUSED)
            const bool fused = !g.spherical && g.nlos == 1 && !std::getenv("SASKTRAN2_AMD_TS_UNFUSED");
            if (fused) {
                {
                    TsTimer t(I, st, solar ? "ts_solve" : "ts_solve_thermal");
                    if (solar) hipLaunchKernelGGL((k_ts_solve<true, true>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                    else hipLaunchKernelGGL((k_ts_solve<false, true>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                }
                {
                    TsTimer t(I, st, solar ? "ts_up" : "ts_up_thermal");
                    if (solar) {
                        if (derivs) hipLaunchKernelGGL((k_ts_up<true, true>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                        else hipLaunchKernelGGL((k_ts_up<true, false>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                    } else {
                        if (derivs) hipLaunchKernelGGL((k_ts_up<false, true>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                        else hipLaunchKernelGGL((k_ts_up<false, false>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                    }
                }
                TS_HIP_CHECK(hipGetLastError());
                ++pass;
                continue;
            }
            {
                TsTimer t(I, st, solar ? "ts_solve" : "ts_solve_thermal");
                if (solar) hipLaunchKernelGGL((k_ts_solve<true, false>), grid_w, dim3(kWave), lds_bytes, st, g, b);
                else hipLaunchKernelGGL((k_ts_solve<false, false>), grid_w, dim3(kWave), lds_bytes, st, g, b);
            }
            {
                TsTimer t(I, st, solar ? "ts_view" : "ts_view_thermal");
                const bool sph = g.spherical != 0;
                if (solar) {
                    if (derivs) launch_view<true, true>(sph, grid_v, st, g, b);
                    else launch_view<true, false>(sph, grid_v, st, g, b);
                } else {
                    if (derivs) launch_view<false, true>(sph, grid_v, st, g, b);
                    else launch_view<false, false>(sph, grid_v, st, g, b);
                }
            }
            if (derivs) {
                TsTimer t(I, st, solar ? "ts_adjoint" : "ts_adjoint_thermal");
                if (solar) launch_adjoint<true>(g.spherical != 0, grid_a, lds_bytes, st, g, b);
                else launch_adjoint<false>(g.spherical != 0, grid_a, lds_bytes, st, g, b);
            }
            TS_HIP_CHECK(hipGetLastError());
            ++pass;
        }
    }
}

}  // namespace skdev
