// HIP kernels + device driver of the exact single-scatter source and the successive-orders ("HR", legacy) multiple-scatter
// source (gfx950).  Reference stages and where they run here:
//
//   solar transmission at every layer boundary        cpp/lib/solar/singlescattersource.cpp:103-160           k_so_march (exact rows / table)
//   single-scatter layer source, ground term           singlescattersource.cpp:574-636,950-1167,
//                                                      cpp/include/sasktran2/solartransmission.h:733-846      k_so_march
//   first-order incoming radiance + accumulation       cpp/lib/sourceintegrator/sourceintegrator.cpp:666-760   k_so_march<incoming>
//   scattering matrices x incoming                     cpp/lib/hr/diffuse_table.cpp:673-716,775-808,
//                                                      cpp/lib/hr/diffuse_point.cpp:199-296                    k_so_scatter
//   incoming = accumulation x outgoing + first order   diffuse_table.cpp:755-770                               k_so_accumulate
//   LOS integration of both sources                    sourceintegrator.cpp:521-580, diffuse_table.cpp:965-1064  k_so_march<los>
//
// MI355X formulation: a LANE IS A WAVELENGTH.  Every per-wavelength array is stored [item][wavelength] (wavelength fastest), so the
// 64 lanes of a wavefront read 512 contiguous bytes; every geometric quantity (layer stencils, interpolation weights, CSR rows of the
// accumulation matrix, Wigner functions of a ray's scattering angle, scattering-matrix basis) depends on the ray / layer / direction
// only, is wave-uniform and comes through the scalar cache.  No LDS, no atomics, no divergence except at ray ends.  The accumulation
// matrix is never stored: an entry is (geometric weight, scalar) x (layer factor omega (1 - e^-od) e^-od_before, one double per layer
// and wavelength), which cuts the per-wavelength HBM stream of an iteration from 8 B per matrix entry to 8 B per ray layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "so_engine.hpp"

namespace skdev {

#define SO_HIP_CHECK(expr)                                                                                     \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + \
                                     ":" + std::to_string(__LINE__));                                          \
    } while (0)

namespace {
constexpr double kPi = 3.14159265358979323846;
constexpr int kWave = 64;

struct DevCsr {
    const int* start;
    const int* idx;
    const double* w;
};
struct DevRays {
    int nrays;
    const int *layer_start, *endpoint_start;
    const unsigned char* ground_is_hit;
    const double* ground_mu_in;
    DevCsr od;
    const int* stencil_idx;                 // [layer][4]
    const double *w_entrance, *w_exit;      // [layer][4]
    const double *quad_start, *quad_end;    // [layer]
    const unsigned char* tiny;
    const int* mid_idx;                     // [layer][2]
    const double* mid_w;
    DevCsr solar;                           // per endpoint
    const unsigned char* solar_blocked;
    const double *C2, *S2, *d00, *d02;      // per ray; d00 / d02 [ray][num_ss_moments]
};
struct SoGeo {
    int ngrid, nlos, S, num_ss_moments, num_ms_moments;
    int npoints, ninterior, nin, nout;
    DevRays los, inc;
    DevCsr table;                           // solar optical depth of the table points over the grid
    int ntable;
    DevCsr acc;                             // accumulation rows per incoming ray
    const int* acc_layer;
    DevCsr los_source, los_ground;
    const int *in_start, *out_start, *point_leg_idx;
    const double* point_leg_w;
    const double* scat_basis;               // [row][col][num_ms_moments * per]
    const double* ground_factor;            // concatenated per ground point
    const int* ground_factor_start;         // [nground + 1]
};
// per-block scratch, every array [item][nwp] with nwp = wavelengths of the block padded to the wavefront size
struct SoBufs {
    int nw, nwp, nleg, w0, nwavel_total;
    const double *ext, *ssa, *leg, *albedo, *irradiance;   // caller's inputs at the first wavelength of the block
    double *kT, *omT, *legT, *tab, *cl, *first, *incoming, *outgoing, *factor, *gfactor;
    double *radiance, *los_od;                               // caller's outputs (whole spectrum)
    int accumulate;
};

// inputs [w][q] -> [q][w]; Legendre / Greek coefficients [w][q][lg] -> [q][lg][w]
__global__ void __launch_bounds__(256) k_so_transpose(SoGeo g, SoBufs b) {
    const int nlg = b.nleg * (g.S == 3 ? 4 : 1);
    const size_t per_w = size_t(g.ngrid) * (2 + nlg), n = per_w * b.nwp;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i % b.nwp);
        const size_t item = i / b.nwp;
        const int ws = w < b.nw ? w : b.nw - 1;   // padding lanes repeat the last wavelength (keeps every lane finite)
        if (item < size_t(g.ngrid)) b.kT[i] = b.ext[size_t(ws) * g.ngrid + item];
        else if (item < size_t(2 * g.ngrid)) b.omT[i - size_t(g.ngrid) * b.nwp] = b.ssa[size_t(ws) * g.ngrid + (item - g.ngrid)];
        else {
            const size_t t = item - 2 * g.ngrid;   // q * nlg + lg
            b.legT[t * b.nwp + w] = b.leg[size_t(ws) * g.ngrid * nlg + t];
        }
    }
}

// solar optical depth at the table points; Legendre coefficients of the diffuse points
__global__ void __launch_bounds__(kWave) k_so_tables(SoGeo g, SoBufs b) {
    const int w = blockIdx.x * kWave + threadIdx.x, item = blockIdx.y, nwp = b.nwp;
    if (item < g.ntable) {
        double od = 0;
        for (int e = g.table.start[item]; e < g.table.start[item + 1]; ++e) od += g.table.w[e] * b.kT[size_t(g.table.idx[e]) * nwp + w];
        b.tab[size_t(item) * nwp + w] = od;
        return;
    }
    const int per = g.S == 3 ? 4 : 1, nlgB = g.num_ms_moments * per, nlg_in = b.nleg * per;
    const int t = item - g.ntable, p = t / nlgB, lg = t % nlgB;
    double c = 0;
    if (lg < nlg_in)
        for (int i = 0; i < 2; ++i)
            c += g.point_leg_w[2 * p + i] * b.legT[(size_t(g.point_leg_idx[2 * p + i]) * nlg_in + lg) * nwp + w];
    b.cl[size_t(t) * nwp + w] = c;
}

// One ray, 64 wavelengths per wavefront, marched from the observer outward.
// INCOMING: first-order incoming radiance of a diffuse point direction (table solar transmission) + the layer factors of the accumulation matrix.
// LOS: radiance of a line of sight = single scatter (exact solar transmission) + diffuse source interpolated from `outgoing`.
template <int S, bool LOS>
__global__ void __launch_bounds__(kWave) k_so_march(SoGeo g, SoBufs b, int single_scatter, int multiple_scatter) {
    const DevRays& rs = LOS ? g.los : g.inc;
    const int w = blockIdx.x * kWave + threadIdx.x, r = blockIdx.y, nwp = b.nwp;
    const int l0 = rs.layer_start[r], n = rs.layer_start[r + 1] - l0, e0 = rs.endpoint_start[r];
    const int per = S == 3 ? 4 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg);
    const int ws = w < b.nw ? w : b.nw - 1;
    const double F = b.irradiance ? b.irradiance[ws] : 1.0, albedo = b.albedo[ws];
    double val[S];
#pragma unroll
    for (int s = 0; s < S; ++s) val[s] = 0;
    double cur = 0, before = 1;

    // small cache of the phase functions at the grid points this ray is passing (a ray crosses one grid level per layer)
    int cq[4] = {-1, -1, -1, -1};
    double c11[4], c12[4];
    int slot = 0;
    auto phase = [&](int q, double& p11, double& p12) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cq[c] == q) { p11 = c11[c]; p12 = c12[c]; return; }
        double s1 = 0, s2 = 0;
        const double* lq = b.legT + size_t(q) * nlg_in * nwp + w;
        for (int l = 0; l < nm; ++l) {
            s1 += lq[size_t(l) * per * nwp] * rs.d00[size_t(r) * g.num_ss_moments + l];
            if (S == 3) s2 += lq[size_t(l * per + 3) * nwp] * rs.d02[size_t(r) * g.num_ss_moments + l];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c == slot) { cq[c] = q; c11[c] = s1; c12[c] = s2; }
        slot = (slot + 1) & 3;
        p11 = s1;
        p12 = s2;
    };
    auto transmission = [&](int e) -> double {
        if (LOS && rs.solar_blocked[e]) return 0.0;
        const double* x = LOS ? b.kT : b.tab;
        double od = 0;
        for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) od += rs.solar.w[t] * x[size_t(rs.solar.idx[t]) * nwp + w];
        return exp(-od) * F;
    };
    auto endpoint = [&](int L, const double* wts, double T, double* out) {
        double ke = 0, oe = 0, a = 0, bb = 0;
        for (int i = 0; i < 4; ++i) {
            const int q = rs.stencil_idx[4 * L + i];
            const double wt = wts[4 * L + i];
            if (q < 0 || wt == 0.0) continue;
            double p11, p12;
            phase(q, p11, p12);
            ke += b.kT[size_t(q) * nwp + w] * wt;
            oe += b.omT[size_t(q) * nwp + w] * wt;
            a += p11 * wt;
            bb += p12 * wt;
        }
        const double amp = ke * oe * T / (kPi * 4);
        out[0] = amp * a;
        if (S == 3) {
            out[1] = amp * (-rs.C2[r] * bb);
            out[2] = amp * (-rs.S2[r] * bb);
        }
    };

    double T_entrance = 0;   // transmission at the entrance of the layer in hand = exit of the layer nearer to the observer
    double total_od = 0;
    if (n > 0 && single_scatter) T_entrance = transmission(e0 + n);
    for (int j = n - 1; j >= 0; --j) {
        const int L = l0 + j;
        double od = 0;
        for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t) od += rs.od.w[t] * b.kT[size_t(rs.od.idx[t]) * nwp + w];
        const double att = exp(-od);
        if (single_scatter) {
            const double T_exit = transmission(e0 + j);
            if (!rs.tiny[L]) {
                double st[S], en[S];
                endpoint(L, rs.w_entrance, T_entrance, st);
                endpoint(L, rs.w_exit, T_exit, en);
                const double sf = (fabs(od) < 1e-12) ? 1.0 : -expm1(-od) / od;
#pragma unroll
                for (int s = 0; s < S; ++s) val[s] += sf * (st[s] * rs.quad_start[L] + en[s] * rs.quad_end[L]) * before;
            }
            T_entrance = T_exit;
        }
        if (multiple_scatter) {
            const double omega = rs.mid_w[2 * L] * b.omT[size_t(rs.mid_idx[2 * L]) * nwp + w] + rs.mid_w[2 * L + 1] * b.omT[size_t(rs.mid_idx[2 * L + 1]) * nwp + w];
            const double f = omega * (1 - att) * before;
            if (LOS) {
                for (int t = g.los_source.start[L]; t < g.los_source.start[L + 1]; ++t) {
                    const double fw = f * g.los_source.w[t];
                    const double* o = b.outgoing + size_t(g.los_source.idx[t]) * S * nwp + w;
#pragma unroll
                    for (int s = 0; s < S; ++s) val[s] += fw * o[size_t(s) * nwp];
                }
            } else {
                b.factor[size_t(L) * nwp + w] = f;
            }
        }
        cur += od;
        total_od += od;
        before = exp(-cur);
    }
    if (n > 0 && rs.ground_is_hit[r]) {
        if (single_scatter && rs.ground_mu_in[r] > 0) val[0] += T_entrance * (albedo / kPi) * rs.ground_mu_in[r] * before;
        if (multiple_scatter) {
            if (LOS) {
                for (int t = g.los_ground.start[r]; t < g.los_ground.start[r + 1]; ++t) {
                    const double fw = before * g.los_ground.w[t];
                    const double* o = b.outgoing + size_t(g.los_ground.idx[t]) * S * nwp + w;
#pragma unroll
                    for (int s = 0; s < S; ++s) val[s] += fw * o[size_t(s) * nwp];
                }
            } else {
                b.gfactor[size_t(r) * nwp + w] = before;
            }
        }
    } else if (!LOS && multiple_scatter) {
        b.gfactor[size_t(r) * nwp + w] = 0.0;
    }
    if (LOS) {
        if (w >= b.nw) return;
        double* out = b.radiance + (size_t(b.w0 + w) * g.nlos + r) * S;
#pragma unroll
        for (int s = 0; s < S; ++s) out[s] = b.accumulate ? out[s] + val[s] : val[s];
        if (b.los_od) b.los_od[size_t(b.w0 + w) * g.nlos + r] = total_od;
    } else {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            b.first[(size_t(r) * S + s) * nwp + w] = val[s];
            b.incoming[(size_t(r) * S + s) * nwp + w] = val[s];
        }
    }
}

// outgoing source of one (point, outgoing row) for 64 wavelengths: row of [sum_l c_l B_l] x incoming (interior points),
// Lambertian reflection of the downwelling flux (ground points)
template <int S>
__global__ void __launch_bounds__(kWave) k_so_scatter(SoGeo g, SoBufs b) {
    const int w = blockIdx.x * kWave + threadIdx.x, row = blockIdx.y, p = blockIdx.z, nwp = b.nwp;
    const int nrow = (g.out_start[p + 1] - g.out_start[p]) * S, ncol = (g.in_start[p + 1] - g.in_start[p]) * S;
    if (row >= nrow) return;
    const double* in = b.incoming + size_t(g.in_start[p]) * S * nwp + w;
    double* out = b.outgoing + (size_t(g.out_start[p]) * S + row) * nwp + w;
    if (p >= g.ninterior) {
        double v = 0;
        if (row % S == 0) {
            const int ws = w < b.nw ? w : b.nw - 1;
            const double* f = g.ground_factor + g.ground_factor_start[p - g.ninterior];
            for (int i = 0; i < ncol / S; ++i) v += f[i] * in[size_t(i) * S * nwp];
            v *= b.albedo[ws] / kPi;
        }
        *out = v;
        return;
    }
    const int per = S == 3 ? 4 : 1, nlgB = g.num_ms_moments * per, nlg = min(g.num_ms_moments, b.nleg) * per;
    const double* cl = b.cl + size_t(p) * nlgB * nwp + w;
    const double* B = g.scat_basis + size_t(row) * ncol * nlgB;
    double sum = 0;
    for (int c0 = 0; c0 < nlg; c0 += 16) {
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0;
        const int nt = min(16, nlg - c0);
        if (nt == 16) {
            for (int i = 0; i < ncol; ++i) {
                const double x = in[size_t(i) * nwp];
                const double* Bi = B + size_t(i) * nlgB + c0;
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] += Bi[t] * x;
            }
        } else {
            for (int i = 0; i < ncol; ++i) {
                const double x = in[size_t(i) * nwp];
                const double* Bi = B + size_t(i) * nlgB + c0;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < nt) acc[t] += Bi[t] * x;
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < nt) sum += cl[size_t(c0 + t) * nwp] * acc[t];
    }
    *out = sum;
}

// incoming = accumulation matrix x outgoing + first order, one incoming ray x 64 wavelengths per wavefront
template <int S>
__global__ void __launch_bounds__(kWave) k_so_accumulate(SoGeo g, SoBufs b) {
    const int w = blockIdx.x * kWave + threadIdx.x, r = blockIdx.y, nwp = b.nwp;
    double acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = b.first[(size_t(r) * S + s) * nwp + w];
    const double gf = b.gfactor[size_t(r) * nwp + w];
    int Lc = -2;
    double f = 0;
    for (int e = g.acc.start[r]; e < g.acc.start[r + 1]; ++e) {
        const int L = g.acc_layer[e];
        if (L != Lc) { Lc = L; f = L >= 0 ? b.factor[size_t(L) * nwp + w] : gf; }
        const double fw = f * g.acc.w[e];
        const double* o = b.outgoing + size_t(g.acc.idx[e]) * S * nwp + w;
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] += fw * o[size_t(s) * nwp];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) b.incoming[(size_t(r) * S + s) * nwp + w] = acc[s];
}

struct DevMem {
    std::vector<void*> ptrs;
    template <typename T>
    T* alloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("sasktran2_amd: hipMalloc failed (successive-orders scratch)");
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
    template <typename T>
    const T* upload(const std::vector<T>& v) {
        T* p = alloc<T>(v.size());
        if (!v.empty()) SO_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
    ~DevMem() { release(); }
};
}  // namespace

struct SoDeviceEngine::Impl {
    SoGeo g{};
    DevMem tables, scratch;
    SoBufs b{};
    int num_iterations = 50;
    bool single_scatter = true, multiple_scatter = true;
    int capacity = 0, cap_nleg = 0;
    size_t n_inc_rays = 0, n_inc_layers = 0, n_acc = 0, n_out = 0, n_table = 0, max_rows = 0;
    size_t scratch_bytes = 0;
    int last_chunk = 0;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, std::pair<double, long>> times;

    DevCsr up(const skso::Csr& c) {
        DevCsr d;
        d.start = tables.upload(c.start);
        d.idx = tables.upload(c.idx);
        d.w = tables.upload(c.w);
        return d;
    }
    static std::vector<unsigned char> bytes(const std::vector<char>& v) { return std::vector<unsigned char>(v.begin(), v.end()); }
    DevRays up(const skso::RaySet& r) {
        DevRays d{};
        d.nrays = r.nrays;
        d.layer_start = tables.upload(r.layer_start);
        d.endpoint_start = tables.upload(r.endpoint_start);
        d.ground_is_hit = tables.upload(bytes(r.ground_is_hit));
        d.ground_mu_in = tables.upload(r.ground_mu_in);
        d.od = up(r.od);
        d.stencil_idx = tables.upload(r.stencil_idx);
        d.w_entrance = tables.upload(r.w_entrance);
        d.w_exit = tables.upload(r.w_exit);
        d.quad_start = tables.upload(r.quad_start);
        d.quad_end = tables.upload(r.quad_end);
        d.tiny = tables.upload(bytes(r.tiny));
        d.mid_idx = tables.upload(r.mid_idx);
        d.mid_w = tables.upload(r.mid_w);
        d.solar = up(r.solar);
        d.solar_blocked = tables.upload(bytes(r.solar_blocked));
        d.C2 = tables.upload(r.C2);
        d.S2 = tables.upload(r.S2);
        d.d00 = tables.upload(r.d00);
        d.d02 = tables.upload(r.d02);
        return d;
    }
};

namespace {
struct SoTimer {
    SoDeviceEngine::Impl& I;
    hipStream_t st;
    const char* name;
    SoTimer(SoDeviceEngine::Impl& i, hipStream_t s, const char* n) : I(i), st(s), name(n) {
        if (I.profiling) (void)hipEventRecord(I.ev0, st);
    }
    ~SoTimer() {
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

SoDeviceEngine::SoDeviceEngine(const skso::SoGeometry& sg, int num_iterations) : impl_(new Impl()) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        delete impl_;
        throw std::runtime_error("sasktran2_amd: no HIP device available (there is no CPU fallback)");
    }
    Impl& I = *impl_;
    SoGeo& g = I.g;
    I.num_iterations = num_iterations;
    I.single_scatter = sg.cfg.single_scatter;
    I.multiple_scatter = sg.cfg.multiple_scatter && sg.npoints > 0 && num_iterations > 0;
    g.ngrid = sg.ngrid;
    g.nlos = sg.nlos;
    g.S = sg.cfg.nstokes;
    g.num_ss_moments = sg.cfg.num_ss_moments;
    g.num_ms_moments = sg.cfg.num_ms_moments;
    g.los = I.up(sg.los);
    g.npoints = g.ninterior = 0;
    if (I.multiple_scatter) {
        g.npoints = sg.npoints;
        g.ninterior = sg.ninterior;
        g.nin = sg.nin;
        g.nout = sg.nout;
        g.inc = I.up(sg.incoming);
        g.table = I.up(sg.table.od);
        g.ntable = sg.table.od.rows();
        g.acc = I.up(sg.accumulation);
        g.acc_layer = I.tables.upload(sg.accumulation_layer);
        g.los_source = I.up(sg.los_source);
        g.los_ground = I.up(sg.los_ground);
        g.in_start = I.tables.upload(sg.in_start);
        g.out_start = I.tables.upload(sg.out_start);
        g.point_leg_idx = I.tables.upload(sg.point_leg_idx);
        g.point_leg_w = I.tables.upload(sg.point_leg_w);
        // basis [l][greek][row][col] -> [row][col][l * per + greek]: the (order, Greek constant) run of one matrix element is contiguous
        const int S = g.S, per = S == 3 ? 4 : 1, rows = sg.nout * S, cols = sg.nin * S, nlgB = sg.cfg.num_ms_moments * per;
        std::vector<double> Bt(size_t(rows) * cols * nlgB);
        for (int lg = 0; lg < nlgB; ++lg)
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols; ++c) Bt[(size_t(r) * cols + c) * nlgB + lg] = sg.scat_basis[(size_t(lg) * rows + r) * cols + c];
        g.scat_basis = I.tables.upload(Bt);
        std::vector<double> gf;
        std::vector<int> gfs{0};
        for (const auto& f : sg.ground_in_factor) {
            gf.insert(gf.end(), f.begin(), f.end());
            gfs.push_back((int)gf.size());
        }
        g.ground_factor = I.tables.upload(gf);
        g.ground_factor_start = I.tables.upload(gfs);
        I.n_inc_rays = sg.incoming.nrays;
        I.n_inc_layers = sg.incoming.nlayers();
        I.n_acc = sg.accumulation.w.size();
        I.n_out = sg.out_start.back();
        I.n_table = g.ntable;
        I.max_rows = size_t(sg.nout) * S;
    }
    SO_HIP_CHECK(hipEventCreate(&I.ev0));
    SO_HIP_CHECK(hipEventCreate(&I.ev1));
}

SoDeviceEngine::~SoDeviceEngine() {
    if (!impl_) return;
    (void)hipDeviceSynchronize();
    if (impl_->ev0) (void)hipEventDestroy(impl_->ev0);
    if (impl_->ev1) (void)hipEventDestroy(impl_->ev1);
    delete impl_;
}

int SoDeviceEngine::nlos() const { return impl_->g.nlos; }
int SoDeviceEngine::ngrid() const { return impl_->g.ngrid; }
int SoDeviceEngine::nstokes() const { return impl_->g.S; }
size_t SoDeviceEngine::scratch_bytes() const { return impl_->scratch_bytes; }
int SoDeviceEngine::last_chunk() const { return impl_->last_chunk; }
void SoDeviceEngine::set_profiling(bool on) { impl_->profiling = on; }
void SoDeviceEngine::reset_profile() { impl_->times.clear(); }
bool SoDeviceEngine::kernel_time(const char* name, double* ms, long* launches) const {
    auto it = impl_->times.find(name);
    if (it == impl_->times.end()) return false;
    *ms = it->second.first;
    *launches = it->second.second;
    return true;
}
std::string SoDeviceEngine::kernel_names() const {
    std::string s;
    for (auto& kv : impl_->times) s += (s.empty() ? "" : ",") + kv.first;
    return s;
}
size_t SoDeviceEngine::num_incoming_rays() const { return impl_->n_inc_rays; }
size_t SoDeviceEngine::num_incoming_layers() const { return impl_->n_inc_layers; }
size_t SoDeviceEngine::num_accumulation_entries() const { return impl_->n_acc; }
size_t SoDeviceEngine::num_outgoing() const { return impl_->n_out; }

static size_t so_doubles_per_wavelength(const SoDeviceEngine::Impl& I, int nleg) {
    const SoGeo& g = I.g;
    const size_t per = g.S == 3 ? 4 : 1, S = g.S;
    size_t d = size_t(g.ngrid) * (2 + size_t(nleg) * per);
    if (I.multiple_scatter)
        d += I.n_table + size_t(g.ninterior) * g.num_ms_moments * per + 2 * I.n_inc_rays * S + I.n_out * S + I.n_inc_layers + I.n_inc_rays;
    return d;
}
size_t SoDeviceEngine::bytes_per_wavelength() const { return so_doubles_per_wavelength(*impl_, std::max(impl_->cap_nleg, impl_->g.num_ss_moments)) * sizeof(double); }

void SoDeviceEngine::solve_device(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg, const double* d_albedo,
                                  const double* d_irradiance, double* d_radiance, double* d_los_od, void* stream, int accumulate) {
    Impl& I = *impl_;
    SoGeo& g = I.g;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nwavel <= 0 || nleg <= 0) throw std::runtime_error("sasktran2_amd: nwavel and nleg must be positive");
    const int S = g.S, per = S == 3 ? 4 : 1;
    // block of wavelengths in flight: half of the free HBM (plus what this engine already holds), balanced blocks, multiple of the wavefront
    const size_t per_w = so_doubles_per_wavelength(I, nleg) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = size_t(0.5 * double(free_b + I.scratch_bytes));
    if (const char* e = std::getenv("SASKTRAN2_AMD_HBM_BUDGET_MB")) budget = size_t(std::max(1, atoi(e))) << 20;
    size_t fit = std::max<size_t>(kWave, budget / per_w / kWave * kWave);
    if (const char* e = std::getenv("SASKTRAN2_AMD_SO_CHUNK")) fit = std::max<size_t>(kWave, size_t(atoi(e)) / kWave * kWave);
    const size_t nblocks = (size_t(nwavel) + fit - 1) / fit;
    int chunk = int((size_t(nwavel) + nblocks - 1) / nblocks);
    chunk = (chunk + kWave - 1) / kWave * kWave;
    I.last_chunk = chunk;
    if (chunk > I.capacity || nleg > I.cap_nleg) {
        I.scratch.release();
        I.capacity = std::max(chunk, I.capacity);
        I.cap_nleg = std::max(nleg, I.cap_nleg);
        const size_t W = I.capacity, nlg = size_t(I.cap_nleg) * per;
        SoBufs& b = I.b;
        b.kT = I.scratch.alloc<double>(size_t(g.ngrid) * W);
        b.omT = I.scratch.alloc<double>(size_t(g.ngrid) * W);
        b.legT = I.scratch.alloc<double>(size_t(g.ngrid) * nlg * W);
        if (I.multiple_scatter) {
            b.tab = I.scratch.alloc<double>(I.n_table * W);
            b.cl = I.scratch.alloc<double>(size_t(g.ninterior) * g.num_ms_moments * per * W);
            b.first = I.scratch.alloc<double>(I.n_inc_rays * S * W);
            b.incoming = I.scratch.alloc<double>(I.n_inc_rays * S * W);
            b.outgoing = I.scratch.alloc<double>(I.n_out * S * W);
            b.factor = I.scratch.alloc<double>(I.n_inc_layers * W);
            b.gfactor = I.scratch.alloc<double>(I.n_inc_rays * W);
        }
        I.scratch_bytes = so_doubles_per_wavelength(I, I.cap_nleg) * sizeof(double) * W;
    }
    for (int w0 = 0; w0 < nwavel; w0 += chunk) {
        SoBufs b = I.b;
        b.nw = std::min(chunk, nwavel - w0);
        b.nwp = (b.nw + kWave - 1) / kWave * kWave;
        b.nleg = nleg;
        b.w0 = w0;
        b.nwavel_total = nwavel;
        b.ext = d_ext + size_t(w0) * g.ngrid;
        b.ssa = d_ssa + size_t(w0) * g.ngrid;
        b.leg = d_leg + size_t(w0) * g.ngrid * nleg * per;
        b.albedo = d_albedo + w0;
        b.irradiance = d_irradiance ? d_irradiance + w0 : nullptr;
        b.radiance = d_radiance;
        b.los_od = d_los_od;
        b.accumulate = accumulate;
        const unsigned wb = unsigned(b.nwp / kWave);
        {
            SoTimer t(I, st, "so_transpose");
            hipLaunchKernelGGL(k_so_transpose, dim3(2048), dim3(256), 0, st, g, b);
        }
        const int ss = I.single_scatter ? 1 : 0, ms = I.multiple_scatter ? 1 : 0;
        if (ms) {
            {
                SoTimer t(I, st, "so_tables");
                hipLaunchKernelGGL(k_so_tables, dim3(wb, unsigned(g.ntable + g.ninterior * g.num_ms_moments * per)), dim3(kWave), 0, st, g, b);
            }
            {
                SoTimer t(I, st, "so_march_incoming");
                if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<1, false>), dim3(wb, unsigned(g.inc.nrays)), dim3(kWave), 0, st, g, b, 1, 1);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<3, false>), dim3(wb, unsigned(g.inc.nrays)), dim3(kWave), 0, st, g, b, 1, 1);
            }
            for (int it = 0; it < I.num_iterations; ++it) {
                if (it > 0) {
                    SoTimer t(I, st, "so_accumulate");
                    if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<1>), dim3(wb, unsigned(g.inc.nrays)), dim3(kWave), 0, st, g, b);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<3>), dim3(wb, unsigned(g.inc.nrays)), dim3(kWave), 0, st, g, b);
                }
                SoTimer t(I, st, "so_scatter");
                if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<1>), dim3(wb, unsigned(I.max_rows), unsigned(g.npoints)), dim3(kWave), 0, st, g, b);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<3>), dim3(wb, unsigned(I.max_rows), unsigned(g.npoints)), dim3(kWave), 0, st, g, b);
            }
        }
        {
            SoTimer t(I, st, "so_march_los");
            if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<1, true>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<3, true>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
        }
        SO_HIP_CHECK(hipGetLastError());
    }
}

}  // namespace skdev
