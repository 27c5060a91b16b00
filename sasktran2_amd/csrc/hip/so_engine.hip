// HIP kernels + device driver of the exact single-scatter source and the successive-orders ("HR", legacy) multiple-scatter
// source (gfx950).  Reference stages and where they run here:
//
//   solar transmission at every layer boundary        cpp/lib/solar/singlescattersource.cpp:103-160           k_so_layers (table) / k_so_march (exact rows)
//   single-scatter layer source, ground term           singlescattersource.cpp:574-636,950-1167,
//                                                      cpp/include/sasktran2/solartransmission.h:733-846      k_so_layers / k_so_march
//   first-order incoming radiance + accumulation       cpp/lib/sourceintegrator/sourceintegrator.cpp:666-760   k_so_layers (per layer) + k_so_scan (along the ray)
//   scattering matrices x incoming                     cpp/lib/hr/diffuse_table.cpp:673-716,775-808,
//                                                      cpp/lib/hr/diffuse_point.cpp:199-296                    k_so_scatter
//   incoming = accumulation x outgoing + first order   diffuse_table.cpp:755-770                               k_so_accumulate
//   LOS integration of both sources                    sourceintegrator.cpp:521-580, diffuse_table.cpp:965-1064  k_so_march
//
// MI355X formulation: a LANE IS A WAVELENGTH.  Every per-wavelength array is stored [item][wavelength] (wavelength fastest), so the
// 64 lanes of a wavefront read 512 contiguous bytes; every geometric quantity (layer stencils, interpolation weights, CSR rows of the
// accumulation matrix, Wigner functions of a ray's scattering angle, scattering-matrix basis) depends on the ray / layer / direction
// only, is wave-uniform and comes through the scalar cache.  No LDS, no atomics, no divergence except at ray ends.  The accumulation
// matrix is never stored: an entry is (geometric weight, scalar) x (layer factor omega (1 - e^-od) e^-od_before, one double per layer
// and wavelength), which cuts the per-wavelength HBM stream of an iteration from 8 B per matrix entry to 8 B per ray layer.
#include <hip/hip_runtime.h>

#include <cmath>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
constexpr int kPad = 128;   // wavelengths of a block are padded to two wavefronts (k_so_scatter serves two slabs per wave)
constexpr int kLayersPerBlock = 2;

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
    const double *ground_mu_out, *ground_phi;   // outgoing zenith cosine / azimuth difference at the ground (ground_scattering_geometry, singlescattersource.cpp:264-279)
    DevCsr od;
    const int* stencil_idx;                 // [layer][4]
    const double *w_entrance, *w_exit;      // [layer][4]
    const double *quad_start, *quad_end;    // [layer]
    const double *frac_start, *frac_end, *distance;   // [layer] quadrature fractions and length (emission sources)
    const unsigned char* tiny;
    const int* mid_idx;                     // [layer][2]
    const double* mid_w;
    DevCsr solar;                           // per endpoint
    const unsigned char* solar_blocked;
    const double *C2, *S2, *d00, *d02;      // per ray; d00 / d02 [ray][num_ss_moments]
};
// everything k_so_layers needs of one incoming-ray layer in one 256-byte record (four s_load_dwordx16): one scalar round trip instead
// of a chain of dependent index loads
struct LayerRec {
    int q[4];            // grid stencil (union of the two boundaries), padded with 0
    int mid[2];          // atmosphere stencil of the mid point
    int tx[4], te[4];    // solar-table stencil of the exit / entrance boundary
    int ray, tiny;
    double wod[4], went[4], wexit[4], midw[2], quad_start, quad_end, twx[4], twe[4];
};
static_assert(sizeof(LayerRec) == 256, "LayerRec layout");
// six accumulation-matrix entries that share one factor slot (a ray layer: 2 altitudes x 3 nearest directions; padded with weight 0)
struct AccGroup {
    int col[6];
    int slot, pad;
    double w[6];
};
static_assert(sizeof(AccGroup) == 80, "AccGroup layout");

struct SoGeo {
    int ngrid, nlos, S, num_ss_moments, num_ms_moments;
    int emission, occultation;              // line-of-sight sources without scattering (skso::Settings)
    const LayerRec* inc_rec;
    const AccGroup* acc_group;
    const int* acc_group_start;             // [incoming rays + 1]
    int npoints, ninterior, nin, nout;
    DevRays los, inc;
    DevCsr table;                           // solar optical depth of the table points over the grid
    int ntable;
    DevCsr acc;                             // accumulation rows per incoming ray
    const int* acc_layer;                   // factor slot of an entry: its ray layer, or nlayers + ray for the ground term
    const int* inc_layer_ray;               // [incoming layers] ray of a layer
    int n_inc_layers, n_out_total;
    DevCsr los_source, los_ground;
    const int *in_start, *out_start, *point_leg_idx;
    const double* point_leg_w;
    const double* scat_basis;               // [row][col][num_ms_moments * per]
    const double *stokes_C, *stokes_S;      // [nlos] rotation of (Q, U) to the configured Stokes basis
    // scalar scattering through spherical-harmonic space (see k_so_sh): nlm = num_ms_moments^2 real harmonics
    int nlm, nout_pad;
    const double* sh_in;                    // [nin][nlm]      w_i Y_lm(in_i) 4 pi / (2l + 1)
    const double* sh_out;                   // [nlm][nout_pad] Y_lm(out_o)
    const int* sh_l;                        // [nlm] degree l of harmonic lm
    const double* ground_factor;            // concatenated per ground point
    const int* ground_factor_start;         // [nground + 1]
};
// per-block scratch.  SLAB-MAJOR layout: every array is [slab][item][64] (slab = 64 consecutive wavelengths = one wavefront), so the
// working set of a slab -- the directions of `outgoing` all rays gather from, the solar table, the optics -- is one contiguous range
// (2.8 MB of `outgoing` at the C5 size).  [item][wavelength] with a 160 KB row stride put the 5 549 rows of a slab on a fraction of the
// L2 sets and the gathers missed (PMC: 2.8 x the algorithmic HBM reads in k_so_accumulate).
struct SoBufs {
    int nw, nwp, nleg, w0, nwavel_total;
    const double *ext, *ssa, *leg, *albedo, *irradiance;   // caller's inputs at the first wavelength of the block
    // non-Lambertian surface: pi x brdf at the ground geometry of every line of sight, [los][nwavel_total] (whole spectrum), or null.  Only the
    // direct bounce of the lines of sight uses it (end_of_ray_source, cpp/lib/solar/singlescattersource.cpp:574-636)
    const double* los_albedo;
    const double *emission, *surface_emission;             // emission_source [w][q], surface emission [w] (emission sources only)
    double* emT;                                            // emission_source, slab major
    double *kT, *omT, *legT, *tab, *cl, *first, *incoming, *outgoing, *factor, *lod, *lsrc;   // factor: [layers + rays] (ground slots last)
    double* sh;                              // [interior point][nlm] c_l x harmonic coefficients of the incoming field
    double *radiance, *los_od;                               // caller's outputs (whole spectrum)
    int accumulate;
    // native weighting functions of the sources without a diffuse field (exact single scatter, emission, occultation): per-thread
    // accumulators [slab][ray][S][3 ngrid + 2][64] -- extinction, ssa, albedo, emission profile, surface emission -- and the caller's outputs
    // wf_ext / wf_ssa / wf_em [ngrid][nwavel][nlos][S], wf_alb / wf_sem [nwavel][nlos][S] (the two emission outputs may be null)
    double *dacc, *wf_ext, *wf_ssa, *wf_alb, *wf_em, *wf_sem;
    // scattering-group derivatives of the single-scatter source (mappings with d_legendre / scat_factor): ngroups directions of the
    // Legendre / Greek coefficients, dleg [group][nwavel_total][ngrid][nleg * per] (caller's layout), their phase values at the lines of
    // sight's scattering angles dphT [slab][((group * nlos + ray) * ngrid + q) * ncomp + c][64] (c = 0: P11 part, 1: the P12 part of
    // NSTOKES = 3), ngroups * ngrid more accumulator slots per (ray, Stokes component) and the caller's output
    // wf_grp [group][ngrid][nwavel][nlos][S]
    int ngroups;
    const double* dleg;
    double *dphT, *wf_grp;
};
__host__ __device__ inline int so_deriv_slots(int ngrid, int ngroups) { return 3 * ngrid + 2 + ngroups * ngrid; }

// pointers of one slab, advanced to this lane: element `item` of array X is X[item * kWave]
__device__ inline SoBufs slab_view(const SoGeo& g, const SoBufs& b, int slab, int lane) {
    SoBufs v = b;
    const size_t o = size_t(slab) * kWave, per = g.S == 3 ? 4 : 1, S = g.S;
    v.kT += o * g.ngrid + lane;
    v.omT += o * g.ngrid + lane;
    v.legT += o * g.ngrid * b.nleg * per + lane;
    if (v.emT) v.emT += o * g.ngrid + lane;
    v.tab += o * g.ntable + lane;
    v.cl += o * g.ninterior * g.num_ms_moments * per + lane;
    v.first += o * g.inc.nrays * S + lane;
    v.incoming += o * g.inc.nrays * S + lane;
    v.outgoing += o * g.n_out_total * S + lane;
    v.factor += o * (g.n_inc_layers + g.inc.nrays) + lane;
    v.lod += o * g.n_inc_layers + lane;
    v.lsrc += o * g.n_inc_layers * S + lane;
    if (v.sh) v.sh += o * g.ninterior * g.nlm + lane;
    if (v.dacc) v.dacc += o * g.nlos * S * so_deriv_slots(g.ngrid, b.ngroups) + lane;
    if (v.dphT) v.dphT += o * size_t(b.ngroups) * g.nlos * g.ngrid * (S == 3 ? 2 : 1) + lane;
    return v;
}

// inputs [w][q] -> [q][w]; Legendre / Greek coefficients [w][q][lg] -> [q][lg][w]
__global__ void __launch_bounds__(256) k_so_transpose(SoGeo g, SoBufs b) {
    const int nlg = b.nleg * (g.S == 3 ? 4 : 1);
    const size_t nk = g.ngrid, nl = size_t(g.ngrid) * nlg, per_w = 2 * nk + nl, n = per_w * b.nwp;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int lane = int(i % kWave);
        const size_t t = i / kWave, slab = t / per_w, item = t % per_w;   // item: extinction | ssa | Legendre (q * nlg + lg)
        const int w = int(slab) * kWave + lane;
        const int ws = w < b.nw ? w : b.nw - 1;   // padding lanes repeat the last wavelength (keeps every lane finite)
        if (item < nk) b.kT[(slab * nk + item) * kWave + lane] = b.ext[size_t(ws) * g.ngrid + item];
        else if (item < 2 * nk) b.omT[(slab * nk + (item - nk)) * kWave + lane] = b.ssa[size_t(ws) * g.ngrid + (item - nk)];
        else b.legT[(slab * nl + (item - 2 * nk)) * kWave + lane] = b.leg[size_t(ws) * nl + (item - 2 * nk)];
        if (item < nk && b.emT) b.emT[(slab * nk + item) * kWave + lane] = b.emission[size_t(ws) * g.ngrid + item];
    }
}

// solar optical depth at the table points; Legendre coefficients of the diffuse points
__global__ void __launch_bounds__(kWave) k_so_tables(SoGeo g, SoBufs b) {
    const int item = blockIdx.y;
    const SoBufs v = slab_view(g, b, blockIdx.x, threadIdx.x);
    if (item < g.ntable) {
        double od = 0;
        for (int e = g.table.start[item]; e < g.table.start[item + 1]; ++e) od += g.table.w[e] * v.kT[size_t(g.table.idx[e]) * kWave];
        v.tab[size_t(item) * kWave] = od;
        return;
    }
    const int per = g.S == 3 ? 4 : 1, nlgB = g.num_ms_moments * per, nlg_in = b.nleg * per;
    const int t = item - g.ntable, p = t / nlgB, lg = t % nlgB;
    double c = 0;
    if (lg < nlg_in)
        for (int i = 0; i < 2; ++i)
            c += g.point_leg_w[2 * p + i] * v.legT[(size_t(g.point_leg_idx[2 * p + i]) * nlg_in + lg) * kWave];
    v.cl[size_t(t) * kWave] = c;
}

// One ray, 64 wavelengths per wavefront, marched from the observer outward.
// INCOMING: first-order incoming radiance of a diffuse point direction (table solar transmission) + the layer factors of the accumulation matrix.
// LOS: radiance of a line of sight = single scatter (exact solar transmission) + diffuse source interpolated from `outgoing`.
template <int S, bool LOS>
__global__ void __launch_bounds__(kWave) k_so_march(SoGeo g, SoBufs b, int single_scatter, int multiple_scatter) {
    const DevRays& rs = LOS ? g.los : g.inc;
    const int w = blockIdx.x * kWave + threadIdx.x, r = blockIdx.y;
    const SoBufs v = slab_view(g, b, blockIdx.x, threadIdx.x);
    const int l0 = rs.layer_start[r], n = rs.layer_start[r + 1] - l0, e0 = rs.endpoint_start[r];
    const int per = S == 3 ? 4 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg);
    const int ws = w < b.nw ? w : b.nw - 1;
    const double F = b.irradiance ? b.irradiance[ws] : 1.0;
    const double albedo = (LOS && b.los_albedo) ? b.los_albedo[size_t(r) * b.nwavel_total + b.w0 + ws] : b.albedo[ws];
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
        const double* lq = v.legT + size_t(q) * nlg_in * kWave;
        for (int l = 0; l < nm; ++l) {
            s1 += lq[size_t(l) * per * kWave] * rs.d00[size_t(r) * g.num_ss_moments + l];
            if (S == 3) s2 += lq[size_t(l * per + 3) * kWave] * rs.d02[size_t(r) * g.num_ss_moments + l];
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
        const double* x = LOS ? v.kT : v.tab;
        double od = 0;
        for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) od += rs.solar.w[t] * x[size_t(rs.solar.idx[t]) * kWave];
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
            ke += v.kT[size_t(q) * kWave] * wt;
            oe += v.omT[size_t(q) * kWave] * wt;
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
    const bool emit = LOS && g.emission != 0;
    if (n > 0 && single_scatter) T_entrance = transmission(e0 + n);
    for (int j = n - 1; j >= 0; --j) {
        const int L = l0 + j;
        double od = 0;
        for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t) od += rs.od.w[t] * v.kT[size_t(rs.od.idx[t]) * kWave];
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
        if (emit && !rs.tiny[L]) {   // EmissionSource::integrated_source (emission_source.cpp:37-100)
            double oms = 0, ome = 0, es = 0, ee = 0;
            for (int i = 0; i < 4; ++i) {
                const int q = rs.stencil_idx[4 * L + i];
                if (q < 0) continue;
                const double om_q = v.omT[size_t(q) * kWave], em_q = v.emT[size_t(q) * kWave];
                oms += om_q * rs.w_entrance[4 * L + i]; es += em_q * rs.w_entrance[4 * L + i];
                ome += om_q * rs.w_exit[4 * L + i]; ee += em_q * rs.w_exit[4 * L + i];
            }
            val[0] += before * (g.emission == 1 ? (1 - att) * ((1 - oms) * es * rs.frac_start[L] + (1 - ome) * ee * rs.frac_end[L])
                                                : rs.distance[L] * (es * rs.frac_start[L] + ee * rs.frac_end[L]));
        }
        if (multiple_scatter) {
            const double omega = rs.mid_w[2 * L] * v.omT[size_t(rs.mid_idx[2 * L]) * kWave] + rs.mid_w[2 * L + 1] * v.omT[size_t(rs.mid_idx[2 * L + 1]) * kWave];
            const double f = omega * (1 - att) * before;
            if (LOS) {
                for (int t = g.los_source.start[L]; t < g.los_source.start[L + 1]; ++t) {
                    const double fw = f * g.los_source.w[t];
                    const double* o = v.outgoing + size_t(g.los_source.idx[t]) * S * kWave;
#pragma unroll
                    for (int s = 0; s < S; ++s) val[s] += fw * o[size_t(s) * kWave];
                }
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
                    const double* o = v.outgoing + size_t(g.los_ground.idx[t]) * S * kWave;
#pragma unroll
                    for (int s = 0; s < S; ++s) val[s] += fw * o[size_t(s) * kWave];
                }
            }
        }
    }
    if (LOS) {
        // end-of-ray sources without scattering: surface emission behind a ray that ends on the ground, unit source of an occultation
        // behind one that leaves the atmosphere
        if (emit && n > 0 && rs.ground_is_hit[r] && b.surface_emission) val[0] += b.surface_emission[ws] * before;
        if (g.occultation && !(n > 0 && rs.ground_is_hit[r])) val[0] += before;
        if (w >= b.nw) return;
        double* out = b.radiance + (size_t(b.w0 + w) * g.nlos + r) * S;
        if (S == 3) {   // configured Stokes basis (OutputC::assign_lane)
            const double C = g.stokes_C[r], Sn = g.stokes_S[r], q = val[1], u = val[S - 1];
            val[1] = C * q - Sn * u;
            val[S - 1] = Sn * q + C * u;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) out[s] = b.accumulate ? out[s] + val[s] : val[s];
        if (b.los_od) b.los_od[size_t(b.w0 + w) * g.nlos + r] = total_od;
    }
}

// Native weighting functions of the exact single-scatter radiance of a line of sight (d/dk_q, d/d omega_q, d/d albedo), reverse mode:
// with B_j the transmission from layer j to the observer, dI = sum_j B_j (dS_j - I_{j-1} e^{-od_j} d od_j) + ground, where dS_j runs through
// the source factor (od row), the solar transmission rows of both boundaries and the local extinction / single scatter albedo
// (singlescattersource.cpp:988-1030, solartransmission.h:852-901, sourceintegrator.cpp:536-579).  One line of sight x 64 wavelengths per
// wavefront; every lane accumulates into its own column of `dacc` (coalesced read-modify-writes, L1 / L2 resident).
template <int S>
__global__ void __launch_bounds__(kWave) k_so_los_deriv(SoGeo g, SoBufs b, int single_scatter, int multiple_scatter) {
    const DevRays& rs = g.los;
    const int w = blockIdx.x * kWave + threadIdx.x, r = blockIdx.y;
    const SoBufs v = slab_view(g, b, blockIdx.x, threadIdx.x);
    const int l0 = rs.layer_start[r], n = rs.layer_start[r + 1] - l0, e0 = rs.endpoint_start[r];
    const int per = S == 3 ? 4 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg), ngrid = g.ngrid, nn = so_deriv_slots(ngrid, b.ngroups);
    const int ncomp = S == 3 ? 2 : 1, grp0 = 3 * ngrid + 2;
    const int ws = w < b.nw ? w : b.nw - 1;
    const double F = b.irradiance ? b.irradiance[ws] : 1.0;
    const double albedo = b.los_albedo ? b.los_albedo[size_t(r) * b.nwavel_total + b.w0 + ws] : b.albedo[ws];   // (slot 2 ngrid: d / d of this value)
    double* acc = v.dacc + size_t(r) * S * nn * kWave;
    for (int x = 0; x < S * nn; ++x) acc[size_t(x) * kWave] = 0.0;
    if (n == 0) return;
    auto add = [&](int s, int x, double val) { acc[(size_t(s) * nn + x) * kWave] += val; };
    auto layer_od = [&](int L) {
        double od = 0;
        for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t) od += rs.od.w[t] * v.kT[size_t(rs.od.idx[t]) * kWave];
        return od;
    };
    auto transmission = [&](int e) -> double {
        if (!single_scatter || rs.solar_blocked[e]) return 0.0;
        double od = 0;
        for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) od += rs.solar.w[t] * v.kT[size_t(rs.solar.idx[t]) * kWave];
        return exp(-od) * F;
    };
    double total = 0;
    for (int j = 0; j < n; ++j) total += layer_od(l0 + j);
    double I[S];
#pragma unroll
    for (int s = 0; s < S; ++s) I[s] = 0;
    double prefix = 0;                       // optical depth of the layers behind the one in hand
    double T_exit = transmission(e0);        // far end of the ray
    // end-of-ray sources without scattering: surface emission (emission_source.cpp:156-173), unit source of an occultation (occultation.cpp:26-35)
    if (g.emission != 0 && rs.ground_is_hit[r] && b.surface_emission) {
        I[0] += b.surface_emission[ws];
        add(0, 3 * ngrid + 1, exp(-total));
    }
    if (g.occultation && !rs.ground_is_hit[r]) I[0] += 1.0;
    if (single_scatter && rs.ground_is_hit[r] && rs.ground_mu_in[r] > 0) {
        const double Bg = exp(-total), base = T_exit * rs.ground_mu_in[r] / kPi;
        I[0] += base * albedo;
        add(0, 2 * ngrid, base * Bg);
        for (int t = rs.solar.start[e0]; t < rs.solar.start[e0 + 1]; ++t) add(0, rs.solar.idx[t], -rs.solar.w[t] * base * albedo * Bg);
    }
    // diffuse field: its source along the ray is part of I (the attenuation derivatives below see it), with the direct dependence of
    // omega (1 - e^-od) on the layer; the dependence through the outgoing sources themselves is the adjoint of so_adjoint.inc
    if (multiple_scatter && rs.ground_is_hit[r])
        for (int e = g.los_ground.start[r]; e < g.los_ground.start[r + 1]; ++e)
#pragma unroll
            for (int s = 0; s < S; ++s) I[s] += g.los_ground.w[e] * v.outgoing[(size_t(g.los_ground.idx[e]) * S + s) * kWave];
    for (int j = 0; j < n; ++j) {
        const int L = l0 + j;
        const double od = layer_od(L), att = exp(-od);
        const double B_in = exp(-(total - prefix));   // transmission of what enters this layer from behind, through it, to the observer
        prefix += od;
        const double Bj = exp(-(total - prefix));
        for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t)
#pragma unroll
            for (int s = 0; s < S; ++s) add(s, rs.od.idx[t], -rs.od.w[t] * I[s] * B_in);
#pragma unroll
        for (int s = 0; s < S; ++s) I[s] *= att;
        if (multiple_scatter) {
            const double omega = rs.mid_w[2 * L] * v.omT[size_t(rs.mid_idx[2 * L]) * kWave] + rs.mid_w[2 * L + 1] * v.omT[size_t(rs.mid_idx[2 * L + 1]) * kWave];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double X = 0;
                for (int e = g.los_source.start[L]; e < g.los_source.start[L + 1]; ++e) X += g.los_source.w[e] * v.outgoing[(size_t(g.los_source.idx[e]) * S + s) * kWave];
                I[s] += omega * (1 - att) * X;
                for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t) add(s, rs.od.idx[t], rs.od.w[t] * omega * att * X * Bj);
                add(s, ngrid + rs.mid_idx[2 * L], rs.mid_w[2 * L] * (1 - att) * X * Bj);
                add(s, ngrid + rs.mid_idx[2 * L + 1], rs.mid_w[2 * L + 1] * (1 - att) * X * Bj);
            }
        }
        const double T_ent = transmission(e0 + j + 1);
        if (g.emission != 0 && !rs.tiny[L]) {
            // EmissionSource::integrated_source with its derivative rows (emission_source.cpp:37-152)
            double oms = 0, ome = 0, es = 0, ee = 0;
            for (int i = 0; i < 4; ++i) {
                const int q = rs.stencil_idx[4 * L + i];
                if (q < 0) continue;
                const double om_q = v.omT[size_t(q) * kWave], em_q = v.emT[size_t(q) * kWave];
                oms += om_q * rs.w_entrance[4 * L + i]; es += em_q * rs.w_entrance[4 * L + i];
                ome += om_q * rs.w_exit[4 * L + i]; ee += em_q * rs.w_exit[4 * L + i];
            }
            const bool standard = g.emission == 1;
            const double factor = standard ? 1 - att : rs.distance[L];
            const double cell = standard ? (1 - oms) * es * rs.frac_start[L] + (1 - ome) * ee * rs.frac_end[L] : es * rs.frac_start[L] + ee * rs.frac_end[L];
            I[0] += factor * cell;
            if (standard)
                for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t) add(0, rs.od.idx[t], rs.od.w[t] * att * cell * Bj);
            for (int i = 0; i < 4; ++i) {
                const int q = rs.stencil_idx[4 * L + i];
                if (q < 0) continue;
                const double wsn = rs.w_entrance[4 * L + i] * rs.frac_start[L], wen = rs.w_exit[4 * L + i] * rs.frac_end[L];
                if (standard) add(0, ngrid + q, -factor * (wsn * es + wen * ee) * Bj);
                add(0, 2 * ngrid + 1 + q, factor * (wsn * (standard ? 1 - oms : 1.0) + wen * (standard ? 1 - ome : 1.0)) * Bj);
            }
        }
        if (single_scatter && !rs.tiny[L]) {
            const double sf = (fabs(od) < 1e-12) ? 1.0 : -expm1(-od) / od;
            const double dsf = (fabs(od) < 1e-12) ? -0.5 : 1 / od - sf * (1 + 1 / od);
            double quad_sum[S];
#pragma unroll
            for (int s = 0; s < S; ++s) quad_sum[s] = 0;
            for (int side = 0; side < 2; ++side) {   // 0: entrance, 1: exit
                const double* wts = side == 0 ? rs.w_entrance : rs.w_exit;
                const double quad = side == 0 ? rs.quad_start[L] : rs.quad_end[L], T = side == 0 ? T_ent : T_exit;
                const int e = e0 + j + (side == 0 ? 1 : 0);
                double ke = 0, oe = 0, ph[S];
#pragma unroll
                for (int s = 0; s < S; ++s) ph[s] = 0;
                for (int i = 0; i < 4; ++i) {
                    const int q = rs.stencil_idx[4 * L + i];
                    const double wt = wts[4 * L + i];
                    if (q < 0 || wt == 0.0) continue;
                    double p11 = 0, p12 = 0;
                    const double* lq = v.legT + size_t(q) * nlg_in * kWave;
                    for (int l = 0; l < nm; ++l) {
                        p11 += lq[size_t(l) * per * kWave] * rs.d00[size_t(r) * g.num_ss_moments + l];
                        if (S == 3) p12 += lq[size_t(l * per + 3) * kWave] * rs.d02[size_t(r) * g.num_ss_moments + l];
                    }
                    ke += v.kT[size_t(q) * kWave] * wt;
                    oe += v.omT[size_t(q) * kWave] * wt;
                    ph[0] += p11 * wt;
                    if (S == 3) { ph[1] += -rs.C2[r] * p12 * wt; ph[2] += -rs.S2[r] * p12 * wt; }
                }
                const double unscaled = T / (kPi * 4), c = sf * quad * Bj;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const double value = ke * oe * unscaled * ph[s];
                    quad_sum[s] += value * quad;
                    I[s] += sf * value * quad;
                    if (!rs.solar_blocked[e])
                        for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) add(s, rs.solar.idx[t], -c * rs.solar.w[t] * value);
                    for (int i = 0; i < 4; ++i) {
                        const int q = rs.stencil_idx[4 * L + i];
                        const double wt = wts[4 * L + i];
                        if (q < 0 || wt == 0.0) continue;
                        add(s, ngrid + q, c * wt * ke * unscaled * ph[s]);
                        add(s, q, c * wt * oe * unscaled * ph[s]);
                    }
                }
                // scattering groups: the phase function moves along the group's direction of the coefficients at every stencil point
                for (int gi = 0; gi < b.ngroups; ++gi)
                    for (int i = 0; i < 4; ++i) {
                        const int q = rs.stencil_idx[4 * L + i];
                        const double wt = wts[4 * L + i];
                        if (q < 0 || wt == 0.0) continue;
                        const double* dp = v.dphT + (size_t(gi * g.nlos + r) * ngrid + q) * ncomp * kWave;
                        const double base = c * wt * ke * oe * unscaled;
                        add(0, grp0 + gi * ngrid + q, base * dp[0]);
                        if (S == 3) {
                            add(1, grp0 + gi * ngrid + q, -rs.C2[r] * base * dp[kWave]);
                            add(2, grp0 + gi * ngrid + q, -rs.S2[r] * base * dp[kWave]);
                        }
                    }
            }
            for (int t = rs.od.start[L]; t < rs.od.start[L + 1]; ++t)
#pragma unroll
                for (int s = 0; s < S; ++s) add(s, rs.od.idx[t], rs.od.w[t] * dsf * quad_sum[s] * Bj);
        }
        T_exit = T_ent;
    }
}

// phase values of the scattering groups' directions at the scattering angle of every line of sight:
//   dphT[group][ray][q][0] = sum_l dleg_l d00_l(ray),   [1] = sum_l dleg(b1)_l d02_l(ray)  (NSTOKES = 3), slab major.
// One (group, grid point) x 64 wavelengths per wavefront; every lane reads its own contiguous row of coefficients.
__global__ void __launch_bounds__(kWave) k_so_group_phase(SoGeo g, SoBufs b) {
    const DevRays& rs = g.los;
    const int S = g.S, per = S == 3 ? 4 : 1, ncomp = S == 3 ? 2 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg);
    const int w = blockIdx.x * kWave + threadIdx.x, gi = blockIdx.y / g.ngrid, q = blockIdx.y % g.ngrid;
    const SoBufs v = slab_view(g, b, blockIdx.x, threadIdx.x);
    const int ws = w < b.nw ? w : b.nw - 1;
    const double* row = b.dleg + ((size_t(gi) * b.nwavel_total + b.w0 + ws) * g.ngrid + q) * nlg_in;
    for (int r = 0; r < g.nlos; ++r) {
        double p11 = 0, p12 = 0;
        for (int l = 0; l < nm; ++l) {
            p11 += row[l * per] * rs.d00[size_t(r) * g.num_ss_moments + l];
            if (S == 3) p12 += row[l * per + 3] * rs.d02[size_t(r) * g.num_ss_moments + l];
        }
        double* dp = v.dphT + (size_t(gi * g.nlos + r) * g.ngrid + q) * ncomp * kWave;
        dp[0] = p11;
        if (S == 3) dp[kWave] = p12;
    }
}

// accumulators -> the caller's layout (wf_ext / wf_ssa [ngrid][nwavel][nlos][S], wf_alb [nwavel][nlos][S]) in the configured Stokes basis
__global__ void __launch_bounds__(256) k_so_deriv_out(SoGeo g, SoBufs b) {
    const int S = g.S, nn = so_deriv_slots(g.ngrid, b.ngroups), nlos = g.nlos;
    const size_t n = size_t(nn) * b.nw * nlos * S;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int s = int(i % S), r = int((i / S) % nlos), w = int((i / S / nlos) % b.nw), x = int(i / S / nlos / b.nw);
        const double* a = b.dacc + (size_t(w / kWave) * nlos * S * nn) * kWave + (w % kWave);
        auto at = [&](int sc) { return a[((size_t(r) * S + sc) * nn + x) * kWave]; };
        double val = at(s);
        if (S == 3 && s > 0) {
            const double C = g.stokes_C[r], Sn = g.stokes_S[r], q = at(1), u = at(2);
            val = (s == 1) ? C * q - Sn * u : Sn * q + C * u;
        }
        double* dst;
        if (x < g.ngrid) dst = b.wf_ext + ((size_t(x) * b.nwavel_total + b.w0 + w) * nlos + r) * S + s;
        else if (x < 2 * g.ngrid) dst = b.wf_ssa + ((size_t(x - g.ngrid) * b.nwavel_total + b.w0 + w) * nlos + r) * S + s;
        else if (x == 2 * g.ngrid) dst = b.wf_alb + (size_t(b.w0 + w) * nlos + r) * S + s;
        else if (x <= 3 * g.ngrid) dst = b.wf_em ? b.wf_em + ((size_t(x - 2 * g.ngrid - 1) * b.nwavel_total + b.w0 + w) * nlos + r) * S + s : nullptr;
        else if (x == 3 * g.ngrid + 1) dst = b.wf_sem ? b.wf_sem + (size_t(b.w0 + w) * nlos + r) * S + s : nullptr;
        else dst = b.wf_grp + ((size_t(x - 3 * g.ngrid - 2) * b.nwavel_total + b.w0 + w) * nlos + r) * S + s;   // x - (3 ngrid + 2) = group * ngrid + q
        if (!dst) continue;
        *dst = b.accumulate ? *dst + val : val;
    }
}

// INCOMING RAYS, stage 1: everything of a ray layer that does not depend on the layers in front of it -- optical depth, the
// unattenuated single-scatter source (table solar transmission at both boundaries) and omega (1 - e^-od).  One layer x 64 wavelengths per
// wavefront; grid x = layer so that the wavefronts in flight share one wavelength slab of the optics / table in L2.
// (the tables come as __restrict__ kernel arguments: only then may the compiler keep them on the scalar path next to the stores)
template <int S>
__global__ void __launch_bounds__(kWave) k_so_layers(SoGeo g, SoBufs b, const LayerRec* __restrict__ recs, const double* __restrict__ d00tab,
                                                     const double* __restrict__ d02tab, const double* __restrict__ C2tab,
                                                     const double* __restrict__ S2tab, double* __restrict__ lod_out,
                                                     double* __restrict__ factor_out, double* __restrict__ lsrc_out, int layers_per_block) {
    const int w = blockIdx.y * kWave + threadIdx.x;
    const SoBufs v = slab_view(g, b, blockIdx.y, threadIdx.x);
    lod_out += size_t(blockIdx.y) * kWave * g.n_inc_layers + threadIdx.x;
    factor_out += size_t(blockIdx.y) * kWave * (g.n_inc_layers + g.inc.nrays) + threadIdx.x;
    lsrc_out += size_t(blockIdx.y) * kWave * g.n_inc_layers * S + threadIdx.x;
    const int per = S == 3 ? 4 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg);
    const int ws = w < b.nw ? w : b.nw - 1;
    const double F = b.irradiance ? b.irradiance[ws] : 1.0;
    // a wavefront works through kLayersPerBlock layers: one workgroup per layer and slab (45 million of them at the C5 size) is bound
    // by the dispatcher, not by the arithmetic
    const int L_end = min(int(blockIdx.x + 1) * layers_per_block, g.n_inc_layers);
  for (int L = blockIdx.x * layers_per_block; L < L_end; ++L) {
    const LayerRec& rec = recs[L];
    const int r = rec.ray;
    double kq[4], oq[4], od = 0, tod[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kq[i] = v.kT[size_t(rec.q[i]) * kWave];
        oq[i] = v.omT[size_t(rec.q[i]) * kWave];
        tod[0] += rec.twx[i] * v.tab[size_t(rec.tx[i]) * kWave];
        tod[1] += rec.twe[i] * v.tab[size_t(rec.te[i]) * kWave];
    }
    const double omega = rec.midw[0] * v.omT[size_t(rec.mid[0]) * kWave] + rec.midw[1] * v.omT[size_t(rec.mid[1]) * kWave];
#pragma unroll
    for (int i = 0; i < 4; ++i) od += rec.wod[i] * kq[i];
    const double em1 = expm1(-od);   // 1 - e^-od = -em1 without cancellation, and the only transcendental this layer needs for its own depth
    // streaming stores: 3 doubles per layer and wavelength pass through once; kept out of L2 they do not evict the slab's table / optics
    __builtin_nontemporal_store(od, &lod_out[size_t(L) * kWave]);
    __builtin_nontemporal_store(omega * (-em1), &factor_out[size_t(L) * kWave]);
    double src[S];
#pragma unroll
    for (int s = 0; s < S; ++s) src[s] = 0;
    if (!rec.tiny) {
        const double T[2] = {exp(-tod[0]) * F, exp(-tod[1]) * F};   // exit, entrance
        double ke[2] = {0, 0}, oe[2] = {0, 0}, a[2] = {0, 0}, bb[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double wx = rec.wexit[i], we = rec.went[i];
            if (wx == 0.0 && we == 0.0) continue;
            double p11 = 0, p12 = 0;
            const double* lq = v.legT + size_t(rec.q[i]) * nlg_in * kWave;
            for (int l = 0; l < nm; ++l) {
                p11 += lq[size_t(l) * per * kWave] * d00tab[size_t(r) * g.num_ss_moments + l];
                if (S == 3) p12 += lq[size_t(l * per + 3) * kWave] * d02tab[size_t(r) * g.num_ss_moments + l];
            }
            ke[0] += kq[i] * wx; oe[0] += oq[i] * wx; a[0] += p11 * wx; bb[0] += p12 * wx;
            ke[1] += kq[i] * we; oe[1] += oq[i] * we; a[1] += p11 * we; bb[1] += p12 * we;
        }
        const double sf = (fabs(od) < 1e-12) ? 1.0 : -em1 / od;
        const double amp_x = ke[0] * oe[0] * T[0] / (kPi * 4) * rec.quad_end, amp_e = ke[1] * oe[1] * T[1] / (kPi * 4) * rec.quad_start;
        src[0] = sf * (amp_e * a[1] + amp_x * a[0]);
        if (S == 3) {
            const double pol = sf * (amp_e * bb[1] + amp_x * bb[0]);
            src[1] = -C2tab[r] * pol;
            src[2] = -S2tab[r] * pol;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) __builtin_nontemporal_store(src[s], &lsrc_out[(size_t(L) * S + s) * kWave]);
  }
}

// INCOMING RAYS, stage 2: the scan along a ray from the diffuse point outward -- attenuation in front of every layer, first-order
// incoming radiance, layer factors of the accumulation matrix (in place over omega (1 - e^-od)), ground term.
template <int S>
__global__ void __launch_bounds__(kWave) k_so_scan(SoGeo g, SoBufs b) {
    const DevRays& rs = g.inc;
    const int r = blockIdx.x, w = blockIdx.y * kWave + threadIdx.x;
    const SoBufs v = slab_view(g, b, blockIdx.y, threadIdx.x);
    const int l0 = rs.layer_start[r], n = rs.layer_start[r + 1] - l0;
    double val[S];
#pragma unroll
    for (int s = 0; s < S; ++s) val[s] = 0;
    double cur = 0, before = 1;
#pragma unroll 4
    for (int j = n - 1; j >= 0; --j) {
        const size_t o = size_t(l0 + j) * kWave;
        const double od = __builtin_nontemporal_load(&v.lod[o]), f = __builtin_nontemporal_load(&v.factor[o]);
#pragma unroll
        for (int s = 0; s < S; ++s) val[s] += __builtin_nontemporal_load(&v.lsrc[(size_t(l0 + j) * S + s) * kWave]) * before;
        __builtin_nontemporal_store(f * before, &v.factor[o]);
        cur += od;
        before = exp(-cur);
    }
    double gf = 0;
    if (n > 0 && rs.ground_is_hit[r]) {
        gf = before;
        if (rs.ground_mu_in[r] > 0) {
            const int ws = w < b.nw ? w : b.nw - 1;
            const double F = b.irradiance ? b.irradiance[ws] : 1.0;
            double t_od = 0;
            const int e = rs.endpoint_start[r];
            for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) t_od += rs.solar.w[t] * v.tab[size_t(rs.solar.idx[t]) * kWave];
            val[0] += exp(-t_od) * F * (b.albedo[ws] / kPi) * rs.ground_mu_in[r] * before;
        }
    }
    v.factor[size_t(g.n_inc_layers + r) * kWave] = gf;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        v.first[(size_t(r) * S + s) * kWave] = val[s];
        v.incoming[(size_t(r) * S + s) * kWave] = val[s];
    }
}

// INCOMING RAYS, both stages in one walk along the ray (default; SASKTRAN2_AMD_SO_SPLIT_LAYERS=1 keeps k_so_layers + k_so_scan): one ray x 64
// wavelengths per wavefront, from the diffuse point outward.  Consecutive layers share an endpoint (endpoint e0 + j is the exit of layer j
// and the entrance of layer j - 1): the boundary source  k omega T / 4 pi x phase  -- one exp and one Legendre sum per stencil point --
// is formed once per endpoint and carried to the next layer instead of twice, and the optical depth / source of a layer never travel
// through HBM (k_so_layers wrote lod and lsrc for k_so_scan to read back: 2 + 2 S doubles per layer and wavelength).  Only the layer
// factors of the accumulation matrix are stored.
template <int S>
__global__ void __launch_bounds__(kWave) k_so_ray(SoGeo g, SoBufs b, const LayerRec* __restrict__ recs, const double* __restrict__ d00tab,
                                                  const double* __restrict__ d02tab, const double* __restrict__ C2tab,
                                                  const double* __restrict__ S2tab, double* __restrict__ factor_base) {
    // two wavelength slabs per wavefront: every scalar-cache operand (a layer record is 256 bytes) feeds both
    constexpr int NS = 2;
    const DevRays& rs = g.inc;
    const int r = blockIdx.x;
    const SoBufs v[NS] = {slab_view(g, b, NS * blockIdx.y, threadIdx.x), slab_view(g, b, NS * blockIdx.y + 1, threadIdx.x)};
    double* factor_out[NS];
    double F[NS], albedo[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int w = (NS * blockIdx.y + k) * kWave + threadIdx.x, ws = w < b.nw ? w : b.nw - 1;
        factor_out[k] = factor_base + size_t(NS * blockIdx.y + k) * kWave * (g.n_inc_layers + g.inc.nrays) + threadIdx.x;
        F[k] = b.irradiance ? b.irradiance[ws] : 1.0;
        albedo[k] = b.albedo[ws];
    }
    const int l0 = rs.layer_start[r], n = rs.layer_start[r + 1] - l0;
    const int per = S == 3 ? 4 : 1, nlg_in = b.nleg * per, nm = min(g.num_ss_moments, b.nleg);
    const double* d00 = d00tab + size_t(r) * g.num_ss_moments;
    const double* d02 = d02tab + size_t(r) * g.num_ss_moments;
    // boundary source of one endpoint from the record of a layer it belongs to (side 0: the layer's exit, 1: its entrance)
    auto boundary = [&](const LayerRec& rec, int side, double* Ba, double* Bb) {
        const double* wts = side == 0 ? rec.wexit : rec.went;
        const int* tq = side == 0 ? rec.tx : rec.te;
        const double* tw = side == 0 ? rec.twx : rec.twe;
        double tod[NS], ke[NS], oe[NS], a[NS], bb[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) tod[k] = ke[k] = oe[k] = a[k] = bb[k] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < NS; ++k) tod[k] += tw[i] * v[k].tab[size_t(tq[i]) * kWave];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double wt = wts[i];
            if (wt == 0.0) continue;
            double p11[NS], p12[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) p11[k] = p12[k] = 0;
            for (int l = 0; l < nm; ++l) {
                const double c00 = d00[l], c02 = S == 3 ? d02[l] : 0.0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const double* lq = v[k].legT + size_t(rec.q[i]) * nlg_in * kWave;
                    p11[k] += lq[size_t(l) * per * kWave] * c00;
                    if (S == 3) p12[k] += lq[size_t(l * per + 3) * kWave] * c02;
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                ke[k] += v[k].kT[size_t(rec.q[i]) * kWave] * wt;
                oe[k] += v[k].omT[size_t(rec.q[i]) * kWave] * wt;
                a[k] += p11[k] * wt;
                bb[k] += p12[k] * wt;
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const double amp = ke[k] * oe[k] * (exp(-tod[k]) * F[k]) / (kPi * 4);
            Ba[k] = amp * a[k];
            Bb[k] = amp * bb[k];
        }
    };
    double val[NS][S], cur[NS], before[NS], ent_a[NS], ent_b[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        cur[k] = 0; before[k] = 1; ent_a[k] = ent_b[k] = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) val[k][s] = 0;
    }
    if (n > 0) boundary(recs[l0 + n - 1], 1, ent_a, ent_b);
    for (int j = n - 1; j >= 0; --j) {
        const LayerRec& rec = recs[l0 + j];
        double ex_a[NS], ex_b[NS];
        boundary(rec, 0, ex_a, ex_b);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            double od = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) od += rec.wod[i] * v[k].kT[size_t(rec.q[i]) * kWave];
            const double omega = rec.midw[0] * v[k].omT[size_t(rec.mid[0]) * kWave] + rec.midw[1] * v[k].omT[size_t(rec.mid[1]) * kWave];
            const double em1 = expm1(-od);
            if (!rec.tiny) {
                const double sf = (fabs(od) < 1e-12) ? 1.0 : -em1 / od;
                val[k][0] += sf * (rec.quad_start * ent_a[k] + rec.quad_end * ex_a[k]) * before[k];
                if (S == 3) {
                    const double pol = sf * (rec.quad_start * ent_b[k] + rec.quad_end * ex_b[k]) * before[k];
                    val[k][1] += -C2tab[r] * pol;
                    val[k][2] += -S2tab[r] * pol;
                }
            }
            __builtin_nontemporal_store(omega * (-em1) * before[k], &factor_out[k][size_t(l0 + j) * kWave]);
            cur[k] += od;
            before[k] = exp(-cur[k]);
            ent_a[k] = ex_a[k];
            ent_b[k] = ex_b[k];
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        double gf = 0;
        if (n > 0 && rs.ground_is_hit[r]) {
            gf = before[k];
            if (rs.ground_mu_in[r] > 0) {
                double t_od = 0;
                const int e = rs.endpoint_start[r];
                for (int t = rs.solar.start[e]; t < rs.solar.start[e + 1]; ++t) t_od += rs.solar.w[t] * v[k].tab[size_t(rs.solar.idx[t]) * kWave];
                val[k][0] += exp(-t_od) * F[k] * (albedo[k] / kPi) * rs.ground_mu_in[r] * before[k];
            }
        }
        factor_out[k][size_t(g.n_inc_layers + r) * kWave] = gf;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v[k].first[(size_t(r) * S + s) * kWave] = val[k][s];
            v[k].incoming[(size_t(r) * S + s) * kWave] = val[k][s];
        }
    }
}

// outgoing source of one (point, outgoing row) for 64 wavelengths: row of [sum_l c_l B_l] x incoming (interior points),
// Lambertian reflection of the downwelling flux (ground points)
template <int S>
__global__ void __launch_bounds__(kWave) k_so_scatter(SoGeo g, SoBufs b, int p0) {
    // two wavelength slabs per wavefront: every scalar-cache operand (one element of the basis) feeds two FMAs
    const int w = blockIdx.y * (2 * kWave) + threadIdx.x, row = blockIdx.x, p = p0 + blockIdx.z;
    const SoBufs v = slab_view(g, b, 2 * blockIdx.y, threadIdx.x), v2 = slab_view(g, b, 2 * blockIdx.y + 1, threadIdx.x);
    const int nrow = (g.out_start[p + 1] - g.out_start[p]) * S, ncol = (g.in_start[p + 1] - g.in_start[p]) * S;
    if (row >= nrow) return;
    const double* __restrict__ in = v.incoming + size_t(g.in_start[p]) * S * kWave;
    const double* __restrict__ in2 = v2.incoming + size_t(g.in_start[p]) * S * kWave;
    double* out = v.outgoing + (size_t(g.out_start[p]) * S + row) * kWave;
    double* out2 = v2.outgoing + (size_t(g.out_start[p]) * S + row) * kWave;
    if (p >= g.ninterior) {
        double v0 = 0, v1 = 0;
        if (row % S == 0) {
            const int wa = min(w, b.nw - 1), wb2 = min(w + kWave, b.nw - 1);
            const double* f = g.ground_factor + g.ground_factor_start[p - g.ninterior];
            for (int i = 0; i < ncol / S; ++i) {
                v0 += f[i] * in[size_t(i) * S * kWave];
                v1 += f[i] * in2[size_t(i) * S * kWave];
            }
            v0 *= b.albedo[wa] / kPi;
            v1 *= b.albedo[wb2] / kPi;
        }
        out[0] = v0;
        out2[0] = v1;
        return;
    }
    const int per = S == 3 ? 4 : 1, nlgB = g.num_ms_moments * per, nlg = min(g.num_ms_moments, b.nleg) * per;
    const double* __restrict__ cl = v.cl + size_t(p) * nlgB * kWave;
    const double* __restrict__ cl2 = v2.cl + size_t(p) * nlgB * kWave;
    const double* __restrict__ B = g.scat_basis + size_t(row) * ncol * nlgB;
    double sum0 = 0, sum1 = 0;
    for (int c0 = 0; c0 < nlg; c0 += 16) {
        double acc0[16], acc1[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc0[t] = acc1[t] = 0;
        const int nt = min(16, nlg - c0);
        if (nt == 16) {
#pragma unroll 2
            for (int i = 0; i < ncol; ++i) {
                const double x0 = in[size_t(i) * kWave], x1 = in2[size_t(i) * kWave];
                const double* Bi = B + size_t(i) * nlgB + c0;
#pragma unroll
                for (int t = 0; t < 16; ++t) { acc0[t] += Bi[t] * x0; acc1[t] += Bi[t] * x1; }
            }
        } else {
            for (int i = 0; i < ncol; ++i) {
                const double x0 = in[size_t(i) * kWave], x1 = in2[size_t(i) * kWave];
                const double* Bi = B + size_t(i) * nlgB + c0;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < nt) { acc0[t] += Bi[t] * x0; acc1[t] += Bi[t] * x1; }
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < nt) { sum0 += cl[size_t(c0 + t) * kWave] * acc0[t]; sum1 += cl2[size_t(c0 + t) * kWave] * acc1[t]; }
    }
    out[0] = sum0;
    out2[0] = sum1;
}

// Scalar scattering of the interior points through spherical-harmonic space.  sum_l c_l w_i P_l(o . i) in_i is, by the addition theorem,
//   b_lm = c_l (4 pi / (2l + 1)) sum_i w_i Y_lm(in_i) in_i      (STAGE 1: 110 -> L^2 numbers per point)
//   out_o = sum_lm Y_lm(out_o) b_lm                               (STAGE 2: L^2 -> 110)
// i.e. 2 x 110 x L^2 multiply-adds instead of 110 x 110 x L (3.4 x fewer at L = 16), with wavelength-independent transform matrices as
// scalar operands.  Both stages are OUT[row][w] = scale x sum_c M[c][row] X[c][w]: 16 rows x two wavelength slabs per wavefront.
// (The transform is verified against the direct basis at engine construction; the oracle keeps the direct product.)
template <int STAGE>
__global__ void __launch_bounds__(kWave) k_so_sh(SoGeo g, SoBufs b, const double* __restrict__ M, const int* __restrict__ row_l) {
    const int chunk = blockIdx.x, p = blockIdx.z;
    const SoBufs v = slab_view(g, b, 2 * blockIdx.y, threadIdx.x), v2 = slab_view(g, b, 2 * blockIdx.y + 1, threadIdx.x);
    const int nrows = STAGE == 1 ? g.nlm : g.nout_pad, ncols = STAGE == 1 ? g.nin : g.nlm;
    const double* __restrict__ x0 = STAGE == 1 ? v.incoming + size_t(g.in_start[p]) * kWave : v.sh + size_t(p) * g.nlm * kWave;
    const double* __restrict__ x1 = STAGE == 1 ? v2.incoming + size_t(g.in_start[p]) * kWave : v2.sh + size_t(p) * g.nlm * kWave;
    double acc0[16], acc1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc0[t] = acc1[t] = 0;
    const double* __restrict__ Mc = M + size_t(chunk) * 16;
#pragma unroll 2
    for (int c = 0; c < ncols; ++c) {
        const double a0 = x0[size_t(c) * kWave], a1 = x1[size_t(c) * kWave];
        const double* __restrict__ Mr = Mc + size_t(c) * nrows;
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc0[t] += Mr[t] * a0; acc1[t] += Mr[t] * a1; }
    }
    if (STAGE == 1) {
        const int nlgB = g.num_ms_moments, nl_in = min(g.num_ms_moments, b.nleg);
        double* o0 = v.sh + (size_t(p) * g.nlm + chunk * 16) * kWave;
        double* o1 = v2.sh + (size_t(p) * g.nlm + chunk * 16) * kWave;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int l = row_l[chunk * 16 + t];
            const double c0 = l < nl_in ? v.cl[(size_t(p) * nlgB + l) * kWave] : 0.0, c1 = l < nl_in ? v2.cl[(size_t(p) * nlgB + l) * kWave] : 0.0;
            o0[size_t(t) * kWave] = acc0[t] * c0;
            o1[size_t(t) * kWave] = acc1[t] * c1;
        }
    } else {
        double* o0 = v.outgoing + (size_t(g.out_start[p]) + chunk * 16) * kWave;
        double* o1 = v2.outgoing + (size_t(g.out_start[p]) + chunk * 16) * kWave;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (chunk * 16 + t < g.nout) { o0[size_t(t) * kWave] = acc0[t]; o1[size_t(t) * kWave] = acc1[t]; }
    }
}

// incoming = accumulation matrix x outgoing + first order, one incoming ray x 64 wavelengths per wavefront.  The matrix entry is
// (geometric weight, scalar cache) x (factor of its ray layer, one coalesced load); grid x = ray, so that the wavefronts in flight gather
// from one wavelength slab of `outgoing` (2.8 MB at 5 549 directions) held in L2.
// Measured (MI355X, C5 size, 11.6 ms per launch): with every gather pointed at ONE line the time does not change, without the factor
// stream it drops to 9 ms -- the kernel is bound by the vector-memory issue rate (7 wave loads of 512 B per group of 6 entries =
// 56 cycles of the CU's 64 B/clk L1 path against 116 measured), not by L2 misses or HBM; limiting the wavefronts in flight only hurts.
template <int S>
__global__ void __launch_bounds__(kWave) k_so_accumulate(SoGeo g, SoBufs b) {
    const int r = blockIdx.x;
    const SoBufs v = slab_view(g, b, blockIdx.y, threadIdx.x);
    const double* __restrict__ factor = v.factor;
    const double* __restrict__ outgoing = v.outgoing;
    double acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = __builtin_nontemporal_load(&v.first[(size_t(r) * S + s) * kWave]);
    // the layer factors are read exactly once per launch (23 GB at the C5 size): streaming loads keep them from evicting the slab of
    // `outgoing` that every wavefront in flight gathers from (PMC: 2.8 x the algorithmic HBM reads without the hint)
    const int g0 = g.acc_group_start[r], g1 = g.acc_group_start[r + 1];
    int gi = g0;
#ifndef SKSO_ACC_UNROLL
#define SKSO_ACC_UNROLL 2
#endif
    constexpr int U = SKSO_ACC_UNROLL;
    for (; gi + U <= g1; gi += U) {   // U groups = U factor loads + 6 U gathers in flight
        double f[U], o[U][6][S];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const AccGroup& G = g.acc_group[gi + u];
            f[u] = __builtin_nontemporal_load(&factor[size_t(G.slot) * kWave]);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int s = 0; s < S; ++s) o[u][t][s] = outgoing[(size_t(G.col[t]) * S + s) * kWave];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const AccGroup& G = g.acc_group[gi + u];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double v = 0;
#pragma unroll
                for (int t = 0; t < 6; ++t) v += G.w[t] * o[u][t][s];
                acc[s] += f[u] * v;
            }
        }
    }
    for (; gi < g1; ++gi) {
        const AccGroup& G = g.acc_group[gi];
        const double f = __builtin_nontemporal_load(&factor[size_t(G.slot) * kWave]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            double v = 0;
#pragma unroll
            for (int t = 0; t < 6; ++t) v += G.w[t] * outgoing[(size_t(G.col[t]) * S + s) * kWave];
            acc[s] += f * v;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) v.incoming[(size_t(r) * S + s) * kWave] = acc[s];
}

#include "so_adjoint.inc"

// One step of the fixed-point driver of MultipleScatterSource::SuccessiveOrders (reference: FixedPointSolver::solve / update_history /
// anderson_update, cpp/lib/successive_orders/fixed_point.h:132-258), one wavelength per LANE: residual and stop rule in the infinity norm,
// the ring of state / residual differences, the <= 3 x 3 normal equations of the Type-II Anderson step (diagonal regularised by
// 1e-14 x its scale; an unsolvable or non-finite step falls back to the damped Picard step), the new state.  The state vectors are slab major
// like every per-wavelength array here ([slab][item][64]), so every pass is a unit-stride stream.  The bookkeeping of the ring (hist, start,
// slot) is the same for all wavelengths still iterating and comes from the host; a wavelength that met the tolerance keeps its state.
struct FpStep {
    int n;                      // state items per wavelength
    int depth, hist, start;     // ring AFTER this step's history update: `hist` valid columns, oldest at `start`
    int slot;                   // column written by this step's history update, -1: none (first step or depth 0)
    int conv_on;
    double rel, abs, damping;
};
__global__ void __launch_bounds__(kWave) k_so_fixed_point(SoBufs b, FpStep P, double* __restrict__ x_, const double* __restrict__ mapped_,
                                                          double* __restrict__ prev_x_, double* __restrict__ prev_r_, double* __restrict__ dX_,
                                                          double* __restrict__ dR_, int* __restrict__ done, int* __restrict__ active) {
    const int lane = threadIdx.x, w = blockIdx.x * kWave + lane;
    const size_t o = size_t(blockIdx.x) * kWave * P.n + lane, col = size_t(b.nwp) * P.n;   // column stride of the difference rings
    double* x = x_ + o;
    const double* mapped = mapped_ + o;
    double *prev_x = prev_x_ + o, *prev_r = prev_r_ + o, *dX = dX_ + o, *dR = dR_ + o;
    if (w >= b.nw || done[w]) return;
    double rn = 0, scale = 0;
    for (int i = 0; i < P.n; ++i) {
        const double xi = x[size_t(i) * kWave], mi = mapped[size_t(i) * kWave];
        rn = fmax(rn, fabs(mi - xi));
        scale = fmax(scale, fmax(fabs(xi), fabs(mi)));
    }
    if (P.conv_on && rn <= P.abs + P.rel * scale) {
        for (int i = 0; i < P.n; ++i) x[size_t(i) * kWave] = mapped[size_t(i) * kWave];
        done[w] = 1;
        return;
    }
    if (P.conv_on) atomicAdd(active, 1);
    // history of differences, then this step's state / residual become the previous ones
    for (int i = 0; i < P.n; ++i) {
        const size_t k = size_t(i) * kWave;
        const double xi = x[k], ri = mapped[k] - xi;
        if (P.slot >= 0) {
            dX[col * P.slot + k] = xi - prev_x[k];
            dR[col * P.slot + k] = ri - prev_r[k];
        }
        prev_x[k] = xi;
        prev_r[k] = ri;
    }
    bool mixed = false;
    double c[3] = {0, 0, 0};
    if (P.hist > 0) {
        double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, rhs[3] = {0, 0, 0};
        const size_t s0 = col * ((P.start + 0) % P.depth), s1 = col * ((P.start + 1) % P.depth), s2 = col * ((P.start + 2) % P.depth);
        for (int i = 0; i < P.n; ++i) {
            const size_t k = size_t(i) * kWave;
            const double ri = prev_r[k];
            const double r0 = dR[s0 + k], r1 = (P.hist > 1) ? dR[s1 + k] : 0.0, r2 = (P.hist > 2) ? dR[s2 + k] : 0.0;
            rhs[0] += r0 * ri; rhs[1] += r1 * ri; rhs[2] += r2 * ri;
            G[0][0] += r0 * r0; G[1][0] += r1 * r0; G[1][1] += r1 * r1; G[2][0] += r2 * r0; G[2][1] += r2 * r1; G[2][2] += r2 * r2;
        }
        G[0][1] = G[1][0]; G[0][2] = G[2][0]; G[1][2] = G[2][1];
        double gmax = 1.0;
        for (int a = 0; a < P.hist; ++a) gmax = fmax(gmax, G[a][a]);
        for (int a = 0; a < P.hist; ++a) G[a][a] += 1e-14 * gmax;
        // symmetric system of order hist <= 3: Gaussian elimination with partial pivoting
        double A[3][4];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = G[i][j]; A[i][3] = rhs[i]; }
        bool ok = true;
        const int n = P.hist;
        for (int k = 0; k < n && ok; ++k) {
            int piv = k;
            for (int i = k + 1; i < n; ++i) if (fabs(A[i][k]) > fabs(A[piv][k])) piv = i;
            if (!(fabs(A[piv][k]) > 0)) { ok = false; break; }
            if (piv != k) for (int j = 0; j < 4; ++j) { const double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
            for (int i = k + 1; i < n; ++i) {
                const double f = A[i][k] / A[k][k];
                for (int j = k; j < 4; ++j) A[i][j] -= f * A[k][j];
            }
        }
        // (the right-hand side sits in column 3 whatever the order: columns n .. 2 of the eliminated rows are zero)
        for (int i = n - 1; i >= 0 && ok; --i) {
            double v = A[i][3];
            for (int j = i + 1; j < n; ++j) v -= A[i][j] * c[j];
            c[i] = v / A[i][i];
            ok = isfinite(c[i]);
        }
        if (ok) {
            for (int i = 0; i < P.n && ok; ++i) {
                const size_t k = size_t(i) * kWave;
                double v = prev_x[k] + P.damping * prev_r[k] - c[0] * (dX[s0 + k] + P.damping * dR[s0 + k]);
                if (P.hist > 1) v -= c[1] * (dX[s1 + k] + P.damping * dR[s1 + k]);
                if (P.hist > 2) v -= c[2] * (dX[s2 + k] + P.damping * dR[s2 + k]);
                ok = isfinite(v);
            }
            if (ok) {
                for (int i = 0; i < P.n; ++i) {
                    const size_t k = size_t(i) * kWave;
                    double v = prev_x[k] + P.damping * prev_r[k] - c[0] * (dX[s0 + k] + P.damping * dR[s0 + k]);
                    if (P.hist > 1) v -= c[1] * (dX[s1 + k] + P.damping * dR[s1 + k]);
                    if (P.hist > 2) v -= c[2] * (dX[s2 + k] + P.damping * dR[s2 + k]);
                    x[k] = v;
                }
                mixed = true;
            }
        }
    }
    if (!mixed)
        for (int i = 0; i < P.n; ++i) {
            const size_t k = size_t(i) * kWave;
            x[k] = (P.damping == 1.0) ? mapped[k] : x[k] + P.damping * prev_r[k];   // damping 1: exactly the legacy pass
        }
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

// pi x brdf and its derivatives wrt the BRDF arguments at the ground geometry of the lines of sight (the reference's surfaces,
// cpp/include/sasktran2/atmosphere/surface.h:151-352; kind 3: the functional BRDF of legacy/test_scalar.cpp:806-845): one thread per (line of sight, wavelength)
__global__ void __launch_bounds__(256) k_so_los_albedo(DevRays rs, int kind, int nargs, const double* args, int nw, double* alb, double* dalb) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(rs.nrays) * nw) return;
    const int r = int(tid / nw), w = int(tid % nw);
    double val = 0, d[3] = {0, 0, 0};
    if (rs.ground_is_hit[r] && rs.ground_mu_in[r] > 0) {
        const double mi = rs.ground_mu_in[r], mo = rs.ground_mu_out[r], phi = rs.ground_phi[r];
        const double* a = args + size_t(w) * nargs;
        if (kind == 2) {   // MODIS: (iso + vol k_vol + geo k_geo) / pi  (surface.h:236-281)
            const double ssza = sqrt(1 - mi * mi), svza = sqrt(1 - mo * mo), tsza = ssza / mi, tvza = svza / mo, craa = -cos(phi), sraa = sin(phi);
            const double csa = fmax(-1.0, fmin(1.0, mi * mo + ssza * svza * craa)), sa = acos(csa);
            const double k_vol = ((0.5 * kPi - sa) * csa + sin(sa)) / (mi + mo) - 0.25 * kPi;
            const double d2 = tsza * tsza + tvza * tvza - 2 * tsza * tvza * craa;
            const double ct = fmax(-1.0, fmin(1.0, 2 * sqrt(d2 + tsza * tsza * tvza * tvza * sraa * sraa) * mi * mo / (mi + mo))), t = acos(ct);
            const double k_geo = (t - sin(t) * ct) * (mi + mo) / (kPi * mi * mo) - (mi + mo - 0.5 * (1 + csa)) / (mi * mo);
            val = a[0] + a[1] * k_vol + a[2] * k_geo;
            d[0] = 1.0; d[1] = k_vol; d[2] = k_geo;
        } else if (kind == 1) {   // SnowKokhanovsky: r0 exp(-alpha K0 K0 / r0) / pi, alpha = sqrt(4 pi arg)  (surface.h:151-229)
            const double ss = sqrt(1 - mi * mi), sv = sqrt(1 - mo * mo), cost = fmax(-1.0, fmin(1.0, -mi * mo + ss * sv * cos(phi)));
            const double theta = acos(cost) * 180.0 / kPi;
            const double r0 = (1.247 + 1.186 * (mi + mo) + 5.157 * mi * mo + 11.1 * exp(-0.087 * theta) + 1.1 * exp(-0.014 * theta)) / (4.0 * (mi + mo));
            const double kk = (3.0 / 7.0) * (1.0 + 2.0 * mi) * (3.0 / 7.0) * (1.0 + 2.0 * mo), alpha = sqrt(4 * kPi * a[0]), e = exp(-alpha * kk / r0);
            val = r0 * e;
            d[0] = -2.0 * kPi * kk * e / alpha;
        } else {                  // kind 3
            const double f = fabs(cos(phi)) * fabs(1 - fabs(fabs(mo) - fabs(mi)));
            val = a[0] * f;
            d[0] = f;
        }
    }
    alb[tid] = val;
    for (int k = 0; k < nargs; ++k) dalb[size_t(k) * rs.nrays * nw + tid] = d[k];
}

struct SoDeviceEngine::Impl {
    int brdf_kind = 0, brdf_nargs = 1;              // non-Lambertian surface of the next solves (set_surface_brdf)
    const double* brdf_args = nullptr;              // [nwavel][nargs] (device)
    DevMem brdf_mem;
    double *los_alb = nullptr, *los_dalb = nullptr; // [los][nwavel], [argument][los][nwavel]
    size_t brdf_cap = 0;
    SoGeo g{};
    DevMem tables, scratch;
    SoBufs b{};
    int num_iterations = 50;
    // fixed-point driver (set_fixed_point): state x, map(x), previous state / residual and the two rings of `depth` difference columns,
    // each [nwp][n_out x S] slab major; `done` per wavelength, `active` = wavelengths that did not meet the tolerance in the last step
    struct FixedPoint { bool enabled = false; double rel = 1e-6, abs = 1e-12, damping = 1.0; int depth = 3; } fp;
    double *fp_x = nullptr, *fp_mapped = nullptr, *fp_prev_x = nullptr, *fp_prev_r = nullptr, *fp_dX = nullptr, *fp_dR = nullptr;
    int *fp_done = nullptr, *fp_active = nullptr;
    int fp_last_iterations = 0;
    // weighting functions of the diffuse field (so_adjoint.inc): the transposed accumulation matrix, the per-seed scratch and the
    // outgoing states x_1 .. x_{n-1} of the forward passes
    const int *accT_start = nullptr, *accT_ray = nullptr, *accT_slot = nullptr, *slot_ray = nullptr;
    const double* accT_w = nullptr;
    const int *tabT_start = nullptr, *tabT_row = nullptr;   // the solar table by grid point
    const double* tabT_w = nullptr;
    const double *shT_out = nullptr, *shT_in = nullptr;     // transposed spherical-harmonic transforms [nout][nlm], [nlm][nin_pad]
    int nin_pad = 0;
    SoAdj adj{};
    double *adj_mu = nullptr, *adj_y = nullptr, *adj_gis = nullptr;   // mu slices [passes + 1], Y per pass, ground flux per pass
    // scattering groups through the diffuse field: directions slab major, their phase values on the incoming rays, the coefficient table of
    // S_{g,i}, its outputs per pass, the seed's per-point sums
    double *adj_dT = nullptr, *adj_dph = nullptr, *adj_gcl = nullptr, *adj_xg = nullptr, *adj_gpt = nullptr;
    int adj_groups = 0;
    size_t adj_passes = 0, adj_terms = 0;
    // fixed-order sums of the chain's contributions (so_adjoint.inc): per incoming layer (first compact slot, used-slot mask | mid slots),
    // compact slots by grid level, rays that see the lit ground, the incoming rays' solar stencils by table row
    const int2* chain_meta = nullptr;
    const int *rowq_start = nullptr, *rowq_c = nullptr, *alb_rays = nullptr, *solT_start = nullptr, *solT_end = nullptr;
    const double* solT_w = nullptr;
    int n_used = 0, n_alb = 0, n_endpoints = 0;
    const int* ray_ground = nullptr;       // [incoming rays] ground point of the ray's diffuse point, or -1
    const double* ray_gfactor = nullptr;   // [incoming rays] its factor in the point's downwelling flux
    std::vector<double*> xhist;
    bool cap_adjoint = false;
    bool single_scatter = true, multiple_scatter = true;
    int capacity = 0, cap_nleg = 0;
    bool cap_derivs = false;
    int cap_groups = 0;   // scattering groups the derivative scratch holds
    const double *d_emission = nullptr, *d_surface_emission = nullptr;
    size_t n_inc_rays = 0, n_inc_layers = 0, n_acc = 0, n_out = 0, n_table = 0, max_rows = 0, max_cols = 0, n_groups = 0;
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
        d.ground_mu_out = tables.upload(r.ground_mu_out);
        d.ground_phi = tables.upload(r.ground_phi);
        d.od = up(r.od);
        d.stencil_idx = tables.upload(r.stencil_idx);
        d.w_entrance = tables.upload(r.w_entrance);
        d.w_exit = tables.upload(r.w_exit);
        d.quad_start = tables.upload(r.quad_start);
        d.quad_end = tables.upload(r.quad_end);
        d.frac_start = tables.upload(r.frac_start);
        d.frac_end = tables.upload(r.frac_end);
        d.distance = tables.upload(r.distance);
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
    g.emission = sg.cfg.emission;
    g.occultation = sg.cfg.occultation ? 1 : 0;
    g.los = I.up(sg.los);
    g.stokes_C = I.tables.upload(sg.stokes_C);
    g.stokes_S = I.tables.upload(sg.stokes_S);
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
        {
            std::vector<int> slot(sg.accumulation_layer.size()), layer_ray(sg.incoming.nlayers());
            const int nl = sg.incoming.nlayers();
            for (int r = 0; r < sg.incoming.nrays; ++r) {
                for (int l = sg.incoming.layer_start[r]; l < sg.incoming.layer_start[r + 1]; ++l) layer_ray[l] = r;
                for (int e = sg.accumulation.start[r]; e < sg.accumulation.start[r + 1]; ++e)
                    slot[e] = sg.accumulation_layer[e] >= 0 ? sg.accumulation_layer[e] : nl + r;
            }
            g.acc_layer = I.tables.upload(slot);
            g.inc_layer_ray = I.tables.upload(layer_ray);
            g.n_inc_layers = nl;
            g.n_out_total = sg.out_start.back();
            // packed layer records
            const skso::RaySet& in = sg.incoming;
            std::vector<LayerRec> rec(nl);
            for (int r = 0; r < in.nrays; ++r)
                for (int l = in.layer_start[r]; l < in.layer_start[r + 1]; ++l) {
                    LayerRec& R = rec[l];
                    std::memset(&R, 0, sizeof(R));
                    R.ray = r;
                    R.tiny = in.tiny[l];
                    std::vector<double> wod(4, 0.0);
                    for (int i = 0; i < 4; ++i) {
                        const int q = in.stencil_idx[4 * l + i];
                        R.q[i] = std::max(q, 0);
                        R.went[i] = q < 0 ? 0.0 : in.w_entrance[4 * l + i];
                        R.wexit[i] = q < 0 ? 0.0 : in.w_exit[4 * l + i];
                    }
                    for (int t = in.od.start[l]; t < in.od.start[l + 1]; ++t) {
                        int slot_i = -1;
                        for (int i = 0; i < 4; ++i)
                            if (in.stencil_idx[4 * l + i] == in.od.idx[t]) slot_i = i;
                        if (slot_i < 0) throw std::runtime_error("sasktran2_amd: optical-depth row outside the layer stencil");
                        R.wod[slot_i] += in.od.w[t];
                    }
                    for (int i = 0; i < 2; ++i) { R.mid[i] = in.mid_idx[2 * l + i]; R.midw[i] = in.mid_w[2 * l + i]; }
                    R.quad_start = in.quad_start[l];
                    R.quad_end = in.quad_end[l];
                    const int e_exit = in.endpoint_start[r] + (l - in.layer_start[r]);
                    for (int side = 0; side < 2; ++side) {
                        const int e = e_exit + side, n = in.solar.start[e + 1] - in.solar.start[e];
                        if (n > 4) throw std::runtime_error("sasktran2_amd: solar-table stencil with more than four points");
                        for (int i = 0; i < n; ++i) {
                            (side == 0 ? R.tx : R.te)[i] = in.solar.idx[in.solar.start[e] + i];
                            (side == 0 ? R.twx : R.twe)[i] = in.solar.w[in.solar.start[e] + i];
                        }
                    }
                }
            // the adjoint's chain (so_adjoint.inc) writes one value per USED stencil slot of a layer and the row kernels sum them per grid
            // level in the order of these lists: used = a valid stencil slot, or a free slot lent to a level of the layer's mid-point stencil
            {
                std::vector<int2> meta(nl);
                std::vector<std::vector<int>> by_q(sg.ngrid);
                int c = 0;
                for (int l = 0; l < nl; ++l) {
                    LayerRec& R = rec[l];
                    int mask = 0, mslot[2] = {0, 0};
                    for (int i = 0; i < 4; ++i)
                        if (in.stencil_idx[4 * l + i] >= 0) mask |= 1 << i;
                    for (int j = 0; j < 2; ++j) {
                        int at = -1;
                        for (int i = 0; i < 4 && at < 0; ++i)
                            if (((mask >> i) & 1) && R.q[i] == R.mid[j]) at = i;
                        for (int i = 0; i < 4 && at < 0; ++i)
                            if (!((mask >> i) & 1)) {   // (its weights are zero: the other kernels read the level and drop it)
                                at = i;
                                R.q[i] = R.mid[j];
                                mask |= 1 << i;
                            }
                        if (at < 0) throw std::runtime_error("sasktran2_amd: mid-point level outside a full layer stencil");
                        mslot[j] = at;
                    }
                    meta[l].x = c;
                    meta[l].y = mask | (mslot[0] << 4) | (mslot[1] << 6);
                    for (int i = 0; i < 4; ++i)
                        if ((mask >> i) & 1) by_q[R.q[i]].push_back(c++);
                }
                I.n_used = c;
                std::vector<int> qs{0}, qc;
                for (const auto& v : by_q) {
                    qc.insert(qc.end(), v.begin(), v.end());
                    qs.push_back((int)qc.size());
                }
                I.chain_meta = I.tables.upload(meta);
                I.rowq_start = I.tables.upload(qs);
                I.rowq_c = I.tables.upload(qc);
                std::vector<int> alb;
                I.n_endpoints = in.endpoint_start.back();
                std::vector<std::vector<std::pair<int, double>>> by_t(sg.table.od.rows());
                for (int r = 0; r < in.nrays; ++r) {
                    const int n = in.layer_start[r + 1] - in.layer_start[r], e0 = in.endpoint_start[r];
                    if (n == 0) continue;
                    if (in.ground_is_hit[r] && in.ground_mu_in[r] > 0) alb.push_back(r);
                    for (int e = e0; e <= e0 + n; ++e)
                        for (int t = in.solar.start[e]; t < in.solar.start[e + 1]; ++t)
                            if (in.solar.w[t] != 0.0) by_t[in.solar.idx[t]].push_back({e, in.solar.w[t]});
                }
                I.n_alb = (int)alb.size();
                if (alb.empty()) alb.push_back(0);
                I.alb_rays = I.tables.upload(alb);
                std::vector<int> ts{0}, te;
                std::vector<double> tw;
                for (const auto& v : by_t) {
                    for (const auto& ew : v) { te.push_back(ew.first); tw.push_back(ew.second); }
                    ts.push_back((int)te.size());
                }
                if (te.empty()) { te.push_back(0); tw.push_back(0.0); }
                I.solT_start = I.tables.upload(ts);
                I.solT_end = I.tables.upload(te);
                I.solT_w = I.tables.upload(tw);
            }
            g.inc_rec = I.tables.upload(rec);
            // accumulation entries in groups of six per factor slot
            std::vector<AccGroup> groups;
            std::vector<int> gstart{0};
            for (int r = 0; r < in.nrays; ++r) {
                int e = sg.accumulation.start[r];
                const int e1 = sg.accumulation.start[r + 1];
                while (e < e1) {
                    AccGroup G;
                    std::memset(&G, 0, sizeof(G));
                    G.slot = slot[e];
                    int n = 0;
                    while (e < e1 && n < 6 && slot[e] == G.slot) {
                        G.col[n] = sg.accumulation.idx[e];
                        G.w[n] = sg.accumulation.w[e];
                        ++n;
                        ++e;
                    }
                    groups.push_back(G);
                }
                gstart.push_back((int)groups.size());
            }
            g.acc_group = I.tables.upload(groups);
            g.acc_group_start = I.tables.upload(gstart);
            I.n_groups = groups.size();
            // the accumulation matrix by COLUMN (outgoing direction): the adjoint's lambda <- A^T mu as a gather
            {
                const int ncol = sg.out_start.back();
                std::vector<int> tstart(ncol + 1, 0), tray(sg.accumulation.idx.size()), tslot(sg.accumulation.idx.size());
                std::vector<double> tw(sg.accumulation.idx.size());
                for (int c : sg.accumulation.idx) ++tstart[c + 1];
                for (int c = 0; c < ncol; ++c) tstart[c + 1] += tstart[c];
                std::vector<int> fill(tstart.begin(), tstart.end() - 1);
                for (int r = 0; r < in.nrays; ++r)
                    for (int e = sg.accumulation.start[r]; e < sg.accumulation.start[r + 1]; ++e) {
                        const int at = fill[sg.accumulation.idx[e]]++;
                        tray[at] = r;
                        tslot[at] = slot[e];
                        tw[at] = sg.accumulation.w[e];
                    }
                I.accT_start = I.tables.upload(tstart);
                I.accT_ray = I.tables.upload(tray);
                I.accT_slot = I.tables.upload(tslot);
                I.accT_w = I.tables.upload(tw);
                std::vector<int> sr(nl + in.nrays);   // ray of a factor slot (layer slots, then the rays' ground slots)
                for (int l = 0; l < nl; ++l) sr[l] = layer_ray[l];
                for (int r = 0; r < in.nrays; ++r) sr[nl + r] = r;
                I.slot_ray = I.tables.upload(sr);
                const skso::Csr& T = sg.table.od;
                std::vector<int> qstart(sg.ngrid + 1, 0), qrow(T.idx.size());
                std::vector<double> qw(T.idx.size());
                for (int c : T.idx) ++qstart[c + 1];
                for (int c = 0; c < sg.ngrid; ++c) qstart[c + 1] += qstart[c];
                std::vector<int> qfill(qstart.begin(), qstart.end() - 1);
                for (int t = 0; t < T.rows(); ++t)
                    for (int e = T.start[t]; e < T.start[t + 1]; ++e) {
                        const int at = qfill[T.idx[e]]++;
                        qrow[at] = t;
                        qw[at] = T.w[e];
                    }
                I.tabT_start = I.tables.upload(qstart);
                I.tabT_row = I.tables.upload(qrow);
                I.tabT_w = I.tables.upload(qw);
                std::vector<int> rg(in.nrays, -1);
                std::vector<double> rf(in.nrays, 0.0);
                for (int p = sg.ninterior; p < sg.npoints; ++p)
                    for (int i = 0; i < sg.ground_nin[p - sg.ninterior]; ++i) {
                        rg[sg.in_start[p] + i] = p - sg.ninterior;
                        rf[sg.in_start[p] + i] = sg.ground_in_factor[p - sg.ninterior][i];
                    }
                I.ray_ground = I.tables.upload(rg);
                I.ray_gfactor = I.tables.upload(rf);
            }
        }
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
        // Scalar scattering through spherical-harmonic space (k_so_sh) when it is cheaper than the direct product: real orthonormal
        // harmonics Y_lm, l < L, sum_m Y_lm(a) Y_lm(b) = (2l + 1) / (4 pi) P_l(a . b).  Verified against the direct basis below.
        g.nlm = 0;
        const int Lm = sg.cfg.num_ms_moments;
        if (S == 1 && Lm * Lm % 16 == 0 && size_t(Lm) * Lm * (sg.nin + sg.nout) < size_t(sg.nin) * sg.nout * Lm && !std::getenv("SASKTRAN2_AMD_SO_DIRECT")) {
            const int nlm = Lm * Lm, nout_pad = (sg.nout + 15) / 16 * 16;
            auto harmonics = [&](const double* d, std::vector<double>& Y) {   // Y[lm], lm = l^2 + (m + l)
                const double x = d[0], y = d[1], z = std::max(-1.0, std::min(1.0, d[2]));
                const double phi = std::atan2(y, x), st = std::sqrt(std::max(0.0, 1 - z * z));
                Y.assign(nlm, 0.0);
                for (int m = 0; m < Lm; ++m) {
                    // associated Legendre P_l^m(z), l = m .. Lm - 1 (without the Condon-Shortley phase: it cancels in the products)
                    double pmm = 1;
                    for (int k = 1; k <= m; ++k) pmm *= (2 * k - 1) * st;
                    double pl2 = 0, pl1 = pmm;
                    for (int l = m; l < Lm; ++l) {
                        double plm;
                        if (l == m) plm = pmm;
                        else if (l == m + 1) plm = z * (2 * m + 1) * pmm;
                        else plm = ((2 * l - 1) * z * pl1 - (l + m - 1) * pl2) / (l - m);
                        if (l > m) { pl2 = pl1; pl1 = plm; }
                        double ratio = 1;   // (l - m)! / (l + m)!
                        for (int k = l - m + 1; k <= l + m; ++k) ratio /= k;
                        const double N = std::sqrt((2 * l + 1) / (4 * kPi) * ratio);
                        if (m == 0) Y[l * l + l] = N * plm;
                        else {
                            Y[l * l + l + m] = std::sqrt(2.0) * N * plm * std::cos(m * phi);
                            Y[l * l + l - m] = std::sqrt(2.0) * N * plm * std::sin(m * phi);
                        }
                    }
                }
            };
            std::vector<double> sh_in(size_t(sg.nin) * nlm), sh_out(size_t(nlm) * nout_pad, 0.0), Y;
            std::vector<int> sh_l(nlm);
            for (int l = 0; l < Lm; ++l)
                for (int m = -l; m <= l; ++m) sh_l[l * l + l + m] = l;
            for (int i = 0; i < sg.nin; ++i) {
                harmonics(&sg.in_dir[3 * i], Y);
                for (int lm = 0; lm < nlm; ++lm) sh_in[size_t(i) * nlm + lm] = sg.in_weight[i] * Y[lm] * 4 * kPi / (2 * sh_l[lm] + 1);
            }
            for (int o = 0; o < sg.nout; ++o) {
                harmonics(&sg.out_dir[3 * o], Y);
                for (int lm = 0; lm < nlm; ++lm) sh_out[size_t(lm) * nout_pad + o] = Y[lm];
            }
            double worst = 0;   // the transform must reproduce every element of the direct basis w_i P_l(o . i)
            for (int l = 0; l < Lm; ++l)
                for (int o = 0; o < sg.nout; ++o)
                    for (int i = 0; i < sg.nin; ++i) {
                        double v = 0;
                        for (int lm = l * l; lm < (l + 1) * (l + 1); ++lm) v += sh_out[size_t(lm) * nout_pad + o] * sh_in[size_t(i) * nlm + lm];
                        worst = std::max(worst, std::abs(v - sg.scat_basis[(size_t(l) * rows + o) * cols + i]));
                    }
            if (worst < 1e-13) {
                g.nlm = nlm;
                g.nout_pad = nout_pad;
                g.sh_in = I.tables.upload(sh_in);
                g.sh_out = I.tables.upload(sh_out);
                g.sh_l = I.tables.upload(sh_l);
                // transposed transforms of the adjoint (k_so_adj_sh)
                const int nin_pad = (sg.nin + 15) / 16 * 16;
                std::vector<double> shT_out(size_t(sg.nout) * nlm), shT_in(size_t(nlm) * nin_pad, 0.0);
                for (int o = 0; o < sg.nout; ++o)
                    for (int lm = 0; lm < nlm; ++lm) shT_out[size_t(o) * nlm + lm] = sh_out[size_t(lm) * nout_pad + o];
                for (int i = 0; i < sg.nin; ++i)
                    for (int lm = 0; lm < nlm; ++lm) shT_in[size_t(lm) * nin_pad + i] = sh_in[size_t(i) * nlm + lm];
                I.shT_out = I.tables.upload(shT_out);
                I.shT_in = I.tables.upload(shT_in);
                I.nin_pad = nin_pad;
            } else {
                std::fprintf(stderr, "sasktran2_amd: spherical-harmonic scattering transform off by %.3e, using the direct product\n", worst);
            }
        }
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
        I.max_cols = size_t(sg.nin) * S;
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
bool SoDeviceEngine::has_successive_orders() const { return impl_->multiple_scatter; }
bool SoDeviceEngine::has_unlinearised_sources() const { return false; }   // the diffuse field has its discrete adjoint (so_adjoint.inc)
bool SoDeviceEngine::has_emission() const { return impl_->g.emission != 0; }
void SoDeviceEngine::set_surface_brdf(int kind, int nargs, const double* d_args) {
    if (kind < 0 || kind > 3 || (kind != 0 && (!d_args || nargs != (kind == 2 ? 3 : 1)))) throw std::runtime_error("sasktran2_amd: bad BRDF kind / argument count");
    if (kind != 0 && has_successive_orders()) throw std::runtime_error("sasktran2_amd: non-Lambertian surfaces are not built for the successive-orders source");
    impl_->brdf_kind = kind;
    impl_->brdf_nargs = kind == 0 ? 1 : nargs;
    impl_->brdf_args = kind == 0 ? nullptr : d_args;
}
const double* SoDeviceEngine::los_albedo_derivative() const { return impl_->los_dalb; }
void SoDeviceEngine::set_fixed_point(bool enabled, double relative_tolerance, double absolute_tolerance, double damping, int anderson_depth) {
    if (!(relative_tolerance >= 0) || !(absolute_tolerance >= 0) || !std::isfinite(relative_tolerance) || !std::isfinite(absolute_tolerance))
        throw std::runtime_error("successive-orders tolerances must be finite and non-negative");
    if (!(damping > 0 && damping <= 1)) throw std::runtime_error("successive-orders damping must lie in (0, 1]");
    if (anderson_depth < 0 || anderson_depth > 3) throw std::runtime_error("successive-orders Anderson depth must be 0 .. 3 in this build");
    Impl& I = *impl_;
    I.fp.enabled = enabled;
    I.fp.rel = relative_tolerance;
    I.fp.abs = absolute_tolerance;
    I.fp.damping = damping;
    I.fp.depth = anderson_depth;
    I.scratch.release();   // the next solve sizes its scratch with the state vectors
    I.capacity = 0;
}
int SoDeviceEngine::last_fixed_point_iterations() const { return impl_->fp_last_iterations; }
void SoDeviceEngine::set_emission(const double* d_emission, const double* d_surface_emission) {
    impl_->d_emission = d_emission;
    impl_->d_surface_emission = d_surface_emission;
}
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
    size_t d = size_t(g.ngrid) * (2 + size_t(nleg) * per) + (g.emission ? g.ngrid : 0);
    if (I.multiple_scatter)
        d += I.n_table + size_t(g.ninterior) * g.num_ms_moments * per + 2 * I.n_inc_rays * S + I.n_out * S + I.n_inc_layers * (2 + S) + I.n_inc_rays +
             size_t(g.ninterior) * g.nlm;
    if (I.multiple_scatter && I.fp.enabled) d += size_t(4 + 2 * I.fp.depth) * I.n_out * S + 1;
    return d;
}
size_t SoDeviceEngine::bytes_per_wavelength() const { return so_doubles_per_wavelength(*impl_, std::max(impl_->cap_nleg, impl_->g.num_ss_moments)) * sizeof(double); }

// The diffuse field's share of the weighting functions of one wavelength block (so_adjoint.inc), seed by seed into the accumulators
// k_so_los_deriv has filled.
template <int S>
static void run_adjoint(SoDeviceEngine::Impl& I, const SoGeo& g, const SoBufs& b, hipStream_t st, unsigned wb, bool adj_final) {
    SoTimer t(I, st, "so_adjoint");
    SoAdj A = I.adj;
    const size_t W = b.nwp, Wc = A.wcap, n_state = size_t(g.n_out_total) * S * W, n_rays = size_t(g.inc.nrays) * S * W;
    const size_t mu_stride = size_t(g.inc.nrays) * S * Wc, y_stride = size_t(g.n_inc_layers + g.inc.nrays) * S * Wc, nground = size_t(g.npoints - g.ninterior);
    const size_t gis_stride = std::max<size_t>(nground, 1) * Wc;
    const int n = I.fp.enabled && !adj_final ? I.fp_last_iterations : I.num_iterations;   // passes to transpose / terms of the series
    const int npass = adj_final ? 1 : std::max(n - 1, 0);
    if (n > kMaxPasses) throw std::runtime_error("sasktran2_amd: successive-orders weighting functions support at most 64 passes");
    int rays_per_wave = 8;
    if (const char* e = std::getenv("SASKTRAN2_AMD_SO_ADJ_RPW")) rays_per_wave = std::max(1, atoi(e));
    const int rays_per_block = rays_per_wave * kChainWaves;
    const unsigned nchunks = unsigned((g.inc.nrays + rays_per_block - 1) / rays_per_block);
    const bool sh = S == 1 && g.nlm > 0;   // scalar interior points through spherical-harmonic space, as the forward passes
    // previous state of pass k (k = 1 .. n), slice of its mu, its Y table and its ground-flux row
    auto xprev_of = [&](int k) -> const double* { return adj_final ? b.outgoing : (k > 1 ? I.xhist[k - 2] : nullptr); };
    auto mu_of = [&](int k) { return I.adj_mu + size_t(adj_final ? 1 : (k >= 2 ? k - 2 : npass)) * mu_stride; };
    auto gis_of = [&](int k) { return I.adj_gis + size_t(adj_final ? 0 : k - 1) * gis_stride; };
    // seed-independent tables, once per solve: Y_k = W x_{k-1} per factor slot, the ground points' incoming flux of every pass
    SO_HIP_CHECK(hipMemsetAsync(I.adj_y, 0, std::max<size_t>(npass, 1) * y_stride * sizeof(double), st));
    for (int k = adj_final ? 1 : n; k >= 1; --k) {
        A.xprev = xprev_of(k);
        A.y = A.xprev ? I.adj_y + size_t(adj_final ? 0 : k - 2) * y_stride : nullptr;
        A.gis = gis_of(k);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_y<S>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b, A, I.ray_ground, I.ray_gfactor);
        if (nground > 0) hipLaunchKernelGGL(k_so_adj_gis, dim3(unsigned(nground), wb), dim3(kWave), 0, st, g, A, I.ray_ground);
    }
    // scattering groups (scalar engine): directions slab major, their phase values on the incoming rays, and per pass the outputs of the
    // scattering operator with the groups' coefficients at either level of every interior point (so_adjoint.inc, "scattering groups")
    const int ngroups = b.ngroups;
    const size_t xg_pass = size_t(ngroups) * 2 * g.n_out_total * S * Wc, xg_one = size_t(g.n_out_total) * S * Wc;
    auto xg_of = [&](int k) { return I.adj_xg + size_t(adj_final ? 0 : k - 1) * xg_pass; };
    if (ngroups > 0) {
        hipLaunchKernelGGL(k_so_adj_dleg_T, dim3(2048), dim3(256), 0, st, g, b, I.adj_dT);
        hipLaunchKernelGGL(k_so_adj_dph_inc, dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b, I.adj_dT, I.adj_dph);
        for (int k = adj_final ? 1 : n; k >= 1; --k) {
            // incoming field of the pass: first + A x_{k-1}
            const double* xp = xprev_of(k);
            if (xp) {
                SoBufs bx = b;
                bx.outgoing = const_cast<double*>(xp);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<S>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, bx);
            } else
                SO_HIP_CHECK(hipMemcpyAsync(b.incoming, b.first, n_rays * sizeof(double), hipMemcpyDeviceToDevice, st));
            for (int gi = 0; gi < ngroups; ++gi)
                for (int side = 0; side < 2; ++side) {
                    hipLaunchKernelGGL(k_so_adj_gcl, dim3(unsigned(g.ninterior * g.num_ms_moments * (S == 3 ? 4 : 1)), wb), dim3(kWave), 0, st, g, b, I.adj_dT, I.adj_gcl, gi, side);
                    SoBufs bg = b;
                    bg.cl = I.adj_gcl;
                    bg.outgoing = xg_of(k) + size_t(gi * 2 + side) * xg_one;
                    if (sh) {
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<1>), dim3(unsigned(g.nlm / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, bg, g.sh_in, g.sh_l);
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<2>), dim3(unsigned(g.nout_pad / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, bg, g.sh_out, g.sh_l);
                    } else
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<S>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, bg, 0);
                }
        }
    }
    A.ngroups = ngroups;
    A.dph_inc = ngroups > 0 ? I.adj_dph : nullptr;
    A.gpt = ngroups > 0 ? I.adj_gpt : nullptr;
    A.xg = nullptr;
    A.n_used = I.n_used;
    A.cont_alb = 2 * I.n_used;
    A.cont_end = A.cont_alb + g.inc.nrays;
    A.cont_grp = A.cont_end + I.n_endpoints;
    A.ncont = size_t(A.cont_grp) + size_t(I.adj_groups) * I.n_used;   // (the stride the scratch was sized with)
    // the chain with mu_k of a ray in registers for the usual pass counts (one about the final state, nine for ten orders of scattering)
    auto launch_chain = [&](const SoAdj& A_) {
        const dim3 grid(wb, nchunks), block(kWave * kChainWaves);
        if (npass == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_chain<S, 1>), grid, block, 0, st, g, b, A_, g.inc_rec, I.chain_meta, rays_per_wave);
        else if (npass == 9 && S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_chain<S, 9>), grid, block, 0, st, g, b, A_, g.inc_rec, I.chain_meta, rays_per_wave);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_chain<S, 0>), grid, block, 0, st, g, b, A_, g.inc_rec, I.chain_meta, rays_per_wave);
    };
    A.y = nullptr;
    A.npass = npass;
    A.y_hist = I.adj_y;
    A.mu_hist = adj_final ? A.gfirst : I.adj_mu;   // about the final state every term meets the same Y: the summed mu is all the chain needs
    for (int ray = 0; ray < g.nlos; ++ray)
        for (int comp = 0; comp < S; ++comp) {
            A.ray = ray;
            A.comp = comp;
            SO_HIP_CHECK(hipMemsetAsync(A.lam, 0, n_state * sizeof(double), st));
            SO_HIP_CHECK(hipMemsetAsync(A.gfirst, 0, n_rays * sizeof(double), st));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_seed<S>), dim3(wb), dim3(kWave), 0, st, g, b, A);
            for (int k = n; k >= 1; --k) {
                A.xprev = xprev_of(k);
                A.mu = mu_of(k);
                A.gis = gis_of(k);
                if (ngroups > 0) {   // lambda_k . S_{g,i}(incoming_k) per interior point
                    A.xg = xg_of(k);
                    hipLaunchKernelGGL(k_so_adj_gdot, dim3(unsigned(g.ninterior), wb, unsigned(ngroups * 2)), dim3(kWave), 0, st, g, b, A);
                }
                if (sh) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_sh<1>), dim3(unsigned(g.nlm / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b, A, I.shT_out, g.sh_l, I.nin_pad);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_sh<2>), dim3(unsigned(I.nin_pad / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b, A, I.shT_in, g.sh_l, I.nin_pad);
                    if (g.npoints > g.ninterior)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_scatter<S>), dim3(unsigned(I.max_cols), wb, unsigned(g.npoints - g.ninterior)), dim3(kWave), 0, st, g, b, A, g.ninterior);
                } else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_scatter<S>), dim3(unsigned(I.max_cols), wb, unsigned(g.npoints)), dim3(kWave), 0, st, g, b, A, 0);
                if (!A.xprev) break;   // the first pass started from x = 0: nothing behind it
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_nu<S>), dim3(2048, wb), dim3(256), 0, st, g, b, A, I.slot_ray);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_adj_accT<S>), dim3(unsigned(g.n_out_total), wb), dim3(kWave), 0, st, g, b, A, I.accT_start, I.accT_slot, I.accT_w);
                std::swap(A.lam, A.lam_new);
            }
            launch_chain(A);
            // the chain's contributions summed in the fixed order of the host's lists (no atomics: bit-reproducible for every launch shape)
            hipLaunchKernelGGL(k_so_adj_gtod, dim3(unsigned(g.ntable), wb), dim3(kWave), 0, st, g, A, I.solT_start, I.solT_end, I.solT_w);
            hipLaunchKernelGGL(k_so_adj_fold, dim3(unsigned(g.ngrid), wb), dim3(kWave), 0, st, g, b, A, I.tabT_start, I.tabT_row, I.tabT_w);
            hipLaunchKernelGGL(k_so_adj_rows, dim3(unsigned((2 + ngroups) * g.ngrid + 1), wb), dim3(4 * kWave), 0, st, g, b, A, I.rowq_start, I.rowq_c, I.alb_rays, I.n_alb);
            if (ngroups > 0) hipLaunchKernelGGL(k_so_adj_gmap, dim3(wb), dim3(kWave), 0, st, g, b, A);
        }
    SO_HIP_CHECK(hipGetLastError());
}

void SoDeviceEngine::solve_device(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg, const double* d_albedo,
                                  const double* d_irradiance, double* d_radiance, double* d_los_od, void* stream, int accumulate,
                                  double* d_wf_ext, double* d_wf_ssa, double* d_wf_alb, double* d_wf_emission, double* d_wf_surface_emission,
                                  int ngroups, const double* d_dleg, double* d_wf_groups) {
    Impl& I = *impl_;
    SoGeo& g = I.g;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nwavel <= 0 || nleg <= 0) throw std::runtime_error("sasktran2_amd: nwavel and nleg must be positive");
    const int S = g.S, per = S == 3 ? 4 : 1;
    const bool derivs = d_wf_ext != nullptr;
    if (derivs && (!d_wf_ssa || !d_wf_alb)) throw std::runtime_error("sasktran2_amd: weighting functions need all three output buffers");
    const bool adjoint = derivs && I.multiple_scatter;   // weighting functions through the diffuse field (so_adjoint.inc)
    // the exact transpose of the n passes needs their states; the Anderson-mixed / damped / tolerance-stopped driver is differentiated about
    // its final state instead (adjoint Neumann series, as many terms as the driver's iteration limit)
    const bool fp_is_legacy = I.fp.depth == 0 && I.fp.damping == 1.0 && I.fp.rel == 0 && I.fp.abs == 0;
    const bool adj_final = adjoint && I.fp.enabled && !fp_is_legacy;
    const size_t n_hist = (adjoint && !adj_final && I.num_iterations > 1) ? size_t(I.num_iterations - 1) : 0;
    if (ngroups < 0 || (ngroups > 0 && (!derivs || !d_dleg || !d_wf_groups))) throw std::runtime_error("sasktran2_amd: scattering groups need weighting-function buffers, d_legendre and an output");
    if (g.emission && !I.d_emission) throw std::runtime_error("sasktran2_amd: the engine has an emission source but no emission_source array was set (set_emission)");
    // block of wavelengths in flight: half of the free HBM (plus what this engine already holds), balanced blocks, multiple of the wavefront
    const size_t deriv_doubles = derivs ? size_t(g.nlos) * S * so_deriv_slots(g.ngrid, ngroups) + size_t(ngroups) * g.nlos * g.ngrid * (S == 3 ? 2 : 1) : 0;
    const size_t n_pass = adjoint ? (adj_final ? 1 : n_hist) : 0;   // passes with a predecessor: one Y table (and one mu slice) each
    const size_t n_terms = adjoint ? size_t(std::max(I.num_iterations, 1)) : 0;
    const size_t adj_group_doubles = (adjoint && ngroups > 0) ? size_t(ngroups) * (size_t(g.ngrid) * nleg * per + I.n_inc_rays * g.ngrid * (S == 3 ? 2 : 1) + size_t(g.ninterior) * 2) +
                                                                   size_t(g.ninterior) * g.num_ms_moments * per + n_terms * size_t(ngroups) * 2 * I.n_out * S : 0;
    const size_t adj_doubles = adj_group_doubles + (adjoint ? (2 + n_hist) * I.n_out * S + (n_pass + 2) * I.n_inc_rays * S + (n_pass + 1) * (I.n_inc_layers + I.n_inc_rays) * S + I.n_table +
                                             (n_terms + 1) * std::max(g.npoints - g.ninterior, 1) + 2 * size_t(I.n_used) + I.n_inc_rays + I.n_endpoints + size_t(ngroups) * I.n_used : 0);
    const size_t per_w = (so_doubles_per_wavelength(I, nleg) + deriv_doubles + adj_doubles) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t budget = size_t(0.5 * double(free_b + I.scratch_bytes));
    if (const char* e = std::getenv("SASKTRAN2_AMD_HBM_BUDGET_MB")) budget = size_t(std::max(1, atoi(e))) << 20;
    size_t fit = std::max<size_t>(kPad, budget / per_w / kPad * kPad);
    if (const char* e = std::getenv("SASKTRAN2_AMD_SO_CHUNK")) fit = std::max<size_t>(kPad, size_t(atoi(e)) / kPad * kPad);
    const size_t nblocks = (size_t(nwavel) + fit - 1) / fit;
    int chunk = int((size_t(nwavel) + nblocks - 1) / nblocks);
    chunk = (chunk + kPad - 1) / kPad * kPad;
    I.last_chunk = chunk;
    if (chunk > I.capacity || nleg > I.cap_nleg || (derivs && !I.cap_derivs) || ngroups > I.cap_groups || (adjoint && !I.cap_adjoint) || n_hist > I.xhist.size() || n_pass > I.adj_passes || n_terms > I.adj_terms || (adjoint && ngroups > I.adj_groups)) {
        I.scratch.release();
        I.cap_derivs = I.cap_derivs || derivs;
        I.cap_adjoint = I.cap_adjoint || adjoint;
        const size_t keep_hist = std::max(n_hist, I.xhist.size());
        I.adj_passes = std::max(n_pass, I.adj_passes);
        I.adj_terms = std::max(n_terms, I.adj_terms);
        if (adjoint) I.adj_groups = std::max(ngroups, I.adj_groups);
        I.cap_groups = std::max(I.cap_groups, ngroups);
        I.capacity = std::max(chunk, I.capacity);
        I.cap_nleg = std::max(nleg, I.cap_nleg);
        const size_t W = I.capacity, nlg = size_t(I.cap_nleg) * per;
        SoBufs& b = I.b;
        b.kT = I.scratch.alloc<double>(size_t(g.ngrid) * W);
        b.omT = I.scratch.alloc<double>(size_t(g.ngrid) * W);
        b.legT = I.scratch.alloc<double>(size_t(g.ngrid) * nlg * W);
        b.emT = g.emission ? I.scratch.alloc<double>(size_t(g.ngrid) * W) : nullptr;
        if (I.multiple_scatter) {
            b.tab = I.scratch.alloc<double>(I.n_table * W);
            b.cl = I.scratch.alloc<double>(size_t(g.ninterior) * g.num_ms_moments * per * W);
            b.first = I.scratch.alloc<double>(I.n_inc_rays * S * W);
            b.incoming = I.scratch.alloc<double>(I.n_inc_rays * S * W);
            b.outgoing = I.scratch.alloc<double>(I.n_out * S * W);
            b.factor = I.scratch.alloc<double>((I.n_inc_layers + I.n_inc_rays) * W);
            b.lod = I.scratch.alloc<double>(I.n_inc_layers * W);
            b.lsrc = I.scratch.alloc<double>(I.n_inc_layers * S * W);
            b.sh = g.nlm > 0 ? I.scratch.alloc<double>(size_t(g.ninterior) * g.nlm * W) : nullptr;
            if (I.fp.enabled) {
                const size_t nst = I.n_out * S * W;
                I.fp_x = I.scratch.alloc<double>(nst);
                I.fp_mapped = I.scratch.alloc<double>(nst);
                I.fp_prev_x = I.scratch.alloc<double>(nst);
                I.fp_prev_r = I.scratch.alloc<double>(nst);
                I.fp_dX = I.scratch.alloc<double>(nst * std::max(I.fp.depth, 1));
                I.fp_dR = I.scratch.alloc<double>(nst * std::max(I.fp.depth, 1));
                I.fp_done = I.scratch.alloc<int>(W);
                I.fp_active = I.scratch.alloc<int>(1);
            }
        }
        I.xhist.clear();
        if (I.cap_adjoint) {
            SoAdj& A = I.adj;
            A.lam = I.scratch.alloc<double>(I.n_out * S * W);
            A.lam_new = I.scratch.alloc<double>(I.n_out * S * W);
            I.adj_mu = I.scratch.alloc<double>((I.adj_passes + 1) * I.n_inc_rays * S * W);
            I.adj_y = I.scratch.alloc<double>(std::max<size_t>(I.adj_passes, 1) * (I.n_inc_layers + I.n_inc_rays) * S * W);
            I.adj_gis = I.scratch.alloc<double>(std::max<size_t>(I.adj_terms, 1) * std::max(g.npoints - g.ninterior, 1) * W);
            A.wcap = W;
            A.gfirst = I.scratch.alloc<double>(I.n_inc_rays * S * W);
            A.nu = I.scratch.alloc<double>((I.n_inc_layers + I.n_inc_rays) * S * W);
            const size_t n_gpt = std::max(g.npoints - g.ninterior, 1);
            A.galb = I.scratch.alloc<double>(n_gpt * W);
            A.gtod = I.scratch.alloc<double>(I.n_table * W);
            SO_HIP_CHECK(hipMemset(A.galb, 0, n_gpt * W * sizeof(double)));            // (k_so_adj_fold leaves it cleared)
            A.cont = I.scratch.alloc<double>((2 * size_t(I.n_used) + I.n_inc_rays + I.n_endpoints + size_t(I.adj_groups) * I.n_used) * W);
            for (size_t k = 0; k < keep_hist; ++k) I.xhist.push_back(I.scratch.alloc<double>(I.n_out * S * W));
            if (I.adj_groups > 0) {
                const size_t G = I.adj_groups;
                I.adj_dT = I.scratch.alloc<double>(G * g.ngrid * nlg * W);
                I.adj_dph = I.scratch.alloc<double>(G * I.n_inc_rays * g.ngrid * (S == 3 ? 2 : 1) * W);
                I.adj_gcl = I.scratch.alloc<double>(size_t(g.ninterior) * g.num_ms_moments * per * W);
                I.adj_xg = I.scratch.alloc<double>(std::max<size_t>(I.adj_terms, 1) * G * 2 * I.n_out * S * W);
                I.adj_gpt = I.scratch.alloc<double>(G * g.ninterior * 2 * W);
                SO_HIP_CHECK(hipMemset(I.adj_gpt, 0, G * g.ninterior * 2 * W * sizeof(double)));   // (k_so_adj_gmap leaves it cleared)
            }
        }
        b.dacc = I.cap_derivs ? I.scratch.alloc<double>(size_t(g.nlos) * S * so_deriv_slots(g.ngrid, I.cap_groups) * W) : nullptr;
        b.dphT = I.cap_groups > 0 ? I.scratch.alloc<double>(size_t(I.cap_groups) * g.nlos * g.ngrid * (S == 3 ? 2 : 1) * W) : nullptr;
        I.scratch_bytes = (so_doubles_per_wavelength(I, I.cap_nleg) + (I.cap_derivs ? size_t(g.nlos) * S * so_deriv_slots(g.ngrid, I.cap_groups) : 0) +
                           size_t(I.cap_groups) * g.nlos * g.ngrid * (S == 3 ? 2 : 1)) * sizeof(double) * W;
    }
    const double* los_albedo = nullptr;
    if (I.brdf_kind != 0) {
        const size_t need = size_t(g.nlos) * nwavel * (1 + I.brdf_nargs);
        if (need > I.brdf_cap) {
            SO_HIP_CHECK(hipStreamSynchronize(st));
            I.brdf_mem.release();
            I.los_alb = I.brdf_mem.alloc<double>(size_t(g.nlos) * nwavel);
            I.los_dalb = I.brdf_mem.alloc<double>(size_t(g.nlos) * nwavel * I.brdf_nargs);
            I.brdf_cap = need;
        }
        hipLaunchKernelGGL(k_so_los_albedo, dim3(unsigned((size_t(g.nlos) * nwavel + 255) / 256)), dim3(256), 0, st, g.los, I.brdf_kind, I.brdf_nargs, I.brdf_args, nwavel,
                           I.los_alb, I.los_dalb);
        los_albedo = I.los_alb;
    }
    for (int w0 = 0; w0 < nwavel; w0 += chunk) {
        SoBufs b = I.b;
        b.los_albedo = los_albedo;
        b.nw = std::min(chunk, nwavel - w0);
        b.nwp = (b.nw + kPad - 1) / kPad * kPad;
        b.nleg = nleg;
        b.w0 = w0;
        b.nwavel_total = nwavel;
        b.ext = d_ext + size_t(w0) * g.ngrid;
        b.ssa = d_ssa + size_t(w0) * g.ngrid;
        b.leg = d_leg + size_t(w0) * g.ngrid * nleg * per;
        b.albedo = d_albedo + w0;
        b.irradiance = d_irradiance ? d_irradiance + w0 : nullptr;
        b.emission = I.d_emission ? I.d_emission + size_t(w0) * g.ngrid : nullptr;
        b.surface_emission = I.d_surface_emission ? I.d_surface_emission + w0 : nullptr;
        if (!g.emission) b.emT = nullptr;
        b.radiance = d_radiance;
        b.los_od = d_los_od;
        b.accumulate = accumulate;
        b.wf_ext = d_wf_ext;
        b.wf_ssa = d_wf_ssa;
        b.wf_alb = d_wf_alb;
        b.wf_em = d_wf_emission;
        b.wf_sem = d_wf_surface_emission;
        b.ngroups = ngroups;
        b.dleg = d_dleg;
        b.wf_grp = d_wf_groups;
        if (ngroups == 0) b.dphT = nullptr;
        if (!derivs) b.dacc = nullptr;
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
            const char* split_env = std::getenv("SASKTRAN2_AMD_SO_SPLIT_LAYERS");
            if (!(split_env && atoi(split_env) != 0)) {
                SoTimer t(I, st, "so_ray");
                const dim3 grid(unsigned(g.inc.nrays), wb / 2);   // two slabs per wavefront (the block is padded to 128 wavelengths)
                if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_ray<1>), grid, dim3(kWave), 0, st, g, b, g.inc_rec, g.inc.d00, g.inc.d02, g.inc.C2, g.inc.S2, b.factor);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_ray<3>), grid, dim3(kWave), 0, st, g, b, g.inc_rec, g.inc.d00, g.inc.d02, g.inc.C2, g.inc.S2, b.factor);
            } else {
            {
                SoTimer t(I, st, "so_layers");
                int lpb = kLayersPerBlock;
                if (const char* e = std::getenv("SASKTRAN2_AMD_SO_LPB")) lpb = std::max(1, atoi(e));
                const dim3 grid(unsigned((I.n_inc_layers + lpb - 1) / lpb), wb);
                if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_layers<1>), grid, dim3(kWave), 0, st, g, b, g.inc_rec, g.inc.d00, g.inc.d02, g.inc.C2, g.inc.S2, b.lod, b.factor, b.lsrc, lpb);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_layers<3>), grid, dim3(kWave), 0, st, g, b, g.inc_rec, g.inc.d00, g.inc.d02, g.inc.C2, g.inc.S2, b.lod, b.factor, b.lsrc, lpb);
            }
            {
                SoTimer t(I, st, "so_scan");
                if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scan<1>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scan<3>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b);
            }
            }
            // one application of the scattering operator S: b_.incoming -> b_.outgoing
            auto scatter = [&](const SoBufs& b_) {
                SoTimer t(I, st, "so_scatter");
                if (S == 1 && g.nlm > 0) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<1>), dim3(unsigned(g.nlm / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b_, g.sh_in, g.sh_l);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<2>), dim3(unsigned(g.nout_pad / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b_, g.sh_out, g.sh_l);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<1>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints - g.ninterior)), dim3(kWave), 0, st, g, b_, g.ninterior);
                } else if (S == 1) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<1>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints)), dim3(kWave), 0, st, g, b_, 0);
                } else {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<3>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints)), dim3(kWave), 0, st, g, b_, 0);
                }
            };
            if (I.fp.enabled) {
                // MultipleScatterSource::SuccessiveOrders: fixed point of x -> S (first + T x) with Anderson mixing (k_so_fixed_point).  The state
                // starts at 0 (the first step is the first scattering order, as the legacy pass); the LOS integration then reads the state.
                const size_t nst = size_t(g.n_out_total) * S * b.nwp;
                SO_HIP_CHECK(hipMemsetAsync(I.fp_x, 0, nst * sizeof(double), st));
                SO_HIP_CHECK(hipMemsetAsync(I.fp_done, 0, size_t(b.nwp) * sizeof(int), st));
                const bool conv_on = I.fp.rel > 0 || I.fp.abs > 0;
                int hist = 0, start = 0;
                bool have_prev = false;
                I.fp_last_iterations = 0;
                for (int it = 0; it < I.num_iterations; ++it) {
                    SoBufs bx = b, bm = b;
                    bx.outgoing = I.fp_x;
                    bm.outgoing = I.fp_mapped;
                    if (it > 0) {
                        SoTimer t(I, st, "so_accumulate");
                        if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<1>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, bx);
                        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<3>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, bx);
                    }
                    scatter(bm);
                    int slot = -1;
                    if (have_prev && I.fp.depth > 0) {
                        if (hist < I.fp.depth) { slot = (start + hist) % I.fp.depth; ++hist; }
                        else { slot = start; start = (start + 1) % I.fp.depth; }
                    }
                    have_prev = true;
                    {
                        SoTimer t(I, st, "so_fixed_point");
                        if (conv_on) SO_HIP_CHECK(hipMemsetAsync(I.fp_active, 0, sizeof(int), st));
                        const FpStep P{int(g.n_out_total) * S, I.fp.depth, hist, start, slot, conv_on ? 1 : 0, I.fp.rel, I.fp.abs, I.fp.damping};
                        hipLaunchKernelGGL(k_so_fixed_point, dim3(wb), dim3(kWave), 0, st, b, P, I.fp_x, I.fp_mapped, I.fp_prev_x, I.fp_prev_r, I.fp_dX, I.fp_dR,
                                           I.fp_done, I.fp_active);
                    }
                    I.fp_last_iterations = it + 1;
                    if (size_t(it) < n_hist) SO_HIP_CHECK(hipMemcpyAsync(I.xhist[it], I.fp_x, nst * sizeof(double), hipMemcpyDeviceToDevice, st));   // x_{it+1}
                    if (conv_on) {
                        int active = 1;
                        SO_HIP_CHECK(hipMemcpyAsync(&active, I.fp_active, sizeof(int), hipMemcpyDeviceToHost, st));
                        SO_HIP_CHECK(hipStreamSynchronize(st));
                        if (active == 0) break;
                    }
                }
                b.outgoing = I.fp_x;
            } else
            for (int it = 0; it < I.num_iterations; ++it) {
                if (it > 0) {
                    SoTimer t(I, st, "so_accumulate");
                    if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<1>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_accumulate<3>), dim3(unsigned(g.inc.nrays), wb), dim3(kWave), 0, st, g, b);
                }
                SoTimer t(I, st, "so_scatter");
                if (S == 1 && g.nlm > 0) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<1>), dim3(unsigned(g.nlm / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b, g.sh_in, g.sh_l);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_sh<2>), dim3(unsigned(g.nout_pad / 16), wb / 2, unsigned(g.ninterior)), dim3(kWave), 0, st, g, b, g.sh_out, g.sh_l);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<1>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints - g.ninterior)), dim3(kWave), 0, st, g, b, g.ninterior);
                } else if (S == 1) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<1>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints)), dim3(kWave), 0, st, g, b, 0);
                } else {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_scatter<3>), dim3(unsigned(I.max_rows), wb / 2, unsigned(g.npoints)), dim3(kWave), 0, st, g, b, 0);
                }
                if (size_t(it) < n_hist)   // x_{it+1}
                    SO_HIP_CHECK(hipMemcpyAsync(I.xhist[it], b.outgoing, size_t(g.n_out_total) * S * b.nwp * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
        }
        {
            SoTimer t(I, st, "so_march_los");
            if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<1, true>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_march<3, true>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
        }
        if (derivs) {
            SoTimer t(I, st, "so_los_derivatives");
            if (ngroups > 0) hipLaunchKernelGGL(k_so_group_phase, dim3(wb, unsigned(ngroups * g.ngrid)), dim3(kWave), 0, st, g, b);
            if (S == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_los_deriv<1>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_so_los_deriv<3>), dim3(wb, unsigned(g.nlos)), dim3(kWave), 0, st, g, b, ss, ms);
            if (adjoint) {
                if (S == 1) run_adjoint<1>(I, g, b, st, wb, adj_final);
                else run_adjoint<3>(I, g, b, st, wb, adj_final);
            }
            hipLaunchKernelGGL(k_so_deriv_out, dim3(2048), dim3(256), 0, st, g, b);
        }
        SO_HIP_CHECK(hipGetLastError());
    }
}

}  // namespace skdev
