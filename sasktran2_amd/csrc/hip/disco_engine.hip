// HIP kernels + device driver of the DISCO hot path (gfx950).
//
// Launch structure per spectrum chunk (NW wavelengths resident in HBM; layer records, block inverses and table values
// are wavelength-major so that the wavefront owning a wavelength streams them with unit stride):
//
//   k0_layer_optics      grid over wavelengths              layer optics, LOS shell optics
//   for each solar-zenith index s:
//     k0_transmission    grid over wavelengths              beam attenuation, average secants
//     for azimuth order m = 0 .. (stop rule):
//       k1_layer_solution  one wavefront per wavelength     lane group per layer: Cholesky + Jacobi eigen-solution in LDS,
//                                                           Green's particular solution (+ d/d omega)
//       k2a_bvp            one wavefront per wavelength     block-tridiagonal factor/solve, blocks staged in LDS
//       k3_table_values    one wavefront per wavelength     diffuse-source table values, convergence flags
//       k2c_adjoint_ops    one wavefront per wavelength     LOS-independent adjoint operators of every layer
//       k2b_adjoint        one lane per LOS, LDS-shared     adjoint sweeps + reduced gradient (small mat-vecs)
//       k_stop_rule        grid over wavelengths            azimuth stop rule, active-wavelength count
//     k4a_chain          grid over wavelength x LOS         reverse chain beam/optics -> native derivatives
//   k4b_los              grid over wavelength x LOS         LOS integration of the table
//   k_output             grid over wavelength x LOS         writes radiance / weighting functions in the caller's layout
//
// The task bodies live in disco_core.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "disco_engine.hpp"
#include "disco_wave_kernels.hpp"
#include "disco_group_kernels.hpp"

namespace skdev {

#define HIP_CHECK(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + \
                                     ":" + std::to_string(__LINE__));                                          \
    } while (0)

using skd::Bufs;
using skd::Geo;

constexpr int kBlock = 64;

__global__ void __launch_bounds__(kBlock) kern_k0(Geo g, Bufs b) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= b.nw) return;
    skd::k0_layer_optics(g, b, w, false);
}
// The layer optics of skd::k0_layer_optics for NSTR streams, 64 wavelengths per wavefront: the Legendre coefficients of a grid level
// ([wavelength][level][order] in the caller's storage: 128-byte runs 6.5 KB apart) come through an LDS tile -- 16 lanes per wavelength fetch
// its run, four wavelengths per load instruction -- instead of one 8-byte gather per lane, order and level; a level shared by two layers is
// fetched once; the mixed coefficients of a layer are formed in registers and stored once (the generic form zeroes, accumulates and
// normalises them in HBM).  Same statements per wavelength, same results.
template <int NSTR> __global__ void __launch_bounds__(kBlock) kern_k0_tile(Geo g, Bufs b) {
    __shared__ double tile[2][kBlock][NSTR + 1];
    __shared__ int tile_q[2];
    const int lane = threadIdx.x, w0 = blockIdx.x * kBlock, w = w0 + lane, nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    const int ws = min(w, nw - 1);
    const double* ext = b.ext + size_t(ws) * ngrid;
    const double* ssa = b.ssa + size_t(ws) * ngrid;
    if (lane < 2) tile_q[lane] = -1;
    __syncthreads();
    auto fetch = [&](int q, int buf) {   // Legendre coefficients of level q of the block's wavelengths -> tile[buf]
        for (int i = 0; i < NSTR; ++i) {
            const int idx = i * kBlock + lane, wl = idx / NSTR, l = idx % NSTR;
            const int wg = min(w0 + wl, nw - 1);
            tile[buf][wl][l] = (l < b.nleg) ? b.leg[(size_t(wg) * ngrid + q) * b.nleg + l] : 0.0;
        }
    };
    double ceiling_depth = 0, floor_depth = 0;
    for (int p = 0; p < nlyr; ++p) {
        // the (at most two) levels of this layer: reuse the tile of a level the previous layer brought in
        int buf_of[2] = {-1, -1};
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * p + t];
            if (q < 0) continue;
            int bsel = (tile_q[0] == q) ? 0 : (tile_q[1] == q) ? 1 : -1;
            if (bsel < 0) {
                // a free buffer: not the one the other level of this layer sits in
                bsel = (buf_of[0] == 0 || (t == 0 && tile_q[0] == g.wq_idx[2 * p + 1])) ? 1 : 0;
                __syncthreads();
                fetch(q, bsel);
                if (lane == 0) tile_q[bsel] = q;
                __syncthreads();
            }
            buf_of[t] = bsel;
        }
        double od = 0, s = 0, acc[NSTR];
#pragma unroll
        for (int l = 0; l < NSTR; ++l) acc[l] = 0;
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * p + t];
            if (q < 0) continue;
            const double wt = g.wq_w[2 * p + t];
            const double kext = ext[q], kscat = ssa[q] * kext;
            od += kext * wt;
            s += kscat * wt;
            const double fq = b.delta_f ? b.delta_f[size_t(ws) * ngrid + q] : 0.0, fsub = fq / (1 - fq);
            const double* row = tile[buf_of[t]][lane];
#pragma unroll
            for (int l = 0; l < NSTR; ++l) acc[l] += wt * kscat * (row[l] - (2 * l + 1) * fsub);
        }
        double* at = b.at + (size_t(ws) * nlyr + p) * (NSTR + 2);   // the same values in one run per (wavelength, layer)
        if (w < nw) {
#pragma unroll
            for (int l = 0; l < NSTR; ++l) {
                const double v = (s > 0) ? acc[l] / s : 0.0;
                b.a[(size_t(p) * NSTR + l) * nw + w] = v;
                at[l] = v;
            }
        }
        s /= od;
        od *= g.dh[p];
        floor_depth += od;
        const double te = od / g.dh[p];
        const double se = fmax(te * s, te * skd::kSSADither);
        double m_ssa = se / te;
        if (1 - m_ssa < skd::kSSADither) m_ssa = 1 - skd::kSSADither;
        if (w < nw) {
            b.kbar[size_t(p) * nw + w] = te;
            b.om[size_t(p) * nw + w] = m_ssa;
            b.tau[size_t(p) * nw + w] = floor_depth - ceiling_depth;
            at[NSTR] = m_ssa;
            at[NSTR + 1] = floor_depth - ceiling_depth;
        }
        ceiling_depth = floor_depth;
    }
}
// LOS shells of k0_layer_optics, one thread per (shell, wavelength), wavelength fastest: optical depth, local ssa and the
// shell transmission exp(-od) (parked in srcw until kern_k0_los turns it into the source weight)
__global__ void __launch_bounds__(256) kern_k0_shells(Geo g, Bufs b) {
    const size_t n = size_t(g.nq_total) * b.nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int q = int(i / b.nw), w = int(i % b.nw);
        const double* ext = b.ext + size_t(w) * g.ngrid;
        const double* ssa = b.ssa + size_t(w) * g.ngrid;
        double od = 0, om = 0;
        for (int t = g.rl_od_start[q]; t < g.rl_od_start[q + 1]; ++t) od += g.od_w[t] * ext[g.od_idx[t]];
        for (int t = g.rl_ssa_start[q]; t < g.rl_ssa_start[q + 1]; ++t) om += ssa[g.ssa_idx[t]] * g.ssa_w[t];
        b.od[i] = od;
        b.omloc[i] = om;
        b.srcw[i] = exp(-od);
    }
}
// attenuation from every shell to the observer, one thread per (LOS, wavelength), wavelength fastest
__global__ void __launch_bounds__(256) kern_k0_los(Geo g, Bufs b) {
    const size_t n = size_t(g.nlos) * b.nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int l = int(i / b.nw), w = int(i % b.nw), nw = b.nw;
        const int q0 = g.ray_qstart[l], q1 = g.ray_qstart[l + 1];
        double att = 1.0, total = 0;
        for (int q = q1 - 1; q >= q0; --q) {
            const double e = b.srcw[size_t(q) * nw + w];
            total += b.od[size_t(q) * nw + w];
            b.srcw[size_t(q) * nw + w] = g.rl_active[q] ? b.omloc[size_t(q) * nw + w] * (1 - e) * att : 0.0;
            att *= e;
        }
        b.att_total[size_t(l) * nw + w] = att;
        b.los_od[size_t(w) * g.nlos + l] = total;
    }
}
// skd::k0_transmission with one thread per (wavelength, layer) -- grid.y = layer: the slant optical depths of a layer's ceiling and of the one
// above it are two dot products of a Chapman row with the layer depths (the per-wavelength form walks the layers one after the other:
// 0.3 ms with 20 wavefronts in flight at 1 250 wavelengths).  Same sums in the same order per level, so the same results.
__global__ void __launch_bounds__(kBlock) kern_k0t(Geo g, Bufs b, int s) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y, nw = b.nw, nlyr = g.nlyr;
    if (w >= nw) return;
    const double F0 = b.irradiance ? b.irradiance[w] : 1.0;
    if (p == 0) {
        b.T[w] = F0;
        b.norders[w] = g.nstr;
        b.notconv[w] = 0;
    }
    const double* ch = g.chapman + (size_t(s) * nlyr + p) * nlyr;
    if (ch[0] < 0) {
        b.T[size_t(p + 1) * nw + w] = 0.0;
        b.sec[size_t(p) * nw + w] = 1e10;
        return;
    }
    double prev = 0;   // slant depth of the level above; a level in darkness counts as -1 (the reference differences against the flag)
    if (p > 0) {
        const double* chm = ch - nlyr;
        if (chm[0] < 0) prev = -1.0;
        else for (int r = 0; r < nlyr; ++r) prev += chm[r] * b.tau[size_t(r) * nw + w];
    }
    double sig = 0;
    for (int r = 0; r < nlyr; ++r) sig += ch[r] * b.tau[size_t(r) * nw + w];
    b.sec[size_t(p) * nw + w] = (sig - prev) / b.tau[size_t(p) * nw + w];
    b.T[size_t(p + 1) * nw + w] = exp(-sig) * F0;
}
__global__ void __launch_bounds__(kBlock) kern_k4a_thermal(Geo g, Bufs b, int w0, int nw_total, double* wf_em, double* wf_sem) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(b.nw) * g.nlos) return;
    skd::k4a_thermal(g, b, int(tid % g.nlos), int(tid / g.nlos), w0, nw_total, wf_em, wf_sem);
}
__global__ void __launch_bounds__(kBlock) kern_k0_thermal(Geo g, Bufs b) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= b.nw) return;
    skd::k0_thermal(g, b, w);
}
__global__ void __launch_bounds__(kBlock) kern_stop(Geo g, Bufs b, int s, int m, int* active) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= b.nw) return;
    skd::k_stop_rule(g, b, s, m, w);
    if (b.norders[w] > m + 1) atomicAdd(active, 1);
}
__global__ void __launch_bounds__(kBlock) kern_k4a(Geo g, Bufs b, int s, int clear) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(b.nw) * g.nlos) return;
    skd::k4a_chain(g, b, s, int(tid % g.nlos), int(tid / g.nlos), clear != 0);
}
// skd::k4a_chain for the LAST solar zenith index of a solve in two steps.  kern_k4a_gt, grid.y = chunks of eight layers r: the derivative
// of everything downstream with respect to tau_r (the walk over all layers p per chunk -- the per-(wavelength, line of sight) form takes the
// chunks one after the other) written over the tau rows of the gradient; kern_k4a_apply: tau_r / omega_r -> extinction / ssa of the grid
// levels in the order of the per-thread form (same sums in the same order: results do not depend on the launch shape).  Nothing is cleared
// (the next solve's initialisation does that).
__global__ void __launch_bounds__(kBlock) kern_k4a_gt(Geo g, Bufs b, int s) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), nw = b.nw, nlyr = g.nlyr;
    auto G = [&](int e) -> double& { return b.grad[size_t(e) * nt + tid]; };
    constexpr int RC = 8;
    const int r0 = blockIdx.y * RC;
    double gt[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) gt[k] = (r0 + k < nlyr) ? G(r0 + k) : 0.0;
    for (int p = 0; p < nlyr; ++p) {
        const double* ch = g.chapman + (size_t(s) * nlyr + p) * nlyr;
        if (ch[0] < 0) continue;
        const double* chm = (p > 0) ? g.chapman + (size_t(s) * nlyr + p - 1) * nlyr : nullptr;
        const bool prev_ok = (p > 0) && !(chm[0] < 0);
        const double inv_tau = 1.0 / b.tau[size_t(p) * nw + w];
        const double gsec = G(3 * nlyr + p), sec_p = b.sec[size_t(p) * nw + w];
        const double gT = (p + 1 < nlyr) ? G(2 * nlyr + p + 1) : G(4 * nlyr + 1);
        const double Tn = b.T[size_t(p + 1) * nw + w];
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const int r = r0 + k;
            if (r >= nlyr) break;
            const double dsig_p = ch[r];
            double dsec = (dsig_p - (prev_ok ? chm[r] : 0.0)) * inv_tau;
            if (r == p) dsec -= sec_p * inv_tau;
            gt[k] += gsec * dsec - gT * Tn * dsig_p;
        }
    }
#pragma unroll
    for (int k = 0; k < RC; ++k)
        if (r0 + k < nlyr) G(r0 + k) = gt[k];
}
__global__ void __launch_bounds__(kBlock) kern_k4a_apply(Geo g, Bufs b) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    auto G = [&](int e) { return b.grad[size_t(e) * nt + tid]; };
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    for (int r = 0; r < nlyr; ++r) {
        const double gtr = G(r), gom = G(nlyr + r);
        const double kb = b.kbar[size_t(r) * nw + w], omr = b.om[size_t(r) * nw + w];
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * r + t];
            if (q < 0) continue;
            const double wt = g.wq_w[2 * r + t];
            b.dnat[size_t(q) * nt + tid] += gtr * wt * g.dh[r] + gom * wt * (ssa[q] - omr) / kb;
            b.dnat[size_t(ngrid + q) * nt + tid] += gom * wt * ext[q] / kb;
        }
    }
    b.dnat[size_t(2 * ngrid) * nt + tid] += G(4 * nlyr);
}
__global__ void __launch_bounds__(kBlock) kern_k4b(Geo g, Bufs b) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(b.nw) * g.nlos) return;
    skd::k4b_los(g, b, int(tid % g.nlos), int(tid / g.nlos));
}
// table-entry source weights of the 16-stream sweeps: cwbuf[w][e] = srcw[ray layer of e][w] * interpolation weight of e
__global__ void __launch_bounds__(256) kern_cw(Geo g, Bufs b) {
    const size_t n = size_t(b.nw) * g.n_lp;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / g.n_lp), e = int(i % g.n_lp);
        b.cwbuf[i] = b.srcw[size_t(g.lp_q[e]) * b.nw + w] * g.lp_w[e];
    }
}
// live flags of the table entries at azimuth order m, one byte per 8 entries: the adjoint sweeps test a bit instead of following
// entry -> table slot -> convergence order (a dependent load per entry, repeated by the four lanes of a line of sight)
__global__ void __launch_bounds__(256) kern_qlmask(Geo g, Bufs b, int m) {
    const int nb = (g.n_lp + 7) / 8;
    const size_t n = size_t(b.nw) * nb;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / nb), k = int(i % nb);
        if (m >= b.norders[w]) continue;
        unsigned bits = 0;
        for (int j = 0; j < 8 && 8 * k + j < g.n_lp; ++j)
            if (skd::conv_at(b, g, g.lp_slot[8 * k + j], w) > m) bits |= 1u << j;
        b.qlmask[i] = uint8_t(bits);
    }
}
// Fluxes at the flux observers from the order-0 solution (DOSourcePlaneParallelPostProcessing::compute_flux, do_source_planeparallel.cpp:221-566):
// inside layer p at optical depth x = fraction x tau the discrete-ordinates intensities at the stream angles are
//   down_l = sum_j [ W+(l, j) cL_j + W-(l, j) cM_j ],   up_l = sum_j [ W-(l, j) cL_j + W+(l, j) cM_j ],
//   cL_j = h+_j(x) L_j + A+_j D-_j(x),   cM_j = h-_j(x) M_j + A-_j D+_j(x)      (multipliers: sktran_do_postprocessing.h:19-188)
// and  downwelling = 2 pi sum_l w_l mu_l down_l  (+ the direct beam  T_ceiling cos(sza)  when the DO source carries the single scatter, as the
// reference adds it: the transmission at the layer's CEILING), upwelling the same with up_l, actinic both without the mu_l (+ T_ceiling).
// One thread per (wavelength, observer); accumulate: the thermal pass adds its solution's share (no direct beam).
template <int N> __global__ void __launch_bounds__(kBlock) kern_flux(Geo g, Bufs b, int s, int nobs, const int* __restrict__ fx_layer,
                                                                     const double* __restrict__ fx_frac, int ntypes, const int* __restrict__ types,
                                                                     double* __restrict__ flux, int w0, int nw_total, int accumulate, int direct) {
    using R = skd::Rec<N>;
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(b.nw) * nobs) return;
    const int w = int(tid / nobs), o = int(tid % nobs), nw = b.nw, nlyr = g.nlyr;
    const int p = fx_layer[o];
    const double lf = fx_frac[o];
    const double* rec = skd::rec_at<N>(b, nlyr, p, w);
    const double* xs = skd::xsol_at<N>(b, nlyr, p, w);
    const double th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    double cL[N], cM[N];
    for (int j = 0; j < N; ++j) {
        skd::PPMult pm;
        skd::pp_multipliers(T, th, sc, rec[R::K + j], lf, pm);
        cL[j] = pm.hp * xs[j] + rec[R::AP + j] * pm.Dm;
        cM[j] = pm.hm * xs[N + j] + rec[R::AM + j] * pm.Dp;
    }
    double dn = 0, up = 0, dn_a = 0, up_a = 0;
    for (int l = 0; l < N; ++l) {
        double d = 0, u = 0;
        for (int j = 0; j < N; ++j) {
            const double wp = rec[R::WP + l + N * j], wm = rec[R::WM + l + N * j];
            d += wp * cL[j] + wm * cM[j];
            u += wm * cL[j] + wp * cM[j];
        }
        const double q = 2.0 * skd::kPi * g.wt[l];
        dn += q * g.mu[l] * d;
        up += q * g.mu[l] * u;
        dn_a += q * d;
        up_a += q * u;
    }
    for (int t = 0; t < ntypes; ++t) {
        double v = 0;
        if (types[t] == 0) v = up;
        else if (types[t] == 1) v = dn + (direct ? T * g.csz[s] : 0.0);
        else if (types[t] == 2) v = dn_a + up_a + (direct ? T : 0.0);
        double* dst = &flux[(size_t(t) * nw_total + w0 + w) * nobs + o];
        *dst = accumulate ? *dst + v : v;
    }
}
// batched azimuth orders: the reduced gradient is the sum of the orders' rows IN ORDER (what the order-by-order path adds up in b.grad)
__global__ void __launch_bounds__(256) kern_grad_orders(const double* __restrict__ gord, double* __restrict__ grad, size_t n, size_t stride, int norders) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        double acc = gord[i];
        for (int m = 1; m < norders; ++m) acc += gord[size_t(m) * stride + i];
        grad[i] = acc;
    }
}
// plane-parallel post-processing: one thread per (wavelength, line of sight), see skd::kpp_los
template <int N> __global__ void __launch_bounds__(kBlock) kern_kpp(Geo g, Bufs b, int s, int m) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= size_t(b.nw) * g.nlos) return;
    skd::kpp_los<N>(g, b, s, m, int(tid % g.nlos), int(tid / g.nlos));
}
__global__ void __launch_bounds__(256) kern_init(Bufs b, const int* slot_is_ground, int nslot, size_t n_grad, size_t n_dnat, size_t n_rad) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = tid; i < size_t(nslot) * b.nw; i += stride) b.conv_order[i] = slot_is_ground[i % nslot] ? 1 : 255;
    for (size_t i = tid; i < n_grad; i += stride) b.grad[i] = 0.0;
    for (size_t i = tid; i < n_dnat; i += stride) b.dnat[i] = 0.0;
    for (size_t i = tid; i < n_rad; i += stride) b.radiance[i] = 0.0;
    for (size_t i = tid; i < size_t(b.nw); i += stride) b.status[i] = 0;
}
// radiance[w][los] and native weighting functions in the reference's user layout [altitude][w][los]
// (rust/sasktran2-rs/src/bindings/output.rs:420-425); w0 = first wavelength of this chunk, nw_total = full spectrum.
// grid.y = altitude planes (one (altitude, wavelength, line of sight) element of each weighting function per thread);
// plane 0 also moves the radiance, optical depth, albedo row and status
__global__ void __launch_bounds__(kBlock) kern_output(Geo g, Bufs b, int w0, int nw_total, double* radiance, double* wf_ext,
                                                      double* wf_ssa, double* wf_alb, double* los_od, int* status) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), l = int(tid % g.nlos), q = blockIdx.y;
    const size_t o = size_t(w0 + w) * g.nlos + l;
    if (q == 0) {
        radiance[o] = b.radiance[tid];
        if (los_od) los_od[o] = b.los_od[tid];
        if (status && l == 0) status[w0 + w] = b.status[w];
        if (b.derivs && wf_alb) wf_alb[o] = b.dnat[size_t(2 * g.ngrid) * nt + tid];
    }
    if (!b.derivs) return;
    const size_t plane = size_t(nw_total) * g.nlos;
    if (wf_ext) __builtin_nontemporal_store(b.dnat[size_t(q) * nt + tid], &wf_ext[size_t(q) * plane + o]);
    if (wf_ssa) __builtin_nontemporal_store(b.dnat[size_t(g.ngrid + q) * nt + tid], &wf_ssa[size_t(q) * plane + o]);
}

// the same results as one wavelength-major slab row per wavelength (DeviceEngine::set_slab_output): fields radiance, los_od and, with
// derivatives, wf_extinction[q], wf_ssa[q], wf_albedo, each nlos doubles
__global__ void __launch_bounds__(kBlock) kern_output_slab(Geo g, Bufs b, int w0, double* slab, int* status) {
    size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), l = int(tid % g.nlos), q = blockIdx.y;   // grid.y = altitude planes, as kern_output
    const int nf = b.derivs ? 2 * g.ngrid + 3 : 2;
    double* row = slab + size_t(w0 + w) * nf * g.nlos + l;
    if (q == 0) {
        row[0] = b.radiance[tid];
        row[g.nlos] = b.los_od[tid];
        if (status && l == 0) status[w0 + w] = b.status[w];
        if (b.derivs) row[size_t(2 + 2 * g.ngrid) * g.nlos] = b.dnat[size_t(2 * g.ngrid) * nt + tid];
    }
    if (!b.derivs) return;
    row[size_t(2 + q) * g.nlos] = b.dnat[size_t(q) * nt + tid];
    row[size_t(2 + g.ngrid + q) * g.nlos] = b.dnat[size_t(g.ngrid + q) * nt + tid];
}

// ------------------------------------------------------------------------------------------------
struct DevMem {
    std::vector<void*> ptrs;
    template <class T> T* alloc(size_t n) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
    template <class T> T* upload(const std::vector<T>& v) {
        T* p = alloc<T>(v.size());
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
    ~DevMem() { release(); }
};

struct DeviceEngine::Impl {
    skhost::GeoTables tab;
    Geo g{};
    DevMem geo_mem;
    int* d_slot_is_ground = nullptr;
    int* d_vec_active = nullptr;  // dummy counter of the stop-rule kernel in the polarised driver
    // One "slot" = scratch buffers + stream for one block of wavelengths in flight.  Blocks of a spectrum are dealt
    // round-robin to the slots; kernels of different slots run concurrently (a single one of these latency-bound
    // kernels cannot fill the SIMDs by itself).
    struct Slot {
        DevMem mem;
        Bufs b{};
        int* d_active = nullptr;
        const double* thermal_surface = nullptr;   // surface emission of the block in flight (thermal pass)
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        // look-ahead of the layer solutions: K1 of order m+1 only needs the layer optics, so it runs on its own stream into
        // the second record buffer while K2a..K2b of order m still work on the first
        hipStream_t k1_stream = nullptr, adj_stream = nullptr;
        // order pipeline: records, BVP factors and solutions of up to kDepth azimuth orders in flight (buffer 0 is Bufs' own)
        static constexpr int kDepth = 4;
        double* recx[kDepth - 1] = {};
        double* dinvx[kDepth - 1] = {};
        double* xsolx[kDepth - 1] = {};
        uint8_t* qlmaskx[kDepth - 1] = {};   // live-flag rows: formed on the BVP stream right after the table values, read by the adjoint stream
        std::vector<hipEvent_t> ev_k1, ev_bvp, ev_rest;  // per azimuth order
        hipEvent_t ev_begin = nullptr;
        // batched azimuth orders (small launches): one set of per-order buffers for every order, allocated when first needed
        DevMem batch_mem;
        int batch_cap = 0;
        double *bt_rec = nullptr, *bt_dinv = nullptr, *bt_xsol = nullptr, *bt_eig = nullptr, *bt_ops = nullptr, *bt_z = nullptr, *bt_hw = nullptr, *bt_grad = nullptr;
        uint8_t* bt_qlmask = nullptr;
        // scattering groups on the matrix-core path (ensure_group_scratch): direction, derivative blocks (one per pipeline buffer), group
        // operator rows, HD rows of the forward sweep, gradient / native-derivative rows extended by the groups' rows
        DevMem group_mem;
        double *ga = nullptr, *recg[kDepth] = {}, *opsg = nullptr, *hg = nullptr, *grad_g = nullptr, *dnat_g = nullptr;
        double *grad0 = nullptr, *dnat0 = nullptr;   // the plain rows (ensure_scratch)
    };
    std::vector<std::unique_ptr<Slot>> slots;
    // scattering groups of the solve in hand (set_scattering_groups): count, d_legendre [group][wavelength][q][l], d_f [group][wavelength][q] or
    // null, output [group][q][wavelength][los]
    int ngroups = 0, group_cap_w = 0, group_cap_g = 0;
    const double *g_dleg = nullptr, *g_dfg = nullptr;
    double* g_wf_scat = nullptr;
    double *d_k1g_pp = nullptr, *d_k1g_q = nullptr;   // kern_k1g_tables, once per engine
    hipEvent_t fork = nullptr;
    int chunk_capacity = 0;
    int last_chunk = 0, last_slots = 0;
    const double* delta_f = nullptr;
    double* slab = nullptr;   // set_slab_output
    const double *thermal_emission = nullptr, *thermal_surface = nullptr;   // EmissionSource::discrete_ordinates inputs (device) or null
    double *thermal_wf_emission = nullptr, *thermal_wf_surface = nullptr;   // their weighting functions [ngrid][nwavel][nlos], [nwavel][nlos] or null
    int thermal_w0 = 0, thermal_nw_total = 0;                               // block in flight of the thermal pass
    // flux observers: layer / fraction per observer (device), requested flux types and the output of the solve in hand
    int nfx = 0, nflux_types = 0;
    int brdf_kind = 0, brdf_nargs = 1;        // non-Lambertian surface of the next polarised / run-time-n solves (set_surface_brdf)
    const double* brdf_args = nullptr;         // [nwavel][nargs] (device)
    int forced_azimuth = 0;   // > 0: the plane-parallel post-processing solves exactly this many azimuth orders (Config::num_do_forced_azimuth)
    int *d_fx_layer = nullptr, *d_flux_types = nullptr;
    double *d_fx_frac = nullptr, *d_flux = nullptr;
    bool profiling = false;
    // in-situ timing (set_profiling(2)): an event pair around every launch ON ITS OWN STREAM, nothing serialised, read back by
    // kernel_time() -- the duration of a kernel as it runs inside the pipelined step (what rocprofv3 --kernel-trace sees)
    bool in_situ = false;
    struct Pending { const char* name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<double, long>> times;  // name -> (ms, launches)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t pooled_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
    void collect() {
        for (const Pending& q : pending) {
            (void)hipEventSynchronize(q.b);
            float ms = 0;
            if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) {
                auto& e = times[q.name];
                e.first += ms;
                e.second += 1;
            }
            event_pool.push_back(q.a);
            event_pool.push_back(q.b);
        }
        pending.clear();
    }
    size_t scratch_bytes = 0;
};

DeviceEngine::DeviceEngine(const skgeo::DiscoGeometry& dg, bool ss_in_do) : impl_(new Impl) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw std::runtime_error("sasktran2_amd: no HIP device available (the MI355X path has no CPU fallback)");
    Impl& I = *impl_;
    I.tab.include_ss = ss_in_do ? 1 : 0;
    I.tab.build(dg);
    const auto& t = I.tab;
    const bool matrix_path = (t.N == 1 || t.N == 2 || t.N == 4 || t.N == 8 || t.N == 16);
    vector_mode_ = (dg.nstokes == 3) || !matrix_path;  // radiances through the run-time-n task bodies (disco_vec_core.hpp)
    if (dg.nstokes != 1 && dg.nstokes != 3) throw std::runtime_error("sasktran2_amd: NSTOKES must be 1 or 3");
    Geo& g = I.g;
    t.fill_scalars(g);
    DevMem& M = I.geo_mem;
    g.mu = M.upload(t.mu); g.wt = M.upload(t.wt); g.lpmu = M.upload(t.lpmu); g.lpcsz = M.upload(t.lpcsz); g.lpang = M.upload(t.lpang);
    g.dh = M.upload(t.dh); g.lfrac = M.upload(t.lfrac); g.wq_idx = M.upload(t.wq_idx); g.wq_w = M.upload(t.wq_w);
    g.chapman = M.upload(t.chapman); g.csz = M.upload(t.csz); g.pt_start = M.upload(t.pt_start); g.pt_angle = M.upload(t.pt_angle);
    g.pt_slot = M.upload(t.pt_slot); g.gpt_start = M.upload(t.gpt_start); g.gpt_slot = M.upload(t.gpt_slot);
    g.ray_qstart = M.upload(t.ray_qstart); g.ray_ground = M.upload(t.ray_ground); g.ray_gsaa = M.upload(t.ray_gsaa);
    g.rl_od_start = M.upload(t.rl_od_start); g.od_idx = M.upload(t.od_idx); g.od_w = M.upload(t.od_w);
    g.rl_ssa_start = M.upload(t.rl_ssa_start); g.ssa_idx = M.upload(t.ssa_idx); g.ssa_w = M.upload(t.ssa_w);
    g.rl_azi = M.upload(t.rl_azi); g.rl_cosm = M.upload(t.rl_cosm); g.rl_cosm_t = M.upload(t.rl_cosm_t); g.rl_active = M.upload(t.rl_active); g.rl_tab_start = M.upload(t.rl_tab_start);
    g.tab_slot = M.upload(t.tab_slot); g.tab_w = M.upload(t.tab_w); g.lp_start = M.upload(t.lp_start); g.lp_q = M.upload(t.lp_q);
    g.lp_angle = M.upload(t.lp_angle); g.lp_slot = M.upload(t.lp_slot); g.lp_w = M.upload(t.lp_w); g.lp_pc = M.upload(t.lp_pc); g.g_start = M.upload(t.g_start);
    g.g_slot = M.upload(t.g_slot); g.g_sza = M.upload(t.g_sza); g.g_w = M.upload(t.g_w);
    g.pp_mu = M.upload(t.pp_mu); g.pp_azi = M.upload(t.pp_azi); g.pp_first = M.upload(t.pp_first); g.pp_lf = M.upload(t.pp_lf);
    g.lplos = M.upload(t.lplos);
    I.d_slot_is_ground = M.upload(t.slot_is_ground);
    I.d_vec_active = M.alloc<int>(1);
    HIP_CHECK(hipEventCreateWithFlags(&I.fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreate(&I.ev0));
    HIP_CHECK(hipEventCreate(&I.ev1));
    init_vector(dg);   // the run-time-n path also serves scattering-group derivatives of engines on the matrix-core path
}

DeviceEngine::~DeviceEngine() {
    if (impl_) {
        for (auto& sl : impl_->slots) {
            if (sl->stream) (void)hipStreamDestroy(sl->stream);
            if (sl->k1_stream) (void)hipStreamDestroy(sl->k1_stream);
            if (sl->adj_stream) (void)hipStreamDestroy(sl->adj_stream);
            for (auto e : sl->ev_bvp) (void)hipEventDestroy(e);
            if (sl->done) (void)hipEventDestroy(sl->done);
            if (sl->ev_begin) (void)hipEventDestroy(sl->ev_begin);
            for (auto e : sl->ev_k1) (void)hipEventDestroy(e);
            for (auto e : sl->ev_rest) (void)hipEventDestroy(e);
        }
        impl_->collect();
        for (hipEvent_t e : impl_->event_pool) (void)hipEventDestroy(e);
        if (impl_->fork) (void)hipEventDestroy(impl_->fork);
        if (impl_->ev0) (void)hipEventDestroy(impl_->ev0);
        if (impl_->ev1) (void)hipEventDestroy(impl_->ev1);
        destroy_vector();
        delete impl_;
    }
}

int DeviceEngine::nlos() const { return impl_->tab.nlos; }
int DeviceEngine::ngrid() const { return impl_->tab.ngrid; }
int DeviceEngine::nstr() const { return impl_->tab.nstr; }
size_t DeviceEngine::scratch_bytes() const { return impl_->scratch_bytes; }
void DeviceEngine::set_delta_m_fraction(const double* d_f) { impl_->delta_f = d_f; }
void DeviceEngine::set_slab_output(double* d_slab) { impl_->slab = d_slab; }
void DeviceEngine::set_thermal_emission(const double* d_emission, const double* d_surface_emission) {
    impl_->thermal_emission = d_emission;
    impl_->thermal_surface = d_surface_emission;
}
void DeviceEngine::set_thermal_wf_outputs(double* d_wf_emission, double* d_wf_surface_emission) {
    impl_->thermal_wf_emission = d_wf_emission;
    impl_->thermal_wf_surface = d_wf_surface_emission;
}
void DeviceEngine::set_flux_observers(const std::vector<int>& layer, const std::vector<double>& fraction) {
    Impl& I = *impl_;
    I.nfx = int(layer.size());
    I.d_fx_layer = I.geo_mem.upload(layer);
    I.d_fx_frac = I.geo_mem.upload(fraction);
    I.d_flux_types = I.geo_mem.alloc<int>(8);
}
void DeviceEngine::set_flux_output(double* d_flux, const std::vector<int>& types) {
    Impl& I = *impl_;
    I.d_flux = d_flux;
    I.nflux_types = int(std::min<size_t>(types.size(), 8));
    if (d_flux && I.nflux_types > 0) HIP_CHECK(hipMemcpy(I.d_flux_types, types.data(), size_t(I.nflux_types) * sizeof(int), hipMemcpyHostToDevice));
}
int DeviceEngine::num_flux_observers() const { return impl_->nfx; }
void DeviceEngine::set_surface_brdf(int kind, int nargs, const double* d_args) {
    if (kind < 0 || kind > 3 || (kind != 0 && (!d_args || nargs != (kind == 2 ? 3 : 1)))) throw std::runtime_error("sasktran2_amd: bad BRDF kind / argument count");
    impl_->brdf_kind = kind;
    impl_->brdf_nargs = kind == 0 ? 1 : nargs;
    impl_->brdf_args = kind == 0 ? nullptr : d_args;
}
void DeviceEngine::set_forced_azimuth(int n) {
    if (n > 0 && !impl_->g.pp_mode) throw std::runtime_error("sasktran2_amd: num_do_forced_azimuth is honoured by the plane-parallel post-processing only");
    impl_->forced_azimuth = n;
}
int DeviceEngine::last_chunk() const { return impl_->last_chunk; }
int DeviceEngine::last_slots() const { return impl_->last_slots; }
void DeviceEngine::set_profiling(bool on) { impl_->collect(); impl_->profiling = on; impl_->in_situ = false; }
void DeviceEngine::set_in_situ_timing(bool on) { impl_->collect(); impl_->in_situ = on; impl_->profiling = false; }
void DeviceEngine::reset_profile() { impl_->collect(); impl_->times.clear(); }
bool DeviceEngine::kernel_time(const char* name, double* ms, long* launches) const {
    impl_->collect();
    auto it = impl_->times.find(name);
    if (it == impl_->times.end()) return false;
    *ms = it->second.first;
    *launches = it->second.second;
    return true;
}
std::string DeviceEngine::kernel_names() const {
    impl_->collect();
    std::string s;
    for (auto& kv : impl_->times) s += (s.empty() ? "" : ",") + kv.first;
    return s;
}

size_t DeviceEngine::bytes_per_wavelength() const {
    const auto& t = impl_->tab;
    const size_t N = t.N, nlyr = t.nlyr, nstr = t.nstr, nlos = t.nlos;
    size_t d = 0;
    d += nlyr * (3 + nstr) + (nlyr + 1) + nlyr;              // tau, om, kbar, a, T, sec
    d += size_t(t.nq_total) * 3 + nlos;                      // od, omloc, srcw, att_total
    if (N == 8) d += t.lp_q.size() + Impl::Slot::kDepth * (t.lp_q.size() / 64 + 1) + nlyr * (N + N * N) + nlyr * 18; // cwbuf, qlmask rows, eig, at
    d += Impl::Slot::kDepth * (nlyr * (4 * N * N + 41 * N) + nlyr * 4 * N * N + nlyr * 2 * N);  // rec, dinv, xsol (kDepth buffers each)
    d += nstr * size_t(t.nslot);                             // J
    const size_t nlos_pad = (nlos + 15) / 16 * 16;
    d += nlyr * (8 * N * N + 26 * N + 5) + nlyr * nstr * nlos_pad;  // ops, qlbuf
    d += nlyr * 2 * N * nlos_pad + (4 * nlyr + 2) * nlos + (2 * t.ngrid + 1) * nlos + 2 * nlos;  // z, grad, dnat, radiance, los_od
    return d * sizeof(double) + size_t(t.nslot) + 3 * sizeof(int);
}

void DeviceEngine::ensure_scratch(int nw, int nslots) {
    Impl& I = *impl_;
    if (nw <= I.chunk_capacity && nslots <= (int)I.slots.size()) return;
    nw = std::max(nw, I.chunk_capacity);
    for (auto& sl : I.slots) sl->mem.release();
    while ((int)I.slots.size() < nslots) {
        auto sl = std::make_unique<Impl::Slot>();
        HIP_CHECK(hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&sl->k1_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&sl->adj_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sl->ev_begin, hipEventDisableTiming));
        sl->ev_k1.resize(I.tab.nstr);
        sl->ev_bvp.resize(I.tab.nstr);
        sl->ev_rest.resize(I.tab.nstr);
        for (int i = 0; i < I.tab.nstr; ++i) {
            HIP_CHECK(hipEventCreateWithFlags(&sl->ev_k1[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&sl->ev_bvp[i], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&sl->ev_rest[i], hipEventDisableTiming));
        }
        I.slots.push_back(std::move(sl));
    }
    const auto& t = I.tab;
    const size_t N = t.N, nlyr = t.nlyr, nstr = t.nstr, nlos = t.nlos, W = nw, nt = W * nlos;
    const size_t nlos_pad = (nlos + 15) / 16 * 16;  // the 16-stream sweeps keep whole 16-LOS tiles
    for (auto& sl : I.slots) {
        DevMem& M = sl->mem;
        Bufs& b = sl->b;
        b.tau = M.alloc<double>(nlyr * W); b.om = M.alloc<double>(nlyr * W); b.kbar = M.alloc<double>(nlyr * W);
        b.a = M.alloc<double>(nlyr * nstr * W); b.T = M.alloc<double>((nlyr + 1) * W); b.sec = M.alloc<double>(nlyr * W);
        b.at = M.alloc<double>(nstr == 16 ? nlyr * 18 * W : 1);
        b.od = M.alloc<double>(size_t(t.nq_total) * W); b.omloc = M.alloc<double>(size_t(t.nq_total) * W);
        b.srcw = M.alloc<double>(size_t(t.nq_total) * W); b.att_total = M.alloc<double>(nlos * W);
        b.cwbuf = M.alloc<double>(t.N == 8 ? t.lp_q.size() * W : 1);
        b.eig = M.alloc<double>(t.N == 8 ? nlyr * (N + N * N) * W : 1);
        b.qlmask = M.alloc<uint8_t>(t.N == 8 ? (t.lp_q.size() + 7) / 8 * W + 8 : 8);
        b.rec = M.alloc<double>(nlyr * (4 * N * N + 41 * N) * W); b.dinv = M.alloc<double>(nlyr * 4 * N * N * W);
        for (int k = 0; k < Impl::Slot::kDepth - 1; ++k) {
            sl->recx[k] = M.alloc<double>(nlyr * (4 * N * N + 41 * N) * W);
            sl->dinvx[k] = M.alloc<double>(nlyr * 4 * N * N * W);
            sl->xsolx[k] = M.alloc<double>(nlyr * 2 * N * W);
            sl->qlmaskx[k] = M.alloc<uint8_t>(t.N == 8 ? (t.lp_q.size() + 7) / 8 * W + 8 : 8);
        }
        b.xsol = M.alloc<double>(nlyr * 2 * N * W); b.J = M.alloc<double>(nstr * size_t(t.nslot) * W);
        b.conv_order = M.alloc<uint8_t>(size_t(t.nslot) * W); b.notconv = M.alloc<int>(W); b.norders = M.alloc<int>(W);
        b.status = M.alloc<int>(W); b.z = M.alloc<double>(nlyr * 2 * N * W * nlos_pad);
        b.ops = M.alloc<double>(nlyr * (8 * N * N + 26 * N + 5) * W); b.qlbuf = M.alloc<double>(nlyr * nstr * W * nlos_pad);
        b.grad = M.alloc<double>((4 * nlyr + 2) * nt);
        b.dnat = M.alloc<double>((2 * t.ngrid + 1) * nt); b.radiance = M.alloc<double>(nt); b.los_od = M.alloc<double>(nt);
        sl->grad0 = b.grad;
        sl->dnat0 = b.dnat;
        sl->d_active = M.alloc<int>(1);
        sl->group_mem.release();
    }
    I.group_cap_w = I.group_cap_g = 0;
    I.chunk_capacity = nw;
    I.scratch_bytes = bytes_per_wavelength() * W * I.slots.size();
}

static bool k1_lane_enabled();
// scratch of the scattering groups on the matrix-core path, allocated when a solve first asks for them (3.6 KB per wavelength, layer and group
// for the derivative blocks of the four pipeline buffers and the operator rows, 0.5 KB per tile for the HD rows)
void DeviceEngine::ensure_group_scratch(int ngroups) {
    Impl& I = *impl_;
    const int nw = I.chunk_capacity;
    if (ngroups <= I.group_cap_g && nw <= I.group_cap_w) return;
    const auto& t = I.tab;
    const size_t N = t.N, nlyr = t.nlyr, nstr = t.nstr, nlos = t.nlos, W = nw, nt = W * nlos, G = ngroups, ntile = (nlos + 15) / 16;
    for (auto& sl : I.slots) {
        DevMem& M = sl->group_mem;
        M.release();
        sl->ga = M.alloc<double>(G * nlyr * nstr * W);
        for (int k = 0; k < Impl::Slot::kDepth; ++k) sl->recg[k] = M.alloc<double>(W * nlyr * G * skd::RecG<8>::SIZE);
        sl->opsg = M.alloc<double>(W * nlyr * G * skd::OpsG<8>::SIZE);
        sl->hg = M.alloc<double>(W * nlyr * ntile * 64);
        sl->grad_g = M.alloc<double>((4 * nlyr + 2 + G * nlyr) * nt);
        sl->dnat_g = M.alloc<double>((2 * t.ngrid + 1 + G * t.ngrid) * nt);
        (void)N;
    }
    I.group_cap_g = ngroups;
    I.group_cap_w = nw;
    if (!I.d_k1g_pp) {
        const size_t NP = N * (N + 1) / 2;
        I.d_k1g_pp = I.geo_mem.alloc<double>(nstr * NP * nstr);
        I.d_k1g_q = I.geo_mem.alloc<double>(size_t(std::max(I.g.nsza, 1)) * nstr * N * nstr);
        hipLaunchKernelGGL(kern_k1g_tables, dim3(64), dim3(256), 0, nullptr, I.g, I.d_k1g_pp, I.d_k1g_q);
        HIP_CHECK(hipDeviceSynchronize());
    }
}
void DeviceEngine::set_scattering_groups(int ngroups, const double* d_dleg, double* d_wf_scat, const double* d_dfg) {
    Impl& I = *impl_;
    if (ngroups > 0 && !groups_on_matrix_path(ngroups)) throw std::runtime_error("sasktran2_amd: scattering groups are not served by the matrix-core kernels of this engine");
    I.ngroups = ngroups;
    I.g_dleg = d_dleg;
    I.g_wf_scat = d_wf_scat;
    I.g_dfg = d_dfg;
}
// 16 streams, spherical post-processing, lane-path records, at most two 16-LOS tiles (the fused forward sweep) and two groups
bool DeviceEngine::groups_on_matrix_path(int ngroups) const {
    const auto& t = impl_->tab;
    return !vector_mode_ && t.N == 8 && !impl_->g.pp_mode && k1_lane_enabled() && t.nlos <= 32 && !t.lp_q.empty() && ngroups <= 2;
}

namespace {
struct Timer {
    DeviceEngine::Impl& I;
    hipStream_t st;
    const char* name;
    hipEvent_t a = nullptr;
    Timer(DeviceEngine::Impl& i, hipStream_t s, const char* n) : I(i), st(s), name(n) {
        if (I.profiling) (void)hipEventRecord(I.ev0, st);
        else if (I.in_situ) {
            a = I.pooled_event();
            (void)hipEventRecord(a, st);
        }
    }
    ~Timer() {
        if (a) {
            hipEvent_t b = I.pooled_event();
            (void)hipEventRecord(b, st);
            I.pending.push_back({name, a, b});
            return;
        }
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
inline unsigned blocks(size_t n, int bs = kBlock) { return unsigned((n + bs - 1) / bs); }
}  // namespace

// With few wavelengths per launch one wavefront per wavelength cannot fill the device: the layer-parallel kernels (K1, K3,
// K2c) then split the layers over blockIdx.y and the 16-stream adjoint sweeps one 16-LOS tile per block.
static int small_batch_split(int nw) {
    if (const char* e = std::getenv("SASKTRAN2_AMD_SPLIT")) return std::max(1, atoi(e));
    // (measured with the fused adjoint kernel, MI355X: 2 500 wavelengths 39.4 ms with two blocks per wavelength, 35.3 with one;
    // 1 250: 22.0 ms with four, 23.0 with two, 23.8 with one)
    return nw >= 2000 ? 1 : (nw >= 1500 ? 2 : 4);
}
// Eigen-solutions ahead of K1: 8 and 16 streams solve the eigenproblems of all (layer, wavelength) pairs one per LANE in registers
// (kern_kj_lane) and the cooperative kernel continues from there.  Returns whether the kernel was launched.
template <int N>
static bool launch_kj(hipStream_t st, const Geo& g, const Bufs& b, int m) {
    if constexpr (N == 4 || N == 8) {
        static const bool lane_jacobi = [] { const char* e = std::getenv("SASKTRAN2_AMD_KJ"); return e ? atoi(e) != 0 : true; }();
        if (lane_jacobi) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_kj_lane<N>), dim3(unsigned((b.nw + 63) / 64), unsigned(g.nlyr)), dim3(64), 0, st, g, b, m);
            return true;
        }
    }
    return false;
}
// 16 streams, spherical post-processing: the whole layer solution one (layer, wavelength) per lane -- kern_kj_lane<8, true> hands the
// eigen-solutions over in b.eig, kern_k1_lane writes the records in the column-block layout skd::RecC (with the coefficient vectors of the
// particular solution in place of its derivative rows, see Bufs::k1_lane); the BVP, table-value and adjoint-operator kernels are
// instantiated for that layout.  Default since round 4 (round 3 wrote the standard layout in 64-byte runs scattered over each lane's
// record: 2.1 x the bytes in HBM write traffic, and lost to the cooperative kernel).  SASKTRAN2_AMD_K1=wave keeps kern_kj_lane +
// kern_k1p_wave with standard records (the parity suite compares the two).
static bool k1_lane_enabled() {
    const char* e = std::getenv("SASKTRAN2_AMD_K1");
    return !(e && std::string(e) == "wave");
}
// (SASKTRAN2_AMD_K1_FUSED=0: the eigen stage as a kernel of its own, kern_kj_lane<8, true>, handing over through b.eig)
static bool k1_fused() {
    static const bool on = [] { const char* e = std::getenv("SASKTRAN2_AMD_K1_FUSED"); return e ? atoi(e) != 0 : true; }();
    return on;
}
static void launch_k1_lane8(hipStream_t st, const Geo& g, const Bufs& b, int s, int m, unsigned norders = 1) {
    const dim3 grid(unsigned((b.nw + 63) / 64), unsigned(g.nlyr), norders);
    if (k1_fused()) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k1_lane<8, true>), grid, dim3(64), 0, st, g, b, s, m);
        return;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_kj_lane<8, true>), grid, dim3(64), 0, st, g, b, m);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k1_lane<8, false>), grid, dim3(64), 0, st, g, b, s, m);
}
// K1 launch: lane-group width per layer and the split of the layers over blockIdx.y
template <int N>
static void launch_k1(hipStream_t st, const Geo& g, const Bufs& b, int s, int m, bool pre) {
    constexpr int LG = (N >= 16) ? 64 : SKD_K1_LG;
    const int split = small_batch_split(b.nw);
    const int ng = 64 / LG;
    int lpb = (g.nlyr + split - 1) / split;
    lpb = (lpb + ng - 1) / ng * ng;
    const int ny = (g.nlyr + lpb - 1) / lpb;
    if constexpr (N == 4 || N == 8) {
        if (pre) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k1p_wave<N, LG>), dim3(unsigned(b.nw), unsigned(ny)), dim3(64), 0, st, g, b, s, m, lpb);
            return;
        }
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k1_wave<N, LG>), dim3(unsigned(b.nw), unsigned(ny)), dim3(64), 0, st, g, b, s, m, lpb);
}

template <int N>
static void launch_k3(hipStream_t st, const Geo& g, const Bufs& b, int s, int m, int split_override = 0) {
    constexpr int NG = (N >= 16) ? 2 : 4;  // layers per pass of kern_k3_wave
    const int split = split_override > 0 ? split_override : small_batch_split(b.nw);
    int lpb = (g.nlyr + split - 1) / split;
    lpb = (lpb + NG - 1) / NG * NG;
    const int ny = (g.nlyr + lpb - 1) / lpb;
    if constexpr (N == 8) {
        if (b.k1_lane) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k3_wave<8, skd::RecC<8>>), dim3(unsigned(b.nw), unsigned(ny)), dim3(64), 0, st, g, b, s, m, lpb);
            return;
        }
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k3_wave<N>), dim3(unsigned(b.nw), unsigned(ny)), dim3(64), 0, st, g, b, s, m, lpb);
}

// Small launches, 16 streams, spherical: ALL azimuth orders of the block in ONE launch per kernel (gridDim.z = order).  The orders are
// independent of each other up to the diffuse-table values -- whose stop rule (do_source_diffuse_storage.cpp:1033-1060) compares order m with
// m - 1 and m - 2 --, so the layer solutions and boundary-value problems of every order run side by side, the table values and the stop rule
// order after order (short kernels), the adjoint operators and sweeps again side by side in groups of four orders behind them, every order
// adding into gradient rows of its own that kern_grad_orders sums in order.  A rank's share of a sharded spectrum (1 250 wavelengths of 10 000 on
// eight GPUs) otherwise is a chain of 16 x 5 kernels that each fill a fraction of the chip: the boundary-value recursion alone 0.9 ms per
// order.  Same kernels, same statements, same order of the sums: bit-identical to the order-by-order path (tests/test_gpu_parity.py).
// Returns false (nothing launched) when the per-order buffers do not fit the free HBM.
static bool run_orders_batched(DeviceEngine::Impl& I, DeviceEngine::Impl::Slot& slot, hipStream_t st, int derivs, int s) {
    using RC = skd::RecC<8>;
    constexpr int N = 8, NORD = 16, GROUP = 4;
    const Geo& g = I.g;
    Bufs& b = slot.b;
    const size_t nw = b.nw, nt = nw * g.nlos, nlyr = g.nlyr;
    const int ntile = (g.nlos + 15) / 16;
    const size_t nqm = (size_t(g.n_lp) + 7) / 8;
    if (int(nw) > slot.batch_cap) {
        const size_t W = nw;
        const size_t per_order = (nlyr * (4 * N * N + 41 * N) + nlyr * 4 * N * N + nlyr * 2 * N + nlyr * (N + N * N) + nlyr * skd::Ops<N>::SIZE +
                                  nlyr * ntile * 256 + nlyr * ntile * 64 + (4 * nlyr + 2) * g.nlos) * sizeof(double) + nqm + 8;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        slot.batch_mem.release();
        slot.batch_cap = 0;
        if (per_order * NORD * W > size_t(0.6 * double(free_b))) return false;
        DevMem& M = slot.batch_mem;
        slot.bt_rec = M.alloc<double>(NORD * nlyr * (4 * N * N + 41 * N) * W);
        slot.bt_dinv = M.alloc<double>(NORD * nlyr * 4 * N * N * W);
        slot.bt_xsol = M.alloc<double>(NORD * nlyr * 2 * N * W);
        slot.bt_eig = M.alloc<double>(NORD * nlyr * (N + N * N) * W);
        slot.bt_ops = M.alloc<double>(NORD * nlyr * skd::Ops<N>::SIZE * W);
        slot.bt_z = M.alloc<double>(NORD * nlyr * ntile * 256 * W);
        slot.bt_hw = M.alloc<double>(NORD * nlyr * ntile * 64 * W);
        slot.bt_grad = M.alloc<double>(NORD * (4 * nlyr + 2) * g.nlos * W);
        slot.bt_qlmask = M.alloc<uint8_t>(NORD * (nqm * W + 8));
        slot.batch_cap = int(W);
    }
    const size_t W = size_t(slot.batch_cap);
    Bufs bb = b;
    bb.rec = slot.bt_rec; bb.dinv = slot.bt_dinv; bb.xsol = slot.bt_xsol; bb.eig = slot.bt_eig; bb.ops = slot.bt_ops;
    bb.z = slot.bt_z; bb.qlbuf = slot.bt_hw; bb.qlmask = slot.bt_qlmask; bb.grad = slot.bt_grad;
    bb.ord_rec = nlyr * (4 * N * N + 41 * N) * W; bb.ord_dinv = nlyr * 4 * N * N * W; bb.ord_xsol = nlyr * 2 * N * W;
    bb.ord_eig = nlyr * (N + N * N) * W; bb.ord_ops = nlyr * skd::Ops<N>::SIZE * W; bb.ord_z = nlyr * ntile * 256 * W;
    bb.ord_qlbuf = nlyr * ntile * 64 * W; bb.ord_qlmask = nqm * W + 8; bb.ord_grad = (4 * nlyr + 2) * g.nlos * W;
    // the sets from order m on (a batched launch that starts at order m: blockIdx.z counts from there), and one order's set alone for the
    // kernels that run order by order
    auto from_order = [&](int m) {
        Bufs o = bb;
        o.rec += m * bb.ord_rec; o.dinv += m * bb.ord_dinv; o.xsol += m * bb.ord_xsol; o.eig += m * bb.ord_eig; o.ops += m * bb.ord_ops;
        o.z += m * bb.ord_z; o.qlbuf += m * bb.ord_qlbuf; o.qlmask += m * bb.ord_qlmask; o.grad += m * bb.ord_grad;
        return o;
    };
    auto order_bufs = [&](int m) {
        Bufs o = from_order(m);
        o.ord_rec = o.ord_dinv = o.ord_xsol = o.ord_eig = o.ord_ops = o.ord_z = o.ord_qlbuf = o.ord_qlmask = o.ord_grad = 0;
        return o;
    };
    const bool adj = derivs && g.n_lp > 0;
    hipStream_t sC = (!I.profiling && slot.adj_stream) ? slot.adj_stream : st;
    if (adj) HIP_CHECK(hipMemsetAsync(slot.bt_grad, 0, size_t(NORD) * bb.ord_grad * sizeof(double), st));
    {
        Timer t(I, st, "k1_layer_solution");
        launch_k1_lane8(st, g, bb, s, 0, NORD);
    }
    {
        Timer t(I, st, "k2a_bvp");
        const dim3 grid(unsigned((nw + 3) / 4), 1, NORD);
        if (size_t(grid.x) * NORD <= 1024) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8_lone<RC>), grid, dim3(64), 0, st, g, bb, s, 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8<RC>), grid, dim3(64), 0, st, g, bb, s, 0);
    }
    const bool fused = ntile <= 2 && adj;
    for (int m = 0; m < NORD; ++m) {
        const Bufs bm = order_bufs(m);
        {
            Timer t(I, st, "k3_table_values");
            // (one pass of four layers per block: the table values are the one stage that runs order after order here, so the length of a
            // block's chain of passes is what the step waits for -- 16 x)
            launch_k3<8>(st, g, bm, s, m, (g.nlyr + 3) / 4);
        }
        if (adj) {
            Timer t(I, st, "k2q_live_flags");
            hipLaunchKernelGGL(kern_qlmask, dim3(2048), dim3(256), 0, st, g, bm, m);
        }
        {
            Timer t(I, st, "k_stop_rule");
            hipLaunchKernelGGL(kern_stop, dim3(blocks(nw)), dim3(kBlock), 0, st, g, bm, s, m, slot.d_active);
        }
        if (adj && (m % GROUP == GROUP - 1 || m == NORD - 1)) {
            // adjoint stage of the orders m0 .. m behind their table values (it only tests conv_order > order, which no later order changes)
            const int m0 = m / GROUP * GROUP, cnt = m - m0 + 1;
            const Bufs bg = from_order(m0);
            if (sC != st) {
                HIP_CHECK(hipEventRecord(slot.ev_bvp[m], st));
                HIP_CHECK(hipStreamWaitEvent(sC, slot.ev_bvp[m], 0));
            }
            {
                Timer t(I, sC, "k2c_adjoint_ops");
                if (fused && ntile == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 2>), dim3(unsigned(nw), 1, unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0, g.nlyr);
                else if (fused) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 1>), dim3(unsigned(nw), 1, unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0, g.nlyr);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true>), dim3(unsigned(nw), 1, unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0, g.nlyr);
            }
            Timer t(I, sC, "k2b_adjoint");
            if (fused && ntile == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<2, true>), dim3(unsigned(nw), 1, unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0);
            else if (fused) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<1, true>), dim3(unsigned(nw), 1, unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<1>), dim3(unsigned(nw), unsigned(std::min(ntile, 4)), unsigned(cnt)), dim3(64), 0, sC, g, bg, s, m0);
        }
    }
    if (s == g.nsza - 1) {
        Timer t(I, st, "k4b_los");
        hipLaunchKernelGGL(kern_k4b, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b);
    }
    if (adj) {
        if (sC != st) {
            HIP_CHECK(hipEventRecord(slot.ev_rest[NORD - 1], sC));
            HIP_CHECK(hipStreamWaitEvent(st, slot.ev_rest[NORD - 1], 0));
        }
        Timer t(I, st, "k4a_chain");
        const size_t n_grad = size_t(4 * nlyr + 2) * nt;
        // (a gradient row is [element][w x nlos + los] with the launch's nt as its row length: the per-order sets are laid out alike)
        hipLaunchKernelGGL(kern_grad_orders, dim3(2048), dim3(256), 0, st, slot.bt_grad, b.grad, n_grad, bb.ord_grad, NORD);
    }
    HIP_CHECK(hipGetLastError());
    return true;
}

template <int N>
static void run_chunk(DeviceEngine::Impl& I, DeviceEngine::Impl::Slot& slot, hipStream_t st, int derivs) {
    const Geo& g = I.g;
    Bufs& b = slot.b;
    const size_t nw = b.nw, nt = nw * g.nlos;
    {
        Timer t(I, st, "init");
        hipLaunchKernelGGL(kern_init, dim3(1024), dim3(256), 0, st, b, I.d_slot_is_ground, g.nslot, size_t(4 * g.nlyr + 2 + b.ngroups * g.nlyr) * nt,
                           size_t(2 * g.ngrid + 1 + b.ngroups * g.ngrid) * nt, nt);
    }
    {
        Timer t(I, st, "k0_layer_optics");
        if (g.nstr == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k0_tile<16>), dim3(blocks(nw)), dim3(kBlock), 0, st, g, b);
        else hipLaunchKernelGGL(kern_k0, dim3(blocks(nw)), dim3(kBlock), 0, st, g, b);
        if (g.nq_total > 0) hipLaunchKernelGGL(kern_k0_shells, dim3(4096), dim3(256), 0, st, g, b);
        hipLaunchKernelGGL(kern_k0_los, dim3(unsigned((size_t(g.nlos) * nw + 255) / 256)), dim3(256), 0, st, g, b);
        if (b.ngroups > 0) hipLaunchKernelGGL(kern_k0g, dim3(blocks(nw, 256), unsigned(g.nlyr)), dim3(256), 0, st, g, b);
    }
    if (g.pp_mode) {
        // plane-parallel post-processing: every azimuth order is solved (do_source_planeparallel.cpp), K3 only fills the
        // record's E / PPM entries that K2c reads (there is no diffuse table), KPP integrates the lines of sight
        {
            Timer t(I, st, "k0_transmission");
            hipLaunchKernelGGL(kern_k0t, dim3(blocks(nw), unsigned(g.nlyr)), dim3(kBlock), 0, st, g, b, 0);
        }
        // do_source_planeparallel.cpp:55-60: num_do_forced_azimuth replaces the stream count as the number of azimuth orders
        const int num_azi = I.forced_azimuth > 0 ? std::min(I.forced_azimuth, g.nstr) : g.nstr;
        for (int m = 0; m < num_azi; ++m) {
            {
                bool pre;
                { Timer t(I, st, "kj_eigen"); pre = launch_kj<N>(st, g, b, m); }
                Timer t(I, st, "k1_layer_solution");
                launch_k1<N>(st, g, b, 0, m, pre);
            }
            {
                Timer t(I, st, "k2a_bvp");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_wave<N>), dim3(unsigned(nw)), dim3(64), 0, st, g, b, 0, m);
            }
            if (m == 0 && I.nfx > 0 && I.d_flux && I.nflux_types > 0) {
                Timer t(I, st, "k_flux");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_flux<N>), dim3(blocks(nw * size_t(I.nfx))), dim3(kBlock), 0, st, g, b, 0, I.nfx, I.d_fx_layer, I.d_fx_frac,
                                   I.nflux_types, I.d_flux_types, I.d_flux, I.thermal_w0, I.thermal_nw_total, 0, g.include_ss);
            }
            {
                Timer t(I, st, "k3_table_values");
                launch_k3<N>(st, g, b, 0, m);
            }
            if (derivs) {
                Timer t(I, st, "k2c_adjoint_ops");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_wave<N>), dim3(unsigned(nw)), dim3(64), 0, st, g, b, 0, m);
            }
            Timer t(I, st, "kpp_los");
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_kpp<N>), dim3(blocks(nt)), dim3(kBlock), 0, st, g, b, 0, m);
        }
        if (derivs) {
            Timer t(I, st, "k4a_chain");
            hipLaunchKernelGGL(kern_k4a, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b, 0, 1);
        }
        if (b.emission) {
            // thermal pass (skd::k0_thermal): the order-0 solve once more with the thermal "beam" in T / sec and the isotropic Legendre row
            // of the extra solar index; radiances accumulate in kpp_los
            Timer t(I, st, "thermal_pass");
            Bufs bt = b;
            bt.thermal_pass = 1;
            bt.surf_em = slot.thermal_surface;
            hipLaunchKernelGGL(kern_k0_thermal, dim3(blocks(nw)), dim3(kBlock), 0, st, g, bt);
            launch_k1<N>(st, g, bt, g.nsza, 0, launch_kj<N>(st, g, bt, 0));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_wave<N>), dim3(unsigned(nw)), dim3(64), 0, st, g, bt, g.nsza, 0);
            if (I.nfx > 0 && I.d_flux && I.nflux_types > 0)   // the thermal solution's share of the fluxes
                hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_flux<N>), dim3(blocks(nw * size_t(I.nfx))), dim3(kBlock), 0, st, g, bt, g.nsza, I.nfx, I.d_fx_layer, I.d_fx_frac,
                                   I.nflux_types, I.d_flux_types, I.d_flux, I.thermal_w0, I.thermal_nw_total, 1, 0);
            if (derivs) {
                // weighting functions: the adjoint operators of the thermal order-0 solve, the sweeps inside kern_kpp, and the chain from the
                // layer gradients (tau, omega, T = b0 (1 - omega) 4 pi / omega, sec = b1) to extinction, ssa, emission profile, surface emission
                launch_k3<N>(st, g, bt, g.nsza, 0);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_wave<N>), dim3(unsigned(nw)), dim3(64), 0, st, g, bt, g.nsza, 0);
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_kpp<N>), dim3(blocks(nt)), dim3(kBlock), 0, st, g, bt, g.nsza, 0);
            if (derivs)
                hipLaunchKernelGGL(kern_k4a_thermal, dim3(blocks(nt)), dim3(kBlock), 0, st, g, bt, I.thermal_w0, I.thermal_nw_total, I.thermal_wf_emission,
                                   I.thermal_wf_surface);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (N == 8 && derivs && g.n_lp > 0) {
        Timer t(I, st, "k0_entry_weights");
        hipLaunchKernelGGL(kern_cw, dim3(4096), dim3(256), 0, st, g, b);
    }
    // Order pipeline (not while profiling, which serialises everything on one stream).  The layer solutions of order m + 1 only
    // need the layer optics and the BVP of order m + 1 only K1(m + 1), so with the record / block-inverse / solution buffers
    // doubled the three stages  K1 | K2a K3 stop | K2c K2b  of consecutive orders run on three streams.  Measured: it pays when few
    // blocks are in flight (a lone block is bound by the single-wave latency of the layer recursions) and costs a little when
    // three or more blocks already overlap their kernels, so it follows the launch shape unless SASKTRAN2_AMD_LOOKAHEAD says else.
    //
    // Cross-stream accesses that carry no event (deliberately: an event after kern_stop(m - 1) would chain K1(m) behind the BVP of the
    // previous order and undo the look-ahead).  They rest on one invariant, kept by k_stop_rule and K3 (k3_table_values):
    //     `norders[w]` and `conv_order[w][slot]` only ever move to values GREATER than the order in flight that could read them.
    //  * K1(m) on the look-ahead stream tests `m >= norders[w]` while kern_stop(m - depth) .. kern_stop(m - 1) on `st` may lower norders
    //    to a value <= m.  A stale (larger) value only makes K1(m) fill record buffer m % depth for an order nobody reads: that buffer
    //    is free (ev_rest[m - depth]), and every consumer of order m (K3, stop on `st`; K2c / K2b behind ev_bvp[m]) is ordered after
    //    kern_stop(m - 1) and sees the final count.
    //  * K2c / K2b(m) on the adjoint stream test `conv_order > m` while kern_stop(m) clamps entries > m + 1 down to m + 1 and K3(m + 1)
    //    writes m + 2: the test's outcome is the same before and after either write (single-byte / 4-byte stores, no tearing).
    // tests/test_gpu_parity.py::test_c2_wavelength_blocks_in_flight_are_equivalent compares LOOKAHEAD=0 and =1 bit for bit.
    const int lookahead_env = [] { const char* e = std::getenv("SASKTRAN2_AMD_LOOKAHEAD"); return e ? atoi(e) : -1; }();
    const bool pipe = !I.profiling && slot.k1_stream != nullptr && (lookahead_env >= 0 ? lookahead_env != 0 : I.last_slots <= 2);
    static constexpr int kDepth = DeviceEngine::Impl::Slot::kDepth;
    double* rec_buf[kDepth] = {b.rec};
    double* dinv_buf[kDepth] = {b.dinv};
    double* xsol_buf[kDepth] = {b.xsol};
    uint8_t* qlmask_buf[kDepth] = {b.qlmask};
    for (int k = 1; k < kDepth; ++k) { rec_buf[k] = slot.recx[k - 1]; dinv_buf[k] = slot.dinvx[k - 1]; xsol_buf[k] = slot.xsolx[k - 1]; qlmask_buf[k] = slot.qlmaskx[k - 1]; }
    // orders in flight.  A full-size launch fills the chip with two (the kernels of one order already overlap those of the next); a small
    // one (a rank's share of a sharded run) is a chain of kernels per order that each fill a fraction of it -- K1, BVP, table values,
    // adjoint operators, sweeps -- and only as many chains run side by side as there are buffers (1 250 wavelengths: 23.3 -> 21.0 ms)
    static const int depth_env = [] { const char* e = std::getenv("SASKTRAN2_AMD_DEPTH"); return e ? std::min(kDepth, std::max(2, atoi(e))) : 0; }();
    const int depth = depth_env ? depth_env : (nw < 4000 ? kDepth : 2);
    hipStream_t sA = pipe ? slot.k1_stream : st, sC = pipe ? slot.adj_stream : st;
    for (int s = 0; s < g.nsza; ++s) {
        {
            Timer t(I, st, "k0_transmission");
            hipLaunchKernelGGL(kern_k0t, dim3(blocks(nw), unsigned(g.nlyr)), dim3(kBlock), 0, st, g, b, s);
        }
        if constexpr (N == 8) {
            // small launches: every azimuth order in one launch per kernel (run_orders_batched); SASKTRAN2_AMD_BATCH_ORDERS = largest block
            // served that way (0: never)
            static const int batch_max = [] { const char* e = std::getenv("SASKTRAN2_AMD_BATCH_ORDERS"); return e ? atoi(e) : 2600; }();
            if (b.k1_lane && b.ngroups == 0 && int(nw) <= batch_max && g.nstr == 16 && run_orders_batched(I, slot, st, derivs, s)) {
                if (derivs) {
                    Timer t(I, st, "k4a_chain");
                    if (s + 1 < g.nsza) hipLaunchKernelGGL(kern_k4a, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b, s, 1);
                    else {
                        hipLaunchKernelGGL(kern_k4a_gt, dim3(blocks(nt), unsigned((g.nlyr + 7) / 8)), dim3(kBlock), 0, st, g, b, s);
                        hipLaunchKernelGGL(kern_k4a_apply, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b);
                    }
                }
                continue;
            }
        }
        if (pipe) {
            HIP_CHECK(hipEventRecord(slot.ev_begin, st));
            HIP_CHECK(hipStreamWaitEvent(sA, slot.ev_begin, 0));
            HIP_CHECK(hipStreamWaitEvent(sC, slot.ev_begin, 0));
        }
        int last_m = -1;
        for (int m = 0; m < g.nstr; ++m) {
            if (pipe) {
                b.rec = rec_buf[m % depth];
                b.dinv = dinv_buf[m % depth];
                b.xsol = xsol_buf[m % depth];
                b.qlmask = qlmask_buf[m % depth];
                if (b.ngroups > 0) b.recg = slot.recg[m % depth];
                // buffers m % depth are free once the adjoint stage of order m - depth is done
                if (m >= depth) {
                    HIP_CHECK(hipStreamWaitEvent(sA, slot.ev_rest[m - depth], 0));
                    HIP_CHECK(hipStreamWaitEvent(st, slot.ev_rest[m - depth], 0));
                }
            }
            if (N == 8 && b.k1_lane) {
                {
                    Timer t(I, sA, "k1_layer_solution");
                    launch_k1_lane8(sA, g, b, s, m);
                }
                if (b.ngroups > 0 && derivs) {   // the layer solutions' derivatives along the scattering groups' directions
                    Timer t(I, sA, "k1g_group_derivatives");
                    constexpr int SETS = 4;   // 32 wavelengths per block behind one load of the order's tables
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k1g_cols8<8, SETS>), dim3(unsigned((nw + 8 * SETS - 1) / (8 * SETS)), unsigned(g.nlyr), unsigned(b.ngroups)), dim3(64), 0,
                                       sA, g, b, s, m, I.d_k1g_pp, I.d_k1g_q);
                }
            } else {
                bool pre;
                { Timer t(I, sA, "kj_eigen"); pre = launch_kj<N>(sA, g, b, m); }
                Timer t(I, sA, "k1_layer_solution");
                launch_k1<N>(sA, g, b, s, m, pre);
            }
            if (pipe) {
                HIP_CHECK(hipEventRecord(slot.ev_k1[m], sA));
                HIP_CHECK(hipStreamWaitEvent(st, slot.ev_k1[m], 0));
            }
            {
                Timer t(I, st, "k2a_bvp");
                if constexpr (N == 8) {
                    // column-per-lane BVP, four wavelengths per wavefront (kern_k2a_cols8); SASKTRAN2_AMD_K2A=mfma keeps the matrix-core form
                    static const bool cols = [] { const char* e = std::getenv("SASKTRAN2_AMD_K2A"); return !(e && std::string(e) == "mfma"); }();
                    static const int lone_env = [] { const char* e = std::getenv("SASKTRAN2_AMD_K2A_LONE"); return e ? atoi(e) : -1; }();
                    const bool lone = lone_env >= 0 ? lone_env != 0 : (nw + 3) / 4 <= 1024;   // at most one wavefront per SIMD: the spill-free instantiation
                    if (b.k1_lane && lone)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8_lone<skd::RecC<8>>), dim3(unsigned((nw + 3) / 4)), dim3(64), 0, st, g, b, s, m);
                    else if (b.k1_lane)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8<skd::RecC<8>>), dim3(unsigned((nw + 3) / 4)), dim3(64), 0, st, g, b, s, m);
                    else if (cols && lone)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8_lone<skd::Rec<8>>), dim3(unsigned((nw + 3) / 4)), dim3(64), 0, st, g, b, s, m);
                    else if (cols)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_cols8<skd::Rec<8>>), dim3(unsigned((nw + 3) / 4)), dim3(64), 0, st, g, b, s, m);
                    else
                        hipLaunchKernelGGL(kern_k2a_mfma8, dim3(unsigned(nw)), dim3(64), 0, st, g, b, s, m);
                } else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2a_wave<N>), dim3(unsigned(nw)), dim3(64), 0, st, g, b, s, m);
            }
            {
                Timer t(I, st, "k3_table_values");
                launch_k3<N>(st, g, b, s, m);
            }
            // live flags of the table entries at this order (they only read what the table values just wrote -- the stop rule's clamp does not
            // change the test `first skipped order > m` -- into the row of this order's buffer set).  Full-size launches: here, off the adjoint
            // stream, whose chain of operator kernel and sweeps is the longer one (113 -> 111 ms); small launches: the BVP stream is the
            // critical one (1 250 wavelengths: 19.4 against 19.8 ms), the flags stay at the head of the adjoint stage
            const bool flags_here = depth == 2;
            if (N == 8 && derivs && g.n_lp > 0 && flags_here) {
                Timer t2(I, st, "k2q_live_flags");
                hipLaunchKernelGGL(kern_qlmask, dim3(2048), dim3(256), 0, st, g, b, m);
            }
            if (pipe) {
                HIP_CHECK(hipEventRecord(slot.ev_bvp[m], st));
                HIP_CHECK(hipStreamWaitEvent(sC, slot.ev_bvp[m], 0));
            }
            if (derivs) {
                int lpw = 1;
                while (lpw < g.nlos && lpw < 64) lpw *= 2;
                if constexpr (N == 8) {
                    // 16 streams: adjoint operators and sweeps as FP64 matrix-core products (v_mfma_f64_16x16x4_f64)
                    const int ntile = (g.nlos + 15) / 16;
                    // full-size launches with at most two 16-LOS tiles: the forward sweep runs inside the operator kernel (FP, SP, HD stay in
                    // registers), kern_k2b_mfma8<.., true> keeps the backward sweep; SASKTRAN2_AMD_FUSE=0 keeps the two separate kernels
                    static const bool fuse_env = [] { const char* e = std::getenv("SASKTRAN2_AMD_FUSE"); return e ? atoi(e) != 0 : true; }();
                    // (scattering groups: their rows exist in the fused kernels only)
                    const bool fused = (fuse_env || b.ngroups > 0) && ntile <= 2 && (small_batch_split(int(nw)) == 1 || b.ngroups > 0) && g.n_lp > 0;
                    if (b.ngroups > 0 && !(fused && b.k1_lane)) throw std::runtime_error("sasktran2_amd: scattering groups need the fused adjoint kernels (at most 32 lines of sight)");
                    if (g.n_lp > 0 && !flags_here) {
                        Timer t2(I, sC, "k2q_live_flags");
                        hipLaunchKernelGGL(kern_qlmask, dim3(2048), dim3(256), 0, sC, g, b, m);
                    }
                    {
                        Timer t2(I, sC, "k2c_adjoint_ops");
                        // the layers of the adjoint operators are independent: three blocks per wavelength also at full size (30 000 wavefronts fill
                        // the 4 096 resident slots in 7.3 rounds instead of 2.4: 36 -> 32 ms per step alone, 0.4 % of the pipelined step)
                        int split = std::max(3, small_batch_split(int(nw)));
                        if (const char* e = std::getenv("SASKTRAN2_AMD_K2C_SPLIT")) split = std::max(1, atoi(e));
                        const int lpb = (g.nlyr + split - 1) / split, ny = (g.nlyr + lpb - 1) / lpb;
                        // (one instantiation per group count: the groups' derivative blocks take 3.6 KB of LDS each)
                        if (b.ngroups == 1 && ntile == 2)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 2, 1>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (b.ngroups == 1)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 1, 1>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (b.ngroups > 0 && ntile == 2)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 2, 2>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (b.ngroups > 0)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 1, 2>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (fused && ntile == 2 && b.k1_lane)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 2>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (fused && b.k1_lane)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true, 1>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (fused && ntile == 2)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<false, 2>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (fused)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<false, 1>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m, g.nlyr);
                        else if (b.k1_lane)
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<true>), dim3(unsigned(nw), unsigned(ny)), dim3(64), 0, sC, g, b, s, m, lpb);
                        else
                            hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_mfma8<false>), dim3(unsigned(nw), unsigned(ny)), dim3(64), 0, sC, g, b, s, m, lpb);
                    }
                    Timer t(I, sC, "k2b_adjoint");
                    if (b.ngroups > 0 && ntile == 2)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<2, true, true>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    else if (b.ngroups > 0)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<1, true, true>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    else if (fused && ntile == 2)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<2, true>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    else if (fused)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<1, true>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    else if (ntile > 1 && small_batch_split(int(nw)) == 1)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<2>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    else  // one 16-LOS tile per block (up to 4 blocks per wavelength)
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_mfma8<1>), dim3(unsigned(nw), unsigned(std::min(ntile, 4))), dim3(64), 0, sC, g, b, s, m);
                } else {
                    {
                        Timer t2(I, sC, "k2c_adjoint_ops");
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2c_wave<N>), dim3(unsigned(nw)), dim3(64), 0, sC, g, b, s, m);
                    }
                    const size_t per_w = skd::Ops<N>::SIZE * sizeof(double);
                    const int wpb = std::max(1, std::min<int>(64 / lpw, int(49152 / per_w)));
                    Timer t(I, sC, "k2b_adjoint");
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(kern_k2b_wave<N>), dim3(unsigned((nw + wpb - 1) / wpb)), dim3(64), wpb * per_w, sC, g, b, s, m, lpw, wpb);
                }
            }
            if (pipe) HIP_CHECK(hipEventRecord(slot.ev_rest[m], sC));
            // stop rule.  Kernels of orders beyond a wavelength's last one return at once (norders), so the host only
            // reads the count of still-active wavelengths back when profiling (keeps the per-kernel launch counts exact);
            // otherwise every order is enqueued without a host round trip.  (The adjoint stage of order m may still be running:
            // it only tests conv_order > m, which neither this kernel's clamp to m + 1 nor K3 of order m + 1 changes.)
            int active = 1;
            {
                Timer t(I, st, "k_stop_rule");
                if (I.profiling) HIP_CHECK(hipMemsetAsync(slot.d_active, 0, sizeof(int), st));
                hipLaunchKernelGGL(kern_stop, dim3(blocks(nw)), dim3(kBlock), 0, st, g, b, s, m, slot.d_active);
                if (I.profiling) {
                    HIP_CHECK(hipMemcpyAsync(&active, slot.d_active, sizeof(int), hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                }
            }
            last_m = m;
            if (active == 0) break;
        }
        b.rec = rec_buf[0];
        b.dinv = dinv_buf[0];
        b.xsol = xsol_buf[0];
        b.qlmask = qlmask_buf[0];
        if (s == g.nsza - 1) {
            // the line-of-sight integration reads the finished diffuse table (J, conv_order: final after this stop rule) and nothing of the
            // adjoint stage, whose last orders are still running on sC: it goes first, the chain from the layer gradients after it.
            // (Both add into dnat, in this order in every launch shape.)
            Timer t(I, st, "k4b_los");
            hipLaunchKernelGGL(kern_k4b, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b);
        }
        if (pipe && last_m >= 0) HIP_CHECK(hipStreamWaitEvent(st, slot.ev_rest[last_m], 0));  // the whole adjoint stage is in order on sC
        if (derivs) {
            Timer t(I, st, "k4a_chain");
            if (s + 1 < g.nsza) hipLaunchKernelGGL(kern_k4a, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b, s, 1);
            else {
                hipLaunchKernelGGL(kern_k4a_gt, dim3(blocks(nt), unsigned((g.nlyr + 7) / 8)), dim3(kBlock), 0, st, g, b, s);
                hipLaunchKernelGGL(kern_k4a_apply, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b);
            }
            if (b.ngroups > 0) hipLaunchKernelGGL(kern_k4g, dim3(blocks(nt, 256)), dim3(256), 0, st, g, b);   // (clears the groups' rows for the next sza)
        }
    }
    if (g.nsza == 0) {
        Timer t(I, st, "k4b_los");
        hipLaunchKernelGGL(kern_k4b, dim3(blocks(nt)), dim3(kBlock), 0, st, g, b);
    }
    HIP_CHECK(hipGetLastError());
}

void DeviceEngine::solve_device(int nwavel, int nleg, const double* d_ext, const double* d_ssa, const double* d_leg,
                                const double* d_albedo, const double* d_irradiance, int derivs, double* d_radiance,
                                double* d_wf_ext, double* d_wf_ssa, double* d_wf_alb, double* d_los_od, int* d_status,
                                void* stream, int max_chunk) {
    Impl& I = *impl_;
    if (vector_mode_) throw std::runtime_error("sasktran2_amd: this stream count / NSTOKES is served by solve_device_vector (radiances only)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const auto& t = I.tab;
    // Launch shape (measured on MI355X, C2, with the three-stage order pipeline of run_chunk): one block of up to 10 000
    // wavelengths per stream set, at most two blocks in flight (10 000 wavelengths: 1 x 10 000 -> 212 ms, 3 x 3334 without the
    // pipeline -> 216, 2 x 5000 -> 225; 5 000: 1 x 5000 -> 114, 2 x 2500 -> 115; 2 500: 60 vs 62; 20 000: 2 x 10 000 -> 427, 4 x 5000 -> 447).
    // Scratch is ~1.55 MB per wavelength, i.e. 15.5 GB per block of 10 000.
    const int chunk_cap = 10000;
    int nslots = std::max(1, std::min(2, (nwavel + chunk_cap - 1) / chunk_cap));
    if (const char* e = std::getenv("SASKTRAN2_AMD_STREAMS")) nslots = std::max(1, atoi(e));
    int chunk = std::min(chunk_cap, (nwavel + nslots - 1) / nslots);
    {
        // HBM budget: all blocks in flight may take half of what is free now plus what this engine already holds (as the polarised
        // driver does); a tall geometry (e.g. 100 layers x 32 streams x 100 lines of sight ~ 10 MB per wavelength) then runs in
        // smaller balanced blocks instead of failing in hipMalloc
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const size_t per_w = bytes_per_wavelength();
        size_t budget = size_t(0.5 * double(free_b + I.scratch_bytes));
        if (const char* e = std::getenv("SASKTRAN2_AMD_HBM_BUDGET_MB")) budget = size_t(std::max(1, atoi(e))) << 20;  // tests
        const size_t fit = std::max<size_t>(1, budget / (per_w * size_t(nslots)));
        if (size_t(chunk) > fit) {
            const size_t per_slot = (size_t(nwavel) + nslots - 1) / nslots, nblocks = (per_slot + fit - 1) / fit;
            chunk = int((per_slot + nblocks - 1) / nblocks);
        }
    }
    if (max_chunk > 0) chunk = std::min(chunk, max_chunk);
    if (const char* e = std::getenv("SASKTRAN2_AMD_CHUNK")) chunk = std::max(1, atoi(e));
    chunk = std::max(1, std::min(chunk, nwavel));
    nslots = std::min(nslots, (nwavel + chunk - 1) / chunk);
    if (I.profiling) nslots = 1;  // per-kernel timing serialises everything on the caller's stream
    I.last_chunk = chunk;
    I.last_slots = nslots;
    ensure_scratch(chunk, nslots);
    const int ng = derivs ? I.ngroups : 0;
    if (ng > 0) ensure_group_scratch(ng);
    const bool fan_out = !I.profiling && nslots > 1;
    if (fan_out) HIP_CHECK(hipEventRecord(I.fork, st));
    int used = 0;
    for (int w0 = 0, c = 0; w0 < nwavel; w0 += chunk, ++c) {
        const int nw = std::min(chunk, nwavel - w0);
        Impl::Slot& slot = *I.slots[c % nslots];
        hipStream_t cs = fan_out ? slot.stream : st;
        if (fan_out && c < nslots) HIP_CHECK(hipStreamWaitEvent(cs, I.fork, 0));
        used = std::max(used, c % nslots + 1);
        Bufs& b = slot.b;
        b.nw = nw;
        b.ext = d_ext + size_t(w0) * t.ngrid;
        b.ssa = d_ssa + size_t(w0) * t.ngrid;
        b.leg = d_leg + size_t(w0) * t.ngrid * nleg;
        b.albedo = d_albedo + w0;
        b.irradiance = d_irradiance ? d_irradiance + w0 : nullptr;
        b.delta_f = I.delta_f ? I.delta_f + size_t(w0) * t.ngrid : nullptr;
        b.emission = I.thermal_emission ? I.thermal_emission + size_t(w0) * t.ngrid : nullptr;
        b.surf_em = nullptr;
        b.thermal_pass = 0;
        slot.thermal_surface = I.thermal_surface ? I.thermal_surface + w0 : nullptr;
        if (b.emission && !I.g.pp_mode) throw std::runtime_error("sasktran2_amd: the discrete-ordinates thermal source needs plane-parallel post-processing");
        I.thermal_w0 = w0;
        I.thermal_nw_total = nwavel;
        b.nleg = nleg;
        b.derivs = derivs;
        b.k1_lane = (t.N == 8 && !I.g.pp_mode && k1_lane_enabled()) ? 1 : 0;
        b.ngroups = ng;
        b.grad = ng > 0 ? slot.grad_g : slot.grad0;
        b.dnat = ng > 0 ? slot.dnat_g : slot.dnat0;
        if (ng > 0) {
            b.dleg = I.g_dleg; b.dfg = I.g_dfg; b.dleg_nw = nwavel; b.dleg_w0 = w0;
            b.ga = slot.ga; b.recg = slot.recg[0]; b.opsg = slot.opsg; b.hg = slot.hg;
        }
        switch (t.N) {
            case 1: run_chunk<1>(I, slot, cs, derivs); break;
            case 2: run_chunk<2>(I, slot, cs, derivs); break;
            case 4: run_chunk<4>(I, slot, cs, derivs); break;
            case 8: run_chunk<8>(I, slot, cs, derivs); break;
            case 16: run_chunk<16>(I, slot, cs, derivs); break;
        }
        Timer tm(I, cs, "k_output");
        if (I.slab)
            hipLaunchKernelGGL(kern_output_slab, dim3(blocks(size_t(nw) * t.nlos), unsigned(b.derivs ? t.ngrid : 1)), dim3(kBlock), 0, cs, I.g, b, w0, I.slab, d_status);
        else
            hipLaunchKernelGGL(kern_output, dim3(blocks(size_t(nw) * t.nlos), unsigned(b.derivs ? t.ngrid : 1)), dim3(kBlock), 0, cs, I.g, b, w0, nwavel, d_radiance,
                               d_wf_ext, d_wf_ssa, d_wf_alb, d_los_od, d_status);
        if (ng > 0 && I.g_wf_scat)
            hipLaunchKernelGGL(kern_output_groups, dim3(blocks(size_t(nw) * t.nlos, 256), unsigned(ng * t.ngrid)), dim3(256), 0, cs, I.g, b, w0, nwavel, I.g_wf_scat);
    }
    if (fan_out)
        for (int i = 0; i < used; ++i) {
            HIP_CHECK(hipEventRecord(I.slots[i]->done, I.slots[i]->stream));
            HIP_CHECK(hipStreamWaitEvent(st, I.slots[i]->done, 0));
        }
    HIP_CHECK(hipGetLastError());
}

#include "disco_vec_engine.inc"

}  // namespace skdev
