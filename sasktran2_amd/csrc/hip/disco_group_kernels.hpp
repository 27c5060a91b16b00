// Scattering-group weighting functions on the 16-stream matrix-core path: the kernels that exist only for them.
//
// A scattering group (a derivative mapping with d_legendre / scat_factor arrays, i.e. the number density of a scattering constituent;
// reference: sktran_do_layerarray.cpp:496-513,759-860, outputc.cpp:76-90) moves the Legendre coefficients a_l of a layer along a direction
// da_l at fixed optical depth and single-scatter albedo (those parts go through the extinction / ssa rows).  What the adjoint machinery
// needs of it per layer and azimuth order is the derivative of the layer's solution along that direction -- dW+-, dk, dA+- and the
// derivative of the particular solution's coefficients -- exactly the quantities kern_k1_lane produces for the omega direction
// (da = a / omega).  They are formed here AFTER kern_k1_lane from its record (the tuned kernel stays as it is):
//
//   B+- = 1/mu - omega sum_{l - m even / odd} a_l PP_l          (symmetrised: PP_l[i][r] = sqrt(w_i w_r / mu_i mu_r) P_l(mu_i) P_l(mu_r))
//   dB+- = -omega sum_{l - m even / odd} da_l PP_l
//   Gamma = B- B+,  Gamma xh_j = k_j^2 xh_j,  uh_j = B+ xh_j  (left eigenvectors: uh_a, uh_a . xh_a = 1)
//   z_aj = uh_a . (dB- uh_j + B- dB+ xh_j) = uh_a . dB- uh_j + k_a^2 xh_a . dB+ xh_j          (B- uh_a = k_a^2 xh_a: B- itself is not needed)
//   dk_j = z_jj / (2 k_j),  C_aj = z_aj / (k_j^2 - k_a^2),  dxh_j = sum_a xh_a C_aj,  duh_j = dB+ xh_j + sum_a uh_a C_aj
//
// and from there on as kern_k1_lane's omega derivative (disco_lane_kernels.hpp; reference sktran_do_rte.cpp:382-553,1039-1237).  The same
// mathematics as skv::kv1's group stage (disco_vec_core.hpp), which is the implementation the tests compare against.
#pragma once

#include "disco_core.hpp"

namespace skdev {

// direction of a layer's Legendre coefficients per group:  d_legendre of the mapping at the layer's last contributing grid point plus the
// distance of that point's coefficients from the layer mean (the reference assigns the direction once per layer, from the last entry of its
// interpolation stencil), with the delta-M terms of a scaled storage (sktran_do_layerarray.cpp:777-790,826-843); as skv::kv0 for NSTOKES = 1
__global__ void __launch_bounds__(256) kern_k0g(skd::Geo g, skd::Bufs b) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y, nw = b.nw, nstr = g.nstr, ngrid = g.ngrid, nlyr = g.nlyr;
    if (w >= nw) return;
    const int qs = (g.wq_idx[2 * p + 1] >= 0) ? g.wq_idx[2 * p + 1] : g.wq_idx[2 * p];
    const double* leg = b.leg + size_t(w) * ngrid * b.nleg;
    const double fq = b.delta_f ? b.delta_f[size_t(w) * ngrid + qs] : 0.0, fsub = fq / (1 - fq);
    for (int gi = 0; gi < b.ngroups; ++gi) {
        const double* dl = b.dleg + ((size_t(gi) * b.dleg_nw + b.dleg_w0 + w) * ngrid + qs) * b.nleg;
        const double dfq = b.dfg ? b.dfg[(size_t(gi) * b.dleg_nw + b.dleg_w0 + w) * ngrid + qs] : 0.0;
        for (int l = 0; l < nstr; ++l) {
            const bool have = l < b.nleg;
            const double dft = b.dfg ? (2 * l + 1) * dfq / ((1 - fq) * (1 - fq)) : 0.0;
            b.ga[((size_t(gi) * nlyr + p) * nstr + l) * nw + w] =
                (have ? dl[l] + leg[size_t(qs) * b.nleg + l] : 0.0) - (2 * l + 1) * fsub - b.a[(size_t(p) * nstr + l) * nw + w] - dft;
        }
    }
}

// Group derivative of the layer solutions: one solution COLUMN per lane, eight lanes per (layer, wavelength), eight problems per wavefront;
// grid (wavelengths / 8, layers, groups).  The problem's Xh, Uh and the dB matrix of the moment live in LDS (read as broadcasts inside the
// eight lanes), a lane keeps its own column in registers.  Reads kern_k1_lane's column-block record (W+-, k, A+-), writes skd::RecG.
template <int N> __global__ void __launch_bounds__(64) kern_k1g_cols8(skd::Geo g, skd::Bufs b, int s, int m) {
    static_assert(N == 8, "eight lanes per problem");
    constexpr int NSTR = 2 * N, NPAIR = N * (N + 1) / 2, NP = 64 / N;   // NP problems per wavefront
    using R = skd::RecC<N>;
    using RG = skd::RecG<N>;
    const int lane = threadIdx.x, j = lane & (N - 1), P = lane / N, p = blockIdx.y, gi = blockIdx.z, nw = b.nw, nlyr = g.nlyr;
    const int w = blockIdx.x * NP + P;
    __shared__ double sPP[NPAIR * NSTR];   // as kern_k1_lane
    __shared__ double sQ[N * NSTR];
    __shared__ double sIm[N], sIt[N], sWm[N];
    __shared__ double sX[NP][N][N + 1], sU[NP][N][N + 1], sD[NP][N][N + 1];   // [problem][stream i][solution / stream]
    __shared__ double sDq[NP][4][N];                                        // q+, q-, dq+, dq- of the problem
    __shared__ double sLam[NP][N];
    {
        const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
        const double* lpc = g.lpcsz + (size_t(s) * NSTR + m) * NSTR;
        const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * skd::kPi);
        for (int e = lane; e < NPAIR * NSTR; e += 64) {
            const int pr = e / NSTR, k = e % NSTR, l = m + k;
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= pr) ++i;
            const int r = pr - i * (i + 1) / 2;
            sPP[e] = (l < NSTR) ? sqrt(g.wt[i] / g.mu[i]) * sqrt(g.wt[r] / g.mu[r]) * lpmu[i * NSTR + l] * lpmu[r * NSTR + l] : 0.0;
        }
        for (int e = lane; e < N * NSTR; e += 64) {
            const int i = e / NSTR, k = e % NSTR, l = m + k;
            sQ[e] = (l < NSTR) ? fm * g.wt[i] * lpmu[i * NSTR + l] * lpc[l] : 0.0;
        }
        if (lane < N) {
            sIm[lane] = 1.0 / g.mu[lane];
            sIt[lane] = 1.0 / sqrt(g.wt[lane] * g.mu[lane]);
            sWm[lane] = g.wt[lane] * g.mu[lane];
        }
    }
    const bool live = (w < nw) && (m < b.norders[min(w, nw - 1)]);
    const int wl = min(w, nw - 1);
    // this lane's column of the record
    const double* rec = skd::rec_at_r<R>(b, nlyr, p, wl);
    double wp[N], wm[N], xh[N], uh[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        wp[i] = rec[R::wp(i, j)];
        wm[i] = rec[R::wm(i, j)];
    }
    const double kj = rec[R::k(j)], Ap = rec[R::ap(j)], Am = rec[R::am(j)];
    const double lamj = kj * kj, ikj = 1.0 / kj;
    const double om = b.om[size_t(p) * nw + wl], th = b.tau[size_t(p) * nw + wl], T = b.T[size_t(p) * nw + wl], sc = b.sec[size_t(p) * nw + wl];
    // a_l and the direction, k = l - m  (the eight lanes of a problem read the same values)
    double a[NSTR], d[NSTR];
#pragma unroll
    for (int k = 0; k < NSTR; ++k) {
        a[k] = (m + k < NSTR) ? b.a[(size_t(p) * NSTR + m + k) * nw + wl] : 0.0;
        d[k] = (m + k < NSTR) ? b.ga[((size_t(gi) * nlyr + p) * NSTR + m + k) * nw + wl] : 0.0;
    }
    __syncthreads();
    // W+- = T^-1 (xh +- uh / k) / 2 with T^-1 = 1 / sqrt(w mu)  ->  xh, uh
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double t = 1.0 / sIt[i];
        xh[i] = (wp[i] + wm[i]) * t;
        uh[i] = (wp[i] - wm[i]) * t * kj;
        sX[P][i][j] = xh[i];
        sU[P][i][j] = uh[i];
    }
    sLam[P][j] = lamj;
    // q+-, dq+- of stream j (every lane of the problem needs all of them)
    {
        double sp = 0, sn = 0, dp = 0, dn = 0;
#pragma unroll
        for (int k = 0; k < NSTR; k += 2) {
            sp = fma(a[k], sQ[j * NSTR + k], sp);
            sn = fma(a[k + 1], sQ[j * NSTR + k + 1], sn);
            dp = fma(d[k], sQ[j * NSTR + k], dp);
            dn = fma(d[k + 1], sQ[j * NSTR + k + 1], dn);
        }
        sDq[P][0][j] = (sp + sn) * om;
        sDq[P][1][j] = (sp - sn) * om;
        sDq[P][2][j] = (dp + dn) * om;
        sDq[P][3][j] = (dp - dn) * om;
    }
    // row j of dB+ (even k)
    auto build = [&](int parity) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const int hi = (j > r) ? j : r, lo = (j > r) ? r : j;
            const double* pp = &sPP[(hi * (hi + 1) / 2 + lo) * NSTR];
            double e = 0;
#pragma unroll
            for (int k = 0; k < NSTR; k += 2) e = fma(d[k + parity], pp[k + parity], e);
            sD[P][j][r] = -om * e;
        }
    };
    build(0);
    __syncthreads();
    double tp[N], mu_[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double t = 0;
#pragma unroll
        for (int r = 0; r < N; ++r) t = fma(sD[P][i][r], xh[r], t);
        tp[i] = t;
    }
    __syncthreads();
    build(1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double t = 0;
#pragma unroll
        for (int r = 0; r < N; ++r) t = fma(sD[P][i][r], uh[r], t);
        mu_[i] = t;
    }
    // z_a, dk, C
    double cc[N], dk = 0, lmax = 0;
#pragma unroll
    for (int aa = 0; aa < N; ++aa) lmax = fmax(lmax, sLam[P][aa]);
#pragma unroll
    for (int aa = 0; aa < N; ++aa) {
        double z1 = 0, z2 = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            z1 = fma(sU[P][i][aa], mu_[i], z1);
            z2 = fma(sX[P][i][aa], tp[i], z2);
        }
        const double z = fma(sLam[P][aa], z2, z1);
        if (aa == j) { dk = 0.5 * z * ikj; cc[aa] = 0; }
        else {
            const double den = lamj - sLam[P][aa];
            cc[aa] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
        }
    }
    // dW+-, the norm's derivative, dA+-
    double dwp[N], dwm[N], nrm = 0, dnrm = 0, dAp = 0, dAm = 0;
    bool allzero = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double dx = 0, du = tp[i];
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
            dx = fma(sX[P][i][aa], cc[aa], dx);   // (cc[j] = 0)
            du = fma(sU[P][i][aa], cc[aa], du);
        }
        const double duk = du * ikj - uh[i] * dk * ikj * ikj, hs = 0.5 * sIt[i];
        dwp[i] = hs * (dx + duk);
        dwm[i] = hs * (dx - duk);
        const double wmu = sWm[i];
        nrm = fma(wmu, wp[i] * wp[i] - wm[i] * wm[i], nrm);
        dnrm = fma(2.0 * wmu, wp[i] * dwp[i] - wm[i] * dwm[i], dnrm);
        const double qp = sDq[P][0][i], qm = sDq[P][1][i], dqp = sDq[P][2][i], dqm = sDq[P][3][i];
        allzero = allzero && (qp == 0.0) && (qm == 0.0);
        dAp += qp * dwp[i] + qm * dwm[i] + dqp * wp[i] + dqm * wm[i];
        dAm += qm * dwp[i] + qp * dwm[i] + dqm * wp[i] + dqp * wm[i];
    }
    double cgm = 0, cgp = 0;
    if (allzero) {
        dAp = dAm = 0;
    } else {
        const double inv = 1.0 / nrm;
        dAp = (dAp - dnrm * Ap) * inv;
        dAm = (dAm - dnrm * Am) * inv;
        skd::Mult Cp, Cm;
        skd::green_multipliers_e(T, th, sc, kj, exp(-th * kj), exp(-th * sc), Cp, Cm);
        cgm = Am * Cm.d_k * dk + dAm * Cm.v;
        cgp = Ap * Cp.d_k * dk + dAp * Cp.v;
    }
    if (!live) return;
    double* out = skd::recg_at<N>(b, nlyr, p, w, gi);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[RG::dwp(i, j)] = dwp[i];
        out[RG::dwm(i, j)] = dwm[i];
    }
    out[RG::at(RG::DK, j)] = dk;
    out[RG::at(RG::DAP, j)] = dAp;
    out[RG::at(RG::DAM, j)] = dAm;
    out[RG::at(RG::CGM, j)] = cgm;
    out[RG::at(RG::CGP, j)] = cgp;
    // the direction itself, by absolute Legendre order (what kern_k2c_mfma8 multiplies its Legendre sums with): lanes j < 2 N / N ...
    out[RG::DA + j] = b.ga[((size_t(gi) * nlyr + p) * NSTR + j) * nw + w];
    out[RG::DA + N + j] = b.ga[((size_t(gi) * nlyr + p) * NSTR + N + j) * nw + w];
}

// reduced gradient of the group rows -> grid (as skv::kv4g_chain): a layer's row spreads over its stencil points with weight
// w_pq omega_q k_q / (layer scattering extinction); dnat rows 2 ngrid + 1 + group ngrid + q
__global__ void __launch_bounds__(256) kern_k4g(skd::Geo g, skd::Bufs b) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    for (int gi = 0; gi < b.ngroups; ++gi)
        for (int p = 0; p < nlyr; ++p) {
            double& row = b.grad[size_t(4 * nlyr + 2 + gi * nlyr + p) * nt + tid];
            const double gr = row;
            row = 0;   // (the next solar zenith angle's adjoint stage adds into it again)
            const double scat = b.om[size_t(p) * nw + w] * b.kbar[size_t(p) * nw + w];
            for (int t = 0; t < 2; ++t) {
                const int q = g.wq_idx[2 * p + t];
                if (q < 0) continue;
                b.dnat[size_t(2 * ngrid + 1 + gi * ngrid + q) * nt + tid] += gr * g.wq_w[2 * p + t] * ssa[q] * ext[q] / scat;
            }
        }
}
// native group derivatives of a block -> [group][q][wavelength of the spectrum][los]
__global__ void __launch_bounds__(256) kern_output_groups(skd::Geo g, skd::Bufs b, int w0, int nw_total, double* wf_scat) {
    const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nt = size_t(b.nw) * g.nlos;
    if (tid >= nt) return;
    const int w = int(tid / g.nlos), l = int(tid % g.nlos), gq = blockIdx.y;   // gq = group ngrid + q
    wf_scat[(size_t(gq) * nw_total + w0 + w) * g.nlos + l] = b.dnat[size_t(2 * g.ngrid + 1 + gq) * nt + tid];
}

}  // namespace skdev
