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
#include "disco_wave_kernels.hpp"   // wave_lds_sync, SKD_QR_FENCE (scheduling fence: keeps the unrolled loops' LDS reads from being hoisted all at once)

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

// Tables of kern_k1g_cols8, once per engine: pp[m][pair (i >= r)][k] = sqrt(w_i w_r / mu_i mu_r) P_{m+k}(mu_i) P_{m+k}(mu_r) and
// q[s][m][i][k] = fm w_i P_{m+k}(mu_i) P_{m+k}(mu_0)  (what kern_k1_lane forms per block of 64 problems; here a block has 8 to 32)
__global__ void __launch_bounds__(256) kern_k1g_tables(skd::Geo g, double* pp, double* qq) {
    const int N = g.nstr / 2, NSTR = g.nstr, NPAIR = N * (N + 1) / 2;
    const int npp = NSTR * NPAIR * NSTR, nq = g.nsza * NSTR * N * NSTR;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < npp + nq; e += gridDim.x * blockDim.x) {
        if (e < npp) {
            const int m = e / (NPAIR * NSTR), pr = (e / NSTR) % NPAIR, k = e % NSTR, l = m + k;
            const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= pr) ++i;
            const int r = pr - i * (i + 1) / 2;
            pp[e] = (l < NSTR) ? sqrt(g.wt[i] / g.mu[i]) * sqrt(g.wt[r] / g.mu[r]) * lpmu[i * NSTR + l] * lpmu[r * NSTR + l] : 0.0;
        } else {
            const int f = e - npp, s = f / (NSTR * N * NSTR), m = (f / (N * NSTR)) % NSTR, i = (f / NSTR) % N, k = f % NSTR, l = m + k;
            const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
            const double* lpc = g.lpcsz + (size_t(s) * NSTR + m) * NSTR;
            const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * skd::kPi);
            qq[f] = (l < NSTR) ? fm * g.wt[i] * lpmu[i * NSTR + l] * lpc[l] : 0.0;
        }
    }
}

// Group derivative of the layer solutions: one solution COLUMN per lane, eight lanes per (layer, wavelength), eight problems per wavefront,
// SETS such sets of eight wavelengths one after the other per block; grid (wavelengths / (8 SETS), layers, groups).  The problem's Xh, Uh
// and the dB matrix of the moment live in LDS (read as broadcasts inside the eight lanes), a lane keeps its own column in registers.
// Reads kern_k1_lane's column-block record (W+-, k, A+-), writes skd::RecG.
template <int N, int SETS> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) kern_k1g_cols8(skd::Geo g, skd::Bufs b, int s, int m, const double* pp_tab, const double* q_tab) {
    static_assert(N == 8, "eight lanes per problem");
    constexpr int NSTR = 2 * N, NPAIR = N * (N + 1) / 2, NP = 64 / N;   // NP problems per wavefront
    constexpr int MS = N * N + 2;   // a problem's matrices 66 doubles apart: the eight problems' broadcast reads fall on different banks
    using R = skd::RecC<N>;
    using RG = skd::RecG<N>;
    const int lane = threadIdx.x, j = lane & (N - 1), P = lane / N, p = blockIdx.y, gi = blockIdx.z, nw = b.nw, nlyr = g.nlyr;
    constexpr int PS = NPAIR + 1;
    __shared__ double sPP[NSTR * PS];
    __shared__ double sQ[N * NSTR];
    __shared__ double sIm[N], sIt[N], sWm[N], sTi[N];
    __shared__ double sX[NP * MS], sU[NP * MS], sD[NP * MS];   // [problem][stream i][solution / stream]
    __shared__ double sAd[NP][2][NSTR + 2];                    // a_k and the direction d_k of the problem, k = l - m
    __shared__ double sDq[NP][4][N];                           // q+, q-, dq+, dq- of the problem
    __shared__ double sLam[NP][N];
    // (order-major in LDS, pairs 37 doubles apart: the eight lanes of a problem read eight different pairs of one order -- pair-major they
    // would sit 32 dwords apart, on two banks)
    for (int e = lane; e < NPAIR * NSTR; e += 64) sPP[(e % NSTR) * PS + e / NSTR] = pp_tab[size_t(m) * NPAIR * NSTR + e];
    for (int e = lane; e < N * NSTR; e += 64) sQ[e] = q_tab[(size_t(s) * NSTR + m) * N * NSTR + e];
    if (lane < N) {
        sIm[lane] = 1.0 / g.mu[lane];
        sIt[lane] = 1.0 / sqrt(g.wt[lane] * g.mu[lane]);
        sWm[lane] = g.wt[lane] * g.mu[lane];
        sTi[lane] = sqrt(g.wt[lane] * g.mu[lane]);
    }
    double* mX = sX + P * MS;
    double* mU = sU + P * MS;
    double* mD = sD + P * MS;
    for (int set = 0; set < SETS; ++set) {
        if ((blockIdx.x * SETS + set) * NP >= nw) break;   // (uniform)
        const int w = (blockIdx.x * SETS + set) * NP + P;
        const bool live = (w < nw) && (m < b.norders[min(w, nw - 1)]);
        const int wl = min(w, nw - 1);
        // this lane's column of the record
        const double* rec = skd::rec_at_r<R>(b, nlyr, p, wl);
        double wp[N], wm[N];
        static_assert(R::wp(1, 0) == R::wp(0, 0) + 1 && R::wm(0, 0) == R::wp(0, 0) + N && (R::CB % 2) == 0, "W+(:, j), W-(:, j) in one run of 2 N doubles");
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const double2 vp = *reinterpret_cast<const double2*>(rec + R::wp(i, j)), vm = *reinterpret_cast<const double2*>(rec + R::wm(i, j));
            wp[i] = vp.x; wp[i + 1] = vp.y;
            wm[i] = vm.x; wm[i + 1] = vm.y;
        }
        const double kj = rec[R::k(j)], Ap = rec[R::ap(j)], Am = rec[R::am(j)];
        const double lamj = kj * kj, ikj = 1.0 / kj;
        const double om = b.om[size_t(p) * nw + wl], th = b.tau[size_t(p) * nw + wl], T = b.T[size_t(p) * nw + wl], sc = b.sec[size_t(p) * nw + wl];
        // a_k and the direction d_k, k = l - m: lane j brings k = j and k = j + 8
        const double a0 = (m + j < NSTR) ? b.a[(size_t(p) * NSTR + m + j) * nw + wl] : 0.0;
        const double a1 = (m + j + N < NSTR) ? b.a[(size_t(p) * NSTR + m + j + N) * nw + wl] : 0.0;
        const double d0 = (m + j < NSTR) ? b.ga[((size_t(gi) * nlyr + p) * NSTR + m + j) * nw + wl] : 0.0;
        const double d1 = (m + j + N < NSTR) ? b.ga[((size_t(gi) * nlyr + p) * NSTR + m + j + N) * nw + wl] : 0.0;
        wave_lds_sync();   // (tables; the previous set's reads.  A one-wavefront block: waiting for its LDS requests is the barrier, and the
                           // stores of the previous set stay in flight)
        sAd[P][0][j] = a0; sAd[P][0][j + N] = a1; sAd[P][1][j] = d0; sAd[P][1][j + N] = d1;
        // W+- = T^-1 (xh +- uh / k) / 2 with T^-1 = 1 / sqrt(w mu)  ->  xh, uh
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double t = sTi[i];
            mX[i * N + j] = (wp[i] + wm[i]) * t;
            mU[i * N + j] = (wp[i] - wm[i]) * t * kj;
        }
        sLam[P][j] = lamj;
        wave_lds_sync();
        // q+-, dq+- of stream j (every lane of the problem needs all of them)
        {
            double sp = 0, sn = 0, dp = 0, dn = 0;
#pragma unroll
            for (int k = 0; k < NSTR; k += 2) {
                sp = fma(sAd[P][0][k], sQ[j * NSTR + k], sp);
                sn = fma(sAd[P][0][k + 1], sQ[j * NSTR + k + 1], sn);
                dp = fma(sAd[P][1][k], sQ[j * NSTR + k], dp);
                dn = fma(sAd[P][1][k + 1], sQ[j * NSTR + k + 1], dn);
            }
            sDq[P][0][j] = (sp + sn) * om;
            sDq[P][1][j] = (sp - sn) * om;
            sDq[P][2][j] = (dp + dn) * om;
            sDq[P][3][j] = (dp - dn) * om;
        }
        // row j of dB+ (parity 0: l - m even) or dB- (parity 1)
        auto build = [&](int parity) {
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const int hi = (j > r) ? j : r, lo = (j > r) ? r : j;
                const double* pp = &sPP[hi * (hi + 1) / 2 + lo];
                double e = 0;
#pragma unroll
                for (int k = 0; k < NSTR; k += 2) e = fma(sAd[P][1][k + parity], pp[(k + parity) * PS], e);
                mD[j * N + r] = -om * e;
                SKD_QR_FENCE
            }
        };
        build(0);
        wave_lds_sync();
        double tp[N], mu_[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0;
#pragma unroll
            for (int r = 0; r < N; ++r) t = fma(mD[i * N + r], mX[r * N + j], t);
            tp[i] = t;
            SKD_QR_FENCE
        }
        wave_lds_sync();
        build(1);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0;
#pragma unroll
            for (int r = 0; r < N; ++r) t = fma(mD[i * N + r], mU[r * N + j], t);
            mu_[i] = t;
            SKD_QR_FENCE
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
                z1 = fma(mU[i * N + aa], mu_[i], z1);
                z2 = fma(mX[i * N + aa], tp[i], z2);
            }
            const double z = fma(sLam[P][aa], z2, z1);
            if (aa == j) { dk = 0.5 * z * ikj; cc[aa] = 0; }
            else {
                const double den = lamj - sLam[P][aa];
                cc[aa] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
            }
            SKD_QR_FENCE
        }
        // dW+- (stored as they are formed), the norm's derivative, dA+-
        double* out = skd::recg_at<N>(b, nlyr, p, wl, gi);
        double nrm = 0, dnrm = 0, dAp = 0, dAm = 0;
        bool allzero = true;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double dx = 0, du = tp[i];
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                dx = fma(mX[i * N + aa], cc[aa], dx);   // (cc[j] = 0)
                du = fma(mU[i * N + aa], cc[aa], du);
            }
            const double duk = du * ikj - mU[i * N + j] * dk * ikj * ikj, hs = 0.5 * sIt[i];
            const double dwp = hs * (dx + duk), dwm = hs * (dx - duk);
            if (live) {
                out[RG::dwp(i, j)] = dwp;
                out[RG::dwm(i, j)] = dwm;
            }
            const double wmu = sWm[i];
            nrm = fma(wmu, wp[i] * wp[i] - wm[i] * wm[i], nrm);
            dnrm = fma(2.0 * wmu, wp[i] * dwp - wm[i] * dwm, dnrm);
            const double qp = sDq[P][0][i], qm = sDq[P][1][i], dqp = sDq[P][2][i], dqm = sDq[P][3][i];
            allzero = allzero && (qp == 0.0) && (qm == 0.0);
            dAp += qp * dwp + qm * dwm + dqp * wp[i] + dqm * wm[i];
            dAm += qm * dwp + qp * dwm + dqm * wp[i] + dqp * wm[i];
            SKD_QR_FENCE
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
        if (live) {
            out[RG::at(RG::DK, j)] = dk;
            out[RG::at(RG::DAP, j)] = dAp;
            out[RG::at(RG::DAM, j)] = dAm;
            out[RG::at(RG::CGM, j)] = cgm;
            out[RG::at(RG::CGP, j)] = cgp;
            // the direction itself, by absolute Legendre order (what kern_k2c_mfma8 multiplies its Legendre sums with)
            out[RG::DA + j] = b.ga[((size_t(gi) * nlyr + p) * NSTR + j) * nw + w];
            out[RG::DA + N + j] = b.ga[((size_t(gi) * nlyr + p) * NSTR + N + j) * nw + w];
        }
    }
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
