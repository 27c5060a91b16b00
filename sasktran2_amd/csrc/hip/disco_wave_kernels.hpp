// Workgroup-per-wavelength HIP kernels (gfx950): one 64-lane wavefront owns one wavelength (K2a, K3) or a small
// group of wavelengths with one lane per line of sight (K2b).  The streams x streams layer matrices and the 2N x 2N
// boundary-value blocks are staged in LDS, records are streamed from HBM with unit stride, and all cross-lane steps
// (pivot search, broadcasts) are wavefront shuffles -- a single-wave workgroup makes every barrier a cheap s_barrier.
//
// HIP only (included by disco_engine.hip).  The arithmetic is the same as the task functions of disco_core.hpp, which
// stay as the CPU-emulated specification; parity of these kernels is asserted on the GPU against the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "disco_core.hpp"

namespace skdev {

using skd::Bufs;
using skd::Geo;

// -------------------------------------------------------------------------------------------------------------------
// K2a: block-tridiagonal BVP, factor + solve.  grid = wavelengths, block = 64.
// LDS: current/previous layer W+-, k, G (ping-pong), augmented [D | I] for the Gauss-Jordan inverse, previous Dinv.
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64) kern_k2a_wave(Geo g, Bufs b, int s, int m) {
    constexpr int M2 = 2 * N, LDA = 2 * M2 + 1, LDI = M2 + 1, NN = N * N;
    using R = skd::Rec<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;  // uniform for the workgroup
    const int lane = threadIdx.x, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sK[2][N], sW[2][2 * NN], sG[2][4 * N], sE[2][N];
    __shared__ double sA[M2 * LDA];
    __shared__ double sDp[M2 * LDI];
    __shared__ double sT1[N * LDI];
    __shared__ double sR[M2], sRp[M2], sF[M2], sS[2][N], sX[M2];
    __shared__ int sSing;
    const double rho = (m == 0) ? b.albedo[w] : 0.0;
    if (lane == 0) sSing = 0;

    for (int p = 0; p < nlyr; ++p) {
        const int cur = p & 1, prv = cur ^ 1;
        const bool last = (p == nlyr - 1);
        const double* rec = skd::rec_at<N>(b, nlyr, p, w);
        for (int i = lane; i < N; i += 64) sK[cur][i] = rec[R::K + i];
        for (int i = lane; i < 2 * NN; i += 64) sW[cur][i] = rec[R::WP + i];  // W+ then W- (contiguous in the record)
        for (int i = lane; i < 4 * N; i += 64) sG[cur][i] = rec[R::G + i];
        const double th = b.tau[size_t(p) * nw + w];
        __syncthreads();
        if (lane < N) sE[cur][lane] = exp(-fabs(sK[cur][lane]) * th);
        if (last && m == 0 && lane < N) {
            double sp = 0, sm = 0;
            for (int t = 0; t < N; ++t) {
                sp += g.wt[t] * g.mu[t] * sW[cur][t + N * lane];
                sm += g.wt[t] * g.mu[t] * sW[cur][NN + t + N * lane];
            }
            sS[0][lane] = sp;
            sS[1][lane] = sm;
        }
        __syncthreads();
        // [D | I]
        for (int idx = lane; idx < M2 * M2; idx += 64) {
            const int i = idx / M2, j = idx % M2;
            double v;
            if (i < N) {
                v = (j < N) ? -sW[cur][i + N * j] : -sW[cur][NN + i + N * (j - N)] * sE[cur][j - N];
            } else {
                const int ii = i - N;
                v = (j < N) ? sW[cur][NN + ii + N * j] * sE[cur][j] : sW[cur][ii + N * (j - N)];
                if (last && m == 0) v -= 2.0 * rho * ((j < N) ? sS[0][j] * sE[cur][j] : sS[1][j - N]);
            }
            sA[i * LDA + j] = v;
            sA[i * LDA + M2 + j] = (i == j) ? 1.0 : 0.0;
        }
        if (lane < N) {
            const int i = lane;
            double rA = sG[cur][i], rB = -sG[cur][3 * N + i];
            if (last) {
                if (m == 0) {
                    double sg = 0;
                    for (int t = 0; t < N; ++t) sg += g.wt[t] * g.mu[t] * sG[cur][2 * N + t];
                    rB += g.csz[s] * rho / skd::kPi * b.T[size_t(nlyr) * nw + w] + 2.0 * rho * sg;
                }
            } else {
                rB += skd::rec_at<N>(b, nlyr, p + 1, w)[R::G + N + i];
            }
            sR[i] = rA;
            sR[N + i] = rB;
        }
        __syncthreads();
        if (p > 0) {
            // T1 = Sub_p Dinv_{p-1},  Sub_p rows A = [W+_{p-1} E_{p-1} | W-_{p-1}]
            for (int idx = lane; idx < N * M2; idx += 64) {
                const int i = idx / M2, j = idx % M2;
                double t = 0;
                for (int q = 0; q < N; ++q)
                    t += sW[prv][i + N * q] * sE[prv][q] * sDp[q * LDI + j] + sW[prv][NN + i + N * q] * sDp[(N + q) * LDI + j];
                sT1[i * LDI + j] = t;
            }
            __syncthreads();
            for (int idx = lane; idx < N * M2; idx += 64) {
                const int i = idx / M2, j = idx % M2;
                double t = 0;
                if (j < N) {
                    for (int q = 0; q < N; ++q) t += sT1[i * LDI + N + q] * sW[cur][NN + q + N * j];
                    sA[i * LDA + j] += t;
                } else {
                    const int jj = j - N;
                    for (int q = 0; q < N; ++q) t += sT1[i * LDI + N + q] * sW[cur][q + N * jj];
                    sA[i * LDA + j] += t * sE[cur][jj];
                }
            }
            if (lane < N) {
                double t = 0;
                for (int q = 0; q < M2; ++q) t += sT1[lane * LDI + q] * sRp[q];
                sR[lane] -= sG[prv][2 * N + lane] + t;
            }
            __syncthreads();
        }
        // Gauss-Jordan inverse with partial pivoting on [D | I]
        for (int c = 0; c < M2; ++c) {
            double v = -1.0;
            int idx = lane;
            if (lane < M2 && lane >= c) v = fabs(sA[lane * LDA + c]);
            for (int off = 32; off >= 1; off >>= 1) {
                double ov = __shfl_xor(v, off);
                int oi = __shfl_xor(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            const int piv = idx;
            if (!(v > 0) && lane == 0) sSing = 1;
            if (piv != c && lane < 2 * M2) {
                double t = sA[c * LDA + lane];
                sA[c * LDA + lane] = sA[piv * LDA + lane];
                sA[piv * LDA + lane] = t;
            }
            __syncthreads();
            const double pv = sA[c * LDA + c];
            __syncthreads();
            if (lane < 2 * M2) sA[c * LDA + lane] /= pv;
            if (lane < M2 && lane != c) sF[lane] = sA[lane * LDA + c];
            __syncthreads();
            for (int e = lane; e < M2 * 2 * M2; e += 64) {
                const int r = e / (2 * M2), j = e % (2 * M2);
                if (r != c) sA[r * LDA + j] -= sF[r] * sA[c * LDA + j];
            }
            __syncthreads();
        }
        double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        for (int idx = lane; idx < M2 * M2; idx += 64) {
            const int i = idx / M2, j = idx % M2;
            const double v = sA[i * LDA + M2 + j];
            dg[idx] = v;
            sDp[i * LDI + j] = v;
        }
        if (lane < M2) {
            skd::xsol_at<N>(b, nlyr, p, w)[lane] = sR[lane];
            sRp[lane] = sR[lane];
        }
        __syncthreads();
    }
    // back substitution  x_p = Dinv_p (r'_p - Sup_p x_{p+1})
    for (int p = nlyr - 1; p >= 0; --p) {
        const double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        for (int idx = lane; idx < M2 * M2; idx += 64) sA[(idx / M2) * LDA + (idx % M2)] = dg[idx];
        if (lane < M2) sR[lane] = xs[lane];
        double thn = 0;
        if (p < nlyr - 1) {
            const double* recn = skd::rec_at<N>(b, nlyr, p + 1, w);
            for (int i = lane; i < N; i += 64) sK[0][i] = recn[R::K + i];
            for (int i = lane; i < 2 * NN; i += 64) sW[0][i] = recn[R::WP + i];
            thn = b.tau[size_t(p + 1) * nw + w];
        }
        __syncthreads();
        if (p < nlyr - 1 && lane < N) {
            double t = 0;
            for (int j = 0; j < N; ++j)
                t += sW[0][NN + lane + N * j] * sX[j] + sW[0][lane + N * j] * exp(-fabs(sK[0][j]) * thn) * sX[N + j];
            sR[N + lane] += t;
        }
        __syncthreads();
        double x = 0;
        if (lane < M2)
            for (int j = 0; j < M2; ++j) x += sA[lane * LDA + j] * sR[j];
        __syncthreads();
        if (lane < M2) {
            sX[lane] = x;
            xs[lane] = x;
        }
        __syncthreads();
    }
    if (lane == 0 && sSing) atomicOr(&b.status[w], skd::ERR_SINGULAR_BVP);
}

// -------------------------------------------------------------------------------------------------------------------
// K3: diffuse-source table values + convergence flags.  grid = wavelengths, block = 64 (lanes = table angles).
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64) kern_k3_wave(Geo g, Bufs b, int s, int m) {
    constexpr int NSTR = 2 * N, NN = N * N;
    using R = skd::Rec<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sK[N], sW[2 * NN], sAP[N], sAM[N], sXs[2 * N], sa[NSTR], sCL[N], sCM[N], sP[N], sQ[N], sCleg[NSTR], sRed[N];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    int unconverged = 0;
    for (int p = 0; p < nlyr; ++p) {
        const double* rec = skd::rec_at<N>(b, nlyr, p, w);
        const double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        for (int i = lane; i < N; i += 64) { sK[i] = rec[R::K + i]; sAP[i] = rec[R::AP + i]; sAM[i] = rec[R::AM + i]; }
        for (int i = lane; i < 2 * NN; i += 64) sW[i] = rec[R::WP + i];
        for (int i = lane; i < 2 * N; i += 64) sXs[i] = xs[i];
        for (int l = lane; l < NSTR; l += 64) sa[l] = b.a[(size_t(p) * NSTR + l) * nw + w];
        const double th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
        __syncthreads();
        if (lane < N) {
            skd::PPMult pm;
            skd::pp_multipliers(T, th, sc, sK[lane], g.lfrac[p], pm);
            double* recw = skd::rec_at<N>(b, nlyr, p, w);
            skd::ppm_store(recw + R::PPM + 14 * lane, pm);
            recw[R::E + lane] = exp(-fabs(sK[lane]) * th);
            sCL[lane] = pm.hp * sXs[lane] + sAP[lane] * pm.Dm;
            sCM[lane] = pm.hm * sXs[N + lane] + sAM[lane] * pm.Dp;
        }
        __syncthreads();
        if (lane < N) {
            double pr = 0, qr = 0;
            for (int j = 0; j < N; ++j) {
                const double wp = sW[lane + N * j], wm = sW[NN + lane + N * j];
                pr += wp * sCL[j] + wm * sCM[j];
                qr += wm * sCL[j] + wp * sCM[j];
            }
            sP[lane] = pr * 0.5 * g.wt[lane];
            sQ[lane] = qr * 0.5 * g.wt[lane];
        }
        __syncthreads();
        if (lane < NSTR) {
            const int l = lane;
            double sp = 0, sq = 0;
            if (l >= m)
                for (int r = 0; r < N; ++r) {
                    sp += lpmu[r * NSTR + l] * sP[r];
                    sq += lpmu[r * NSTR + l] * sQ[r];
                }
            sCleg[l] = sa[l] * ((((l - m) & 1) ? -sp : sp) + sq);
        }
        __syncthreads();
        const int t0 = g.pt_start[s * nlyr + p], t1 = g.pt_start[s * nlyr + p + 1];
        for (int t = t0 + lane; t < t1; t += 64) {
            const int slot = g.pt_slot[t];
            if (skd::conv_at(b, g, slot, w) <= m) continue;
            const double* lpa = g.lpang + (size_t(m) * g.nangle + g.pt_angle[t]) * NSTR;
            double val = 0;
            for (int l = m; l < NSTR; ++l) val += sCleg[l] * lpa[l];
            skd::J_at(b, g, m, slot, w) = val;
            if (m >= 2) {
                const double prev = skd::J_at(b, g, m - 1, slot, w), prev2 = skd::J_at(b, g, m - 2, slot, w);
                if ((fabs(val / prev) < 1e-4 || prev < 1e-10) && (fabs(val / prev2) < 1e-4 || prev2 < 1e-10))
                    skd::conv_at(b, g, slot, w) = (uint8_t)(m + 1);
                else
                    unconverged = 1;
            }
        }
        if (p == nlyr - 1 && m == 0) {
            if (lane < N) {
                double sv = rec[R::G + 2 * N + lane];
                for (int j = 0; j < N; ++j)
                    sv += sXs[j] * sW[lane + N * j] * exp(-fabs(sK[j]) * th) + sXs[N + j] * sW[NN + lane + N * j];
                sRed[lane] = 2.0 * g.mu[lane] * g.wt[lane] * sv * b.albedo[w];
            }
            __syncthreads();
            double diffuse = 0;
            for (int i = 0; i < N; ++i) diffuse += sRed[i];
            for (int t = g.gpt_start[s] + lane; t < g.gpt_start[s + 1]; t += 64) skd::J_at(b, g, m, g.gpt_slot[t], w) = diffuse;
        }
        __syncthreads();
    }
    if (__any(unconverged) && lane == 0) b.notconv[w] = 1;
}

// -------------------------------------------------------------------------------------------------------------------
// K2c: LOS-independent adjoint operators of every layer (see skd::k2c_layer_ops for the definitions).
// grid = wavelengths, block = 64; the layer record, its block inverse and the work matrices live in LDS.
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64) kern_k2c_wave(Geo g, Bufs b, int s, int m) {
    constexpr int NSTR = 2 * N, M2 = 2 * N, NN = N * N;
    using R = skd::Rec<N>;
    using O = skd::Ops<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sRec[R::SIZE];
    __shared__ double sDinv[M2 * M2], sXs[M2], sa[NSTR];
    __shared__ double sF[M2 * NSTR], sH[5 * N * NSTR], sTmp[M2], sIpb[N], sDIpb[4 * N];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    const double rho = (m == 0) ? b.albedo[w] : 0.0;
    for (int p = 0; p < nlyr; ++p) {
        const bool last = (p == nlyr - 1);
        const double* rec = skd::rec_at<N>(b, nlyr, p, w);
        const double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        const double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        double* o = skd::ops_at<N>(b, nlyr, p, w);
        for (int i = lane; i < R::SIZE; i += 64) sRec[i] = rec[i];
        for (int i = lane; i < M2 * M2; i += 64) sDinv[i] = dg[i];
        if (lane < M2) sXs[lane] = xs[lane];
        if (lane < NSTR) sa[lane] = (lane >= m) ? b.a[(size_t(p) * NSTR + lane) * nw + w] : 0.0;
        const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w];
        __syncthreads();
        // phase A: F = [UL; UM] and the per-(j,l) pieces of the direct-dependence rows
        for (int idx = lane; idx < N * NSTR; idx += 64) {
            const int j = idx / NSTR, l = idx % NSTR;
            const double sg = ((l - m) & 1) ? -1.0 : 1.0;
            double yp = 0, ym = 0, dyp = 0, dym = 0;
            for (int r = 0; r < N; ++r) {
                const double cm = 0.5 * g.wt[r] * om * sa[l] * lpmu[r * NSTR + l];
                const double wp = sRec[R::WP + r + N * j], wm = sRec[R::WM + r + N * j];
                const double dwp = sRec[R::DWP + r + N * j] + wp / om, dwm = sRec[R::DWM + r + N * j] + wm / om;
                yp += cm * (sg * wp + wm);
                ym += cm * (sg * wm + wp);
                dyp += cm * (sg * dwp + dwm);
                dym += cm * (sg * dwm + dwp);
            }
            skd::PPMult pm;
            skd::ppm_load(&sRec[R::PPM + 14 * j], pm);
            sF[j * NSTR + l] = yp * pm.hp / om;
            sF[(N + j) * NSTR + l] = ym * pm.hm / om;
            const double L = sXs[j], Mv = sXs[N + j], dk = sRec[R::DK + j];
            const double Ap = sRec[R::AP + j], Am = sRec[R::AM + j], dAp = sRec[R::DAP + j], dAm = sRec[R::DAM + j];
            const double tL = pm.hp * L + Ap * pm.Dm, tM = pm.hm * Mv + Am * pm.Dp;
            sH[(4 * N + j) * NSTR + l] = yp * tL + ym * tM;
            sH[(skd::TH_TAU * N + j) * NSTR + l] = yp * (pm.hp_tau * L + Ap * pm.Dm_tau) + ym * (pm.hm_tau * Mv + Am * pm.Dp_tau);
            sH[(skd::TH_SEC * N + j) * NSTR + l] = yp * Ap * pm.Dm_sec + ym * Am * pm.Dp_sec;
            sH[(skd::TH_T * N + j) * NSTR + l] = (T != 0) ? (yp * Ap * pm.Dm + ym * Am * pm.Dp) / T : 0.0;
            sH[(skd::TH_OM * N + j) * NSTR + l] = dyp * tL + dym * tM + yp * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) +
                                                  ym * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
        }
        // edge-intensity derivatives at fixed (L, M): lanes (i, e)
        for (int idx = lane; idx < 4 * N; idx += 64) {
            const int i = idx / 4, e = idx % 4;
            double d_tau = 0, d_om = 0, ipb_i = sRec[R::G + 2 * N + i];
            for (int j = 0; j < N; ++j) {
                const double wp = sRec[R::WP + i + N * j], wm = sRec[R::WM + i + N * j];
                const double dwp = sRec[R::DWP + i + N * j], dwm = sRec[R::DWM + i + N * j];
                const double E = sRec[R::E + j], L = sXs[j], Mv = sXs[N + j];
                const double dE_tau = -sRec[R::K + j] * E, dE_om = -sRec[R::DK + j] * th * E;
                if (e == 0) { d_tau += wm * dE_tau * Mv; d_om += dwp * L + dwm * E * Mv + wm * dE_om * Mv; }
                if (e == 1) { d_tau += wp * dE_tau * Mv; d_om += dwm * L + dwp * E * Mv + wp * dE_om * Mv; }
                if (e == 2) { d_tau += wp * dE_tau * L; d_om += dwp * E * L + wp * dE_om * L + dwm * Mv; ipb_i += wp * E * L + wm * Mv; }
                if (e == 3) { d_tau += wm * dE_tau * L; d_om += dwm * E * L + wm * dE_om * L + dwp * Mv; }
            }
            const double v_tau = sRec[R::DG + (skd::TH_TAU * 4 + e) * N + i] + d_tau;
            const double v_om = sRec[R::DG + (skd::TH_OM * 4 + e) * N + i] + d_om;
            const double v_T = sRec[R::DG + (skd::TH_T * 4 + e) * N + i], v_sec = sRec[R::DG + (skd::TH_SEC * 4 + e) * N + i];
            o[O::DI + (skd::TH_TAU * 4 + e) * N + i] = v_tau;
            o[O::DI + (skd::TH_OM * 4 + e) * N + i] = v_om;
            o[O::DI + (skd::TH_T * 4 + e) * N + i] = v_T;
            o[O::DI + (skd::TH_SEC * 4 + e) * N + i] = v_sec;
            if (e == 2) {
                sIpb[i] = g.wt[i] * g.mu[i] * ipb_i;
                sDIpb[skd::TH_TAU * N + i] = g.wt[i] * g.mu[i] * v_tau;
                sDIpb[skd::TH_OM * N + i] = g.wt[i] * g.mu[i] * v_om;
                sDIpb[skd::TH_T * N + i] = g.wt[i] * g.mu[i] * v_T;
                sDIpb[skd::TH_SEC * N + i] = g.wt[i] * g.mu[i] * v_sec;
            }
        }
        // ground part of u
        if (lane < N) {
            double sp = 0, sm = 0;
            if (last && m == 0)
                for (int i = 0; i < N; ++i) {
                    sp += g.mu[i] * g.wt[i] * sRec[R::WP + i + N * lane];
                    sm += g.mu[i] * g.wt[i] * sRec[R::WM + i + N * lane];
                }
            sTmp[lane] = 2.0 * rho * sp * sRec[R::E + lane];
            sTmp[N + lane] = 2.0 * rho * sm;
        }
        __syncthreads();
        // phase B: HD rows, FP = Dinv^T F, SP / BP, UG, GD, IPB
        for (int idx = lane; idx < 4 * NSTR; idx += 64) {
            const int t = idx / NSTR, l = idx % NSTR;
            double acc = 0, val = 0;
            for (int j = 0; j < N; ++j) {
                acc += sH[(t * N + j) * NSTR + l];
                val += sH[(4 * N + j) * NSTR + l];
            }
            acc /= om;
            if (t == skd::TH_OM) acc -= val / (om * om);
            o[O::HD + idx] = acc;
        }
        for (int idx = lane; idx < M2 * NSTR; idx += 64) {
            const int i = idx / NSTR, l = idx % NSTR;
            double t = 0;
            for (int j = 0; j < M2; ++j) t += sDinv[j * M2 + i] * sF[j * NSTR + l];
            o[O::FP + idx] = t;
        }
        for (int idx = lane; idx < M2 * N; idx += 64) {
            const int i = idx / N, cidx = idx % N;
            double t1 = 0, t2 = 0;
            for (int j = 0; j < N; ++j) {
                const double wp = sRec[R::WP + cidx + N * j], wm = sRec[R::WM + cidx + N * j], E = sRec[R::E + j];
                t1 += sDinv[j * M2 + i] * wm + sDinv[(N + j) * M2 + i] * wp * E;
                t2 += sDinv[j * M2 + i] * wp * E + sDinv[(N + j) * M2 + i] * wm;
            }
            o[O::SP + idx] = t1;
            o[O::BP + idx] = t2;
        }
        if (lane < M2) {
            double t = 0;
            for (int j = 0; j < M2; ++j) t += sDinv[j * M2 + lane] * sTmp[j];
            o[O::UG + lane] = t;
        }
        if (lane < 4) {
            double sacc = 0;
            if (last && m == 0)
                for (int i = 0; i < N; ++i) sacc += sDIpb[lane * N + i];
            o[O::GD + lane] = 2.0 * rho * sacc;
        }
        if (lane == 4) {
            double ipb = 0;
            for (int i = 0; i < N; ++i) ipb += sIpb[i];
            o[O::IPB] = ipb;
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------------------------
// K2b: adjoint sweeps + reduced gradient.  block = 64 lanes = wpb wavelengths x lpw lanes (one lane per LOS);
// the K2c operators of each of the workgroup's wavelengths are staged in LDS and read by all of its LOS lanes as
// broadcasts.  dynamic LDS: wpb * Ops::SIZE doubles.
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64) kern_k2b_wave(Geo g, Bufs b, int s, int m, int lpw, int wpb) {
    using O = skd::Ops<N>;
    extern __shared__ double smem[];
    const int lane = threadIdx.x, nlyr = g.nlyr;
    const int sub = lane / lpw, l = lane % lpw;
    const int w_lane = blockIdx.x * wpb + sub;
    const bool w_ok = (sub < wpb) && (w_lane < b.nw) && (m < b.norders[w_lane < b.nw ? w_lane : 0]);
    double* myOps = smem + size_t(sub < wpb ? sub : 0) * O::SIZE;
    auto stage = [&](int p, int first, int count) {
        __syncthreads();
        if (w_ok) {
            const double* src = skd::ops_at<N>(b, nlyr, p, w_lane);
            for (int i = first + l; i < first + count; i += lpw) myOps[i] = src[i];
        }
        __syncthreads();
    };
    for (int l0 = 0; l0 < g.nlos; l0 += lpw) {
        const int los = l0 + l;
        const bool active = w_ok && los < g.nlos;
        skd::K2bTask<N> t(g, b, s, m, active ? los : 0, active ? w_lane : 0);
        for (int p = 0; p < nlyr; ++p) {
            stage(p, O::FP, O::DI - O::FP);
            if (active) t.forward(p, myOps);
        }
        for (int i = 0; i < N; ++i) t.carry[i] = 0;
        for (int p = nlyr - 1; p >= 0; --p) {
            stage(p, O::BP, O::UG - O::BP);
            if (active) t.backward(p, myOps);
        }
        for (int p = 0; p < nlyr; ++p) {
            stage(p, O::DI, O::SIZE - O::DI);
            if (active) t.gradient(p, myOps);
        }
    }
}

}  // namespace skdev
