// Workgroup-per-wavelength HIP kernels (gfx950): one 64-lane wavefront owns one wavelength (K2a, K3) or a small
// group of wavelengths with one lane per line of sight (K2b).  The streams x streams layer matrices and the 2N x 2N
// boundary-value blocks are staged in LDS, records are streamed from HBM with unit stride, and all cross-lane steps
// (pivot search, broadcasts) are wavefront shuffles -- a single-wave workgroup makes every barrier a cheap s_barrier.
//
// HIP only (included by disco_engine.hip).  The arithmetic is the same as the task functions of disco_core.hpp, which
// stay as the CPU-emulated specification; parity of these kernels is asserted on the GPU against the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "disco_core.hpp"
#include "disco_lane_kernels.hpp"

// minimum waves per SIMD requested from the register allocator (second __launch_bounds__ argument)
// Jacobi stops when sum(off-diagonal^2) <= tol * sum(diagonal^2): 1e-30 is an off-diagonal norm of 1e-15 of the diagonal,
// i.e. the rounding level of the rotations themselves (a tighter test only buys sweeps that change nothing)
#ifndef SKD_JACOBI_TOL
#define SKD_JACOBI_TOL 1e-30
#endif
#ifndef SKD_EXP_MAXSWEEP
#define SKD_EXP_MAXSWEEP 60
#endif
#ifndef SKD_WPE_K2AM
#define SKD_WPE_K2AM 4
#endif
// (the backward sweep with the next layer's operands requested one layer ahead into a second set of registers -- 3 or 2 wavefronts per SIMD --
// measured no faster, 16.8 / 16.2 against 16.5 ms per step: the sweep runs at 58 % of the HBM peak on its sweep states, it does not wait on latency)
#ifndef SKD_WPE_K2BM
#define SKD_WPE_K2BM 4
#endif
// (round 3: 3 -- at 128 registers the fused kernel spilled 33 and lost: 76 k wavelengths/s at 4 waves per SIMD, 86 k at 3.  Round 4: 2.  The
// kernel is bound by instruction issue (about 1 500 instructions per layer of which 280 are FP64 / matrix-core work: per-phase counts
// from the assembly between the clock64 stamps of SKD_K2C_TIMING), not by memory latency -- the next layer's inputs arrive by LDS-DMA --,
// so a third wavefront buys nothing and costs the registers: 41.3 ms per step alone at 3 with 12 spilled, 37.2 ms at 2 with none)
#ifndef SKD_WPE_K2CM
#define SKD_WPE_K2CM 2
#endif
#ifndef SKD_K1_LG
#define SKD_K1_LG 16
#endif
#ifndef SKD_WPE_K3
#define SKD_WPE_K3 4
#endif

#ifndef SKD_WPE_K1
#define SKD_WPE_K1 2
#endif
#ifndef SKD_WPE_K1P
#define SKD_WPE_K1P 2
#endif
#ifndef SKD_WPE_K2A
#define SKD_WPE_K2A 1
#endif
#ifndef SKD_WPE_K2C
#define SKD_WPE_K2C 1
#endif
#ifndef SKD_WPE_K2B
#define SKD_WPE_K2B 1
#endif

#ifdef SKD_K1_TIMING
#define K1T(i) { long long _t = clock64(); tacc[i] += _t - tlast; tlast = _t; }
#else
#define K1T(i)
#endif

namespace skdev {

using skd::Bufs;
using skd::Geo;

// Ordering of LDS writes and reads between the lanes of a ONE-WAVEFRONT block (every kernel that calls it is launched with 64 threads):
// the LDS serves a wavefront's requests in order, so it is enough to wait for them (and to keep the compiler from moving memory
// operations across).  __syncthreads() would add a workgroup fence that drains the global loads, stores and LDS-DMA requests in flight.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// -------------------------------------------------------------------------------------------------------------------
// K1: homogeneous + particular solution of every layer.  grid = wavelengths, block = 64 lanes = NG lane groups of LG
// lanes; each group owns one layer per pass and keeps the layer's streams x streams matrices in LDS
// (B+, B-, Cholesky factor, Jacobi work matrix, eigenvectors, ...).  Same mathematics as skd::k1_layer_solution.
// -------------------------------------------------------------------------------------------------------------------
// PRE = true: the eigen-solution (k_j^2 and X) of every layer was left in the record by kern_kj_lane (lane-per-problem Jacobi,
// disco_lane_kernels.hpp); this kernel then skips its own Cholesky / Jacobi / back-substitution stages.
template <int N, int LG, bool PRE> __device__ __forceinline__ void k1_wave_body(const Geo& g, const Bufs& b, int s, int m, int layers_per_block) {
    constexpr int NSTR = 2 * N, NN = N * N, NG = 64 / LG;
    static_assert(LG >= N && (LG == 16 || LG == 32 || LG == 64), "lane group must cover one matrix row");
    using R = skd::Rec<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, grp = lane / LG, gl = lane % LG, nw = b.nw, nlyr = g.nlyr;
    // (PRE: no Cholesky factor -- seven work matrices, 25 instead of 27 KB of LDS per wavefront: a sixth wavefront per CU)
    constexpr int NMAT = PRE ? 7 : 8;
    __shared__ double sMat[NG][NMAT][NN];
    __shared__ double sVec[NG][18][NSTR];
    __shared__ double sLp[N * NSTR], sLc[NSTR], sMu[N], sWt[N], sSq[N], sTt[N], sIm[N], sIt[N];
    __shared__ int sErr;
    for (int i = lane; i < N * NSTR; i += 64) sLp[i] = g.lpmu[size_t(m) * N * NSTR + i];
    for (int i = lane; i < NSTR; i += 64) sLc[i] = g.lpcsz[(size_t(s) * NSTR + m) * NSTR + i];
    if (lane < N) {
        sMu[lane] = g.mu[lane];
        sWt[lane] = g.wt[lane];
        sSq[lane] = sqrt(g.wt[lane] / g.mu[lane]);
        sTt[lane] = sqrt(g.wt[lane] * g.mu[lane]);
        sIm[lane] = 1.0 / g.mu[lane];
        sIt[lane] = 1.0 / sqrt(g.wt[lane] * g.mu[lane]);
    }
    if (lane == 0) sErr = 0;
    double* Bp = sMat[grp][0];
    double* Bm = sMat[grp][1];
    double* M3 = sMat[grp][2];
    double* M4 = sMat[grp][3];
    double* M5 = sMat[grp][4];
    double* M6 = sMat[grp][5];
    double* M7 = sMat[grp][6];
    double* Rc = sMat[grp][NMAT - 1];   // Cholesky factor (PRE: never touched, aliases M7)
    double* va = sVec[grp][0];    // Legendre coefficients a_l
    double* vlam = sVec[grp][1];
    double* vk = sVec[grp][2];
    double* vdk = sVec[grp][3];
    double* vnrm = sVec[grp][4];
    double* vdnrm = sVec[grp][5];
    double* vQp = sVec[grp][6];
    double* vQm = sVec[grp][7];
    double* vAp = sVec[grp][8];
    double* vAm = sVec[grp][9];
    double* vdAp = sVec[grp][10];
    double* vdAm = sVec[grp][11];
    double* vC = sVec[grp][12];   // Cp.v, Cm.v interleaved: [2j], [2j+1]
    double* vdC = sVec[grp][13];  // dCp[4], dCm[4] per j -> 8N doubles (rows 13..16)
    double* vRinv = vdnrm;   // reciprocal diagonal of the Cholesky factor
    double* vIk = vAm;       // 1 / k_j
    double* vE = vnrm;       // exp(-k_j tau), j < N, and exp(-sec tau) at [N]
    double* vCoef = vC;      // 10 N doubles: rows 12..16 of sVec
    double* vAl = vQp;       // rotation parameters of the Jacobi rounds (the Q / A rows are written later)
    double* vBe = vQm;
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * skd::kPi);
    __syncthreads();

#ifndef SKD_K1_JACOBI_LDS
    // Register-resident Jacobi (see the eigen stage below): the matrix lives in position space -- every round rotates the position
    // pairs (2k, 2k+1), between rounds the data move by the fixed round-robin permutation -- so a lane owns the same 2x2 blocks (pair k
    // rows x pair k' columns, the diagonal blocks first) and eigenvector pairs in every round; their LDS places are computed once here.
    constexpr int JH = (N > 1) ? N / 2 : 1, JNB = JH * JH, JNV = N * JH, JUA = (JNB + LG - 1) / LG, JUV = (JNV + LG - 1) / LG;
    int j_own[JUA], j_d00[JUA], j_d01[JUA], j_d10[JUA], j_d11[JUA], j_k1[JUA], j_k2[JUA], jv_own[JUV], jv_dp[JUV], jv_dq[JUV], jv_k2[JUV];
    bool j_on[JUA], jv_on[JUV];
    {
        auto jperm = [](int i) {   // round-robin move of position i: 0 stays, even positions shift up, odd ones down, the ends cross over
            if (i == 0 || N == 2) return i;
            if (i & 1) return (i == 1) ? 2 : i - 2;
            return (i == N - 2) ? N - 1 : i + 2;
        };
#pragma unroll
        for (int u = 0; u < JUA; ++u) {
            const int bk = gl + u * LG;
            j_on[u] = bk < JNB;
            int k = 0, k2 = 0;
            if (j_on[u]) {
                if (bk < JH) {
                    k = k2 = bk;
                } else {
                    const int o = bk - JH, j = o % (JH > 1 ? JH - 1 : 1);
                    k = o / (JH > 1 ? JH - 1 : 1);
                    k2 = j + (j >= k ? 1 : 0);
                }
            }
            j_k1[u] = k;
            j_k2[u] = k2;
            j_own[u] = (2 * k) * N + 2 * k2;
            const int pr0 = jperm(2 * k), pr1 = jperm(2 * k + 1), pc0 = jperm(2 * k2), pc1 = jperm(2 * k2 + 1);
            j_d00[u] = pr0 * N + pc0;
            j_d01[u] = pr0 * N + pc1;
            j_d10[u] = pr1 * N + pc0;
            j_d11[u] = pr1 * N + pc1;
        }
#pragma unroll
        for (int u = 0; u < JUV; ++u) {
            const int e = gl + u * LG;
            jv_on[u] = e < JNV;
            const int r = jv_on[u] ? e / JH : 0, k2 = jv_on[u] ? e % JH : 0;
            jv_k2[u] = k2;
            jv_own[u] = r * N + 2 * k2;
            jv_dp[u] = r * N + jperm(2 * k2);
            jv_dq[u] = r * N + jperm(2 * k2 + 1);
        }
    }
#endif
    const int p_begin = blockIdx.y * layers_per_block, p_end = min(nlyr, p_begin + layers_per_block);
#ifdef SKD_K1_TIMING
    long long tacc[11] = {0,0,0,0,0,0,0,0,0,0,0}, tlast = clock64();
#endif
    // inputs of a pass (layer optics; with PRE the eigen-solution left by kern_kj_lane), requested one pass ahead when the register budget
    // allows it (PRE: the Jacobi state is gone): every pass otherwise starts with two dependent global loads nothing else can hide
    constexpr int AU = (NSTR + LG - 1) / LG, XU = (NN + LG - 1) / LG;
    struct PassIn { double om, th, T, sc, a[AU], x[XU], lam; };
    auto fetch = [&](int pbase_, PassIn& in) {
        const int pp = (pbase_ + grp < p_end) ? pbase_ + grp : p_end - 1;
        if constexpr (N == 8) {
            const double* o = b.at + (size_t(w) * nlyr + pp) * 18;
            in.om = o[16]; in.th = o[17];
#pragma unroll
            for (int u = 0; u < AU; ++u) in.a[u] = (gl + u * LG < NSTR) ? o[gl + u * LG] : 0.0;
        } else {
            in.om = b.om[size_t(pp) * nw + w]; in.th = b.tau[size_t(pp) * nw + w];
#pragma unroll
            for (int u = 0; u < AU; ++u) in.a[u] = (gl + u * LG < NSTR) ? b.a[(size_t(pp) * NSTR + gl + u * LG) * nw + w] : 0.0;
        }
        in.T = b.T[size_t(pp) * nw + w]; in.sc = b.sec[size_t(pp) * nw + w];
        if (PRE) {
            const double* r = skd::rec_at<N>(b, nlyr, pp, w);
#pragma unroll
            for (int u = 0; u < XU; ++u) in.x[u] = (gl + u * LG < NN) ? r[R::DWP + gl + u * LG] : 0.0;
            in.lam = (gl < N) ? r[R::K + gl] : 0.0;
        }
    };
    PassIn cur;
    fetch(p_begin, cur);
    for (int pbase = p_begin; pbase < p_end; pbase += NG) {
        const bool valid = (pbase + grp) < p_end;
        const int p = valid ? pbase + grp : p_end - 1;
        const double om = cur.om, th = cur.th, T = cur.T, sc = cur.sc;
        const double inv_om_l = 1.0 / om;
        double* rec = skd::rec_at<N>(b, nlyr, p, w);
#pragma unroll
        for (int u = 0; u < AU; ++u) if (gl + u * LG < NSTR) va[gl + u * LG] = cur.a[u];
        const double lam_pre = cur.lam;
        double x_pre[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) x_pre[u] = cur.x[u];
        if (PRE && pbase + NG < p_end) fetch(pbase + NG, cur);
        __syncthreads();
        // B+-, copy of B+ for the Cholesky factor
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            double ep = 0, em = 0;
            for (int l = m; l < NSTR; ++l) {
                const double t = va[l] * sLp[i * NSTR + l] * sLp[j * NSTR + l];
                if ((l - m) & 1) em += t; else ep += t;
            }
            const double dlt = (i == j) ? sIm[i] : 0.0;
            const double bp = dlt - sSq[i] * sSq[j] * om * ep;
            Bp[e] = bp;
            Bm[e] = dlt - sSq[i] * sSq[j] * om * em;
            if (!PRE) Rc[e] = (j <= i) ? bp : 0.0;
        }
        if (PRE) {
            // eigen-solution from kern_kj_lane: X(i, j) at [R::DWP + i N + j] (rewritten with dW+ below), k_j^2 at [R::K + j]
#pragma unroll
            for (int u = 0; u < XU; ++u) if (gl + u * LG < NN) M5[gl + u * LG] = x_pre[u];
        }
        __syncthreads();
        K1T(0)
        // Cholesky B+ = R R^T (lower), one column per step
        for (int j = 0; j < N && !PRE; ++j) {
            double tj = Rc[j * N + j];
            for (int k = 0; k < j; ++k) tj -= Rc[j * N + k] * Rc[j * N + k];
            if (!(tj > 0)) { if (gl == 0) sErr = skd::ERR_CHOLESKY; tj = 1.0; }
            const double d = sqrt(tj), rd = 1.0 / d;
            double val = 0;
            const int i = gl;
            const bool mine = (i < N && i >= j);
            if (mine) {
                double t = Rc[i * N + j];
                for (int k = 0; k < j; ++k) t -= Rc[i * N + k] * Rc[j * N + k];
                val = (i == j) ? d : t * rd;
            }
            __syncthreads();
            if (gl == 0) vRinv[j] = rd;
            if (mine) Rc[i * N + j] = val;
            __syncthreads();
        }
        K1T(1)
        // M3 = B- R ;  M4 = R^T M3 (symmetric)
        for (int e = gl; e < NN && !PRE; e += LG) {
            const int i = e / N, j = e % N;
            double t = 0;
            for (int r = j; r < N; ++r) t += Bm[i * N + r] * Rc[r * N + j];
            M3[e] = t;
        }
        if (!PRE) __syncthreads();
        for (int e = gl; e < NN && !PRE; e += LG) {
            const int i = e / N, j = e % N;
            if (j <= i) {
                double t = 0;
                for (int r = i; r < N; ++r) t += Rc[r * N + i] * M3[r * N + j];
                M4[i * N + j] = t;
                M4[j * N + i] = t;
            }
            M5[e] = (i == j) ? 1.0 : 0.0;
        }
        if (!PRE) __syncthreads();
        K1T(2)
        K1T(3)
        // Jacobi on M4 with the round-robin ("tournament") ordering: every round applies N/2 disjoint rotations at once,
        // A <- J^T A J and V <- V J with J the product of the round's rotations; eigenvectors accumulate in M5.
        // Default: register-resident blocks in position space (ownership tables above): rotate in registers, write to the permuted
        // places, read the own places back.  SKD_K1_JACOBI_LDS keeps the earlier form for comparison (index i of a rotated pair (p, q)
        // has partner i' and coefficients (al, be) = (c, -s) for i = p, (c, +s) for i = q; M6 / M7 are the ping-pong write buffers).
        bool jac_ok = (N == 1) || PRE;
#ifndef SKD_K1_JACOBI_LDS
        if (N > 1 && !PRE) {
            double a00[JUA], a01[JUA], a10[JUA], a11[JUA], evp[JUV], evq[JUV];
#pragma unroll
            for (int u = 0; u < JUA; ++u) {
                a00[u] = a01[u] = a10[u] = a11[u] = 0.0;
                if (j_on[u]) {
                    a00[u] = M4[j_own[u]];
                    a01[u] = M4[j_own[u] + 1];
                    a10[u] = M4[j_own[u] + N];
                    a11[u] = M4[j_own[u] + N + 1];
                }
            }
#pragma unroll
            for (int u = 0; u < JUV; ++u) {
                evp[u] = evq[u] = 0.0;
                if (jv_on[u]) {
                    evp[u] = M5[jv_own[u]];
                    evq[u] = M5[jv_own[u] + 1];
                }
            }
            for (int sweep = 0; sweep < SKD_EXP_MAXSWEEP && !jac_ok; ++sweep) {
                // convergence: off-diagonal mass (upper triangle) against the diagonal, from the registers
                double off = 0, dia = 0;
#pragma unroll
                for (int u = 0; u < JUA; ++u)
                    if (j_on[u]) {
                        if (j_k1[u] == j_k2[u]) {
                            dia += a00[u] * a00[u] + a11[u] * a11[u];
                            off += a01[u] * a01[u];
                        } else if (j_k1[u] < j_k2[u]) {
                            off += a00[u] * a00[u] + a01[u] * a01[u] + a10[u] * a10[u] + a11[u] * a11[u];
                        }
                    }
                for (int o = LG / 2; o >= 1; o >>= 1) {
                    off += __shfl_xor(off, o, LG);
                    dia += __shfl_xor(dia, o, LG);
                }
                const bool conv = (off <= SKD_JACOBI_TOL * dia) || (off == 0);
                if (__all(conv)) { jac_ok = true; break; }
                for (int round = 0; round < N - 1; ++round) {
                    // rotations of the position pairs (2k, 2k+1), by the lanes holding the diagonal blocks (slot 0 of the first N/2 lanes)
                    if (gl < JH) {
                        const double apq = a01[0], app = a00[0], aqq = a11[0];
                        double c = 1.0, sn = 0.0;
                        if (apq != 0.0) {
                            // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without the
                            // division by apq:  t = sgn(d) h / (|d| + sqrt(d^2 + h^2)),  d = aqq - app, h = 2 apq;  c = 1 / sqrt(1 + t^2).
                            // Reciprocal square roots / reciprocals: hardware estimate + two Newton steps (full double precision).
                            const double d = aqq - app, h = 2.0 * apq, x = d * d + h * h;
                            double y = __builtin_amdgcn_rsq(x);
                            y = y * fma(-0.5 * x * y, y, 1.5);
                            y = y * fma(-0.5 * x * y, y, 1.5);
                            const double den = fabs(d) + x * y;
                            double id = __builtin_amdgcn_rcp(den);
                            id = fma(fma(-den, id, 1.0), id, id);
                            id = fma(fma(-den, id, 1.0), id, id);
                            const double tt = ((d >= 0) ? h : -h) * id;
                            const double x2 = fma(tt, tt, 1.0);
                            double y2 = __builtin_amdgcn_rsq(x2);
                            y2 = y2 * fma(-0.5 * x2 * y2, y2, 1.5);
                            y2 = y2 * fma(-0.5 * x2 * y2, y2, 1.5);
                            c = y2;
                            sn = tt * c;
                        }
                        vAl[gl] = c;
                        vBe[gl] = sn;
                    }
                    __syncthreads();
                    // A <- J^T A J, V <- V J on the registers, results to their next places
#pragma unroll
                    for (int u = 0; u < JUA; ++u) {
                        const double c1 = vAl[j_k1[u]], s1 = vBe[j_k1[u]], c2 = vAl[j_k2[u]], s2 = vBe[j_k2[u]];
                        const double t00 = c2 * a00[u] - s2 * a01[u], t01 = s2 * a00[u] + c2 * a01[u];
                        const double t10 = c2 * a10[u] - s2 * a11[u], t11 = s2 * a10[u] + c2 * a11[u];
                        const bool zero = (u == 0) && (gl < JH) && (s1 != 0.0);   // the rotated pair's off-diagonal element is annihilated exactly
                        const double r00 = c1 * t00 - s1 * t10, r11 = s1 * t01 + c1 * t11;
                        const double r10 = zero ? 0.0 : s1 * t00 + c1 * t10, r01 = zero ? 0.0 : c1 * t01 - s1 * t11;
                        if (j_on[u]) {
                            M4[j_d00[u]] = r00;
                            M4[j_d01[u]] = r01;
                            M4[j_d10[u]] = r10;
                            M4[j_d11[u]] = r11;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < JUV; ++u) {
                        const double c2 = vAl[jv_k2[u]], s2 = vBe[jv_k2[u]];
                        if (jv_on[u]) {
                            M5[jv_dp[u]] = c2 * evp[u] - s2 * evq[u];
                            M5[jv_dq[u]] = s2 * evp[u] + c2 * evq[u];
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < JUA; ++u)
                        if (j_on[u]) {
                            a00[u] = M4[j_own[u]];
                            a01[u] = M4[j_own[u] + 1];
                            a10[u] = M4[j_own[u] + N];
                            a11[u] = M4[j_own[u] + N + 1];
                        }
#pragma unroll
                    for (int u = 0; u < JUV; ++u)
                        if (jv_on[u]) {
                            evp[u] = M5[jv_own[u]];
                            evq[u] = M5[jv_own[u] + 1];
                        }
                }
            }
        }
#else
        constexpr int NM1 = (N > 1) ? N - 1 : 1;
        if (N > 1 && !PRE) {
            double* Acur = M4;
            double* Vcur = M5;
            double* Anew = M6;
            double* Vnew = M7;
            for (int sweep = 0; sweep < SKD_EXP_MAXSWEEP && !jac_ok; ++sweep) {
                double off = 0, dia = 0;
                for (int e = gl; e < NN; e += LG) {
                    const int i = e / N, j = e % N;
                    const double v = Acur[e] * Acur[e];
                    if (i == j) dia += v; else if (j > i) off += v;
                }
                for (int o = LG / 2; o >= 1; o >>= 1) {
                    off += __shfl_xor(off, o, LG);
                    dia += __shfl_xor(dia, o, LG);
                }
                const bool conv = (off <= SKD_JACOBI_TOL * dia) || (off == 0);
                if (__all(conv)) { jac_ok = true; break; }
                for (int round = 0; round < N - 1; ++round) {
                    // pairs of this round: (N-1, round) and ((round + k) mod (N-1), (round - k) mod (N-1)), k = 1 .. N/2-1
                    for (int t = gl; t < N / 2; t += LG) {
                        int pp, qq;
                        if (t == 0) { pp = round; qq = N - 1; }
                        else { pp = (round + t) % NM1; qq = (round + NM1 - t) % NM1; }
                        if (pp > qq) { const int tmp = pp; pp = qq; qq = tmp; }
                        const double apq = Acur[pp * N + qq], app = Acur[pp * N + pp], aqq = Acur[qq * N + qq];
                        double c = 1.0, sn = 0.0;
                        if (apq != 0.0) {
                            // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq), written without the
                            // division by apq:  t = sgn(d) h / (|d| + sqrt(d^2 + h^2)),  d = aqq - app, h = 2 apq;  c = 1 / sqrt(1 + t^2).
                            // Reciprocal square roots / reciprocals: hardware estimate + two Newton steps (full double precision).
                            const double d = aqq - app, h = 2.0 * apq, x = d * d + h * h;
                            double y = __builtin_amdgcn_rsq(x);
                            y = y * fma(-0.5 * x * y, y, 1.5);
                            y = y * fma(-0.5 * x * y, y, 1.5);
                            const double den = fabs(d) + x * y;
                            double id = __builtin_amdgcn_rcp(den);
                            id = fma(fma(-den, id, 1.0), id, id);
                            id = fma(fma(-den, id, 1.0), id, id);
                            const double tt = ((d >= 0) ? h : -h) * id;
                            const double x2 = fma(tt, tt, 1.0);
                            double y2 = __builtin_amdgcn_rsq(x2);
                            y2 = y2 * fma(-0.5 * x2 * y2, y2, 1.5);
                            y2 = y2 * fma(-0.5 * x2 * y2, y2, 1.5);
                            c = y2;
                            sn = tt * c;
                        }
                        vAl[pp] = c; vBe[pp] = -sn;
                        vAl[qq] = c; vBe[qq] = sn;
                    }
                    __syncthreads();
                    for (int e = gl; e < NN; e += LG) {
                        const int i = e / N, j = e % N;
                        // partner of index x in this round: N-1 <-> round, otherwise (2 round - x) mod (N-1)
                        auto partner = [&](int x) { return (x == N - 1) ? round : ((x == round) ? N - 1 : (2 * round + 2 * NM1 - x) % NM1); };
                        const int ip = partner(i), jp = partner(j);
                        const double ai = vAl[i], bi = vBe[i], aj = vAl[j], bj = vBe[j];
                        double v;
                        if (jp == i) v = 0.0;  // the rotated pair's off-diagonal element is annihilated exactly
                        else v = ai * (aj * Acur[i * N + j] + bj * Acur[i * N + jp]) + bi * (aj * Acur[ip * N + j] + bj * Acur[ip * N + jp]);
                        Anew[e] = v;
                        Vnew[e] = aj * Vcur[i * N + j] + bj * Vcur[i * N + jp];
                    }
                    __syncthreads();
                    { double* tmp = Acur; Acur = Anew; Anew = tmp; }
                    { double* tmp = Vcur; Vcur = Vnew; Vnew = tmp; }
                }
            }
            // leave the result in M4 / M5 (an odd number of rounds ends in the other pair of buffers)
            if (Acur != M4) {
                for (int e = gl; e < NN; e += LG) { M4[e] = Acur[e]; M5[e] = Vcur[e]; }
                __syncthreads();
            }
        }
#endif
        K1T(4)
        if (!jac_ok && gl == 0) sErr = skd::ERR_EIGEN;
        if (gl < N) {
            double lam = PRE ? lam_pre : M4[gl * N + gl];
            if (!(lam > 0)) { sErr = skd::ERR_EIGEN; lam = fabs(lam) + 1e-300; }
            vlam[gl] = lam;
            vk[gl] = sqrt(lam);
            vIk[gl] = 1.0 / sqrt(lam);
            // xhat = R^-T v (column gl, in place), x = T^-1 xhat
            const int j = gl;
            for (int i = N - 1; i >= 0 && !PRE; --i) {
                double t = M5[i * N + j];
                for (int r = i + 1; r < N; ++r) t -= Rc[r * N + i] * M5[r * N + j];
                M5[i * N + j] = t * vRinv[i];
            }
            for (int i = 0; i < N && !PRE; ++i) M5[i * N + j] *= sIt[i];
        }
        __syncthreads();
        K1T(5)
        // U (M3) = S+ X with S+-_ir = B+-_ir tt_r / tt_i
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            double t = 0;
            for (int r = 0; r < N; ++r) t += Bp[i * N + r] * sTt[r] * M5[r * N + j];
            M3[e] = t * sIt[i];
        }
        __syncthreads();
        // d(Gamma) X (M4)
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            double hm = 0;
            for (int r = 0; r < N; ++r) hm += Bm[i * N + r] * (sTt[r] * sIm[r]) * M5[r * N + j];
            hm *= sIt[i];
            M4[e] = (2 * M5[e] * vlam[j] - M3[e] * sIm[i] - hm) * inv_om_l;
        }
        __syncthreads();
        // C (M6), dk
        double lmax = 0;
        for (int j = 0; j < N; ++j) lmax = fmax(lmax, vlam[j]);
        for (int e = gl; e < NN; e += LG) {
            const int aa = e / N, bb = e % N;
            double z = 0;
            for (int i = 0; i < N; ++i) z += sWt[i] * sMu[i] * M3[i * N + aa] * M4[i * N + bb];
            if (aa == bb) {
                vdk[bb] = 0.5 * z * vIk[bb];
                M6[e] = 0;
            } else {
                const double den = vlam[bb] - vlam[aa];
                M6[e] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
            }
        }
        __syncthreads();
        // dX (M7) = X C
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            double t = 0;
            for (int r = 0; r < N; ++r) t += M5[i * N + r] * M6[r * N + j];
            M7[e] = t;
        }
        __syncthreads();
        // dU (M6) = (U - M^-1 X)/om + S+ dX
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            double t = 0;
            for (int r = 0; r < N; ++r) t += Bp[i * N + r] * sTt[r] * M7[r * N + j];
            M6[e] = (M3[e] - M5[e] * sIm[i]) * inv_om_l + t * sIt[i];
        }
        __syncthreads();
        K1T(6)
        // W+-, dW+- (element-wise, in place: M5 <- W+, M3 <- W-, M7 <- dW+, M6 <- dW-), stored as W(i, solution j) at i + N j
        for (int e = gl; e < NN; e += LG) {
            const int i = e / N, j = e % N;
            const double X = M5[e], U = M3[e], dX = M7[e], dU = M6[e];
            const double ik = vIk[j];
            const double uk = U * ik, duk = dU * ik - U * vdk[j] * ik * ik;
            const double wp = 0.5 * (X + uk), wm = 0.5 * (X - uk), dwp = 0.5 * (dX + duk), dwm = 0.5 * (dX - duk);
            M5[e] = wp; M3[e] = wm; M7[e] = dwp; M6[e] = dwm;
            if (valid) {
                rec[R::WP + i + N * j] = wp;
                rec[R::WM + i + N * j] = wm;
                rec[R::DWP + i + N * j] = dwp;
                rec[R::DWM + i + N * j] = dwm;
            }
        }
        K1T(7)
        // solar pseudo-source Q+-
        if (gl < N) {
            const int i = gl;
            double sp = 0, sn = 0;
            for (int l = m; l < NSTR; ++l) {
                const double t = va[l] * sLp[i * NSTR + l] * sLc[l];
                sp += t;
                sn += ((l - m) & 1) ? -t : t;
            }
            vQp[i] = sp * fm * sWt[i] * om;
            vQm[i] = sn * fm * sWt[i] * om;
        }
        // exp(-k_j tau) of every solution and exp(-sec tau) of the layer: one instruction stream for N + 1 lanes
        if (gl <= N) vE[gl] = exp(-th * ((gl < N) ? vk[gl] : sc));
        __syncthreads();
        K1T(8)
        // Green's function coefficients of solution j: norm, A+-, multipliers C+-, and the coefficient vectors of the
        // particular solution  G = W (A C)  with their derivatives:  coef[(type * 5 + k) * N + j], type 0: (A-, C-), 1: (A+, C+);
        // k = 0: A C,  k = 1 + theta: d(A C)/d theta at fixed W
        if (gl < N) {
            const int j = gl;
            const double inv_om = 1.0 / om;
            bool allzero = true;
            for (int i = 0; i < N; ++i) if (vQp[i] != 0 || vQm[i] != 0) allzero = false;
            double nrm = 0, dnrm = 0, Ap = 0, Am = 0, dAp = 0, dAm = 0;
            for (int i = 0; i < N; ++i) {
                const double wp = M5[i * N + j], wm = M3[i * N + j], dwp = M7[i * N + j], dwm = M6[i * N + j];
                const double wmu = sWt[i] * sMu[i], qp = vQp[i], qm = vQm[i];
                nrm += wmu * (wp * wp - wm * wm);
                dnrm += wmu * 2.0 * (wp * dwp - wm * dwm);
                Ap += qp * wp + qm * wm;
                Am += qm * wp + qp * wm;
                dAp += qp * (dwp + inv_om * wp) + qm * (dwm + inv_om * wm);
                dAm += qm * (dwp + inv_om * wp) + qp * (dwm + inv_om * wm);
            }
            skd::Mult Cp{0, 0, 0, 0}, Cm{0, 0, 0, 0};
            if (allzero) {
                Ap = Am = dAp = dAm = 0;
            } else {
                const double inv = 1.0 / nrm;
                Ap *= inv;
                Am *= inv;
                dAp = (dAp - dnrm * Ap) * inv;
                dAm = (dAm - dnrm * Am) * inv;
                skd::green_multipliers_e(T, th, sc, vk[j], vE[j], vE[N], Cp, Cm);
            }
            const double dk = vdk[j], invT = (T != 0) ? 1.0 / T : 0.0;
            const double dCm[4] = {Cm.d_tau, Cm.d_k * dk, Cm.v * invT, Cm.d_sec};
            const double dCp[4] = {Cp.d_tau, Cp.d_k * dk, Cp.v * invT, Cp.d_sec};
            vCoef[(0 * 5 + 0) * N + j] = Am * Cm.v;
            vCoef[(1 * 5 + 0) * N + j] = Ap * Cp.v;
            for (int t = 0; t < 4; ++t) {
                vCoef[(0 * 5 + 1 + t) * N + j] = Am * dCm[t] + ((t == skd::TH_OM) ? dAm * Cm.v : 0.0);
                vCoef[(1 * 5 + 1 + t) * N + j] = Ap * dCp[t] + ((t == skd::TH_OM) ? dAp * Cp.v : 0.0);
            }
            if (valid) {
                rec[R::K + j] = vk[j];
                rec[R::E + j] = vE[j];
                rec[R::DK + j] = dk;
                rec[R::AP + j] = Ap; rec[R::AM + j] = Am; rec[R::DAP + j] = dAp; rec[R::DAM + j] = dAm;
            }
        }
        __syncthreads();
        K1T(9)
        // G (4 vectors) and dG (4 params x 4 vectors).  A lane owns row i of W+ or W- (and of its omega-derivative) and one
        // coefficient type: G[c] = Wsel coef_0, dG[theta][c] = Wsel coef_{1+theta} (+ dWsel coef_0 for theta = omega),
        // c = 0: (W-, type 0), 1: (W+, type 0), 2: (W+, type 1), 3: (W-, type 1)
        for (int it = gl; it < 4 * N; it += LG) {
            const int i = it % N, sel = (it / N) & 1, type = it / (2 * N);
            const double* Wsel = sel ? M3 : M5;
            const double* dWsel = sel ? M6 : M7;
            const int c = (type == 0) ? (sel ? 0 : 1) : (sel ? 3 : 2);
            const double* cf = vCoef + (type * 5) * N;
            double acc[5] = {0, 0, 0, 0, 0}, accd = 0;
            for (int j = 0; j < N; ++j) {
                const double wv = Wsel[i * N + j], dwv = dWsel[i * N + j];
                for (int k = 0; k < 5; ++k) acc[k] += wv * cf[k * N + j];
                accd += dwv * cf[j];
            }
            if (valid) {
                rec[R::G + c * N + i] = acc[0];
                for (int t = 0; t < 4; ++t) rec[R::DG + (t * 4 + c) * N + i] = acc[1 + t] + ((t == skd::TH_OM) ? accd : 0.0);
            }
        }
        K1T(10)
        __syncthreads();
        if (!PRE && pbase + NG < p_end) fetch(pbase + NG, cur);
    }
#ifdef SKD_K1_TIMING
    if (lane == 0 && w == 0 && blockIdx.y == 0 && m == 1) printf("K1T build %lld chol %lld m3m4 %lld warm %lld jac %lld backsub %lld UdU %lld W %lld Q %lld coef %lld G %lld\n", tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6], tacc[7], tacc[8], tacc[9], tacc[10]);
#endif
    if (lane == 0 && sErr) atomicOr(&b.status[w], sErr);
}
// (the occupancy request of a kernel cannot depend on a template parameter: one entry point per register budget)
template <int N, int LG, bool PRE = false> __global__ void __launch_bounds__(64, SKD_WPE_K1) kern_k1_wave(Geo g, Bufs b, int s, int m, int layers_per_block) {
    static_assert(!PRE, "the continuation from kern_kj_lane is kern_k1p_wave");
    k1_wave_body<N, LG, false>(g, b, s, m, layers_per_block);
}
template <int N, int LG> __global__ void __launch_bounds__(64, SKD_WPE_K1P) kern_k1p_wave(Geo g, Bufs b, int s, int m, int layers_per_block) {
    k1_wave_body<N, LG, true>(g, b, s, m, layers_per_block);
}

// -------------------------------------------------------------------------------------------------------------------
// K2a: block-tridiagonal BVP, factor + solve.  grid = wavelengths, block = 64.
// LDS: current/previous layer W+-, k, G (ping-pong), augmented [D | I] for the Gauss-Jordan inverse, previous Dinv.
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64, SKD_WPE_K2A) kern_k2a_wave(Geo g, Bufs b, int s, int m) {
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
                    rB += g.csz[s] * rho / skd::kPi * b.T[size_t(nlyr) * nw + w] + 2.0 * rho * sg + (b.surf_em ? b.surf_em[w] : 0.0);
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
template <int N, class R = skd::Rec<N>> __global__ void __launch_bounds__(64, SKD_WPE_K3) kern_k3_wave(Geo g, Bufs b, int s, int m, int layers_per_block) {
    // NG lane groups of LG lanes, one layer per group and pass (the per-layer phases only have N .. 2N lanes of work)
    constexpr int NSTR = 2 * N, NN = N * N, LG = (N >= 16) ? 32 : 16, NG = 64 / LG;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, grp = lane / LG, gl = lane % LG, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sKg[NG][N], sWg[NG][2 * NN], sAPg[NG][N], sAMg[NG][N], sXsg[NG][2 * N], sag[NG][NSTR], sCLg[NG][N], sCMg[NG][N],
        sPg[NG][N], sQg[NG][N], sClegg[NG][NSTR], sRedg[NG][N], sExg[NG][N + 2];
    double* sK = sKg[grp]; double* sW = sWg[grp]; double* sAP = sAPg[grp]; double* sAM = sAMg[grp]; double* sXs = sXsg[grp];
    double* sa = sag[grp]; double* sCL = sCLg[grp]; double* sCM = sCMg[grp]; double* sP = sPg[grp]; double* sQ = sQg[grp];
    double* sCleg = sClegg[grp]; double* sRed = sRedg[grp]; double* sEx = sExg[grp];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    int unconverged = 0;
    const int p_begin = blockIdx.y * layers_per_block, p_end = min(nlyr, p_begin + layers_per_block);
    for (int pbase = p_begin; pbase < p_end; pbase += NG) {
        const bool valid = (pbase + grp) < p_end;
        const int p = valid ? pbase + grp : p_end - 1;
        const double* rec = skd::rec_at_r<R>(b, nlyr, p, w);
        const double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        for (int i = gl; i < N; i += LG) { sK[i] = rec[R::k(i)]; sAP[i] = rec[R::ap(i)]; sAM[i] = rec[R::am(i)]; }
        for (int i = gl; i < 2 * NN; i += LG) sW[i] = rec[(i < NN) ? R::wp(i % N, i / N) : R::wm(i % N, (i - NN) / N)];   // sW: W+(i, j) at i + N j, then W-
        for (int i = gl; i < 2 * N; i += LG) sXs[i] = xs[i];
        double th;
        if constexpr (N == 8) {
            const double* o = b.at + (size_t(w) * nlyr + p) * 18;
            for (int l = gl; l < NSTR; l += LG) sa[l] = o[l];
            th = o[17];
        } else {
            for (int l = gl; l < NSTR; l += LG) sa[l] = b.a[(size_t(p) * NSTR + l) * nw + w];
            th = b.tau[size_t(p) * nw + w];
        }
        const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
        wave_lds_sync();
        // exponentials of the layer in one instruction stream: exp(-k_j x) (lanes < N), exp(-sec x), exp(-sec th);  x = th lfrac
        const double lf = g.lfrac[p];
        if (gl < N + 2) sEx[gl] = exp(-((gl < N) ? sK[gl] : sc) * ((gl == N + 1) ? th : th * lf));
        wave_lds_sync();
        if (gl < N) {
            skd::PPMult pm;
            const double Ej = rec[R::e(gl)];  // exp(-k_j th), written by K1;  exp(-k (th - x)) = E / exp(-k x)
            skd::pp_multipliers_e(T, th, sc, sK[gl], lf, sEx[gl], Ej / sEx[gl], sEx[N], sEx[N + 1], pm);
            if (valid) skd::ppm_store(skd::rec_at_r<R>(b, nlyr, p, w) + R::ppm(gl), pm);
            sCL[gl] = pm.hp * sXs[gl] + sAP[gl] * pm.Dm;
            sCM[gl] = pm.hm * sXs[N + gl] + sAM[gl] * pm.Dp;
        }
        wave_lds_sync();
        if (gl < N) {
            double pr = 0, qr = 0;
            for (int j = 0; j < N; ++j) {
                const double wp = sW[gl + N * j], wm = sW[NN + gl + N * j];
                pr += wp * sCL[j] + wm * sCM[j];
                qr += wm * sCL[j] + wp * sCM[j];
            }
            sP[gl] = pr * 0.5 * g.wt[gl];
            sQ[gl] = qr * 0.5 * g.wt[gl];
        }
        wave_lds_sync();
        for (int l = gl; l < NSTR; l += LG) {
            double sp = 0, sq = 0;
            if (l >= m)
                for (int r = 0; r < N; ++r) {
                    sp += lpmu[r * NSTR + l] * sP[r];
                    sq += lpmu[r * NSTR + l] * sQ[r];
                }
            sCleg[l] = sa[l] * ((((l - m) & 1) ? -sp : sp) + sq);
        }
        wave_lds_sync();
        // (the thermal pass runs with the extra solar index s = nsza, which has no diffuse table: only the record entries above are needed)
        if (valid && s < g.nsza) {
            const int t0 = g.pt_start[s * nlyr + p], t1 = g.pt_start[s * nlyr + p + 1];
            for (int t = t0 + gl; t < t1; t += LG) {
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
        }
        // upwelling diffuse reflection of the Lambertian surface (bottom layer, order 0): uniform over the group
        const bool ground = valid && (p == nlyr - 1) && (m == 0);
        if (ground && gl < N) {
            double sv = rec[R::g(2 * N + gl)];
            for (int j = 0; j < N; ++j)
                sv += sXs[j] * sW[gl + N * j] * rec[R::e(j)] + sXs[N + j] * sW[NN + gl + N * j];
            sRed[gl] = 2.0 * g.mu[gl] * g.wt[gl] * sv * b.albedo[w];
        }
        wave_lds_sync();
        if (ground && s < g.nsza) {
            double diffuse = 0;
            for (int i = 0; i < N; ++i) diffuse += sRed[i];
            for (int t = g.gpt_start[s] + gl; t < g.gpt_start[s + 1]; t += LG) skd::J_at(b, g, m, g.gpt_slot[t], w) = diffuse;
        }
        wave_lds_sync();
    }
    if (__any(unconverged) && lane == 0) b.notconv[w] = 1;
}

// -------------------------------------------------------------------------------------------------------------------
// K2c: LOS-independent adjoint operators of every layer (see skd::k2c_layer_ops for the definitions).
// grid = wavelengths, block = 64; the layer record, its block inverse and the work matrices live in LDS.
// -------------------------------------------------------------------------------------------------------------------
template <int N> __global__ void __launch_bounds__(64, SKD_WPE_K2C) kern_k2c_wave(Geo g, Bufs b, int s, int m) {
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
        const double inv_om = 1.0 / om, inv_T = (T != 0) ? 1.0 / T : 0.0;
        __syncthreads();
        // phase A: F = [UL; UM] and the per-(j,l) pieces of the direct-dependence rows
        for (int idx = lane; idx < N * NSTR; idx += 64) {
            const int j = idx / NSTR, l = idx % NSTR;
            const double sg = ((l - m) & 1) ? -1.0 : 1.0;
            double yp = 0, ym = 0, dyp = 0, dym = 0;
            for (int r = 0; r < N; ++r) {
                const double cm = 0.5 * g.wt[r] * om * sa[l] * lpmu[r * NSTR + l];
                const double wp = sRec[R::WP + r + N * j], wm = sRec[R::WM + r + N * j];
                const double dwp = sRec[R::DWP + r + N * j] + wp * inv_om, dwm = sRec[R::DWM + r + N * j] + wm * inv_om;
                yp += cm * (sg * wp + wm);
                ym += cm * (sg * wm + wp);
                dyp += cm * (sg * dwp + dwm);
                dym += cm * (sg * dwm + dwp);
            }
            skd::PPMult pm;
            skd::ppm_load(&sRec[R::PPM + 14 * j], pm);
            sF[j * NSTR + l] = yp * pm.hp * inv_om;
            sF[(N + j) * NSTR + l] = ym * pm.hm * inv_om;
            const double L = sXs[j], Mv = sXs[N + j], dk = sRec[R::DK + j];
            const double Ap = sRec[R::AP + j], Am = sRec[R::AM + j], dAp = sRec[R::DAP + j], dAm = sRec[R::DAM + j];
            const double tL = pm.hp * L + Ap * pm.Dm, tM = pm.hm * Mv + Am * pm.Dp;
            sH[(4 * N + j) * NSTR + l] = yp * tL + ym * tM;
            sH[(skd::TH_TAU * N + j) * NSTR + l] = yp * (pm.hp_tau * L + Ap * pm.Dm_tau) + ym * (pm.hm_tau * Mv + Am * pm.Dp_tau);
            sH[(skd::TH_SEC * N + j) * NSTR + l] = yp * Ap * pm.Dm_sec + ym * Am * pm.Dp_sec;
            sH[(skd::TH_T * N + j) * NSTR + l] = (yp * Ap * pm.Dm + ym * Am * pm.Dp) * inv_T;
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
            acc *= inv_om;
            if (t == skd::TH_OM) acc -= val * inv_om * inv_om;
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
template <int N> __global__ void __launch_bounds__(64, SKD_WPE_K2B) kern_k2b_wave(Geo g, Bufs b, int s, int m, int lpw, int wpb) {
    using O = skd::Ops<N>;
    extern __shared__ double smem[];
    const int lane = threadIdx.x, nlyr = g.nlyr;
    const int sub = lane / lpw, l = lane % lpw;
    const int w_lane = blockIdx.x * wpb + sub;
    const bool w_ok = (sub < wpb) && (w_lane < b.nw) && (m < b.norders[w_lane < b.nw ? w_lane : 0]);
    double* myOps = smem + size_t(sub < wpb ? sub : 0) * O::SIZE;
    auto stage = [&](int p, int first, int count) {
#ifdef SKD_EXP_NOSTAGE
        if (p != 0 && p != nlyr - 1) return;
#endif
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

// ===================================================================================================================
// 16-stream path (N = 8, 2N = 16): the adjoint operators and sweeps as 16 x 16 FP64 matrix-core products.
//   v_mfma_f64_16x16x4_f64 operand maps (gfx950): A: lane l holds A[row = l & 15][k = l >> 4]; B: lane l holds
//   B[k = l >> 4][col = l & 15]; C/D: register r of lane l holds D[row = (l >> 4) + 4 r][col = l & 15].
//   Consequences used below: (1) the C/D image of X^T is the A-operand image of X, k-step r in register r;
//   (2) rows 8..15 (0..7) of a C/D tile are registers 2,3 (0,1) and are the B operand of the next product as they are.
// ===================================================================================================================
typedef double mf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mf4 mfma16(double a, double bb, mf4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0); }

// 64-bit data-parallel-primitive move and the max all-reduce of a 16-lane row built from it (quad_perm / row_ror patterns)
template <int CTRL> __device__ __forceinline__ double dpp64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ unsigned dpp32u(unsigned v) {
    return unsigned(__builtin_amdgcn_mov_dpp(int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, dpp32u<0xb1>(v));
    v = max(v, dpp32u<0x4e>(v));
    v = max(v, dpp32u<0x124>(v));
    v = max(v, dpp32u<0x128>(v));
    return v;
}
__device__ __forceinline__ double row16_max(double v) {
    v = fmax(v, dpp64<0xb1>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp64<0x4e>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp64<0x124>(v));  // row_ror 4
    v = fmax(v, dpp64<0x128>(v));  // row_ror 8
    return v;
}

// In-place Gauss-Jordan inverse of a 16 x 16 matrix held as one matrix-core accumulator tile (C/D image: register r of lane
// (col, q) = element (4 r + q, col)).  Partial (row) pivoting without moving rows: physical row pi(k) is the pivot of column k,
// one rank-1 matrix-core product per step, pivot row / column fixed up in registers.  On return the tile holds
// B[rho][j] = Ainv[pi^-1(rho)][pi(j)]; gj16_scatter() writes the natural layout through 2 KB of LDS.  Returns 1 if singular.
__device__ __forceinline__ int gj16_inplace(mf4& acc, int col, int q, unsigned long long& permP, unsigned long long& permI, double* trace = nullptr) {
    constexpr int M2 = 16;
    unsigned used = 0;
    int sing = 0;
    permP = 0;
    permI = 0;
#ifdef SKD_GJ_TIMING
    long long tg[6] = {0, 0, 0, 0, 0, 0}, tl = clock64();
#define GJT(i) { long long _t = clock64(); tg[i] += _t - tl; tl = _t; }
#else
#define GJT(i)
#endif
    // column c of the tile -> every 16-lane row: D[i = col][c] lives in lane (c, i & 3), register i >> 2
    auto gather_col = [&](int c) {
        const int src = c + 16 * (col & 3);
        const double c0 = __shfl(acc[0], src), c1 = __shfl(acc[1], src), c2 = __shfl(acc[2], src), c3 = __shfl(acc[3], src);
        const int ir = col >> 2;
        return (ir == 0) ? c0 : (ir == 1) ? c1 : (ir == 2) ? c2 : c3;
    };
    double colv = gather_col(0);
    for (int k = 0; k < M2; ++k) {
        // look-ahead: column k + 1 as it is BEFORE this step's update is requested now (its latency hides behind the pivot
        // search) and updated analytically below, so the next step does not wait for this step's product and a gather
        const double cnext = gather_col((k + 1) & 15);
        GJT(0)
        // partial pivoting: arg max |D[i][k]| over the rows not used yet, as an unsigned max-reduction of 32-bit keys inside
        // the 16-lane row (v_max_u32 with DPP operands, no LDS): key = high word of |value| (exponent + 20 mantissa bits,
        // ordered like the magnitude) with its low four bits replaced by 15 - i.  The pivot only has to be close to the
        // largest candidate; its exact value is read back below.
        const unsigned hi = unsigned(__double2hiint(colv)) & 0x7fffffffu;
        const unsigned key = ((used >> col) & 1u) ? 0u : ((hi & ~15u) | unsigned(15 - col));
        const unsigned kmb = row16_max_u32(key);
        GJT(1)
        const int pivrow = __builtin_amdgcn_readfirstlane(15 - int(kmb & 15u));
        double pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(colv), pivrow), __builtin_amdgcn_readlane(__double2loint(colv), pivrow));
        if (!(fabs(pv) > 0.0)) { sing = 1; pv = 1.0; }
        GJT(2)
        // reciprocal of the pivot: hardware estimate + two Newton steps (full double precision, short dependency chain)
        double ip = __builtin_amdgcn_rcp(pv);
        ip = fma(fma(-pv, ip, 1.0), ip, ip);
        ip = fma(fma(-pv, ip, 1.0), ip, ip);
        GJT(3)
        const int pr = pivrow >> 2, pq = pivrow & 3;
        // rank-1 update through k-slot pq of the product: the lanes (col, q = pq) already hold the pivot row D[pivrow][col]
        // in register pr, and the gathered column is replicated over q, so neither operand needs a cross-lane move
        const double rowsel = (pr == 0) ? acc[0] : (pr == 1) ? acc[1] : (pr == 2) ? acc[2] : acc[3];
        const double aop = (q == pq && col != pivrow) ? -colv * ip : 0.0;
        const double bop = (q == pq && col != k) ? rowsel : 0.0;
        // the rank-1 product leaves the pivot row and column untouched; they are scaled by one fused multiply-add per
        // register whose factors do not depend on the product (computed while the matrix core works)
        double ff[4], fg[4];
        {
            const double cs = (col == k) ? -ip : 1.0;                 // column k: -D[i][k] / p ; other columns unchanged
            const double fixv = (col == k) ? ip : rowsel * ip;        // pivot row: D[pivrow][j] / p, (pivrow, k) -> 1 / p
            for (int r = 0; r < 4; ++r) {
                const bool sel = (r == pr) && (q == pq);              // r == pr is uniform
                ff[r] = sel ? 0.0 : cs;
                fg[r] = sel ? fixv : 0.0;
            }
        }
        acc = mfma16(aop, bop, acc);
        GJT(4)
        for (int r = 0; r < 4; ++r) acc[r] = fma(acc[r], ff[r], fg[r]);
        {
            // column k + 1 after this step: D[i][k+1] - D[i][k] / p * D[pivrow][k+1], pivot row scaled by 1 / p
            const double b1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cnext), pivrow), __builtin_amdgcn_readlane(__double2loint(cnext), pivrow));
            colv = (col == pivrow) ? cnext * ip : fma(-colv * ip, b1, cnext);
        }
        GJT(5)
        if (trace && col == 0 && q == 0) { trace[2 * k] = pivrow; trace[2 * k + 1] = pv; }
        used |= 1u << pivrow;
        permP |= (unsigned long long)pivrow << (4 * k);
        permI |= (unsigned long long)k << (4 * pivrow);
    }
#ifdef SKD_GJ_TIMING
    if (trace && col == 0 && q == 0) for (int i = 0; i < 6; ++i) trace[40 + i] = double(tg[i]);
#endif
    return sing;
}
__device__ __forceinline__ void gj16_scatter(const mf4& acc, int col, int q, unsigned long long permP, unsigned long long permI, double* sP) {
    for (int r = 0; r < 4; ++r) {
        const int prow = 4 * r + q;
        const int i = int((permI >> (4 * prow)) & 15ull), c = int((permP >> (4 * col)) & 15ull);
        sP[i * 16 + c] = acc[r];
    }
}

// K2a for N = 8: block-tridiagonal boundary value problem with the 16 x 16 blocks held in matrix-core accumulators.
//   per layer   T1^T = Dinv'_{p-1}^T Sub_p^T (4 products), D'_p = D_p + T1[:, 8:16] [W- | W+ E]_p (2), r'_p = r_p - T1 r'_{p-1} (4),
//               Dinv'_p by in-place Gauss-Jordan with partial (row) pivoting: one rank-1 product per pivot step, the pivot
//               row / column are fixed up in registers, rows are never moved (the final permutation goes through 2 KB of LDS)
//   backward    x_p = Dinv'_p (r'_p - Sup_p x_{p+1})   (4 + 4 products)
//   Vectors are kept "replicated": every lane (col, q) holds entries 4 r + q, r = 0..3, so a vector is a B operand as it is.
//   Same equations as skd::k2a_bvp / kern_k2a_wave (reference: sktran_do_rte.cpp:1620-1790,1897-2294).
__global__ void __launch_bounds__(64, SKD_WPE_K2AM) kern_k2a_mfma8(Geo g, Bufs b, int s, int m) {
    constexpr int N = 8, M2 = 16;
    using R = skd::Rec<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, col = lane & 15, q = lane >> 4, jc = col & 7, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sP[M2 * M2];
    const double rho = (m == 0) ? b.albedo[w] : 0.0;
    mf4 dprev = {0, 0, 0, 0};
    double rprev[4] = {0, 0, 0, 0};
    int sing = 0;
    for (int p = 0; p < nlyr; ++p) {
        const bool last = (p == nlyr - 1);
        const double* rec = skd::rec_at<N>(b, nlyr, p, w);
        const double Ecol = rec[R::E + jc];  // exp(-k_j tau), written by K1
        // D_p and r_p, rows 4 r + q
        mf4 acc;
        double rv[4];
        double gsp = 0, gsm = 0, gsg = 0;
        if (last && m == 0) {
            for (int t = 0; t < N; ++t) {
                const double wm_ = g.wt[t] * g.mu[t];
                gsp += wm_ * rec[R::WP + t + N * jc];
                gsm += wm_ * rec[R::WM + t + N * jc];
                gsg += wm_ * rec[R::G + 2 * N + t];
            }
        }
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + q;
            double v, rr;
            if (row < N) {
                v = (col < N) ? -rec[R::WP + row + N * jc] : -rec[R::WM + row + N * jc] * Ecol;
                rr = rec[R::G + row];
                if (p > 0) rr -= skd::rec_at<N>(b, nlyr, p - 1, w)[R::G + 2 * N + row];
            } else {
                const int ii = row - N;
                v = (col < N) ? rec[R::WM + ii + N * jc] * Ecol : rec[R::WP + ii + N * jc];
                rr = -rec[R::G + 3 * N + ii];
                if (last) {
                    if (m == 0) {
                        v -= 2.0 * rho * ((col < N) ? gsp * Ecol : gsm);
                        rr += g.csz[s] * rho / skd::kPi * b.T[size_t(nlyr) * nw + w] + 2.0 * rho * gsg;
                    }
                } else {
                    rr += skd::rec_at<N>(b, nlyr, p + 1, w)[R::G + N + ii];
                }
            }
            acc[r] = v;
            rv[r] = rr;
        }
        if (p > 0) {
            const double* recm = skd::rec_at<N>(b, nlyr, p - 1, w);
            mf4 t1 = {0, 0, 0, 0};
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + q;
                double bop = 0.0;  // Sub_p[i = col][j]: [W+_{p-1} E_{p-1} | W-_{p-1}]
                if (col < N) bop = (j < N) ? recm[R::WP + col + N * j] * recm[R::E + j] : recm[R::WM + col + N * (j - N)];
                t1 = mfma16(dprev[kk], bop, t1);
            }
            for (int kk = 0; kk < 2; ++kk) {
                const int jq = 4 * kk + q;  // row of [W-_p | W+_p E_p]
                const double bop = (col < N) ? rec[R::WM + jq + N * jc] : rec[R::WP + jq + N * jc] * Ecol;
                acc = mfma16(t1[2 + kk], bop, acc);
            }
            mf4 tr = {0, 0, 0, 0};
            for (int kk = 0; kk < 4; ++kk) tr = mfma16(t1[kk], rprev[kk], tr);
            for (int r = 0; r < 4; ++r) rv[r] -= tr[r];
        }
        // Dinv'_p: in-place Gauss-Jordan with row pivoting, then the permutation to the natural layout through LDS
        unsigned long long permP, permI;
        sing |= gj16_inplace(acc, col, q, permP, permI);
        __syncthreads();
        gj16_scatter(acc, col, q, permP, permI, sP);
        __syncthreads();
        double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        for (int r = 0; r < 4; ++r) {
            dprev[r] = sP[(4 * r + q) * M2 + col];
            dg[64 * r + lane] = dprev[r];
        }
        double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        for (int r = 0; r < 4; ++r) {
            rprev[r] = rv[r];
            if (col == 0) xs[4 * r + q] = rv[r];
        }
    }
    // back substitution  x_p = Dinv'_p (r'_p - Sup_p x_{p+1})
    double xn[4] = {0, 0, 0, 0};
    for (int p = nlyr - 1; p >= 0; --p) {
        const double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        double v[4];
        for (int r = 0; r < 4; ++r) v[r] = (col == 0) ? xs[4 * r + q] : 0.0;
        for (int r = 0; r < 4; ++r) v[r] = __shfl(v[r], 16 * q);  // replicate over the 16 columns
        if (p < nlyr - 1) {
            const double* recn = skd::rec_at<N>(b, nlyr, p + 1, w);
            mf4 sx = {0, 0, 0, 0};
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + q;
                double aop = 0.0;  // rows 8..15: [W-_{p+1} | W+_{p+1} E_{p+1}]
                if (col >= N) {
                    const int ii = col - N;
                    aop = (j < N) ? recn[R::WM + ii + N * j] : recn[R::WP + ii + N * (j - N)] * recn[R::E + j - N];
                }
                sx = mfma16(aop, xn[kk], sx);
            }
            for (int r = 0; r < 4; ++r) v[r] += sx[r];
        }
        mf4 x = {0, 0, 0, 0};
        for (int kk = 0; kk < 4; ++kk) x = mfma16(dg[col * M2 + 4 * kk + q], v[kk], x);
        for (int r = 0; r < 4; ++r) {
            xn[r] = x[r];
            if (col == 0) xs[4 * r + q] = x[r];
        }
    }
    if (__any(sing) && lane == 0) atomicOr(&b.status[w], skd::ERR_SINGULAR_BVP);
}

// K2c for N = 8.  Same operators as skd::k2c_layer_ops; FP, SP and BP are stored as ready-made A-operand images
// (FP: k-step r at [FP + 64 r + lane], r < 4;  SP / BP: [SP + 64 r + lane], [BP + 64 r + lane], r < 2) for kern_k2b_mfma8.
//   FP^T = F^T Dinv  and  [SP | BP]^T = [SupT | SubT]^T Dinv  are two chains of four matrix-core products.
// K1LANE: the records come from kern_k1_lane (column blocks with the coefficient vectors of the particular solution; the derivative rows
// of G are formed here, along the loop that needs them); a template parameter, not a run-time branch
// FTILES > 0 (full-size launches, up to 32 lines of sight): the FORWARD sweep of the adjoint (kern_k2b_mfma8's first half) runs inside this
// kernel, layer by layer, with FP, SP and HD taken from the registers they are formed in -- they never go to HBM (-26 % of the adjoint
// stage's traffic); only what the backward sweep needs (BP, DI, UG, GD, IPB) is written, and kern_k2b_mfma8<TILES, true> does the backward
// sweep.  Every statement of the sweep is the one of kern_k2b_mfma8 (same contraction setting), so the two paths agree bit for bit.
// streamed once between the sweeps: optional non-temporal access (experiment switch SKD_EXP_NT)
#ifdef SKD_EXP_NT
#define SKD_NT_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define SKD_NT_LD(ptr) __builtin_nontemporal_load((ptr))
#else
#define SKD_NT_ST(ptr, val) (*(ptr) = (val))
#define SKD_NT_LD(ptr) (*(ptr))
#endif
// NGM > 0 (K1LANE, FTILES > 0 only): up to NGM scattering groups (b.ngroups): per group one more HD row (rows 8 .. of the forward sweep's second
// product), DI rows and GD (skd::OpsG) from the group's derivative block (skd::RecG, kern_k1g_cols8), which arrives by LDS-DMA like the record
template <bool K1LANE, int FTILES = 0, int NGM = 0> __global__ void __launch_bounds__(64, SKD_WPE_K2CM) kern_k2c_mfma8(Geo g, Bufs b, int s, int m, int layers_per_block) {
    constexpr int N = 8, NSTR = 16, M2 = 16, FT = (FTILES > 0) ? FTILES : 1;
    static_assert(NGM == 0 || (K1LANE && FTILES > 0), "scattering groups: lane-path records and the fused forward sweep");
    using RG = skd::RecG<N>;
    using OG = skd::OpsG<N>;
    constexpr int NGS = (NGM > 0) ? NGM : 1;
    const int ng = (NGM > 0) ? min(b.ngroups, NGM) : 0;
    m += blockIdx.z;   // (batched azimuth orders: skd::order_view)
    b = skd::order_view(b, blockIdx.z);
    using R = std::conditional_t<K1LANE, skd::RecC<N>, skd::Rec<N>>;       // kern_k1_lane writes column-block records
    // the record's image in LDS: column blocks 52 doubles apart (104 dwords = 40 mod 64: the eight solutions' entries fall on eight different
    // bank windows; 48 puts them on two)
    using RL = std::conditional_t<K1LANE, skd::RecC<N, 4>, skd::Rec<N>>;
    constexpr int NB = K1LANE ? 2 : 1;   // K1LANE: two record buffers, the next layer's record arrives by LDS-DMA while this one is worked on
    using O = skd::Ops<N>;
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, col = lane & 15, q = lane >> 4, jc = col & 7, nw = b.nw, nlyr = g.nlyr;
    __shared__ double sRecAll[NB * RL::SIZE];
    // per buffer: the BVP solution (16) and, K1LANE, a_l (16), omega, tau, T of the layer behind it
    __shared__ double sSmall[NB * 36], sTmp[M2], sIpb[N], sDIpb[4 * N];
    __shared__ double sDinv[K1LANE ? M2 * M2 : 1];   // K1LANE: block inverse
    __shared__ double sHD[4 * NSTR], sUG[M2];
    __shared__ __attribute__((aligned(16))) double sGrp[(NGM > 0) ? NB * NGS * RG::SIZE : 1];   // the groups' derivative blocks, two buffers like the record
    __shared__ double sHDG[NGS * NSTR], sDIpbG[NGS * N];
    // ---- state of the fused forward sweep (FTILES > 0), as in kern_k2b_mfma8
    const int nlos = g.nlos, ntile = (nlos + 15) / 16;
    double* zw = b.z + size_t(w) * nlyr * ntile * 256;
    double* hw = b.qlbuf + size_t(w) * nlyr * ntile * 64;
    const double* cwp = b.cwbuf + size_t(w) * g.n_lp;
    const uint8_t* mrow = b.qlmask + size_t(w) * ((g.n_lp + 7) / 8);
    const double* pcm = g.lp_pc + size_t(m) * g.n_lp * NSTR;
    int los[FT];
    bool ok[FT];
    double gw[FT];
    mf4 yprev[FT];
    if constexpr (FTILES > 0) {
        for (int t = 0; t < FT; ++t) {
            los[t] = t * 16 + col;
            ok[t] = t < ntile && los[t] < nlos;
            gw[t] = 0;
            yprev[t] = mf4{0, 0, 0, 0};
            if (ok[t] && m == 0 && g.ray_ground[los[t]]) {
                double acc = 0;
                for (int e = g.g_start[los[t]]; e < g.g_start[los[t] + 1]; ++e)
                    if (g.g_sza[e] == s && skd::conv_at(b, g, g.g_slot[e], w) > m) acc += g.g_w[e];
                gw[t] = acc * b.att_total[size_t(los[t]) * nw + w];
            }
        }
    }
    const double rho = (m == 0) ? b.albedo[w] : 0.0;
    // Legendre sums as a 16 x 16 product:  [YP; YM] = [W+^T W-^T; W-^T W+^T] [P sg; P],  P[r][l] = 1/2 w_r omega a_l P_l^m(mu_r);
    // this lane's B operand rows are r = q and r = q + 4 of column l = col (constant over the layers up to omega a_l)
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    const double pc0 = 0.5 * g.wt[q] * lpmu[q * NSTR + col], pc1 = 0.5 * g.wt[q + 4] * lpmu[(q + 4) * NSTR + col];
    const double sg = ((col - m) & 1) ? -1.0 : 1.0;
    const int p_begin = blockIdx.y * layers_per_block, p_end = min(nlyr, p_begin + layers_per_block);
    // a one-wavefront block needs no barrier between its LDS phases: the LDS serves a wavefront's requests in order, so waiting for them is
    // enough -- and, unlike __syncthreads(), it leaves the global loads and LDS-DMA requests in flight (the workgroup fence drains them)
#define K2C_SYNC() do { if constexpr (K1LANE) wave_lds_sync(); else __syncthreads(); } while (0)
#ifdef SKD_K2C_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define K2CT(i) { long long _t = clock64(); tacc[i] += _t - tlast; tlast = _t; }
#else
#define K2CT(i)
#endif
    // K1LANE: LDS-DMA requests for layer pn into buffer `buf`.  One request writes 1 KB of LDS, lane L at byte 16 L of it, from a per-lane
    // source address: the record's image (eight column blocks 416 bytes apart, then the tail) is 4.5 KB = five requests whose lanes
    // read the record at offsets worked out once (a lane that lands in the padding between two blocks re-reads its block's first bytes);
    // the block inverse (2 KB, ONE buffer: its B-operand registers are read before the request goes out) two; the BVP solution and a_l /
    // omega / tau one; T one.  (One request per column block and per small array was 15 requests and 428 instructions a layer.)
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    int roff[5];
    if constexpr (K1LANE) {
        constexpr int BLK = RL::CB * 8, DAT = R::CB * 8, NBLK = N * BLK;   // bytes: padded block, its data, all blocks
        static_assert(RL::SIZE * 8 > 4 * 1024 && RL::SIZE * 8 <= 5 * 1024 && RL::G == N * RL::CB, "five requests for the record's image");
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int at = 16 * (64 * u + lane);   // byte of the image
            const int j = at / BLK, o = at - j * BLK;
            roff[u] = (at < NBLK) ? j * DAT + ((o < DAT) ? o : 0) : R::G * 8 + (at - NBLK);
        }
    }
    [[maybe_unused]] auto request_layer = [&](int pn, int buf) {
        if constexpr (K1LANE) {
            const char* src = reinterpret_cast<const char*>(skd::rec_at_r<R>(b, nlyr, pn, w));
            double* dst = sRecAll + buf * RL::SIZE;
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_amdgcn_global_load_lds((gptr)(src + roff[u]), (lptr)(dst + 128 * u), 16, 0, 0);
            if (lane < (RL::SIZE * 8 - 4096) / 16) __builtin_amdgcn_global_load_lds((gptr)(src + roff[4]), (lptr)(dst + 512), 16, 0, 0);
            const char* dsrc = reinterpret_cast<const char*>(skd::dinv_at<N>(b, nlyr, pn, w));
            __builtin_amdgcn_global_load_lds((gptr)(dsrc + lane * 16), (lptr)(sDinv), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(dsrc + 1024 + lane * 16), (lptr)(sDinv + 128), 16, 0, 0);
            // solution (128 bytes, lanes 0 .. 7) and a_l / omega / tau (144 bytes, lanes 8 .. 16) side by side; T (8 bytes, two 4-byte lanes) behind
            const char* xsrc = reinterpret_cast<const char*>(skd::xsol_at<N>(b, nlyr, pn, w));
            const char* asrc = reinterpret_cast<const char*>(b.at + (size_t(w) * nlyr + pn) * 18);
            if (lane < 17) __builtin_amdgcn_global_load_lds((gptr)((lane < 8) ? xsrc + lane * 16 : asrc + (lane - 8) * 16), (lptr)(sSmall + buf * 36), 16, 0, 0);
            if (lane < 2)
                __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(b.T + size_t(pn) * nw + w) + lane * 4), (lptr)(sSmall + buf * 36 + 34), 4, 0, 0);
            if constexpr (NGM > 0) {
                static_assert(RG::SIZE * 8 > 1024 && RG::SIZE * 8 <= 2048 && (RG::SIZE * 8) % 16 == 0, "two requests per group block");
                for (int gi = 0; gi < ng; ++gi) {
                    const char* gsrc = reinterpret_cast<const char*>(skd::recg_at<N>(b, nlyr, pn, w, gi));
                    double* gdst = sGrp + (buf * NGS + gi) * RG::SIZE;
                    __builtin_amdgcn_global_load_lds((gptr)(gsrc + lane * 16), (lptr)(gdst), 16, 0, 0);
                    if (lane < (RG::SIZE * 8 - 1024) / 16) __builtin_amdgcn_global_load_lds((gptr)(gsrc + 1024 + lane * 16), (lptr)(gdst + 128), 16, 0, 0);
                }
            }
        }
    };
    if constexpr (K1LANE) request_layer(p_begin, 0);
    for (int p = p_begin; p < p_end; ++p) {
        const bool last = (p == nlyr - 1);
        const double* rec = skd::rec_at_r<R>(b, nlyr, p, w);
        const double* dg = skd::dinv_at<N>(b, nlyr, p, w);
        const double* xs = skd::xsol_at<N>(b, nlyr, p, w);
        double* o = skd::ops_at<N>(b, nlyr, p, w);
        double* sRec = sRecAll;
        double* sXs = sSmall;
        double dinv_b[4];  // B operand of the Dinv chains: Dinv[4 kk + (lane >> 4)][lane & 15] = dg[64 kk + lane]
        double om, th, T, al;
        if constexpr (K1LANE) {
            // everything of this layer was requested one layer ago (LDS-DMA: global_load_lds, no registers in between); the next layer's
            // requests go out now and land while this layer is worked on.  A wavefront is alone with its chain of dependent layers, three of
            // them per SIMD: without the look-ahead every layer starts with a round trip to HBM (24 % of the kernel's cycles, clock64 phases).
            const int cur = (p - p_begin) & 1;
            sRec = sRecAll + cur * RL::SIZE;
            sXs = sSmall + cur * 36;
            // everything older than the previous layer's last 5 FTILES stores (the sweep states and HD parts, issued after every load of that
            // layer: vector-memory operations complete in issue order) must be back -- this layer's LDS-DMA data among it; the stores stay in flight
            if (FTILES > 0 && p > p_begin) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * FTILES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int kk = 0; kk < 4; ++kk) dinv_b[kk] = sDinv[64 * kk + lane];
            const double* sLo = sSmall + cur * 36 + M2;
            om = sLo[16]; th = sLo[17]; T = sLo[18];
            al = (col >= m) ? sLo[col] : 0.0;
            if (p + 1 < p_end) {   // (the last layer's ground term reads the block inverse once more, below: nothing overwrites it)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                request_layer(p + 1, cur ^ 1);
            }
        } else {
            {
                // the record -> LDS through registers, all loads in flight at once (as a rolled loop every element waits for its own load:
                // measured 52 instead of 41 ms per step for the fused kernel)
                constexpr int RU = (R::SIZE + 63) / 64;
                double stage[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) stage[u] = (lane + 64 * u < R::SIZE) ? rec[lane + 64 * u] : 0.0;
#pragma unroll
                for (int u = 0; u < RU; ++u) if (lane + 64 * u < R::SIZE) sRec[lane + 64 * u] = stage[u];
            }
            if (lane < M2) sXs[lane] = xs[lane];
            for (int kk = 0; kk < 4; ++kk) dinv_b[kk] = dg[64 * kk + lane];
            const double* lo = b.at + (size_t(w) * nlyr + p) * 18;
            om = lo[16]; th = lo[17]; T = b.T[size_t(p) * nw + w];
            al = (col >= m) ? lo[col] : 0.0;
            __syncthreads();
        }
        const double inv_om = 1.0 / om, inv_T = (T != 0) ? 1.0 / T : 0.0;
        K2CT(0)
        // ---- Y = [YP; YM] and its omega derivative DY (rows: solution j / 8 + j, columns: Legendre order l)
        mf4 Y = {0, 0, 0, 0}, DY = {0, 0, 0, 0};
        [[maybe_unused]] mf4 DYG[NGS];
        {
            const bool hi = (col >= N);
            const double wp0 = sRec[RL::wp(q, jc)], wp1 = sRec[RL::wp(4 + q, jc)];
            const double wm0 = sRec[RL::wm(q, jc)], wm1 = sRec[RL::wm(4 + q, jc)];
            const double dp0 = sRec[RL::dwp(q, jc)] + wp0 * inv_om, dp1 = sRec[RL::dwp(4 + q, jc)] + wp1 * inv_om;
            const double dm0 = sRec[RL::dwm(q, jc)] + wm0 * inv_om, dm1 = sRec[RL::dwm(4 + q, jc)] + wm1 * inv_om;
            const double c0 = pc0 * om * al, c1 = pc1 * om * al;
            const double bP[4] = {sg * c0, sg * c1, c0, c1};
            const double aW[4] = {hi ? wm0 : wp0, hi ? wm1 : wp1, hi ? wp0 : wm0, hi ? wp1 : wm1};
            const double aD[4] = {hi ? dm0 : dp0, hi ? dm1 : dp1, hi ? dp0 : dm0, hi ? dp1 : dm1};
            for (int kk = 0; kk < 4; ++kk) {
                Y = mfma16(aW[kk], bP[kk], Y);
                DY = mfma16(aD[kk], bP[kk], DY);
            }
            // group direction: d[YP; YM] = [dW] P(a) + [W] P(da)
            if constexpr (NGM > 0)
                for (int gi = 0; gi < ng; ++gi) {
                    const double* sG = sGrp + ((((p - p_begin) & 1) * NGS) + gi) * RG::SIZE;
                    const double alg = (col >= m) ? sG[RG::DA + col] : 0.0;
                    const double g0 = pc0 * om * alg, g1 = pc1 * om * alg;
                    const double bG[4] = {sg * g0, sg * g1, g0, g1};
                    const double gp0 = sG[RG::dwp(q, jc)], gp1 = sG[RG::dwp(4 + q, jc)], gm0 = sG[RG::dwm(q, jc)], gm1 = sG[RG::dwm(4 + q, jc)];
                    const double aG[4] = {hi ? gm0 : gp0, hi ? gm1 : gp1, hi ? gp0 : gm0, hi ? gp1 : gm1};
                    mf4 t = {0, 0, 0, 0};
                    for (int kk = 0; kk < 4; ++kk) {
                        t = mfma16(aG[kk], bP[kk], t);
                        t = mfma16(aW[kk], bG[kk], t);
                    }
                    DYG[gi] = t;
                }
        }
        // ---- F = [UL; UM] (C/D image, in place of Y) and the direct-dependence rows HD: this lane holds solutions j = q, q + 4
        double hd[4] = {0, 0, 0, 0}, val = 0, hdg[NGS];
        for (int gi = 0; gi < NGS; ++gi) hdg[gi] = 0;
        mf4 Fr;
        for (int jx = 0; jx < 2; ++jx) {
            const int j = q + 4 * jx;
            skd::PPMult pm;
            skd::ppm_load(&sRec[RL::ppm(j)], pm);
            const double yp = Y[jx], ym = Y[2 + jx], dyp = DY[jx], dym = DY[2 + jx];
            Fr[jx] = yp * pm.hp * inv_om;
            Fr[2 + jx] = ym * pm.hm * inv_om;
            const double L = sXs[j], Mv = sXs[N + j], dk = sRec[RL::dk(j)];
            const double Ap = sRec[RL::ap(j)], Am = sRec[RL::am(j)], dAp = sRec[RL::dap(j)], dAm = sRec[RL::dam(j)];
            const double tL = pm.hp * L + Ap * pm.Dm, tM = pm.hm * Mv + Am * pm.Dp;
            val += yp * tL + ym * tM;
            hd[skd::TH_TAU] += yp * (pm.hp_tau * L + Ap * pm.Dm_tau) + ym * (pm.hm_tau * Mv + Am * pm.Dp_tau);
            hd[skd::TH_SEC] += yp * Ap * pm.Dm_sec + ym * Am * pm.Dp_sec;
            hd[skd::TH_T] += (yp * Ap * pm.Dm + ym * Am * pm.Dp) * inv_T;
            hd[skd::TH_OM] += dyp * tL + dym * tM + yp * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) +
                              ym * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
            if constexpr (NGM > 0)
                for (int gi = 0; gi < ng; ++gi) {
                    const double* sG = sGrp + ((((p - p_begin) & 1) * NGS) + gi) * RG::SIZE;
                    const double dkg = sG[RG::at(RG::DK, j)], dApg = sG[RG::at(RG::DAP, j)], dAmg = sG[RG::at(RG::DAM, j)];
                    hdg[gi] += DYG[gi][jx] * tL + DYG[gi][2 + jx] * tM + yp * (pm.hp_k * dkg * L + dApg * pm.Dm + Ap * pm.Dm_k * dkg) +
                               ym * (pm.hm_k * dkg * Mv + dAmg * pm.Dp + Am * pm.Dp_k * dkg);
                }
        }
        for (int off = 16; off <= 32; off <<= 1) {
            for (int t = 0; t < 4; ++t) hd[t] += __shfl_xor(hd[t], off);
            val += __shfl_xor(val, off);
            if constexpr (NGM > 0)
                for (int gi = 0; gi < ng; ++gi) hdg[gi] += __shfl_xor(hdg[gi], off);
        }
        if constexpr (NGM > 0)
            if (q == 0)
                for (int gi = 0; gi < ng; ++gi) sHDG[gi * NSTR + col] = hdg[gi] * inv_om;
        if (q == 0)
            for (int t = 0; t < 4; ++t) {
                const double hv = hd[t] * inv_om - ((t == skd::TH_OM) ? val * inv_om * inv_om : 0.0);
                if constexpr (FTILES > 0) sHD[t * NSTR + col] = hv;
                else o[O::HD + t * NSTR + col] = hv;
            }
        K2CT(1)
        // ---- FP^T = F^T Dinv (the C/D image of F is the A operand of F^T) and [SP | BP]^T = [SupT | SubT]^T Dinv
        mf4 x1, x2;
        {
            const bool sup = (col < N);
            const double wm0 = sRec[RL::wm(jc, q)], wm1 = sRec[RL::wm(jc, q + 4)];
            const double we0 = sRec[RL::wp(jc, q)] * sRec[RL::e(q)], we1 = sRec[RL::wp(jc, q + 4)] * sRec[RL::e(q + 4)];
            // [SupT | SubT][jj][c']: SupT[j][c] = W-(c,j), SupT[N+j][c] = W+(c,j) E_j ; SubT[j][c] = W+(c,j) E_j, SubT[N+j][c] = W-(c,j)
            const double aS[4] = {sup ? wm0 : we0, sup ? wm1 : we1, sup ? we0 : wm0, sup ? we1 : wm1};
            x1 = mf4{0, 0, 0, 0};
            x2 = mf4{0, 0, 0, 0};
            for (int kk = 0; kk < 4; ++kk) {
                x1 = mfma16(Fr[kk], dinv_b[kk], x1);   // (F^T Dinv)[l][i] = FP[i][l]
                x2 = mfma16(aS[kk], dinv_b[kk], x2);   // rows 0..7: SP^T, rows 8..15: BP^T
            }
            if constexpr (FTILES == 0) {
                for (int r = 0; r < 4; ++r) o[O::FP + 64 * r + lane] = x1[r];
                o[O::SP + lane] = x2[0];
                o[O::SP + 64 + lane] = x2[1];
            }
            o[O::BP + lane] = x2[2];
            o[O::BP + 64 + lane] = x2[3];
        }
        K2CT(2)
        // ---- edge-intensity derivatives at fixed (L, M).  All 64 lanes: (stream i, edge e, half h of the solutions), one branch-free form
        // for the four edges -- with X the edge's "own" W (e = 0, 3: W-, else W+), Y the other one, p = (e < 2 ? M : L), o = the other amplitude:
        //   d/dtau += X dE/dtau p,   d/domega += dX E p + X dE/domega p + dY o,   I+bottom (e = 2) = G+bottom + X E p + Y o
        // (the four branches of the per-edge form ran one after the other on a half-empty wavefront: 28 % of this kernel's cycles)
        {
            const int i = lane & 7, e = (lane >> 3) & 3, h = lane >> 5;
            const bool own_minus = (e == 0 || e == 3), pm = (e < 2);
            // K1LANE (records of kern_k1_lane): the derivative rows of G are not in the record; edge e of stream i is
            //   DG[theta][e][i] = sum_j X(i, j) coef_type(e)[1 + theta][j] + (theta = omega) sum_j dX(i, j) coef_type(e)[0][j],  type = A- C- (e < 2), A+ C+
            const int tk0 = (e >= 2) ? 5 : 0;
            double d_tau = 0, d_om = 0, ipb_i = 0, dg[4] = {0, 0, 0, 0}, dgg[NGS];
            for (int gi = 0; gi < NGS; ++gi) dgg[gi] = 0;
            [[maybe_unused]] const double* sG0 = sGrp + (((p - p_begin) & 1) * NGS) * RG::SIZE;
#pragma unroll 2
            for (int jj = 0; jj < N / 2; ++jj) {
                const int j = h * (N / 2) + jj;
                const double X = sRec[own_minus ? RL::wm(i, j) : RL::wp(i, j)], Y = sRec[own_minus ? RL::wp(i, j) : RL::wm(i, j)];
                const double dX = sRec[own_minus ? RL::dwm(i, j) : RL::dwp(i, j)], dY = sRec[own_minus ? RL::dwp(i, j) : RL::dwm(i, j)];
                const double E = sRec[RL::e(j)], L = sXs[j], Mv = sXs[N + j];
                const double pa = pm ? Mv : L, oa = pm ? L : Mv;
                const double t_tau = -sRec[RL::k(j)] * E * pa, t_e = E * pa, t_om = -sRec[RL::dk(j)] * th * E * pa;
                d_tau += X * t_tau;
                d_om += dX * t_e + X * t_om + dY * oa;
                ipb_i += X * t_e + Y * oa;
                if constexpr (K1LANE) {
                    const double c0 = sRec[RL::coef(tk0, j)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) dg[t] += X * sRec[RL::coef(tk0 + 1 + t, j)];
                    dg[skd::TH_OM] += dX * c0;
                    // group rows: the omega row's form with the group's dW, dk and coefficient derivatives
                    if constexpr (NGM > 0)
#pragma unroll
                        for (int gi = 0; gi < NGM; ++gi) {
                            if (gi >= ng) break;
                            const double* sG = sG0 + gi * RG::SIZE;
                            const double gX = sG[own_minus ? RG::dwm(i, j) : RG::dwp(i, j)], gY = sG[own_minus ? RG::dwp(i, j) : RG::dwm(i, j)];
                            dgg[gi] += gX * t_e + X * (-sG[RG::at(RG::DK, j)] * th * t_e) + gY * oa + X * sG[RG::at((e >= 2) ? RG::CGP : RG::CGM, j)] + gX * c0;
                        }
                }
            }
            d_tau += __shfl_xor(d_tau, 32);
            d_om += __shfl_xor(d_om, 32);
            ipb_i += __shfl_xor(ipb_i, 32);
            if constexpr (K1LANE) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dg[t] += __shfl_xor(dg[t], 32);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) dg[t] = sRec[R::DG + (t * 4 + e) * N + i];
            }
            const double v_tau = dg[skd::TH_TAU] + d_tau, v_om = dg[skd::TH_OM] + d_om, v_T = dg[skd::TH_T], v_sec = dg[skd::TH_SEC];
            if (h == 0) {
                o[O::DI + (skd::TH_TAU * 4 + e) * N + i] = v_tau;
                o[O::DI + (skd::TH_OM * 4 + e) * N + i] = v_om;
                o[O::DI + (skd::TH_T * 4 + e) * N + i] = v_T;
                o[O::DI + (skd::TH_SEC * 4 + e) * N + i] = v_sec;
                if (e == 2) {
                    const double wmu = g.wt[i] * g.mu[i];
                    sIpb[i] = wmu * (sRec[RL::g(2 * N + i)] + ipb_i);
                    sDIpb[skd::TH_TAU * N + i] = wmu * v_tau;
                    sDIpb[skd::TH_OM * N + i] = wmu * v_om;
                    sDIpb[skd::TH_T * N + i] = wmu * v_T;
                    sDIpb[skd::TH_SEC * N + i] = wmu * v_sec;
                }
            }
            if constexpr (NGM > 0)
                for (int gi = 0; gi < ng; ++gi) {
                    double acc = dgg[gi];
                    acc += __shfl_xor(acc, 32);
                    if (h == 0) {
                        skd::opsg_at<N>(b, nlyr, p, w, gi)[OG::DI + e * N + i] = acc;
                        if (e == 2) sDIpbG[gi * N + i] = g.wt[i] * g.mu[i] * acc;
                    }
                }
        }
        K2CT(3)
        // ground part of u
        if (lane < N) {
            double sp = 0, sm = 0;
            if (last && m == 0)
                for (int i = 0; i < N; ++i) {
                    sp += g.mu[i] * g.wt[i] * sRec[RL::wp(i, lane)];
                    sm += g.mu[i] * g.wt[i] * sRec[RL::wm(i, lane)];
                }
            sTmp[lane] = 2.0 * rho * sp * sRec[RL::e(lane)];
            sTmp[N + lane] = 2.0 * rho * sm;
        }
        K2C_SYNC();
        // UG, GD, IPB
        if (lane < M2) {
            double t = 0;
            if (last && m == 0)
                for (int j = 0; j < M2; ++j) t += (K1LANE ? sDinv[j * M2 + lane] : dg[j * M2 + lane]) * sTmp[j];
            o[O::UG + lane] = t;
            if constexpr (FTILES > 0) sUG[lane] = t;
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
        if constexpr (NGM > 0)
            if (lane >= 8 && lane < 8 + ng) {
                double sacc = 0;
                if (last && m == 0)
                    for (int i = 0; i < N; ++i) sacc += sDIpbG[(lane - 8) * N + i];
                skd::opsg_at<N>(b, nlyr, p, w, lane - 8)[OG::GD] = 2.0 * rho * sacc;
            }
        K2C_SYNC();
        K2CT(4)
        if constexpr (FTILES > 0) {
            // ---- forward sweep of the adjoint at this layer:  Y_p = FP QL_p + SP Y_{p-1}[8:16]  (+ HD QL_p), statement for statement the
            // forward half of kern_k2b_mfma8 with its operands taken from x1 (FP), x2[0:2] (SP), sHD and sUG
#pragma clang fp contract(off)
            double hda[4];
            for (int kk = 0; kk < 4; ++kk) hda[kk] = (col < 4) ? sHD[col * NSTR + 4 * kk + q] : 0.0;
            if constexpr (NGM > 0)
                if (col >= 8 && col < 8 + ng)
                    for (int kk = 0; kk < 4; ++kk) hda[kk] = sHDG[(col - 8) * NSTR + 4 * kk + q];
            for (int t = 0; t < FT; ++t) {
                if (t >= ntile) continue;
                double ql4[4] = {0, 0, 0, 0};
                if (ok[t]) {
                    const int base = (los[t] * g.nsza + s) * nlyr + p;
                    const int e0 = g.lp_start[base], e1 = g.lp_start[base + 1];
                    for (int e = e0; e < e1; e += 4) {
                        const unsigned mk = unsigned(mrow[e >> 3]) >> (e & 4);
                        const double2* cw2 = reinterpret_cast<const double2*>(cwp + e);
                        const double2* pc2 = reinterpret_cast<const double2*>(pcm + (size_t(e) * 4 + q) * 4);
                        const double2 cv0 = cw2[0], cv1 = cw2[1];
                        double2 pa[4], pb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            pa[u] = pc2[8 * u];
                            pb[u] = pc2[8 * u + 1];
                        }
                        const double c[4] = {(mk & 1u) ? cv0.x : 0.0, (mk & 2u) ? cv0.y : 0.0, (mk & 4u) ? cv1.x : 0.0, (mk & 8u) ? cv1.y : 0.0};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ql4[0] += c[u] * pa[u].x;
                            ql4[1] += c[u] * pa[u].y;
                            ql4[2] += c[u] * pb[u].x;
                            ql4[3] += c[u] * pb[u].y;
                        }
                    }
                }
                mf4 acc = {0, 0, 0, 0}, hacc = {0, 0, 0, 0};
                for (int kk = 0; kk < 4; ++kk) acc = mfma16(x1[kk], ql4[kk], acc);
                for (int kk = 0; kk < 2; ++kk) acc = mfma16(x2[kk], yprev[t][2 + kk], acc);
                for (int kk = 0; kk < 4; ++kk) hacc = mfma16(hda[kk], ql4[kk], hacc);
                if (last && m == 0)
                    for (int r = 0; r < 4; ++r) acc[r] += sUG[q + 4 * r] * gw[t];
                double* zt = zw + (size_t(p) * ntile + t) * 256;
                for (int r = 0; r < 4; ++r) SKD_NT_ST(&zt[64 * r + lane], acc[r]);
                SKD_NT_ST(&hw[(size_t(p) * ntile + t) * 64 + lane], hacc[0]);
                if constexpr (NGM > 0)
                    if (q < ng) SKD_NT_ST(&b.hg[((size_t(w) * nlyr + p) * ntile + t) * 64 + lane], hacc[2]);   // (rows 8 .. 8 + ng - 1 only: 128 bytes per group)
                yprev[t] = acc;
            }
        }
        K2CT(5)
    }
#ifdef SKD_K2C_TIMING
    if (lane == 0 && w == 100 && m == 1) printf("K2CT stage %lld  Y/F/HD %lld  FP/SP/BP %lld  edges %lld  ground/UG %lld  forward sweep %lld\n", tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5]);
#endif
}

// K2b for N = 8: one wavefront per wavelength, 16 lines of sight per matrix-core tile (lane & 15 = line of sight).
//   forward   Y_p = FP QL_p + SP Y_{p-1}[8:16]            (6 products / tile;  + HD QL_p, 4 products, reduced-gradient part)
//   backward  Z_p = Y_p - BP Z_{p+1}[0:8]                  (2 products / tile)
//   gradient  G_p = DI_p . [zA_p, zB_{p-1}, -zA_{p+1}, -zB_p] + HD QL_p   (8 products / tile, rows 0..3 = tau, omega, T, sec;
//             rows 4..7 carry sum_i zB_i for the ground terms)
//   Y / Z tiles and the HD part are kept in HBM between the sweeps in their C/D register images
//   (b.z: [w][p][tile][4][64], b.qlbuf: [w][p][tile][64]); nothing goes through LDS.
#ifndef SKD_WPE_K2BG
#define SKD_WPE_K2BG 4   // (146 registers with the groups' rows, 20 spilled at four wavefronts per SIMD -- and still faster than three: 21.8 / 24.7 ms)
#endif
// GROUPS (BWD_ONLY): rows 8 .. 8 + b.ngroups - 1 of the gradient tile are the scattering groups' rows (skd::OpsG, the forward sweep's b.hg)
template <int TILES, bool BWD_ONLY = false, bool GROUPS = false> __global__ void __launch_bounds__(64, GROUPS ? SKD_WPE_K2BG : SKD_WPE_K2BM) kern_k2b_mfma8(Geo g, Bufs b, int s, int m) {
    // the one- and two-tile instantiations must round identically (results do not depend on the launch shape): no implicit
    // contraction of a * b + c in this kernel, fused multiply-adds are written out
#pragma clang fp contract(off)
    static_assert(!GROUPS || BWD_ONLY, "group rows: the forward sweep runs in kern_k2c_mfma8");
    using OG = skd::OpsG<8>;
    const int ng = GROUPS ? min(b.ngroups, 4) : 0;
    constexpr int N = 8, NSTR = 16;
    using O = skd::Ops<N>;
    m += blockIdx.z;   // (batched azimuth orders: skd::order_view; every order then adds into its own gradient rows, kern_grad_orders sums them)
    b = skd::order_view(b, blockIdx.z);
    const int w = blockIdx.x;
    if (m >= b.norders[w]) return;
    const int lane = threadIdx.x, col = lane & 15, q = lane >> 4, nw = b.nw, nlyr = g.nlyr, nlos = g.nlos;
    const size_t nt = size_t(nw) * nlos;
    const int ntile = (nlos + 15) / 16;
    double* zw = b.z + size_t(w) * nlyr * ntile * 256;
    double* hw = b.qlbuf + size_t(w) * nlyr * ntile * 64;
    const double* cwp = b.cwbuf + size_t(w) * g.n_lp;
    const uint8_t* mrow = b.qlmask + size_t(w) * ((g.n_lp + 7) / 8);
    const double* pcm = g.lp_pc + size_t(m) * g.n_lp * NSTR;
    const bool last_is_ground = (m == 0);
    for (int tile0 = blockIdx.y * TILES; tile0 < ntile; tile0 += gridDim.y * TILES) {
        int los[TILES];
        bool ok[TILES];
        double gw[TILES];
        for (int t = 0; t < TILES; ++t) {
            los[t] = (tile0 + t) * 16 + col;
            ok[t] = (tile0 + t) < ntile && los[t] < nlos;
            gw[t] = 0;
            if (ok[t] && m == 0 && g.ray_ground[los[t]]) {
                double acc = 0;
                for (int e = g.g_start[los[t]]; e < g.g_start[los[t] + 1]; ++e)
                    if (g.g_sza[e] == s && skd::conv_at(b, g, g.g_slot[e], w) > m) acc += g.g_w[e];
                gw[t] = acc * b.att_total[size_t(los[t]) * nw + w];
            }
        }
        // ---------------- forward sweep (BWD_ONLY: done by kern_k2c_mfma8<.., TILES>, the Y tiles and the HD part are in HBM already)
        if constexpr (!BWD_ONLY) {
        mf4 yprev[TILES];
        for (int t = 0; t < TILES; ++t) yprev[t] = mf4{0, 0, 0, 0};
        // A operands of the layer (FP, SP, HD); the next layer's are requested before this layer's work starts
        double fpa[4], spa[2], hda[4], nfpa[4], nspa[2], nhda[4];
        auto load_fwd = [&](int p, double* f, double* sp_, double* h) {
            const double* ops = skd::ops_at<N>(b, nlyr, p, w);
            for (int kk = 0; kk < 4; ++kk) f[kk] = ops[O::FP + 64 * kk + lane];
            for (int kk = 0; kk < 2; ++kk) sp_[kk] = ops[O::SP + 64 * kk + lane];
            for (int kk = 0; kk < 4; ++kk) h[kk] = (col < 4) ? ops[O::HD + col * NSTR + 4 * kk + q] : 0.0;
        };
        load_fwd(0, fpa, spa, hda);
        for (int p = 0; p < nlyr; ++p) {
            const double* ops = skd::ops_at<N>(b, nlyr, p, w);
            if (p + 1 < nlyr) load_fwd(p + 1, nfpa, nspa, nhda);
            const bool last = (p == nlyr - 1);
            for (int t = 0; t < TILES; ++t) {
                if ((tile0 + t) >= ntile) continue;
                // Legendre-space LOS weights: this lane holds ql[4 kk + q] of its line of sight.  Four table entries per
                // step, branch free, so that the dependent loads of a step are in flight together.
                double ql4[4] = {0, 0, 0, 0};
#ifdef SKD_EXP_NOQL
                if (false) {
#else
                if (ok[t]) {
#endif
                    // groups are padded to multiples of 4 entries (zero weights): per 4 entries two 16-byte loads of weights, eight of the
                    // cos(m phi) P_l^m table (this lane's 32 bytes per entry) and the live flags of the 4 entries in half a byte
                    const int base = (los[t] * g.nsza + s) * nlyr + p;
                    const int e0 = g.lp_start[base], e1 = g.lp_start[base + 1];
                    for (int e = e0; e < e1; e += 4) {
                        const unsigned mk = unsigned(mrow[e >> 3]) >> (e & 4);
                        const double2* cw2 = reinterpret_cast<const double2*>(cwp + e);
                        const double2* pc2 = reinterpret_cast<const double2*>(pcm + (size_t(e) * 4 + q) * 4);
                        const double2 cv0 = cw2[0], cv1 = cw2[1];
                        double2 pa[4], pb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            pa[u] = pc2[8 * u];
                            pb[u] = pc2[8 * u + 1];
                        }
                        const double c[4] = {(mk & 1u) ? cv0.x : 0.0, (mk & 2u) ? cv0.y : 0.0, (mk & 4u) ? cv1.x : 0.0, (mk & 8u) ? cv1.y : 0.0};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ql4[0] += c[u] * pa[u].x;
                            ql4[1] += c[u] * pa[u].y;
                            ql4[2] += c[u] * pb[u].x;
                            ql4[3] += c[u] * pb[u].y;
                        }
                    }
                }
                mf4 acc = {0, 0, 0, 0}, hacc = {0, 0, 0, 0};
                for (int kk = 0; kk < 4; ++kk) acc = mfma16(fpa[kk], ql4[kk], acc);
                for (int kk = 0; kk < 2; ++kk) acc = mfma16(spa[kk], yprev[t][2 + kk], acc);
                for (int kk = 0; kk < 4; ++kk) hacc = mfma16(hda[kk], ql4[kk], hacc);
                if (last && last_is_ground)
                    for (int r = 0; r < 4; ++r) acc[r] += ops[O::UG + q + 4 * r] * gw[t];
                double* zt = zw + (size_t(p) * ntile + tile0 + t) * 256;
                for (int r = 0; r < 4; ++r) SKD_NT_ST(&zt[64 * r + lane], acc[r]);
                SKD_NT_ST(&hw[(size_t(p) * ntile + tile0 + t) * 64 + lane], hacc[0]);
                yprev[t] = acc;
            }
            for (int kk = 0; kk < 4; ++kk) { fpa[kk] = nfpa[kk]; hda[kk] = nhda[kk]; }
            for (int kk = 0; kk < 2; ++kk) spa[kk] = nspa[kk];
        }
        }
        // ---------------- backward sweep with the reduced gradient
        mf4 znext[TILES], gprev[TILES];
        double e1prev[2] = {0, 0};  // A operand of the zB_{p-1} term of the layer below (DI edge 1), k-steps 0,1
        for (int t = 0; t < TILES; ++t) { znext[t] = mf4{0, 0, 0, 0}; gprev[t] = mf4{0, 0, 0, 0}; }
        for (int p = nlyr - 1; p >= 0; --p) {
            const double* ops = skd::ops_at<N>(b, nlyr, p, w);
            const bool last = (p == nlyr - 1);
            double bpa[2], e0[2], e1[2], e2[2], e3[2];
            for (int kk = 0; kk < 2; ++kk) {
                bpa[kk] = -ops[O::BP + 64 * kk + lane];
                const int i = 4 * kk + q;
                e0[kk] = (col < 4) ? ops[O::DI + (col * 4 + 0) * N + i] : 0.0;
                e1[kk] = (col < 4) ? ops[O::DI + (col * 4 + 1) * N + i] : 0.0;
                e2[kk] = (col < 4) ? -ops[O::DI + (col * 4 + 2) * N + i] : 0.0;
                e3[kk] = (col < 4) ? -ops[O::DI + (col * 4 + 3) * N + i] : ((col < 8 && last && last_is_ground) ? 1.0 : 0.0);
                if constexpr (GROUPS) {   // (every lane loads, from a valid group's rows; lanes of the groups' rows keep the values)
                    const bool isg = col >= 8 && col < 8 + ng;
                    const double* og = skd::opsg_at<N>(b, nlyr, p, w, isg ? col - 8 : 0) + OG::DI + i;
                    const double g0 = og[0], g1 = og[N], g2 = og[2 * N], g3 = og[3 * N];
                    e0[kk] = isg ? g0 : e0[kk];
                    e1[kk] = isg ? g1 : e1[kk];
                    e2[kk] = isg ? -g2 : e2[kk];
                    e3[kk] = isg ? -g3 : e3[kk];
                }
            }
            const double gd = (last && last_is_ground && q < 4) ? ops[O::GD + q] : 0.0;
            double gdg = 0;
            if constexpr (GROUPS)
                if (last && last_is_ground && q < ng) gdg = skd::opsg_at<N>(b, nlyr, p, w, q)[OG::GD];
            const double ipb = ops[O::IPB];
            for (int t = 0; t < TILES; ++t) {
                if ((tile0 + t) >= ntile) continue;
                const double* zt = zw + (size_t(p) * ntile + tile0 + t) * 256;
                mf4 z;
                for (int r = 0; r < 4; ++r) z[r] = SKD_NT_LD(&zt[64 * r + lane]);
                if (!last)
                    for (int kk = 0; kk < 2; ++kk) z = mfma16(bpa[kk], znext[t][kk], z);
                // finish the layer below (p + 1) with zB_p and write it out
                if (!last) {
                    mf4 gp = gprev[t];
                    for (int kk = 0; kk < 2; ++kk) gp = mfma16(e1prev[kk], z[2 + kk], gp);
                    if (ok[t]) b.grad[size_t(q * nlyr + p + 1) * nt + size_t(w) * nlos + los[t]] += gp[0];
#ifndef SKD_EXP_NOGRMW
                    if constexpr (GROUPS)
                        if (ok[t] && q < ng) b.grad[size_t(4 * nlyr + 2 + q * nlyr + p + 1) * nt + size_t(w) * nlos + los[t]] += gp[2];
#endif
                }
                // this layer: HD part from the forward sweep, zA_p, zB_p, zA_{p+1}
                mf4 ga = {SKD_NT_LD(&hw[(size_t(p) * ntile + tile0 + t) * 64 + lane]), 0, 0, 0};
                if constexpr (GROUPS)
                    if (q < ng) ga[2] = SKD_NT_LD(&b.hg[((size_t(w) * nlyr + p) * ntile + tile0 + t) * 64 + lane]);
                for (int kk = 0; kk < 2; ++kk) ga = mfma16(e0[kk], z[kk], ga);
                for (int kk = 0; kk < 2; ++kk) ga = mfma16(e3[kk], z[2 + kk], ga);
                if (!last)
                    for (int kk = 0; kk < 2; ++kk) ga = mfma16(e2[kk], znext[t][kk], ga);
                if (last && last_is_ground && ok[t]) {
                    const double zBsum = ga[1];  // rows 4..7 of the tile
                    ga[0] += (zBsum + gw[t]) * gd;
                    if constexpr (GROUPS) ga[2] += (zBsum + gw[t]) * gdg;
                    if (q == 0) {
                        const size_t tid = size_t(w) * nlos + los[t];
                        const double rho = b.albedo[w];
                        const double tf = g.csz[s] / skd::kPi * b.T[size_t(nlyr) * nw + w];
                        b.grad[size_t(4 * nlyr) * nt + tid] += gw[t] * 2.0 * ipb + zBsum * (2.0 * ipb + tf);
                        b.grad[size_t(4 * nlyr + 1) * nt + tid] += zBsum * g.csz[s] / skd::kPi * rho;
                    }
                }
                gprev[t] = ga;
                znext[t] = z;
            }
            for (int kk = 0; kk < 2; ++kk) e1prev[kk] = e1[kk];
        }
        for (int t = 0; t < TILES; ++t) {
            if (ok[t]) b.grad[size_t(q * nlyr) * nt + size_t(w) * nlos + los[t]] += gprev[t][0];
            if constexpr (GROUPS)
                if (ok[t] && q < ng) b.grad[size_t(4 * nlyr + 2 + q * nlyr) * nt + size_t(w) * nlos + los[t]] += gprev[t][2];
        }
    }
}

}  // namespace skdev
