// Lane-per-problem HIP kernels (gfx950): one LANE owns one small dense problem and keeps it entirely in registers, every
// index a compile-time constant (fully unrolled), so the instruction stream is almost only FP64 fused multiply-adds with no
// cross-lane traffic, no LDS round trips and no barriers.  64 problems per wavefront instead of 4: the cooperative form of
// kern_k1_wave spends 57 % of its time in the Jacobi rounds (two barriers and an LDS exchange per round for 16 lanes).
//
// HIP only (included by disco_wave_kernels.hpp).  Same mathematics as skd::k1_layer_solution (disco_core.hpp), which stays the
// CPU-emulated specification; parity is asserted on the GPU against the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "disco_core.hpp"

// scheduling fence between the steps of the unrolled factorisations: without it the scheduler hoists the cross-lane moves of
// all later steps to the front of the (very long) basic block and spills hundreds of registers
#ifndef SKD_QR_NOFENCE
#define SKD_QR_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define SKD_QR_FENCE
#endif
#ifndef SKD_WPE_KJ
#define SKD_WPE_KJ 1
#endif
#ifndef SKD_KJ_MAXSWEEP
#define SKD_KJ_MAXSWEEP 60
#endif
#ifndef SKD_KJ_TOL
#define SKD_KJ_TOL 1e-30
#endif

namespace skdev {

using skd::Bufs;
using skd::Geo;

// 1 / sqrt(x) and 1 / x in full double precision: hardware estimate + two Newton steps
__device__ __forceinline__ double kj_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}
__device__ __forceinline__ double kj_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

// Eigen stage of one (layer, wavelength) problem in this lane's registers: B+- from the Legendre coefficients a'[k] = a_{m+k} (sPP, sIm: the
// block's LDS tables), Cholesky B+ = R R^T, A = R^T B- R, cyclic Jacobi.  Out: Rc (the factor R, lower), rinv (1 / R_ii), V (the eigenvectors of
// A, NOT yet back-substituted: R^T Xhat), lam (k_j^2, made positive); store_bm(pair, value) receives the entries of B- (lower triangle).
// Returns the error bits.  Shared by kern_kj_lane (eigen-solutions for the cooperative K1) and kern_k1_lane (whole layer solution per lane).
template <int N, class BMSTORE>
__device__ __forceinline__ int kj_eigen_stage(const double (&a)[2 * N], double om, bool live, const double* sPP, const double* sIm, double (&Rc)[N][N],
                                              double (&rinv)[N], double (&V)[N][N], double (&lam)[N], BMSTORE&& store_bm) {
    constexpr int NSTR = 2 * N, NM1 = N - 1, JH = N / 2;
    int err = 0;
    // ---- B+ (lower triangle, becomes the Cholesky factor) and B- (lower triangle)
    double Bm[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const double* pp = &sPP[(i * (i + 1) / 2 + j) * NSTR];
            double ep = 0, em = 0;
#pragma unroll
            for (int k = 0; k < NSTR; k += 2) {
                ep = fma(a[k], pp[k], ep);
                em = fma(a[k + 1], pp[k + 1], em);
            }
            const double dlt = (i == j) ? sIm[i] : 0.0;
            Rc[i][j] = fma(-om, ep, dlt);
            Bm[i][j] = fma(-om, em, dlt);
            store_bm(i * (i + 1) / 2 + j, Bm[i][j]);
        }
    // ---- Cholesky B+ = R R^T (lower, in place), reciprocal diagonal kept
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double tj = Rc[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) tj = fma(-Rc[j][k], Rc[j][k], tj);
        if (!(tj > 0)) { err |= skd::ERR_CHOLESKY; tj = 1.0; }
        const double rd = kj_rsqrt(tj);
        rinv[j] = rd;
        Rc[j][j] = tj * rd;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double t = Rc[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = fma(-Rc[i][k], Rc[j][k], t);
            Rc[i][j] = t * rd;
        }
    }
    // ---- A = R^T B- R (upper triangle A[i][j], i <= j), column by column: m3 = B- R[:, j], A[i][j] = sum_{r >= i} R[r][i] m3[r]
    double A[N][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double m3[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0;
#pragma unroll
            for (int r = j; r < N; ++r) t = fma((i >= r) ? Bm[i][r] : Bm[r][i], Rc[r][j], t);
            m3[i] = t;
        }
#pragma unroll
        for (int i = 0; i <= j; ++i) {
            double t = 0;
#pragma unroll
            for (int r = i; r < N; ++r) t = fma(Rc[r][i], m3[r], t);
            A[i][j] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    // ---- cyclic Jacobi, A <- J^T A J, V <- V J, rotation (p, q): new_p = c x_p - s x_q, new_q = s x_p + c x_q
    bool done = !live;
    for (int sweep = 0; sweep < SKD_KJ_MAXSWEEP; ++sweep) {
        double offp[4] = {0, 0, 0, 0}, dia = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dia = fma(A[i][i], A[i][i], dia);
#pragma unroll
            for (int j = i + 1; j < N; ++j) offp[(i + j) & 3] = fma(A[i][j], A[i][j], offp[(i + j) & 3]);
        }
        const double off = (offp[0] + offp[1]) + (offp[2] + offp[3]);
        done = done || (off <= SKD_KJ_TOL * dia) || (off == 0);
        if (__all(done)) break;
        // round-robin ("tournament") ordering: a round rotates N / 2 disjoint pairs, J = product of the round's rotations,
        // A <- J^T A J, V <- V J.  The N / 2 parameter chains of a round are independent (each is ~25 dependent FP64 operations).
#pragma unroll
        for (int round = 0; round < NM1; ++round) {
            int pp[JH], qq[JH];
            double cc[JH], ss[JH], tt[JH];
            bool rr[JH];
#pragma unroll
            for (int t = 0; t < JH; ++t) {
                int x, y;
                if (t == 0) { x = round; y = N - 1; }
                else { x = (round + t) % NM1; y = (round + NM1 - t) % NM1; }
                pp[t] = (x < y) ? x : y;
                qq[t] = (x < y) ? y : x;
            }
#pragma unroll
            for (int t = 0; t < JH; ++t) {
                const double apq = A[pp[t]][qq[t]], app = A[pp[t]][pp[t]], aqq = A[qq[t]][qq[t]];
                // t = sgn(d) h / (|d| + sqrt(d^2 + h^2)),  d = aqq - app, h = 2 apq;  c = 1 / sqrt(1 + t^2),  s = t c
                const double d = aqq - app, h = 2.0 * apq, x = fma(d, d, h * h);
                const bool rot = !done && (apq != 0.0) && (x > 0.0);
                const double xs = rot ? x : 1.0;
                const double den = fabs(d) + xs * kj_rsqrt(xs);
                double tn = ((d >= 0) ? h : -h) * kj_rcp(den);
                tn = rot ? tn : 0.0;
                const double c = kj_rsqrt(fma(tn, tn, 1.0));
                cc[t] = c; ss[t] = tn * c; tt[t] = tn; rr[t] = rot;
                A[pp[t]][pp[t]] = fma(-tn, apq, app);
                A[qq[t]][qq[t]] = fma(tn, apq, aqq);
                A[pp[t]][qq[t]] = rot ? 0.0 : apq;
            }
            // off-diagonal 2 x 2 blocks (pair t1 rows, pair t2 columns), t1 < t2:  B <- J1^T B J2
#pragma unroll
            for (int t1 = 0; t1 < JH; ++t1)
#pragma unroll
                for (int t2 = t1 + 1; t2 < JH; ++t2) {
                    const int p1 = pp[t1], q1 = qq[t1], p2 = pp[t2], q2 = qq[t2];
                    double& b00 = (p1 < p2) ? A[p1][p2] : A[p2][p1];
                    double& b01 = (p1 < q2) ? A[p1][q2] : A[q2][p1];
                    double& b10 = (q1 < p2) ? A[q1][p2] : A[p2][q1];
                    double& b11 = (q1 < q2) ? A[q1][q2] : A[q2][q1];
                    const double c1 = cc[t1], s1 = ss[t1], c2 = cc[t2], s2 = ss[t2];
                    const double u00 = fma(c2, b00, -s2 * b01), u01 = fma(s2, b00, c2 * b01);
                    const double u10 = fma(c2, b10, -s2 * b11), u11 = fma(s2, b10, c2 * b11);
                    b00 = fma(c1, u00, -s1 * u10);
                    b01 = fma(c1, u01, -s1 * u11);
                    b10 = fma(s1, u00, c1 * u10);
                    b11 = fma(s1, u01, c1 * u11);
                }
#pragma unroll
            for (int t = 0; t < JH; ++t)
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    const double vp = V[r][pp[t]], vq = V[r][qq[t]];
                    V[r][pp[t]] = fma(cc[t], vp, -ss[t] * vq);
                    V[r][qq[t]] = fma(ss[t], vp, cc[t] * vq);
                }
        }
    }
    if (!done) err |= skd::ERR_EIGEN;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double l = A[j][j];
        if (!(l > 0)) { err |= skd::ERR_EIGEN; l = fabs(l) + 1e-300; }
        lam[j] = l;
    }
    return err;
}

// -------------------------------------------------------------------------------------------------------------------
// KJ: homogeneous eigenproblem of every layer (reference: sktran_do_rte.cpp:382-553, the eigen-decomposition of S- S+).
//   grid = (ceil(nw / 64), nlyr), block = 64: lane = wavelength (coalesced reads of the [layer][l][w] layer optics), blockIdx.y = layer.
//   B+- = delta / mu - sqrt(w_i w_j / mu_i mu_j) omega sum_l a_l P_l^m(mu_i) P_l^m(mu_j)   (l - m even: B+, odd: B-), symmetric
//   B+ = R R^T (Cholesky), A = R^T B- R, A V = V diag(k^2) by cyclic Jacobi, X = T^-1 R^-T V   (the formulation of kern_k1_wave)
//   Output into the layer record for kern_k1_wave<.., PRE = true>: k_j^2 at [R::K + j], X(i, j) at [R::DWP + i * N + j]
//   (both fields are rewritten by K1 itself with their final contents), transposed through LDS so that a record's 72 doubles
//   leave as contiguous runs.
// A lane whose problem has converged is frozen (identity rotations), so a problem's result does not depend on the other
// problems of its wavefront.
// -------------------------------------------------------------------------------------------------------------------
// SOA = true: the eigen-solution goes to b.eig[(layer * (N + N N) + e) * nw + w] (e < N: k_j^2; then Xhat(i, j) = (T X)(i, j), the
// eigenvectors before the T^-1 scaling) with plain coalesced stores, for kern_k1_lane.
template <int N, bool SOA = false> __global__ void __launch_bounds__(64, SKD_WPE_KJ) kern_kj_lane(Geo g, Bufs b, int m) {
    m += blockIdx.z;   // (batched azimuth orders: skd::order_view)
    b = skd::order_view(b, blockIdx.z);
    constexpr int NSTR = 2 * N, NPAIR = N * (N + 1) / 2, NOUT = N + N * N, HALF = NOUT / 2, STRIDE = HALF + 1, NM1 = N - 1, JH = N / 2;
    static_assert(NOUT % 2 == 0, "two output halves");
    using R = skd::Rec<N>;
    const int lane = threadIdx.x, p = blockIdx.y, nw = b.nw, nlyr = g.nlyr;
    const int w0 = blockIdx.x * 64, w = w0 + lane;
    __shared__ double sPP[NPAIR * NSTR];   // sqrt(w_i w_j / mu_i mu_j) P_{m+k}(mu_i) P_{m+k}(mu_j), pair (i >= j), k = l - m (0 past the last order)
    __shared__ double sIm[N], sIt[N];
    __shared__ double sOut[64 * STRIDE];
    __shared__ int sLive[64];
    const bool live = (w < nw) && (m < b.norders[w]);
    if (!__any(live)) return;
    {
        const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
        for (int e = lane; e < NPAIR * NSTR; e += 64) {
            const int pr = e / NSTR, k = e % NSTR, l = m + k;
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= pr) ++i;
            const int j = pr - i * (i + 1) / 2;
            sPP[e] = (l < NSTR) ? sqrt(g.wt[i] / g.mu[i]) * sqrt(g.wt[j] / g.mu[j]) * lpmu[i * NSTR + l] * lpmu[j * NSTR + l] : 0.0;
        }
        if (lane < N) {
            sIm[lane] = 1.0 / g.mu[lane];
            sIt[lane] = 1.0 / sqrt(g.wt[lane] * g.mu[lane]);
        }
        sLive[lane] = live ? 1 : 0;
    }
    // Legendre coefficients from order m on (a'[k] = a_{m+k}); a lane without a problem solves the identity-like a = 0 case
    double a[NSTR];
#pragma unroll
    for (int k = 0; k < NSTR; ++k) a[k] = (live && m + k < NSTR) ? b.a[(size_t(p) * NSTR + m + k) * nw + w] : 0.0;
    const double om = live ? b.om[size_t(p) * nw + w] : 0.0;
    __syncthreads();

    double Rc[N][N], rinv[N], V[N][N], lam[N];
    int err = kj_eigen_stage<N>(a, om, live, sPP, sIm, Rc, rinv, V, lam, [](int, double) {});
    // ---- X = T^-1 R^-T V (column by column), staged in LDS in two halves and written as contiguous runs
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double t = V[i][j];
#pragma unroll
            for (int r = i + 1; r < N; ++r) t = fma(-Rc[r][i], V[r][j], t);
            V[i][j] = t * rinv[i];
        }
    }
    if (live && err) atomicOr(&b.status[w], err);
    if constexpr (SOA) {
        if (live) {
            double* out = b.eig + size_t(p) * NOUT * nw + w;
#pragma unroll
            for (int j = 0; j < N; ++j) out[size_t(j) * nw] = lam[j];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = 0; j < N; ++j) out[size_t(N + i * N + j) * nw] = V[i][j];
        }
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < HALF; ++e) {
            const int ee = e + half * HALF;   // 0 .. N - 1: k^2, then X(i, j) at N + i N + j
            double v;
            if (ee < N) v = lam[ee];
            else { const int i = (ee - N) / N, j = (ee - N) % N; v = V[i][j] * sIt[i]; }
            sOut[lane * STRIDE + e] = v;
        }
        __syncthreads();
        for (int idx = lane; idx < 64 * HALF; idx += 64) {
            const int rl = idx / HALF, e = idx % HALF, ee = e + half * HALF;
            if (!sLive[rl]) continue;
            double* rec = skd::rec_at<N>(b, nlyr, p, w0 + rl);
            rec[(ee < N) ? (R::K + ee) : (R::DWP + ee - N)] = sOut[rl * STRIDE + e];
        }
    }
}


// -------------------------------------------------------------------------------------------------------------------
// Column-per-lane 16 x 16 linear algebra for the 16-stream boundary value problem: a 16-lane DPP row owns one matrix, lane c
// holds COLUMN c in registers (compile-time row indices), four matrices (four wavelengths) per wavefront.  A product
// C = A B needs A[i][j] (register i of lane j, a row_newbcast:j move) and B[j][c] (own register j): no LDS, no shuffles
// through the LDS crossbar, no run-time register index anywhere.
// -------------------------------------------------------------------------------------------------------------------
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int CNT, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, CNT>{}, f);
}
// value of lane L of the own 16-lane row (v_mov_b64_dpp row_newbcast:L)
template <int L> __device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double row_dpp64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a row, in every lane (the order of the additions depends on the lane; each lane only uses its own)
__device__ __forceinline__ double row16_sum(double v) {
    v += row_dpp64<0xb1>(v);   // quad_perm [1,0,3,2]
    v += row_dpp64<0x4e>(v);   // quad_perm [2,3,0,1]
    v += row_dpp64<0x124>(v);  // row_ror 4
    v += row_dpp64<0x128>(v);  // row_ror 8
    return v;
}

// -------------------------------------------------------------------------------------------------------------------
// K1 lane per problem: everything of skd::k1_layer_solution after the eigen-decomposition -- U = S+ X, the eigen-pair derivatives
// by bi-orthogonal perturbation sums, W+-, dW+-, the Green's function coefficients and the particular solution G -- for one
// (layer, wavelength) per LANE (reference: sktran_do_rte.cpp:164-298,555-580,902-1332).  Works in the symmetrised variables
// Xhat = T X, Uhat = T U (T = diag sqrt(w mu)):  Uhat = B+ Xhat,  Mhat_j = (2 k_j^2 xhat_j - xhat_j / mu ... ) see below,
// z = Uhat^T Mhat,  C_aj = z_aj / (k_j^2 - k_a^2),  dXhat = Xhat C,  dUhat = (Uhat - Xhat / mu) / omega + Uhat C   (B+ dXhat = Uhat C).
// Solution columns are processed one at a time against the resident Xhat and Uhat (128 registers pairs); B- waits in LDS
// ([entry][lane], conflict free).  Records are written in the COLUMN-BLOCK layout skd::RecC: the 48 doubles of solution column j are one
// contiguous run of three 128-byte lines, staged through a 25-double LDS row per lane in two halves and stored as 16-byte pieces
// (12 per lane and half), so every line leaves whole.  The derivative rows of G (Rec::DG) are NOT formed here: the record holds the
// coefficient vectors coef[type * 5 + k] of every solution instead and kern_k2c_mfma8 forms DG = [W+; W-] coef as matrix-core
// products where it needs them (Bufs::k1_lane).
//   grid = (ceil(nw / 64), nlyr), block = 64, one wavefront per SIMD (the kernel owns the register file).
// -------------------------------------------------------------------------------------------------------------------
// FUSED: the eigen stage runs here too (kj_eigen_stage) -- no eigen-solution through HBM, B- straight into its LDS table, and
// Uhat = B+ Xhat = R (R^T Xhat) = R V from the Cholesky factor and the Jacobi eigenvectors at hand (288 instead of 512 multiply-adds, no B+)
template <int N, bool FUSED = true> __global__ void __launch_bounds__(64, 1) kern_k1_lane(Geo g, Bufs b, int s, int m) {
    m += blockIdx.z;   // (batched azimuth orders: skd::order_view)
    b = skd::order_view(b, blockIdx.z);
    constexpr int NSTR = 2 * N, NPAIR = N * (N + 1) / 2, NOUT = N + N * N, CH = 3 * N + 1, HB = 3 * N / 2;   // CH: odd staging stride; HB: 16-byte pieces per half block
    using R = skd::RecC<N>;
    static_assert(R::CB == 6 * N && N % 2 == 0, "two halves of 3 N doubles per column block");
    const int lane = threadIdx.x, p = blockIdx.y, nw = b.nw, nlyr = g.nlyr;
    const int w0 = blockIdx.x * 64, w = w0 + lane;
    __shared__ double sPP[NPAIR * NSTR];   // sqrt(w_i w_j / mu_i mu_j) P_{m+k}(mu_i) P_{m+k}(mu_j), pair (i >= j), k = l - m
    __shared__ double sQ[N * NSTR];        // fm w_i P_{m+k}(mu_i) P_{m+k}(mu_0)
    __shared__ double sIm[N], sIt[N], sWm[N];
    __shared__ double sBm[NPAIR * 64];
    __shared__ double sOut[64 * CH];
    const bool live = (w < nw) && (m < b.norders[min(w, nw - 1)]);
    if (!__any(live)) return;
    const int wl = min(w, nw - 1);
    const unsigned long long livemask = __ballot(live);
    // 3 N staged doubles of every lane -> element `off` .. of the lanes' records, 16 bytes per store and lane
    auto store_half = [&](int off, int count2) {
#pragma unroll 1
        for (int it = 0; it < HB; ++it) {
            const int idx = lane + 64 * it, rl = idx / HB, e2 = idx - rl * HB;
            if (e2 >= count2 || !((livemask >> rl) & 1ull)) continue;
            const double2 v = make_double2(sOut[rl * CH + 2 * e2], sOut[rl * CH + 2 * e2 + 1]);
            *reinterpret_cast<double2*>(skd::rec_at_r<R>(b, nlyr, p, w0 + rl) + off + 2 * e2) = v;
        }
    };
    {
        const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
        const double* lpc = g.lpcsz + (size_t(s) * NSTR + m) * NSTR;
        const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * skd::kPi);
        for (int e = lane; e < NPAIR * NSTR; e += 64) {
            const int pr = e / NSTR, k = e % NSTR, l = m + k;
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= pr) ++i;
            const int j = pr - i * (i + 1) / 2;
            sPP[e] = (l < NSTR) ? sqrt(g.wt[i] / g.mu[i]) * sqrt(g.wt[j] / g.mu[j]) * lpmu[i * NSTR + l] * lpmu[j * NSTR + l] : 0.0;
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
    double a[NSTR];
#pragma unroll
    for (int k = 0; k < NSTR; ++k) a[k] = (m + k < NSTR) ? b.a[(size_t(p) * NSTR + m + k) * nw + wl] : 0.0;
    const double om = b.om[size_t(p) * nw + wl], th = b.tau[size_t(p) * nw + wl];
    const double T = b.T[size_t(p) * nw + wl], sc = b.sec[size_t(p) * nw + wl];
    const double inv_om = 1.0 / om;
    double xh[N][N], lam[N], uh[N][N], qp[N], qm[N];
    bool allzero = true;
    int err = 0;
    if constexpr (!FUSED) {
        const double* in = b.eig + size_t(p) * NOUT * nw + wl;
#pragma unroll
        for (int j = 0; j < N; ++j) lam[j] = in[size_t(j) * nw];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) xh[i][j] = in[size_t(N + i * N + j) * nw];
    }
    __syncthreads();
    if constexpr (FUSED) {
        // ---- eigen stage (B- goes to LDS as it is formed), then Uhat = R V and Xhat = R^-T V
        double Rc[N][N], rinv[N];
        SKD_QR_FENCE
        err = kj_eigen_stage<N>(a, om, live, sPP, sIm, Rc, rinv, xh, lam, [&](int pair, double v) { sBm[pair * 64 + lane] = v; });
        SKD_QR_FENCE
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double t = 0;
#pragma unroll
                for (int r = 0; r <= i; ++r) t = fma(Rc[i][r], xh[r][j], t);
                uh[i][j] = t;
            }
        SKD_QR_FENCE
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                double t = xh[i][j];
#pragma unroll
                for (int r = i + 1; r < N; ++r) t = fma(-Rc[r][i], xh[r][j], t);
                xh[i][j] = t * rinv[i];
            }
        }
    } else {
        // ---- B+ (registers, lower triangle) and B- (LDS);  Uhat = B+ Xhat
        double bp[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double* pp = &sPP[(i * (i + 1) / 2 + j) * NSTR];
                double ep = 0, em = 0;
#pragma unroll
                for (int k = 0; k < NSTR; k += 2) {
                    ep = fma(a[k], pp[k], ep);
                    em = fma(a[k + 1], pp[k + 1], em);
                }
                const double dlt = (i == j) ? sIm[i] : 0.0;
                bp[i][j] = fma(-om, ep, dlt);
                sBm[(i * (i + 1) / 2 + j) * 64 + lane] = fma(-om, em, dlt);
            }
        SKD_QR_FENCE
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double t = 0;
#pragma unroll
                for (int r = 0; r < N; ++r) t = fma((i >= r) ? bp[i][r] : bp[r][i], xh[r][j], t);
                uh[i][j] = t;
            }
    }
    // ---- solar pseudo-source Q+- (sktran_do_rte.cpp:555-580); FUSED: the Legendre coefficients are read again here -- sixteen coalesced loads
    // instead of sixteen register pairs held through the Jacobi sweeps
    if constexpr (FUSED) {
        SKD_QR_FENCE
#pragma unroll
        for (int k = 0; k < NSTR; ++k) a[k] = (m + k < NSTR) ? b.a[(size_t(p) * NSTR + m + k) * nw + wl] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sp = 0, sn = 0;
#pragma unroll
        for (int k = 0; k < NSTR; k += 2) {
            sp = fma(a[k], sQ[i * NSTR + k], sp);
            sn = fma(a[k + 1], sQ[i * NSTR + k + 1], sn);
        }
        qp[i] = (sp + sn) * om;
        qm[i] = (sp - sn) * om;
        allzero = allzero && (qp[i] == 0.0) && (qm[i] == 0.0);
    }
    // ---- separation constants
    double lmax = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double l = lam[j];
        if (!(l > 0)) { err |= skd::ERR_EIGEN; l = fabs(l) + 1e-300; lam[j] = l; }
        lmax = fmax(lmax, l);
    }
    const double es = exp(-th * sc), invT = (T != 0) ? 1.0 / T : 0.0;
    double cf0[2][N];   // coefficient of the particular solution itself, type 0: A- C-, type 1: A+ C+
    // ---- solution columns
    static_for<N>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        SKD_QR_FENCE
        const double ikj = kj_rsqrt(lam[j]), kj = lam[j] * ikj, ej = exp(-th * kj);
        // Mhat_j = (2 k_j^2 xhat_j - uhat_j / mu - B- (xhat_j / mu)) / omega
        double mh[N];
        {
            double y[N], hh[N];
#pragma unroll
            for (int r = 0; r < N; ++r) { y[r] = sIm[r] * xh[r][j]; hh[r] = 0; }
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int r = 0; r <= i; ++r) {
                    const double v = sBm[(i * (i + 1) / 2 + r) * 64 + lane];
                    hh[i] = fma(v, y[r], hh[i]);
                    if (r != i) hh[r] = fma(v, y[i], hh[r]);
                }
#pragma unroll
            for (int i = 0; i < N; ++i) mh[i] = (2.0 * xh[i][j] * lam[j] - sIm[i] * uh[i][j] - hh[i]) * inv_om;
        }
        // z_a = uhat_a . Mhat_j;  dk_j = z_j / (2 k_j);  C_aj = z_a / (k_j^2 - k_a^2)
        double cc[N], dk = 0;
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
            double z = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) z = fma(uh[i][aa], mh[i], z);
            if (aa == j) { dk = 0.5 * z * ikj; cc[aa] = 0; }
            else {
                const double den = lam[j] - lam[aa];
                const bool okd = fabs(den) > 1e-13 * lmax;
                cc[aa] = okd ? z * kj_rcp(okd ? den : 1.0) : 0.0;
            }
        }
        // dxhat = Xhat C_j,  duhat = (uhat_j - xhat_j / mu) / omega + Uhat C_j,  W+- = T^-1 (xhat +- uhat / k) / 2 and derivatives;
        // the W+, W-, dW+ columns go straight to the staging row (the previous chunk has left it), dW- waits for the second chunk;
        // the sums of the Green's function coefficients of solution j run along (sktran_do_rte.cpp:1039-1237)
        __syncthreads();
        double dwm[N];
        double nrm = 0, dnrm = 0, Ap = 0, Am = 0, dAp = 0, dAm = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double dx = 0, du = (uh[i][j] - sIm[i] * xh[i][j]) * inv_om;
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                if (aa == j) continue;
                dx = fma(xh[i][aa], cc[aa], dx);
                du = fma(uh[i][aa], cc[aa], du);
            }
            const double uk = uh[i][j] * ikj, duk = du * ikj - uh[i][j] * dk * ikj * ikj;
            const double hs = 0.5 * sIt[i];
            const double wp = hs * (xh[i][j] + uk), wm = hs * (xh[i][j] - uk), dwp = hs * (dx + duk);
            dwm[i] = hs * (dx - duk);
            sOut[lane * CH + i] = wp;
            sOut[lane * CH + N + i] = wm;
            sOut[lane * CH + 2 * N + i] = dwp;
            const double wmu = sWm[i];
            nrm = fma(wmu, wp * wp - wm * wm, nrm);
            dnrm = fma(2.0 * wmu, wp * dwp - wm * dwm[i], dnrm);
            Ap += qp[i] * wp + qm[i] * wm;
            Am += qm[i] * wp + qp[i] * wm;
            const double ep_ = fma(inv_om, wp, dwp), em_ = fma(inv_om, wm, dwm[i]);
            dAp += qp[i] * ep_ + qm[i] * em_;
            dAm += qm[i] * ep_ + qp[i] * em_;
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
            skd::green_multipliers_e(T, th, sc, kj, ej, es, Cp, Cm);
        }
        const double dCm[4] = {Cm.d_tau, Cm.d_k * dk, Cm.v * invT, Cm.d_sec};
        const double dCp[4] = {Cp.d_tau, Cp.d_k * dk, Cp.v * invT, Cp.d_sec};
        double coef[10];
        coef[0] = Am * Cm.v;
        coef[5] = Ap * Cp.v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            coef[1 + t] = Am * dCm[t] + ((t == skd::TH_OM) ? dAm * Cm.v : 0.0);
            coef[6 + t] = Ap * dCp[t] + ((t == skd::TH_OM) ? dAp * Cp.v : 0.0);
        }
        cf0[0][j] = coef[0];
        cf0[1][j] = coef[5];
        // first half of the column block: W+ W- dW+ columns; second half: dW- column, the ten coefficients, k, dk, A+, A-, dA+, dA-
        __syncthreads();
        store_half(j * R::CB, HB);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) sOut[lane * CH + i] = dwm[i];
#pragma unroll
        for (int t = 0; t < 10; ++t) sOut[lane * CH + N + t] = coef[t];
        sOut[lane * CH + N + 10] = kj;
        sOut[lane * CH + N + 11] = dk;
        sOut[lane * CH + N + 12] = Ap;
        sOut[lane * CH + N + 13] = Am;
        sOut[lane * CH + N + 14] = dAp;
        sOut[lane * CH + N + 15] = dAm;
        __syncthreads();
        store_half(j * R::CB + 3 * N, HB);
    });
    // ---- particular solution at the layer edges, G[c][i] = sum_j Wsel(i, j) coef0_type(j):  W+- coef = T^-1 (Xhat coef +- Uhat (coef / k)) / 2
    //      c = 0: (W-, A- C-), 1: (W+, A- C-), 2: (W+, A+ C+), 3: (W-, A+ C+)
    {
        double gx[2][N], gu[2][N], cfk[2][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double ikj = kj_rsqrt(lam[j]);
            cfk[0][j] = cf0[0][j] * ikj;
            cfk[1][j] = cf0[1][j] * ikj;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double sx = 0, su = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    sx = fma(xh[i][j], cf0[t][j], sx);
                    su = fma(uh[i][j], cfk[t][j], su);
                }
                gx[t][i] = sx;
                gu[t][i] = su;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double hs = 0.5 * sIt[i];
            sOut[lane * CH + i] = hs * (gx[0][i] - gu[0][i]);
            sOut[lane * CH + N + i] = hs * (gx[0][i] + gu[0][i]);
            sOut[lane * CH + 2 * N + i] = hs * (gx[1][i] + gu[1][i]);
        }
        __syncthreads();
        store_half(R::G, HB);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sOut[lane * CH + i] = 0.5 * sIt[i] * (gx[1][i] - gu[1][i]);
            sOut[lane * CH + N + i] = exp(-th * (lam[i] * kj_rsqrt(lam[i])));   // exp(-k_j tau): the expression of the solution columns, formed again (no registers across them)
        }
        __syncthreads();
        static_assert(R::E == R::G + 4 * N, "exp(-k tau) follows G");
        store_half(R::G + 3 * N, N);
    }
    if (live && err) atomicOr(&b.status[w], err);
}

// Inverse of a 16 x 16 matrix by Householder QR, column per lane:  H_15 .. H_0 [A | I] = [R | Q^T],  A^-1 = R^-1 Q^T.
// a[] (column c of A) is destroyed, inv[] receives column c of the inverse.  No pivot search and no data-dependent control
// flow: unconditionally backward stable, the same instruction stream for every matrix.  Returns 1 if a column norm vanished.
// sR (optional, LDS, 16 x SKD_K2A_RS doubles of this lane's 16-lane row): the back substitution reads R[r][i] from there -- every lane
// writes its column of R once, the reads are broadcasts inside the row, two entries per 16-byte read -- instead of one cross-lane move per
// entry (120 of the 557 v_mov_b64_dpp of a layer; the moves, not the multiply-adds, bound this kernel's issue).
#ifndef SKD_K2A_RS
#define SKD_K2A_RS 18
#endif
template <bool LDS = false, class MID> __device__ __forceinline__ int qr16_inverse_cols(double (&a)[16], double (&inv)[16], int c, MID&& mid, double* sR = nullptr) {
    constexpr int M2 = 16;
    double rinv[LDS ? 1 : M2];   // 1 / R[k][k], the same in every lane of the row (LDS: kept in the padding behind column k of sR)
    int sing = 0;
#pragma unroll
    for (int i = 0; i < M2; ++i) inv[i] = (i == c) ? 1.0 : 0.0;
    static_for<M2>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k == 8) mid();
        SKD_QR_FENCE
        // Householder vector of column k (every lane computes it from its own column, lane k's is broadcast):
        // alpha = -sgn(x_k) |x|, v = x - alpha e_k, beta = 2 / v.v = -1 / (alpha v_k),  1 / alpha = -beta v_k
        // (sums of squares and dot products in four partial accumulators: a chain of dependent FP64 operations costs its full latency per link)
        const double xk = a[k];
        double sg[4] = {xk * xk, 0, 0, 0};
#pragma unroll
        for (int i = k + 1; i < M2; ++i) sg[(i - k) & 3] = fma(a[i], a[i], sg[(i - k) & 3]);
        const double n2 = (sg[0] + sg[1]) + (sg[2] + sg[3]);
        const bool zero = !(n2 > 0.0);
        const double n2s = zero ? 1.0 : n2;
        const double nrm = n2s * kj_rsqrt(n2s);
        const double alpha = (xk >= 0) ? -nrm : nrm;
        const double vk = xk - alpha;
        const double beta = -kj_rcp(alpha * vk);
        if constexpr (k < M2 - 1) {
            const double vkb = row_bcast<k>(vk), betab = row_bcast<k>(beta);
            double vb[M2];
#pragma unroll
            for (int i = k + 1; i < M2; ++i) vb[i] = row_bcast<k>(a[i]);
            const bool own = (c == k);
            if constexpr (LDS) sR[k * SKD_K2A_RS + 16] = -betab * vkb; else rinv[k] = -betab * vkb;
            sing |= (row_bcast<k>(n2) > 0.0) ? 0 : 1;
            double da[4] = {vkb * a[k], 0, 0, 0}, dq[4] = {vkb * inv[k], 0, 0, 0};
#pragma unroll
            for (int i = k + 1; i < M2; ++i) {
                da[(i - k) & 3] = fma(vb[i], a[i], da[(i - k) & 3]);
                dq[(i - k) & 3] = fma(vb[i], inv[i], dq[(i - k) & 3]);
            }
            const double fa = betab * ((da[0] + da[1]) + (da[2] + da[3])), fq = betab * ((dq[0] + dq[1]) + (dq[2] + dq[3]));
            a[k] = own ? alpha : fma(-fa, vkb, a[k]);
            inv[k] = fma(-fq, vkb, inv[k]);
#pragma unroll
            for (int i = k + 1; i < M2; ++i) {
                a[i] = own ? 0.0 : fma(-fa, vb[i], a[i]);
                inv[i] = fma(-fq, vb[i], inv[i]);
            }
        } else {
            // last column: a 1 x 1 reflection is the identity, R[15][15] = x_15
            const double xb = row_bcast<k>(xk);
            sing |= (xb == 0.0) ? 1 : 0;
            if constexpr (LDS) sR[k * SKD_K2A_RS + 16] = kj_rcp((xb == 0.0) ? 1.0 : xb); else rinv[k] = kj_rcp((xb == 0.0) ? 1.0 : xb);
        }
    });
    // back substitution  x = R^-1 y  on the own column of Q^T, column oriented:  x_i = y_i / R[i][i],  y_r -= R[r][i] x_i (r < i)
    // with R[r][i] = register r of lane i
    if constexpr (LDS) {
        // column c of R (rows r < c; the rest is not read) -> sR[c][r]
#pragma unroll
        for (int r = 0; r < M2 - 1; r += 2) *reinterpret_cast<double2*>(&sR[c * SKD_K2A_RS + r]) = make_double2(a[r], a[r + 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<M2>([&](auto ic) {
            constexpr int i = M2 - 1 - decltype(ic)::value;
            SKD_QR_FENCE
            const double xi = inv[i] * sR[i * SKD_K2A_RS + 16];
            inv[i] = xi;
            static_for<(i + 1) / 2>([&](auto rc) {
                constexpr int r = 2 * decltype(rc)::value;
                const double2 rr = *reinterpret_cast<const double2*>(&sR[i * SKD_K2A_RS + r]);
                inv[r] = fma(-rr.x, xi, inv[r]);
                if constexpr (r + 1 < i) inv[r + 1] = fma(-rr.y, xi, inv[r + 1]);
            });
        });
        return sing;
    } else {
    static_for<M2>([&](auto ic) {
        constexpr int i = M2 - 1 - decltype(ic)::value;
        SKD_QR_FENCE
        const double xi = inv[i] * rinv[i];
        inv[i] = xi;
        static_for<i>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            inv[r] = fma(-row_bcast<i>(a[r]), xi, inv[r]);
        });
    });
    return sing;
    }
}

#ifndef SKD_WPE_K2AC
#define SKD_WPE_K2AC 2
#endif
// K2a for N = 8 (16 streams), column per lane: the block-tridiagonal boundary value problem of FOUR wavelengths per wavefront
// (16-lane row = wavelength, lane c = column c of the 16 x 16 blocks).  Same recursion as kern_k2a_mfma8 / skd::k2a_bvp
// (reference: sktran_do_rte.cpp:1620-1790,1897-2294):
//   forward   T1 = Sub_p Dinv'_{p-1} (8 x 16),  D'_p = D_p + T1[:, 8:16] S_p,  r'_p = r_p - T1 r'_{p-1},  Dinv'_p = inverse (Householder QR)
//   backward  x_p = Dinv'_p (r'_p + [0; S_{p+1} x_{p+1}])
//   with, for column c (j = c & 7, E_j = exp(-k_j tau)):   D[r][c] = -(c < 8 ? W+(r, j) : W-(r, j) E_j), D[8 + i][c] = (c < 8 ? W-(i, j) E_j : W+(i, j)),
//   Sub_{p+1}[i][c] = (c < 8 ? W+(i, j) E_j : W-(i, j)),  S_p[i][c] = (c < 8 ? W-(i, j) : W+(i, j) E_j)
// Vectors are distributed (entry c in lane c).  Block inverses leave in the row-major layout kern_k2c_mfma8 reads.
template <class R> __device__ __forceinline__ void k2a_cols8_body(const Geo& g, const Bufs& b, int s, int m) {
    constexpr int N = 8, M2 = 16;
    const int lane = threadIdx.x, c = lane & 15, grp = lane >> 4, jc = c & 7, nw = b.nw, nlyr = g.nlyr;
    const bool lo = (c < N);
    const int w = blockIdx.x * 4 + grp;
    const bool live = (w < nw) && (m < b.norders[min(w, nw - 1)]);
    if (!__any(live)) return;
    const int wl = min(w, nw - 1);   // a row without a wavelength works on a valid record and stores nothing
    const double rho = (m == 0) ? b.albedo[wl] : 0.0;
    // cross-lane operands of the layer products through LDS (broadcast reads inside a 16-lane row) instead of row_newbcast moves:
    // Sub (8 x 16), T1 (8 x 16) and the R factor of the block (16 x 16) per wavelength row; row strides padded against bank conflicts
    // between the four rows of a wavefront
    constexpr int SS = 8 * 16 + 2, RS = 16 * SKD_K2A_RS + 2;
    __shared__ __attribute__((aligned(16))) double sSubAll[4 * SS], sT1All[4 * SS], sRAll[4 * RS];
    double* sSub = sSubAll + grp * SS;
    double* sT1 = sT1All + grp * SS;
    double* sRm = sRAll + grp * RS;
    double subc[N], dinv[M2], rprev = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) subc[i] = 0;
#pragma unroll
    for (int i = 0; i < M2; ++i) dinv[i] = 0;
    int sing = 0;
    // column data of a layer: W+(:, j), W-(:, j), E_j and this lane's entry of G_p (rows < 8: Gplus_top, rows >= 8: Gminus_bottom)
    double wp[N], wm[N], Ec, gown, gup;   // gup: the neighbour's G entry of the right-hand side (layer above: Gplus_bottom, layer below: Gminus_top)
    auto load_layer = [&](int p, double (&wp_)[N], double (&wm_)[N], double& E_, double& g_, double& gn_) {
        const double* rec = skd::rec_at_r<R>(b, nlyr, p, wl);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            wp_[i] = rec[R::wp(i, jc)];
            wm_[i] = rec[R::wm(i, jc)];
        }
        E_ = rec[R::e(jc)];
        g_ = lo ? rec[R::g(c)] : -rec[R::g(3 * N + jc)];
        gn_ = 0;
        if (lo) { if (p > 0) gn_ = -skd::rec_at_r<R>(b, nlyr, p - 1, wl)[R::g(2 * N + c)]; }
        else if (p < nlyr - 1) gn_ = skd::rec_at_r<R>(b, nlyr, p + 1, wl)[R::g(N + jc)];
    };
    load_layer(0, wp, wm, Ec, gown, gup);
    for (int p = 0; p < nlyr; ++p) {
        const bool last = (p == nlyr - 1);
        // the next layer's columns are requested now: their latency hides behind the factorisation (requesting them half way through
        // it looked better on paper and made the register allocator spill 270 instead of 60 registers)
        double nwp[N], nwm[N], nE = 0, ngown = 0, ngup = 0;
        if (!last) load_layer(p + 1, nwp, nwm, nE, ngown, ngup);
        double a[M2], sc[N], subn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double wpe = wp[i] * Ec, wme = wm[i] * Ec;
            a[i] = -(lo ? wp[i] : wme);
            a[N + i] = lo ? wme : wp[i];
            sc[i] = lo ? wm[i] : wpe;
            subn[i] = lo ? wpe : wm[i];
        }
        double rv = gown + gup;
        if (last && m == 0) {
            double gsp = 0, gsm = 0, gsg = 0;
            const double* rec = skd::rec_at_r<R>(b, nlyr, p, wl);
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const double wmu = g.wt[t] * g.mu[t];
                gsp = fma(wmu, wp[t], gsp);
                gsm = fma(wmu, wm[t], gsm);
                gsg = fma(wmu, rec[R::g(2 * N + t)], gsg);
            }
            const double corr = 2.0 * rho * (lo ? gsp * Ec : gsm);
#pragma unroll
            for (int i = 0; i < N; ++i) a[N + i] -= corr;
            if (!lo) rv += g.csz[s] * rho / skd::kPi * b.T[size_t(nlyr) * nw + wl] + 2.0 * rho * gsg;
        }
        if (p > 0) {
            // T1[i][c] = sum_j Sub_p[i][j] Dinv'_{p-1}[j][c]
            double t1[N];
#pragma unroll
            for (int i = 0; i < N; ++i) t1[i] = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) sSub[i * 16 + c] = subc[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < M2; j += 2) {
                SKD_QR_FENCE   // (keeps the scheduler from requesting all 64 operand pairs at once: registers)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double2 sv = *reinterpret_cast<const double2*>(&sSub[i * 16 + j]);
                    t1[i] = fma(sv.x, dinv[j], t1[i]);
                    t1[i] = fma(sv.y, dinv[j + 1], t1[i]);
                }
            }
            // D'[i][c] += sum_q T1[i][8 + q] S_p[q][c]
#pragma unroll
            for (int i = 0; i < N; ++i) sT1[i * 16 + c] = t1[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < N; q += 2) {
                SKD_QR_FENCE
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double2 tv = *reinterpret_cast<const double2*>(&sT1[i * 16 + N + q]);
                    a[i] = fma(tv.x, sc[q], a[i]);
                    a[i] = fma(tv.y, sc[q + 1], a[i]);
                }
            }
            // r'[i] -= sum_c T1[i][c] r'_{p-1}[c]
            double tr = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double sm = row16_sum(t1[i] * rprev);
                tr = (c == i) ? sm : tr;
            }
            rv -= tr;
        }
        sing |= qr16_inverse_cols<true>(a, dinv, c, [] {}, sRm);
        if (live) {
            double* dg = skd::dinv_at<N>(b, nlyr, p, w);
#pragma unroll
            for (int i = 0; i < M2; ++i) dg[i * M2 + c] = dinv[i];
            skd::xsol_at<N>(b, nlyr, p, w)[c] = rv;
        }
        rprev = rv;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            subc[i] = subn[i];
            wp[i] = nwp[i];
            wm[i] = nwm[i];
        }
        Ec = nE; gown = ngown; gup = ngup;
    }
    // back substitution: row per lane (lane c loads row c of Dinv', 128 contiguous bytes), x distributed
    double xn = 0;
    for (int p = nlyr - 1; p >= 0; --p) {
        const double* dg = skd::dinv_at<N>(b, nlyr, p, wl);
        double* xs = skd::xsol_at<N>(b, nlyr, p, wl);
        double drow[M2];
#pragma unroll
        for (int j = 0; j < M2; ++j) drow[j] = dg[c * M2 + j];
        double v = xs[c];
        if (p < nlyr - 1) {
            const double* recn = skd::rec_at_r<R>(b, nlyr, p + 1, wl);
            const double En = recn[R::e(jc)];
            double sx = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double sci = lo ? recn[R::wm(i, jc)] : recn[R::wp(i, jc)] * En;
                const double sm = row16_sum(sci * xn);
                sx = (c == N + i) ? sm : sx;
            }
            v += sx;
        }
        double x = 0;
        static_for<M2>([&](auto jx) {
            constexpr int j = decltype(jx)::value;
            x = fma(drow[j], row_bcast<j>(v), x);
        });
        if (live) xs[c] = x;
        xn = x;
    }
    if (live && sing && c == 0) atomicOr(&b.status[w], skd::ERR_SINGULAR_BVP);
}
template <class R> __global__ void __launch_bounds__(64, SKD_WPE_K2AC) kern_k2a_cols8(Geo g, Bufs b, int s, int m) { k2a_cols8_body<R>(g, skd::order_view(b, blockIdx.z), s, m + blockIdx.z); }
// the same with the whole register file (one wavefront per SIMD, nothing spilled to scratch) for launches that do not fill the SIMDs anyway:
// few wavelengths per rank of a sharded run, where the layer recursion of this kernel is the critical path of the order pipeline
template <class R> __global__ void __launch_bounds__(64, 1) kern_k2a_cols8_lone(Geo g, Bufs b, int s, int m) { k2a_cols8_body<R>(g, skd::order_view(b, blockIdx.z), s, m + blockIdx.z); }

}  // namespace skdev
