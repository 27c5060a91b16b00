// DISCO hot path for MI355X -- polarised (NSTOKES = 3) radiances and weighting functions, run-time matrix size.
//
// Same formulation as disco_core.hpp (symmetrised eigenproblem + Jacobi, Green's function particular solution,
// block-tridiagonal boundary value problem, Legendre-space source table) written for a run-time matrix size
// n = NSTOKES * N (row r = NSTOKES * stream + Stokes component).  The two heavy stages (layer eigen-solution, boundary value
// problem) are written for a team of lanes sharing the work matrices of one task (a wavefront / workgroup with the matrices in
// LDS when they fit, in HBM otherwise) with the matrix products on the FP64 matrix cores (team_gemm); the light stages run one thread
// per task with a thread-strided HBM scratch.  The C ABI uses it for NSTOKES = 3, for stream counts off the matrix-core path of the
// scalar kernels and for scattering-group weighting functions.
//
// Facts the formulation rests on (checked numerically against the oracle, tests/test_vector_emulation.py):
//   * with the Greek matrix B_l = [[a1, -b1, 0], [-b1, a2, 0], [0, 0, a3]], Pi_l = [[P, 0, 0], [0, R, -T], [0, -T, R]] and
//     D = diag(1, 1, -1) the reference's 3x3 products (lp_triple_product, sktran_do_lpproduct.h:22-79) are
//     Pi_l(out) B_l Pi_l(in)  and, negated,  (-1)^(l-m) Pi_l(out) B_l D Pi_l(in);  both are symmetric under
//     (out, s) <-> (in, s'), so S+- = M^-1 (I - omega E+- W) is symmetrised by diag sqrt(w mu) exactly as in the scalar case
//     and the Cholesky + Jacobi route applies with n = 3 N (RTESolver::solveHomogeneous, sktran_do_rte.cpp:382-553);
//   * the D factors of the reference's boundary conditions (sktran_do_rte.cpp:1239-1251,1971-1990,2083-2128) cancel when
//     the "minus" particular solution is kept without D, so the block-tridiagonal system has no sign bookkeeping;
//   * a Lambertian surface couples intensity rows only (v_plus / v_minus, sktran_do_rte.h:116-150).
// The same file is compiled with g++ by tests/emulate_vector.cpp (test infrastructure) to run the task bodies on the CPU.
#pragma once
#include <type_traits>
#include "disco_core.hpp"

namespace skv {

// team kernel bodies are always inlined into their __global__ wrappers: the address space of the work matrices (LDS / HBM) is then known
// to the compiler (ds_ instead of flat_ instructions) and no call frame is needed
#if defined(__HIPCC__)
#define SKV_TEAM_BODY __host__ __device__ inline __attribute__((always_inline))
#else
#define SKV_TEAM_BODY inline
#endif

using skd::Bufs;
using skd::Geo;
using skd::kPi;

// polarised companions of the Legendre tables of skd::Geo (same shapes) and sine azimuth factors of the ray layers
struct VGeo {
    int S;                  // NSTOKES (1 or 3)
    const double* lpmuR;    // [m][i<N][l]
    const double* lpmuT;
    const double* lpangR;   // [m][a][l]
    const double* lpangT;
    const double* lplosR;   // [los][m][l]
    const double* lplosT;
    const double* rl_sinm_t;  // [q][m]  sin(m * azi_q)
    const double* stokes_c;   // [los] rotation of (Q, U) into the configured Stokes basis, cpp/lib/output/outputc.cpp:223-228
    const double* stokes_s;
};

struct VBufs {
    int S, n;               // n = S * N
    const double* leg;      // [w][q][nleg][4]  (a1, a2, a3, b1 as stored by the caller, cpp/include/c_api/atmosphere.h:77-83)
    int nleg;
    double* a4;             // [p][l][4][w]  layer Greek coefficients (b1 with the sign flip of sktran_do_layerarray.cpp:418)
    double* rec;            // [w][p][REC]   REC = 4 n^2 + 41 n, see VRec
    double* dinv;           // [w][p][4 n^2]
    double* xsol;           // [w][p][2 n]
    double* J;              // [w][slot][m][S]
    double* radiance;       // [w*nlos+los][S]
    double* ops;            // [w][p][VOps::size]  LOS-independent adjoint operators (weighting functions)
    double* z;              // [p][2 n][tid3]   tid3 = (w * nlos + los) * S + Stokes component
    double* qlbuf;          // [p][NL][tid3]    Legendre-space LOS weights
    double* grad;           // [4 nlyr + 2][tid3] reduced gradient of the current solar zenith index
    double* dnat;           // [2 ngrid + 1][tid3] native derivatives
    int derivs;
    int k1_compact;         // eigen-stage work space in its compact layout (kv1_scratch(n, true)); set by the launcher
    // scattering-group derivatives (directional derivatives wrt the layer Greek coefficients, sktran_do_layerarray.cpp:496-513,759-860)
    int ngroups;            // G
    const double* dleg;     // [g][wavelength of the whole spectrum][q][nleg][4 | 1]  d_legendre of the scattering derivative mappings
    int dleg_nw, dleg_w0;   // wavelengths of the whole spectrum and offset of this block in it
    const double* dfg;      // [g][wavelength of the whole spectrum][q] d(delta-M truncation fraction) along the group, or null
    double* da4;            // [g][p][l][4][w]  direction of the layer coefficients for group g
    double* scratch;        // [element][thread] work matrices of the running kernel
    size_t scratch_stride;  // threads of the running kernel
    // non-Lambertian surface (plane-parallel / pseudo-spherical post-processing): azimuth expansion of the BRDF per wavelength and order
    int brdf_kind;          // 0 Lambertian (Bufs::albedo), 1 SnowKokhanovsky, 2 MODIS (cpp/include/sasktran2/atmosphere/surface.h:151-352)
    int brdf_nargs;
    int surf_nlos;          // "line of sight" rows of the table: nlos with plane-parallel post-processing; in spherical geometry the GROUND POINTS of the
                            // current solar zenith index (one per needed cos-angle grid point: RL(k, t) = factor from stream t into grid angle k)
    int surf_slot0;         // spherical geometry: table slot of ground point 0 of the current solar zenith index (the ground slots follow one another)
    const double* surf;     // [w][m][1 + nargs][VSurf::NP]  reflection factors and their derivatives wrt the BRDF arguments (VSurf)
    double* surf_ops;       // [w][N][2 n + 1]  adjoint operators of the ground: Dinv^T e_t per incoming stream t, then I-_t at the ground
};

// Reflection factors of one (wavelength, azimuth order): the ground condition of stream i and the reflected intensity of a line of sight are
//   I+_i = sum_t RS(i, t) I-_t + BS(i) T_ground,     I_los = sum_t RL(los, t) I-_t + BL(los) T_ground
// with RS(i, t) = (1 + delta_m0) rho_m(mu_i, mu_t) w_t mu_t, BS(i) = cos(sza) rho_m(mu_i, mu_0) / pi and rho_m the azimuth expansion of
// the BRDF (SurfaceStorage::compute_expansion, cpp/include/sktran_disco/sktran_do_surface.h:49-94; v_plus / v_minus / ground_direct_sun,
// sktran_do_rte.h:116-345).  Block 0 holds the values, block 1 + a their derivatives wrt BRDF argument a.
struct VSurf {
    int N, nlos, nargs;
    SKD_HD VSurf(int N_, int nlos_, int nargs_) : N(N_), nlos(nlos_), nargs(nargs_) {}
    SKD_HD int NP() const { return N * N + N + nlos * N + nlos; }
    SKD_HD int RS(int i, int t) const { return i * N + t; }
    SKD_HD int BS(int i) const { return N * N + i; }
    SKD_HD int RL(int los, int t) const { return N * N + N + los * N + t; }
    SKD_HD int BL(int los) const { return N * N + N + nlos * N + los; }
    SKD_HD size_t per_order() const { return size_t(1 + nargs) * NP(); }
};
SKD_HD const double* vsurf_at(const VBufs& v, const Geo& g, int m, int w) {
    return v.surf + (size_t(w) * g.nstr + m) * VSurf(g.N, v.surf_nlos, v.brdf_nargs).per_order();
}
constexpr int kMaxGroupsWithBrdf = 8;   // scattering groups served together with a non-Lambertian surface
// gradient / native-derivative rows of BRDF argument a: the Lambertian albedo's rows for a = 0, rows behind the scattering groups' otherwise
SKD_HD size_t surf_grad_row(const Geo& g, const VBufs& v, int a) { return a == 0 ? size_t(4 * g.nlyr) : size_t(4 * g.nlyr + 2 + v.ngroups * g.nlyr + (a - 1)); }
SKD_HD size_t surf_dnat_row(const Geo& g, const VBufs& v, int a) { return a == 0 ? size_t(2 * g.ngrid) : size_t(2 * g.ngrid + 1 + v.ngroups * g.ngrid + (a - 1)); }

constexpr int kMaxSurfStreams = 32;
// Weights c_t with which a line of sight sees the downwelling stream intensities at the ground through the surface: block 0 of `sf` for the
// factors themselves, block 1 + a for their derivatives wrt BRDF argument a.  Plane-parallel post-processing: gw RL(los, t).  Spherical:
// the ray's ground interpolation over the cos-angle grid points with the order's azimuth factor cos(m (pi - saa)) (create_ground_source_
// interpolator, do_source_diffuse_storage.cpp:213-265), gw = transmission from the ground to the observer.
SKD_HD void ground_los_weights(const Geo& g, const Bufs& b, const VBufs& v, const double* sf, int block, int s, int m, int los, int w, double gw, double* c) {
    const int N = g.N;
    const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
    const double* f = sf + size_t(block) * SF.NP();
    for (int t = 0; t < N; ++t) c[t] = 0;
    if (gw == 0.0) return;
    if (g.pp_mode) {
        for (int t = 0; t < N; ++t) c[t] = gw * f[SF.RL(los, t)];
        return;
    }
    if (!g.ray_ground[los]) return;
    const double az = cos(m * (kPi - g.ray_gsaa[los]));
    for (int e = g.g_start[los]; e < g.g_start[los + 1]; ++e) {
        if (g.g_sza[e] != s || skd::conv_at(b, g, g.g_slot[e], w) <= m) continue;
        const int k = g.g_slot[e] - v.surf_slot0;
        const double we = gw * g.g_w[e] * az;
        for (int t = 0; t < N; ++t) c[t] += we * f[SF.RL(k, t)];
    }
}

// ---- the reference's BRDFs (surface.h): geometry-only parts, so that the per-wavelength work is a weighted sum.
// MODIS (surface.h:236-281): brdf = (iso + vol k_vol + geo k_geo) / pi, linear in its arguments
SKD_HD void modis_kernels(double mu_in, double mu_out, double phi, double& k_vol, double& k_geo) {
    const double csza = mu_in, cvza = mu_out, ssza = sqrt(1 - csza * csza), svza = sqrt(1 - cvza * cvza);
    const double tsza = ssza / csza, tvza = svza / cvza, craa = -cos(phi), sraa = sin(phi);
    const double csa = fmax(-1.0, fmin(1.0, csza * cvza + ssza * svza * craa)), sa = acos(csa);
    k_vol = ((0.5 * kPi - sa) * csa + sin(sa)) / (csza + cvza) - 0.25 * kPi;
    const double d2 = tsza * tsza + tvza * tvza - 2 * tsza * tvza * craa;
    const double ct = fmax(-1.0, fmin(1.0, 2 * sqrt(d2 + tsza * tsza * tvza * tvza * sraa * sraa) * csza * cvza / (csza + cvza)));
    const double t = acos(ct);
    k_geo = (t - sin(t) * ct) * (csza + cvza) / (kPi * csza * cvza) - (csza + cvza - 0.5 * (1 + csa)) / (csza * cvza);
}
// SnowKokhanovsky (surface.h:151-229): brdf = r0 exp(-alpha K0(mu_s) K0(mu_v) / r0) / pi, alpha = sqrt(4 pi arg); r0 and the K0 product are geometry
SKD_HD void kokhanovsky_parts(double mu_in, double mu_out, double phi, double& r0, double& kk) {
    const double ss = sqrt(1 - mu_in * mu_in), sv = sqrt(1 - mu_out * mu_out);
    const double cost = fmax(-1.0, fmin(1.0, -mu_in * mu_out + ss * sv * cos(phi)));
    const double theta = acos(cost) * 180.0 / kPi;
    r0 = (1.247 + 1.186 * (mu_in + mu_out) + 5.157 * mu_in * mu_out + 11.1 * exp(-0.087 * theta) + 1.1 * exp(-0.014 * theta)) / (4.0 * (mu_in + mu_out));
    kk = (3.0 / 7.0) * (1.0 + 2.0 * mu_in) * (3.0 / 7.0) * (1.0 + 2.0 * mu_out);
}
// Geometry tables of the surface, built once per engine on the host (skhost::SurfaceTables): the NP (outgoing, incoming) direction pairs of
// VSurf, the NQ = 1024 azimuth samples a_q of the two 512-point Gauss-Legendre panels with weights cw[m][q] = (w_q / 2) cos(m pi a_q) pi (2 -
// delta_m0) / 2, per pair the factor that turns rho_m into RS / BS / RL / BL, and per BRDF kind its geometry-only samples.
struct SurfGeo {
    int NP, NQ, nstr;
    const double* cw;        // [m][q]
    const double* factor0;   // [pair] order 0
    const double* factor;    // [pair] orders > 0
    const double* modis_ke;  // [3][m][pair] expansions of k_vol, k_geo (already without the 1 / pi) and of the test function of kind 3
    const double* kok_r0;    // [pair][q]
    const double* kok_kk;    // [pair]
};
// Reflection factors of one (pair, wavelength) for every azimuth order: out[(m * (1 + nargs) + block) * NP + pair]
SKD_HD void surface_expand_pair(const SurfGeo& sg, int kind, int nargs, const double* args, int pair, double* out) {
    const int NP = sg.NP, nstr = sg.nstr;
    const size_t po = size_t(1 + nargs) * NP;
    if (kind == 3) {
        // the functional BRDF of the reference's "BRDF surface" known answer (legacy/test_scalar.cpp:806-845), scaled by its one argument: only
        // reachable through sk_amd_brdf_create_legacy_test_function, for the parity test of that case
        for (int m = 0; m < nstr; ++m) {
            const double f = (m == 0) ? sg.factor0[pair] : sg.factor[pair], ke = sg.modis_ke[(size_t(2) * nstr + m) * NP + pair];
            out[m * po + pair] = f * args[0] * ke;
            out[m * po + size_t(1) * NP + pair] = f * ke;
        }
        return;
    }
    if (kind == 2) {
        for (int m = 0; m < nstr; ++m) {
            const double f = (m == 0) ? sg.factor0[pair] : sg.factor[pair];
            const double kv = sg.modis_ke[(size_t(0) * nstr + m) * NP + pair], kg = sg.modis_ke[(size_t(1) * nstr + m) * NP + pair];
            const double iso = (m == 0) ? 1.0 : 0.0;   // a constant BRDF c / pi expands to c at order 0 (the Lambertian special case)
            out[m * po + pair] = f * (args[0] * iso + args[1] * kv + args[2] * kg);
            out[m * po + size_t(1) * NP + pair] = f * iso;
            out[m * po + size_t(2) * NP + pair] = f * kv;
            out[m * po + size_t(3) * NP + pair] = f * kg;
        }
        return;
    }
    // SnowKokhanovsky: f_q = r0_q exp(-alpha kk / r0_q) / pi,  d f_q / d arg = -2 kk exp(..) / alpha  (surface.h:206-229)
    const double alpha = sqrt(4 * kPi * args[0]), kk = sg.kok_kk[pair];
    for (int m = 0; m < nstr; ++m) { out[m * po + pair] = 0; out[m * po + NP + pair] = 0; }
    for (int q = 0; q < sg.NQ; ++q) {
        const double r0 = sg.kok_r0[size_t(pair) * sg.NQ + q], e = exp(-alpha * kk / r0);
        const double f = r0 * e / kPi, df = -2.0 * kk * e / alpha;
        for (int m = 0; m < nstr; ++m) {
            const double c = sg.cw[size_t(m) * sg.NQ + q];
            out[m * po + pair] += c * f;
            out[m * po + NP + pair] += c * df;
        }
    }
    for (int m = 0; m < nstr; ++m) {
        const double f = (m == 0) ? sg.factor0[pair] : sg.factor[pair];
        out[m * po + pair] *= f;
        out[m * po + NP + pair] *= f;
    }
}

struct VRec {
    int n, ngrp;
    SKD_HD explicit VRec(int n_, int G_ = 0) : n(n_), ngrp(G_) {}
    // same record as skd::Rec with N -> n: eigenvalues, exp(-k tau), d k / d omega, W+-, d W+- / d omega, Green's coefficients and
    // their omega derivatives, edge values of the particular solution G[4][n] (0 G+top, 1 G-top, 2 G+bottom, 3 G-bottom, no D factor),
    // DG[theta][4][n] (theta: tau, omega, T, sec), post-processing multipliers PPM[n][14]
    SKD_HD int K() const { return 0; }
    SKD_HD int E() const { return n; }
    SKD_HD int DK() const { return 2 * n; }
    SKD_HD int WP() const { return 3 * n; }
    SKD_HD int WM() const { return 3 * n + n * n; }
    SKD_HD int DWP() const { return 3 * n + 2 * n * n; }
    SKD_HD int DWM() const { return 3 * n + 3 * n * n; }
    SKD_HD int AP() const { return 3 * n + 4 * n * n; }
    SKD_HD int AM() const { return AP() + n; }
    SKD_HD int DAP() const { return AP() + 2 * n; }
    SKD_HD int DAM() const { return AP() + 3 * n; }
    SKD_HD int G() const { return AP() + 4 * n; }
    SKD_HD int DG() const { return G() + 4 * n; }
    SKD_HD int PPM() const { return DG() + 16 * n; }
    // per scattering group g: d k, d W+-, d A+-, d G[4] along the group's coefficient direction
    SKD_HD int GRP(int g) const { return PPM() + 14 * n + g * (2 * n * n + 7 * n); }
    SKD_HD int GDK(int g) const { return GRP(g); }
    SKD_HD int GDWP(int g) const { return GRP(g) + n; }
    SKD_HD int GDWM(int g) const { return GRP(g) + n + n * n; }
    SKD_HD int GDAP(int g) const { return GRP(g) + n + 2 * n * n; }
    SKD_HD int GDAM(int g) const { return GDAP(g) + n; }
    SKD_HD int GDG(int g) const { return GDAP(g) + 2 * n; }   // [4 edges][n]
    SKD_HD int size() const { return GRP(ngrp); }
};
SKD_HD double* vrec_at(const VBufs& v, int nlyr, int p, int w) { return v.rec + (size_t(w) * nlyr + p) * VRec(v.n, v.ngroups).size(); }
SKD_HD double* vdinv_at(const VBufs& v, int nlyr, int p, int w) { return v.dinv + (size_t(w) * nlyr + p) * (4 * size_t(v.n) * v.n); }
SKD_HD double* vxsol_at(const VBufs& v, int nlyr, int p, int w) { return v.xsol + (size_t(w) * nlyr + p) * (2 * v.n); }
SKD_HD double& vJ_at(const VBufs& v, const Geo& g, int m, int slot, int w, int s) {
    return v.J[((size_t(w) * g.nslot + slot) * g.nstr + m) * v.S + s];
}

// thread-strided scratch array
struct Scr {
    double* p;
    size_t stride;
    SKD_HD double& operator[](size_t i) const { return p[i * stride]; }
    SKD_HD Scr sub(size_t off) const { return Scr{p + off * stride, stride}; }
};

// Greek-matrix product of one order:  out(3) = B_l x,  with b1 as stored (sign already flipped)
SKD_HD void greek_apply(const double* c4, const double* x, double* out) {
    out[0] = c4[0] * x[0] - c4[3] * x[1];
    out[1] = -c4[3] * x[0] + c4[1] * x[1];
    out[2] = c4[2] * x[2];
}
// y = Pi_l x  with (P, R, T) of one direction and order
SKD_HD void pi_apply(double P, double R, double T, const double* x, double* y) {
    y[0] = P * x[0];
    y[1] = R * x[1] - T * x[2];
    y[2] = -T * x[1] + R * x[2];
}

// generalized spherical functions (P, R, T)[l] of stream i / table angle a / line of sight for azimuth order m
struct Fun3 {
    const double *P, *R, *T;
};
SKD_HD Fun3 fun_mu(const Geo& g, const VGeo& vg, int m, int i) {
    const size_t o = (size_t(m) * g.N + i) * g.nstr;
    return Fun3{g.lpmu + o, vg.lpmuR + o, vg.lpmuT + o};
}
SKD_HD Fun3 fun_ang(const Geo& g, const VGeo& vg, int m, int a) {
    const size_t o = (size_t(m) * g.nangle + a) * g.nstr;
    return Fun3{g.lpang + o, vg.lpangR + o, vg.lpangT + o};
}
SKD_HD Fun3 fun_los(const Geo& g, const VGeo& vg, int m, int los) {
    const size_t o = (size_t(los) * g.nstr + m) * g.nstr;
    return Fun3{g.lplos + o, vg.lplosR + o, vg.lplosT + o};
}

// =================================================================================================
// KV0: layer Greek coefficients                                              (task = wavelength)
//   reference: OpticalLayerArray ctor, sktran_do_layerarray.cpp:385-447 (a's carry the delta-M subtraction, b1 does not)
// =================================================================================================
SKD_HD void kv0_coefficients(const Geo& g, const Bufs& b, const VBufs& v, int w) {
    const int nw = b.nw, nlyr = g.nlyr, nstr = g.nstr, ngrid = g.ngrid, S = v.S;
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    const int per = (S == 3) ? 4 : 1;
    const double* leg = v.leg + size_t(w) * ngrid * v.nleg * per;
    for (int p = 0; p < nlyr; ++p) {
        double s = 0;
        for (int l = 0; l < nstr * 4; ++l) v.a4[(size_t(p) * nstr * 4 + l) * nw + w] = 0;
        for (int t = 0; t < 2; ++t) {
            int q = g.wq_idx[2 * p + t];
            if (q < 0) continue;
            const double wt = g.wq_w[2 * p + t], kscat = ssa[q] * ext[q];
            s += kscat * wt;
            const double fq = b.delta_f ? b.delta_f[size_t(w) * ngrid + q] : 0.0, fsub = fq / (1 - fq);
            for (int l = 0; l < nstr; ++l) {
                const double* c = leg + (size_t(q) * v.nleg + l) * per;
                const bool have = l < v.nleg;
                double* a = v.a4 + (size_t(p) * nstr + l) * 4 * nw + w;
                a[0] += wt * kscat * ((have ? c[0] : 0.0) - (2 * l + 1) * fsub);
                if (S == 3) {
                    a[size_t(1) * nw] += wt * kscat * ((have ? c[1] : 0.0) - (2 * l + 1) * fsub);
                    a[size_t(2) * nw] += wt * kscat * ((have ? c[2] : 0.0) - (2 * l + 1) * fsub);
                    a[size_t(3) * nw] += -wt * kscat * (have ? c[3] : 0.0);
                }
            }
        }
        if (s > 0) {
            for (int l = 0; l < nstr * 4; ++l) v.a4[(size_t(p) * nstr * 4 + l) * nw + w] /= s;
        } else {
            v.a4[(size_t(p) * nstr * 4) * nw + w] = 0;
        }
        // scattering-group directions of the layer coefficients: d_legendre of the mapping at the layer's last contributing grid
        // point plus the distance of that point's coefficients from the layer mean (sktran_do_layerarray.cpp:759-860; the reference
        // assigns the direction once per layer, from the last entry of its interpolation stencil)
        if (v.derivs && v.ngroups > 0) {
            const int qs = (g.wq_idx[2 * p + 1] >= 0) ? g.wq_idx[2 * p + 1] : g.wq_idx[2 * p];
            const double fq = b.delta_f ? b.delta_f[size_t(w) * ngrid + qs] : 0.0, fsub = fq / (1 - fq);
            for (int gi = 0; gi < v.ngroups; ++gi) {
                const double* dl = v.dleg + ((size_t(gi) * v.dleg_nw + v.dleg_w0 + w) * ngrid + qs) * v.nleg * per;
                for (int l = 0; l < nstr; ++l) {
                    const bool have = l < v.nleg;
                    const double* c = leg + (size_t(qs) * v.nleg + l) * per;
                    const double* a = v.a4 + (size_t(p) * nstr + l) * 4 * nw + w;
                    double* d = v.da4 + ((size_t(gi) * nlyr + p) * nstr + l) * 4 * nw + w;
                    for (int cc = 0; cc < 4; ++cc) d[size_t(cc) * nw] = 0;
                    // the group also moves the truncation fraction: -(2l + 1) d_f / (1 - f)^2 on a1, a2, a3 (sktran_do_layerarray.cpp:777-790,826-843)
                    const double dft = v.dfg ? (2 * l + 1) * v.dfg[(size_t(gi) * v.dleg_nw + v.dleg_w0 + w) * ngrid + qs] / ((1 - fq) * (1 - fq)) : 0.0;
                    d[0] = (have ? dl[l * per] + c[0] : 0.0) - (2 * l + 1) * fsub - a[0] - dft;
                    if (S == 3) {
                        d[size_t(1) * nw] = (have ? dl[l * per + 1] + c[1] : 0.0) - (2 * l + 1) * fsub - a[size_t(1) * nw] - dft;
                        d[size_t(2) * nw] = (have ? dl[l * per + 2] + c[2] : 0.0) - (2 * l + 1) * fsub - a[size_t(2) * nw] - dft;
                        d[size_t(3) * nw] = (have ? -dl[l * per + 3] - c[3] : 0.0) - a[size_t(3) * nw];
                    }
                }
            }
        }
    }
}

SKD_HD void load_coef(const Bufs& b, const VBufs& v, int nstr, int p, int l, int w, double* c4) {
    const double* a = v.a4 + (size_t(p) * nstr + l) * 4 * b.nw + w;
    for (int c = 0; c < 4; ++c) c4[c] = a[size_t(c) * b.nw];
}
// direction of the layer coefficients for scattering group gi
SKD_HD void load_dcoef(const Geo& g, const Bufs& b, const VBufs& v, int gi, int p, int l, int w, double* c4) {
    const double* a = v.da4 + ((size_t(gi) * g.nlyr + p) * g.nstr + l) * 4 * b.nw + w;
    for (int c = 0; c < 4; ++c) c4[c] = a[size_t(c) * b.nw];
}

// A team of lanes (a wavefront or a workgroup on the device, one "lane" on the CPU) works on one task: every lane calls the
// task function with the same arguments, loops written with SKV_PAR are shared out over the lanes and team.sync() separates a
// phase that writes shared work arrays from the next one that reads them.
struct Team {
    int lane, width;
    SKD_HD void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    }
};
#define SKV_PAR(i, count) for (int i = team.lane; i < (count); i += team.width)

// sum of f(q), q in [lo, hi), with four independent partial sums (the loads of four terms are in flight together)
template <class F> SKD_HD double sum4(int lo, int hi, F f) {
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int q = lo;
    for (; q + 3 < hi; q += 4) {
        t0 += f(q);
        t1 += f(q + 1);
        t2 += f(q + 2);
        t3 += f(q + 3);
    }
    for (; q < hi; ++q) t0 += f(q);
    return (t0 + t1) + (t2 + t3);
}

SKD_HD double recip_sqrt(double x) {
#ifdef __HIP_DEVICE_COMPILE__
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

// Team matrix product  C(i, j) = sum_{k < K} A(i, k) B(k, j),  i < M, j < N, with operands and result given by functors.  On the device
// every wavefront of the team takes strips of up to four 16 x 16 output tiles of one row block in turn and accumulates them on the
// FP64 matrix cores (v_mfma_f64_16x16x4_f64: A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15], D register r =
// D[(l >> 4) + 4 r][l & 15]): one A operand feeds the strip's independent accumulators (the matrix-core pipeline stays busy between
// dependent instructions, a quarter of the A traffic), two k-steps of operand loads are in flight at a time.  Every output element is
// accumulated in k order in one accumulator, so the result does not depend on the strip shape.  On the CPU (emulation) it is the
// plain triple loop.  The caller separates it from neighbouring phases with team.sync().
template <class FA, class FB, class FC> SKV_TEAM_BODY void team_gemm(const Team& team, int M, int N, int K, FA a, FB b, FC store) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SKV_GEMM_PLAIN)
    typedef double mf4v __attribute__((ext_vector_type(4)));
    const int wave = team.lane >> 6, nwaves = (team.width + 63) >> 6, l = team.lane & 63;
    const int tn = (N + 15) / 16, tm = (M + 15) / 16;
    const int ngrp = (tn + 3) / 4, gs = (tn + ngrp - 1) / ngrp;   // column tiles per strip (<= 4), strips per row block
    const int lr = l & 15, kq = l >> 4;
    for (int t = wave; t < tm * ngrp; t += nwaves) {
        const int i0 = (t / ngrp) * 16, jt0 = (t % ngrp) * gs;
        const int gw = (tn - jt0 < gs) ? tn - jt0 : gs;           // tiles of this strip (wave-uniform)
        const int ar = i0 + lr, bc = jt0 * 16 + lr;
        const bool arow = ar < M;
        mf4v acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = mf4v{0, 0, 0, 0};
        int k0 = 0;
        for (; k0 + 8 <= K; k0 += 8) {
            const int ka = k0 + kq, kb = k0 + 4 + kq;
            const double av0 = arow ? a(ar, ka) : 0.0, av1 = arow ? a(ar, kb) : 0.0;
            double bv0[4], bv1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = u < gw && bc + 16 * u < N;
                bv0[u] = on ? b(ka, bc + 16 * u) : 0.0;
                bv1[u] = on ? b(kb, bc + 16 * u) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < gw) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bv0[u], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < gw) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bv1[u], acc[u], 0, 0, 0);
        }
        for (; k0 < K; k0 += 4) {
            const int k = k0 + kq;
            const double av = (arow && k < K) ? a(ar, k) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < gw) {
                    const double bv = (bc + 16 * u < N && k < K) ? b(k, bc + 16 * u) : 0.0;
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[u], 0, 0, 0);
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < gw) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + kq + 4 * r, col = (jt0 + u) * 16 + lr;
                    if (row < M && col < N) store(row, col, acc[u][r]);
                }
            }
    }
#else
    SKV_PAR(e, M * N) {
        const int i = e / N, j = e % N;
        double t = 0;
        for (int k = 0; k < K; ++k) t += a(i, k) * b(k, j);
        store(i, j, t);
    }
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// Register-resident form of team_jacobi for even n with at most 4 blocks and 8 eigenvector pairs per lane.  The matrix lives in POSITION
// space: the rotations of every round pair the positions (2k, 2k+1), and between rounds the data move by the fixed round-robin permutation
// pi (position 0 stays, the even positions shift up, the odd ones down, the ends cross over), so after the n - 1 rounds of a sweep every
// entry is back at its own index.  A lane owns the same 2x2 blocks (pair k rows x pair k' columns) and eigenvector pairs in every round:
// it keeps them in registers, rotates them, writes the results to their permuted places in LDS, and after the barrier reads its own
// places again; all addresses are computed once.  The lanes that own the diagonal blocks compute the rotations from their registers.
// A round costs two barriers and no index arithmetic.  The pairing sequence differs from the indexed form (same sweeps, other order).
SKV_TEAM_BODY bool team_jacobi_registers(const Team& team, int n, double* Mx, double* V, double* rot) {
    constexpr int UA = 4, UV = 8;
    const int half = n / 2, nblk = half * half, nvj = n * half;
    const int nua = (nblk + team.width - 1) / team.width, nuv = (nvj + team.width - 1) / team.width;   // slots in use (team-uniform)
    double* cs = rot;            // (cos, sin) of pair k at [2 k], [2 k + 1]
    double* red = rot + n;       // [2][16] partial sums of the convergence test, one pair per wavefront
    auto perm = [&](int i) {     // round-robin move of position i
        if (i == 0 || n == 2) return i;
        if (i & 1) return (i == 1) ? 2 : i - 2;
        return (i == n - 2) ? n - 1 : i + 2;
    };
    double a00[UA], a01[UA], a10[UA], a11[UA], vp[UV], vq[UV];
    int own[UA], dst00[UA], dst01[UA], dst10[UA], dst11[UA], kk1[UA], kk2[UA], vown[UV], vdp[UV], vdq[UV], vk2[UV];
    bool aon[UA], von[UV];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int bk = team.lane + u * team.width;
        aon[u] = u < nua && bk < nblk;
        // the diagonal blocks come first (slot 0 of the first lanes: only they compute rotations), then the off-diagonal ones row by row
        int k = 0, k2 = 0;
        if (aon[u]) {
            if (bk < half) {
                k = k2 = bk;
            } else {
                const int o = bk - half, j = o % (half - 1);
                k = o / (half - 1);
                k2 = j + (j >= k ? 1 : 0);
            }
        }
        kk1[u] = k;
        kk2[u] = k2;
        own[u] = (2 * k) * n + 2 * k2;
        const int pr0 = perm(2 * k), pr1 = perm(2 * k + 1), pc0 = perm(2 * k2), pc1 = perm(2 * k2 + 1);
        dst00[u] = pr0 * n + pc0;
        dst01[u] = pr0 * n + pc1;
        dst10[u] = pr1 * n + pc0;
        dst11[u] = pr1 * n + pc1;
        a00[u] = a01[u] = a10[u] = a11[u] = 0.0;
        if (aon[u]) {
            a00[u] = Mx[own[u]];
            a01[u] = Mx[own[u] + 1];
            a10[u] = Mx[own[u] + n];
            a11[u] = Mx[own[u] + n + 1];
        }
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
        const int e = team.lane + u * team.width;
        von[u] = u < nuv && e < nvj;
        const int r = von[u] ? e / half : 0, k2 = von[u] ? e % half : 0;
        vk2[u] = k2;
        vown[u] = r * n + 2 * k2;
        vdp[u] = r * n + perm(2 * k2);
        vdq[u] = r * n + perm(2 * k2 + 1);
        vp[u] = vq[u] = 0.0;
        if (von[u]) {
            vp[u] = V[vown[u]];
            vq[u] = V[vown[u] + 1];
        }
    }
    bool converged = false;
    for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        // convergence: off-diagonal mass (upper triangle) against the diagonal, from the registers
        double off = 0, dia = 0;
#pragma unroll
        for (int u = 0; u < UA; ++u)
            if (aon[u]) {
                if (kk1[u] == kk2[u]) {
                    dia += a00[u] * a00[u] + a11[u] * a11[u];
                    off += a01[u] * a01[u];
                } else if (kk1[u] < kk2[u]) {
                    off += a00[u] * a00[u] + a01[u] * a01[u] + a10[u] * a10[u] + a11[u] * a11[u];
                }
            }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            off += __shfl_xor(off, o);
            dia += __shfl_xor(dia, o);
        }
        team.sync();
        if ((team.lane & 63) == 0) {
            red[team.lane >> 6] = off;
            red[16 + (team.lane >> 6)] = dia;
        }
        team.sync();
        off = 0;
        dia = 0;
        for (int wv = 0; wv < (team.width + 63) / 64; ++wv) {
            off += red[wv];
            dia += red[16 + wv];
        }
#ifdef SKV_EXP_JAC_SWEEPS
        if (sweep >= SKV_EXP_JAC_SWEEPS) { converged = true; break; }
#else
        if (off <= 1e-30 * dia || off == 0) { converged = true; break; }
#endif
        for (int step = 0; step < n - 1; ++step) {
            // ---- rotations from the diagonal blocks (slot 0 of the lanes below half)
            if (team.lane < half) {
                double c = 1.0, sn = 0.0;
                const double apq = a01[0];
                if (apq != 0.0) {
                    const double d = 0.5 * (a11[0] - a00[0]);
                    const double inv_r = recip_sqrt(d * d + apq * apq);
                    const double h = 0.5 + 0.5 * fabs(d) * inv_r;
                    const double inv_c = recip_sqrt(h);
                    c = h * inv_c;
                    sn = (d >= 0 ? 0.5 : -0.5) * apq * inv_r * inv_c;
                }
                cs[2 * team.lane] = c;
                cs[2 * team.lane + 1] = sn;
            }
            team.sync();
            // ---- rotate the blocks and the eigenvector pairs, results to their next places
#pragma unroll
            for (int u = 0; u < UA; ++u)
                if (u < nua) {
                    const double c1 = cs[2 * kk1[u]], s1 = cs[2 * kk1[u] + 1], c2 = cs[2 * kk2[u]], s2 = cs[2 * kk2[u] + 1];
                    const double t00 = c2 * a00[u] - s2 * a01[u], t01 = s2 * a00[u] + c2 * a01[u];
                    const double t10 = c2 * a10[u] - s2 * a11[u], t11 = s2 * a10[u] + c2 * a11[u];
                    const bool zero = u == 0 && team.lane < half && s1 != 0.0;   // the rotated pair is annihilated exactly
                    const double r00 = c1 * t00 - s1 * t10, r11 = s1 * t01 + c1 * t11;
                    const double r10 = zero ? 0.0 : s1 * t00 + c1 * t10, r01 = zero ? 0.0 : c1 * t01 - s1 * t11;
                    if (aon[u]) {
                        Mx[dst00[u]] = r00;
                        Mx[dst01[u]] = r01;
                        Mx[dst10[u]] = r10;
                        Mx[dst11[u]] = r11;
                    }
                }
#pragma unroll
            for (int u = 0; u < UV; ++u)
                if (u < nuv) {
                    const double c2 = cs[2 * vk2[u]], s2 = cs[2 * vk2[u] + 1];
                    const double np_ = c2 * vp[u] - s2 * vq[u], nq_ = s2 * vp[u] + c2 * vq[u];
                    if (von[u]) {
                        V[vdp[u]] = np_;
                        V[vdq[u]] = nq_;
                    }
                }
            team.sync();
#pragma unroll
            for (int u = 0; u < UA; ++u)
                if (u < nua && aon[u]) {
                    a00[u] = Mx[own[u]];
                    a01[u] = Mx[own[u] + 1];
                    a10[u] = Mx[own[u] + n];
                    a11[u] = Mx[own[u] + n + 1];
                }
#pragma unroll
            for (int u = 0; u < UV; ++u)
                if (u < nuv && von[u]) {
                    vp[u] = V[vown[u]];
                    vq[u] = V[vown[u] + 1];
                }
        }
    }
    team.sync();
    return converged;
}
#endif

// Symmetric eigenproblem of a team: cyclic Jacobi with the round-robin ("tournament") ordering, n/2 disjoint rotations per round applied
// as A <- J^T A J, V <- V J (V must hold the identity on entry; the eigenvalues end on the diagonal of Mx).  Each round is one pass over
// the 2x2 blocks (pair k rows x pair k' columns): a block is read, rotated from both sides and written by one lane, so the update is in
// place and needs no barrier between the column and the row rotation.  rot: n + 64 doubles of team-shared work space.
// Returns false when 60 sweeps did not converge.
SKV_TEAM_BODY bool team_jacobi(const Team& team, int n, double* Mx, double* V, double* rot) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SKV_JACOBI_INDEXED)
    if ((n & 1) == 0 && (n / 2) * (n / 2) <= 4 * team.width && n * (n / 2) <= 8 * team.width && team.width <= 1024)
        return team_jacobi_registers(team, n, Mx, V, rot);
#endif
    const int np = (n + 1) & ~1, half = np / 2;
    double* cs = rot;                                  // (cos, sin) of pair k at [2 k], [2 k + 1] (sin = 0 when the pair is skipped)
    int* pq = reinterpret_cast<int*>(rot + np);        // members of pair k in this round: p | q << 16  (np + half / 2 <= n + 64 doubles)
    const float inv_half = 1.0f / float(half);
    bool converged = false;
    for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        // convergence: off-diagonal mass against the diagonal (row sums in parallel, then every lane adds the n row sums)
        team.sync();
        SKV_PAR(i, n) rot[i] = sum4(i + 1, n, [&](int j) { return Mx[i * n + j] * Mx[i * n + j]; });
        team.sync();
        double off = 0, dia = 0;
        for (int i = 0; i < n; ++i) {
            dia += Mx[i * n + i] * Mx[i * n + i];
            off += rot[i];
        }
#ifdef SKV_EXP_JAC_SWEEPS
        if (sweep >= SKV_EXP_JAC_SWEEPS) { converged = true; break; }
#else
        if (off <= 1e-30 * dia || off == 0) { converged = true; break; }
#endif
        for (int step = 0; step < np - 1; ++step) {
            auto pair_of = [&](int k, int& pp, int& q) {
                // branch-free: k = 0 pairs the fixed index np - 1 with `step`, k > 0 pairs (step + k) with (step - k) modulo np - 1
                const int up = step + k, dn = step - k;
                const int a_ = (k == 0) ? np - 1 : (up >= np - 1 ? up - (np - 1) : up);
                const int b_ = (k == 0) ? step : (dn < 0 ? dn + (np - 1) : dn);
                pp = a_ < b_ ? a_ : b_;
                q = a_ < b_ ? b_ : a_;
            };
            team.sync();
            SKV_PAR(k, half) {
                int pp, q;
                pair_of(k, pp, q);
                double c = 1.0, sn = 0.0;
#ifndef SKV_EXP_JAC_NOROT
                if (q < n)
#else
                if (q < 0)
#endif
                {
                    const double apq = Mx[pp * n + q];
                    if (apq != 0.0) {
                        // tan(2 phi) = apq / d, |phi| <= pi/4:  cos(2 phi) = |d| / r,  c = sqrt((1 + cos 2phi) / 2),
                        // s = sin(2 phi) / (2 c) -- two reciprocal square roots, no cancellation
                        const double d = 0.5 * (Mx[q * n + q] - Mx[pp * n + pp]);
                        const double inv_r = recip_sqrt(d * d + apq * apq);
                        const double h = 0.5 + 0.5 * fabs(d) * inv_r;
                        const double inv_c = recip_sqrt(h);
                        c = h * inv_c;
                        sn = (d >= 0 ? 0.5 : -0.5) * apq * inv_r * inv_c;
                    }
                }
                cs[2 * k] = c;
                cs[2 * k + 1] = sn;
                pq[k] = pp | (q << 16);
            }
            team.sync();
            // A <- J^T A J by blocks, four blocks in flight per lane.  Index arithmetic and loads are branch-free (slots beyond the
            // last block repeat the lane's first block, the phantom index of an odd n is clamped: its rotation is the identity, so the
            // clamped operands only reach results that are not stored); only the stores are predicated.
#ifdef SKV_EXP_JAC_NOA
            const int nblk = 0;
#else
            const int nblk = half * half;
#endif
            auto block_phase = [&](auto UC) {
              constexpr int U = decltype(UC)::value;
              for (int b0 = team.lane; b0 < nblk; b0 += U * team.width) {
                double r00[U], r01[U], r10[U], r11[U];
                int i00[U], i01[U], i10[U], i11[U];
                bool on[U], v1[U], v2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int bk_ = b0 + u * team.width;
                    on[u] = bk_ < nblk;
                    const int bk = on[u] ? bk_ : b0;
                    const int k = int((float(bk) + 0.5f) * inv_half), k2 = bk - k * half;
                    const int w1 = pq[k], w2 = pq[k2];
                    const int P1 = w1 & 0xFFFF, Q1 = w1 >> 16, P2 = w2 & 0xFFFF, Q2 = w2 >> 16;
                    v1[u] = Q1 < n;
                    v2[u] = Q2 < n;
                    const int Q1c = v1[u] ? Q1 : n - 1, Q2c = v2[u] ? Q2 : n - 1;
                    i00[u] = P1 * n + P2;
                    i01[u] = P1 * n + Q2c;
                    i10[u] = Q1c * n + P2;
                    i11[u] = Q1c * n + Q2c;
                    const double c1 = cs[2 * k], s1 = cs[2 * k + 1], c2 = cs[2 * k2], s2 = cs[2 * k2 + 1];
                    const double a00 = Mx[i00[u]], a01 = Mx[i01[u]], a10 = Mx[i10[u]], a11 = Mx[i11[u]];
                    // columns (pair k'), then rows (pair k)
                    const double t00 = c2 * a00 - s2 * a01, t01 = s2 * a00 + c2 * a01;
                    const double t10 = c2 * a10 - s2 * a11, t11 = s2 * a10 + c2 * a11;
                    const bool zero = (k == k2) && s1 != 0.0;   // the rotated pair is annihilated exactly
                    r00[u] = c1 * t00 - s1 * t10;
                    r10[u] = zero ? 0.0 : s1 * t00 + c1 * t10;
                    r01[u] = zero ? 0.0 : c1 * t01 - s1 * t11;
                    r11[u] = s1 * t01 + c1 * t11;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (on[u]) {
                        Mx[i00[u]] = r00[u];
                        if (v2[u]) Mx[i01[u]] = r01[u];
                        if (v1[u]) Mx[i10[u]] = r10[u];
                        if (v1[u] && v2[u]) Mx[i11[u]] = r11[u];
                    }
                }
              }
            };
            // blocks in flight per lane: as many as the lane has (1, 2 or 4)
            if (nblk <= team.width) block_phase(std::integral_constant<int, 1>{});
            else if (nblk <= 2 * team.width) block_phase(std::integral_constant<int, 2>{});
            else block_phase(std::integral_constant<int, 4>{});
            // V <- V J, four (row, pair) items in flight per lane
#ifdef SKV_EXP_JAC_NOV
            const int nvj = 0;
#else
            const int nvj = n * half;
#endif
            auto vector_phase = [&](auto UC) {
              constexpr int U = decltype(UC)::value;
              for (int e0 = team.lane; e0 < nvj; e0 += U * team.width) {
                double np_[U], nq_[U];
                int ip[U], iq[U];
                bool on[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e_ = e0 + u * team.width;
                    const bool in = e_ < nvj;
                    const int e = in ? e_ : e0;
                    const int r = int((float(e) + 0.5f) * inv_half), k2 = e - r * half;
                    const int w2 = pq[k2];
                    const int P2 = w2 & 0xFFFF, Q2 = w2 >> 16;
                    on[u] = in && Q2 < n;
                    ip[u] = r * n + P2;
                    iq[u] = r * n + (Q2 < n ? Q2 : n - 1);
                    const double c2 = cs[2 * k2], s2 = cs[2 * k2 + 1];
                    const double vp_ = V[ip[u]], vq_ = V[iq[u]];
                    np_[u] = c2 * vp_ - s2 * vq_;
                    nq_[u] = s2 * vp_ + c2 * vq_;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (on[u]) {
                        V[ip[u]] = np_[u];
                        V[iq[u]] = nq_[u];
                    }
                }
              }
            };
            if (nvj <= team.width) vector_phase(std::integral_constant<int, 1>{});
            else if (nvj <= 2 * team.width) vector_phase(std::integral_constant<int, 2>{});
            else vector_phase(std::integral_constant<int, 4>{});
        }
        team.sync();
    }
    return converged;
}

// =================================================================================================
// KV1: homogeneous + particular solution of one layer for azimuth order m    (task = wavelength x layer, one team)
//   ws: 8 n^2 + 20 n + 64 doubles of team-shared work space (LDS when it fits, HBM otherwise)
// =================================================================================================
//   `compact`: without scattering groups the Green's-function coefficients (12 n doubles) live in the storage of the Cholesky factor, which
//   is dead by then -- at 16 polarised streams that is what lets a fourth team onto the CU (38.9 instead of 41.2 KB).  Only for n >= 12
//   (12 n <= n n): the launcher sets VBufs::k1_compact accordingly
SKD_HD size_t kv1_scratch(int n, bool compact = false) { return 8 * size_t(n) * n + (compact ? 8 : 20) * size_t(n) + 64; }
// whether the eigen stage runs in the compact layout (the device launcher and the CPU emulation of the tests both ask here)
SKD_HD bool kv1_compact_layout(int n, int ngroups) { return ngroups == 0 && n >= 12; }
#if defined(SKV_K1_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define KV1T(i) { const long long t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define KV1T(i)
#endif

//   `park` (no scattering groups): E+ and E- go to HBM / L2 instead of the team's work space and the six other matrices share FOUR slots
//   by lifetime -- M0: B+ -> its Cholesky factor -> d(Gamma) X -> dX;  M1: B- -> the symmetrised matrix of the Jacobi sweeps -> U;
//   M2: the product's temporary -> X;  M3: eigenvectors -> C -> dU.  4 n^2 + 19 n + 64 doubles: 79.6 KB at 32 polarised streams (n = 48),
//   i.e. TWO teams per CU where the 8-matrix layout (148 KB) admits one.  Same operations in the same order: bit-identical results.
SKD_HD size_t kv1_scratch_parked(int n) { return 4 * size_t(n) * n + 19 * size_t(n) + 64; }
SKV_TEAM_BODY void kv1_layer_solution(const Team& team, const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int p, int w, double* ws,
                                   bool derivs = false, double* park = nullptr) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr, nw = b.nw;
    const VRec R(n);
    double* rec = vrec_at(v, g.nlyr, p, w);
    const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
    const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    const int nn = n * n;
    const bool parked = park != nullptr;
    double *Ep = parked ? park : ws, *Em = parked ? park + nn : ws + nn;
    double *Bp = parked ? ws : ws + 2 * nn, *Bm = parked ? ws + nn : ws + 3 * nn, *Tm = parked ? ws + 2 * nn : ws + 4 * nn, *V = parked ? ws + 3 * nn : ws + 5 * nn;
    double *X = parked ? Tm : ws + 6 * nn, *U = parked ? Bm : ws + 7 * nn;
    double *lam = ws + (parked ? 4 : 8) * nn, *kk = lam + n, *nrm = kk + n, *Qp = nrm + n, *Qm = Qp + n, *acp = Qm + n, *acm = acp + n;
    double *rot = parked ? acm : acm + n;   // acm is only used by the scattering groups, which keep the 8-matrix layout
    double *coef = (parked || v.ngroups > 0 || !v.k1_compact) ? rot + n + 64 : Bp;  // rot: n + 64, coef: 12 n (see kv1_scratch)
    (void)acm;
    auto rw = [&](int r) { return g.wt[r / S]; };
    auto rmu = [&](int r) { return g.mu[r / S]; };
#if defined(SKV_K1_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
    // ---- E+ = (plain + negated) / 2, E- = (plain - negated) / 2 of the Greek-matrix products: one (i, j) block per lane
    SKV_PAR(blk, N * N) {
        const int i = blk / N, j = blk % N;
        double ep[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, em[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const Fun3 fi = fun_mu(g, vg, m, i), fj = fun_mu(g, vg, m, j);
        for (int l = m; l < nstr; ++l) {
            double c4[4];
            load_coef(b, v, nstr, p, l, w, c4);
            const double sg = ((l - m) & 1) ? -1.0 : 1.0;
            if (S == 1) {
                const double t = c4[0] * fi.P[l] * fj.P[l];
                ep[0] += 0.5 * (t + sg * t);
                em[0] += 0.5 * (t - sg * t);
                continue;
            }
            // columns of Pi_l(mu_j), pushed through B_l and B_l D, then Pi_l(mu_i)
            for (int s2 = 0; s2 < 3; ++s2) {
                double ex[3] = {0, 0, 0}, col[3], bc[3], bdc[3], y0[3], y1[3];
                ex[s2] = 1;
                pi_apply(fj.P[l], fj.R[l], fj.T[l], ex, col);
                greek_apply(c4, col, bc);
                const double dcol[3] = {col[0], col[1], -col[2]};
                greek_apply(c4, dcol, bdc);
                pi_apply(fi.P[l], fi.R[l], fi.T[l], bc, y0);
                pi_apply(fi.P[l], fi.R[l], fi.T[l], bdc, y1);
                for (int s1 = 0; s1 < 3; ++s1) {
                    ep[s1 * 3 + s2] += 0.5 * (y0[s1] + sg * y1[s1]);
                    em[s1 * 3 + s2] += 0.5 * (y0[s1] - sg * y1[s1]);
                }
            }
        }
        for (int s1 = 0; s1 < S; ++s1)
            for (int s2 = 0; s2 < S; ++s2) {
                const int r = i * S + s1, c = j * S + s2;
                const double e_p = ep[s1 * 3 + s2], e_m = em[s1 * 3 + s2];
                Ep[r * n + c] = e_p;
                Em[r * n + c] = e_m;
                const double dlt = (r == c) ? 1.0 / rmu(r) : 0.0;
                const double sq = sqrt(rw(r) / rmu(r)) * sqrt(rw(c) / rmu(c));
                Bp[r * n + c] = dlt - sq * om * e_p;
                Bm[r * n + c] = dlt - sq * om * e_m;
            }
    }
    team.sync();
    int err = 0;
    KV1T(0)
    // ---- Cholesky B+ = L L^T (lower, in place), right-looking: step j scales column j and takes its outer product off the trailing
    //      lower triangle.  The column is staged in a contiguous vector first: the trailing update then reads rows contiguously and the
    //      staged column as a broadcast / contiguous vector (a dot-product form walks the rows with stride n: 24- to 32-way LDS bank
    //      conflicts at n = 24 / 48, measured 11-13 % of the whole eigen stage).
    {
        double* colj = rot;   // staged column j, already divided by the pivot's square root (free until the Jacobi sweeps)
        for (int j = 0; j < n; ++j) {
            team.sync();      // trailing update of step j - 1 visible
            double d = Bp[j * n + j];
            if (!(d > 0)) { err |= skd::ERR_CHOLESKY; d = 1.0; }
            d = sqrt(d);
            const double inv_d = 1.0 / d;
            SKV_PAR(i, n) {
                const double l_ij = (i > j) ? Bp[i * n + j] * inv_d : (i == j ? d : 0.0);
                colj[i] = l_ij;
                if (i != j) Bp[i * n + j] = l_ij;   // final column j of the factor (zero above the diagonal); the pivot itself is still being read
            }
            team.sync();
            if (team.lane == 0) Bp[j * n + j] = d;
            const int mrem = n - j - 1;
            const float inv_m = mrem > 0 ? 1.0f / float(mrem) : 0.0f;
            SKV_PAR(e, mrem * mrem) {
                const int ii = int((float(e) + 0.5f) * inv_m), i = j + 1 + ii, k = j + 1 + (e - ii * mrem);
                if (k <= i) Bp[i * n + k] -= colj[i] * colj[k];
            }
        }
        team.sync();
    }
    KV1T(1)
    // ---- As = L^T B- L (into Bm)
    // (the strict upper triangle of the Cholesky factor is zero, so the full-range products are the triangular ones)
    team_gemm(team, n, n, n, [&](int i, int r) { return Bm[i * n + r]; }, [&](int r, int j) { return Bp[r * n + j]; },
              [&](int i, int j, double val) { Tm[i * n + j] = val; });
    team.sync();
    team_gemm(team, n, n, n, [&](int i, int r) { return Bp[r * n + i]; }, [&](int r, int j) { return Tm[r * n + j]; },
              [&](int i, int j, double val) {
                  if (j <= i) {   // one writer per unordered pair: exactly symmetric storage
                      Bm[i * n + j] = val;
                      Bm[j * n + i] = val;
                  }
              });
    SKV_PAR(e, nn) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    team.sync();
    KV1T(2)
    // ---- eigen-decomposition of the symmetrised matrix (team_jacobi)
    auto jacobi = [&](double* Mx) {
        if (!team_jacobi(team, n, Mx, V, rot)) err |= skd::ERR_EIGEN;
    };
#ifdef SKV_EXP_JAC2X
    SKV_PAR(e, nn) Tm[e] = Bm[e];
    team.sync();
    jacobi(Tm);
    team.sync();
    SKV_PAR(e, nn) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    team.sync();
#endif
    jacobi(Bm);
    team.sync();
    KV1T(3)
    SKV_PAR(j, n) {
        double l_ = Bm[j * n + j];
        if (!(l_ > 0)) { l_ = fabs(l_) + 1e-300; lam[j] = -1.0; } else { lam[j] = l_; }
        kk[j] = sqrt(l_);
    }
    team.sync();
    for (int j = 0; j < n; ++j)
        if (lam[j] < 0) err |= skd::ERR_EIGEN;
    // ---- X = T^-1 L^-T V (one column per lane) ;  U = S+ X with S+ = M^-1 (I - om E+ W)
    SKV_PAR(j, n) {
        for (int i = n - 1; i >= 0; --i) {
            const double t = V[i * n + j] - sum4(i + 1, n, [&](int r) { return Bp[r * n + i] * X[r * n + j]; });
            X[i * n + j] = t / Bp[i * n + i];
        }
        for (int i = 0; i < n; ++i) X[i * n + j] /= sqrt(rw(i) * rmu(i));
    }
    team.sync();
    team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * Ep[i * n + r]; }, [&](int r, int j) { return X[r * n + j]; },
              [&](int i, int j, double val) { U[i * n + j] = (X[i * n + j] - om * val) / rmu(i); });
    team.sync();
    KV1T(4)
    // ---- omega derivatives by bi-orthogonal perturbation theory (left vectors W M S+ X), as skd::k1_layer_solution:
    //      d(Gamma) X_j = [2 lam_j X_j - M^-1 U_j - S- M^-1 X_j] / om ;  dk_j = z_jj / (2 k_j) ;  dX = X C, C_ab = z_ab / (lam_b - lam_a)
    double* dX = parked ? Bp : V;    // V is free once X is known (parked: the slot of d(Gamma) X, consumed by the product that forms C)
    double* dU = Bm;   // As was consumed by the Jacobi sweeps
    double* dk = acp;  // reused before the Green's coefficients need it
    double* DGX = parked ? Bp : Tm;   // d(Gamma) X (parked: the Cholesky factor is dead once X is known)
    double* Cab = parked ? V : Bm;    // C
    double* dUo = parked ? V : Tm;    // dU (parked: C is dead once dX is formed)
    if (derivs) {
        team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * Em[i * n + r] / rmu(r); }, [&](int r, int j) { return X[r * n + j]; },
                  [&](int i, int j, double val) {
                      const double hm = X[i * n + j] / (rmu(i) * rmu(i)) - om / rmu(i) * val;
                      DGX[i * n + j] = (2 * X[i * n + j] * lam[j] - U[i * n + j] / rmu(i) - hm) / om;   // d(Gamma) X
                  });
        team.sync();
        double lmax = 0;
        for (int j = 0; j < n; ++j) lmax = fmax(lmax, lam[j]);
        team_gemm(team, n, n, n, [&](int a_, int i) { return rw(i) * rmu(i) * U[i * n + a_]; }, [&](int i, int b_) { return DGX[i * n + b_]; },
                  [&](int a_, int b_, double z) {
                      if (a_ == b_) {
                          dk[b_] = z / (2 * kk[b_]);
                          Cab[a_ * n + b_] = 0;
                      } else {
                          const double den = lam[b_] - lam[a_];
                          Cab[a_ * n + b_] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
                      }
                  });
        team.sync();
        team_gemm(team, n, n, n, [&](int i, int r) { return X[i * n + r]; }, [&](int r, int j) { return Cab[r * n + j]; },
                  [&](int i, int j, double val) { dX[i * n + j] = val; });
        team.sync();
        team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * Ep[i * n + r]; }, [&](int r, int j) { return dX[r * n + j]; },
                  [&](int i, int j, double val) { dUo[i * n + j] = (U[i * n + j] - X[i * n + j] / rmu(i)) / om + (dX[i * n + j] - om * val) / rmu(i); });
        team.sync();
        dU = dUo;
    }
    KV1T(5)
    // ---- W+-, dW+-, norms (one solution per lane)
    double* dnrm = rot;  // the rotation parameters of the Jacobi sweeps are free now (n + 64 >= n doubles)
    SKV_PAR(j, n) {
        double nr = 0, dnr = 0;
        rec[R.K() + j] = kk[j];
        rec[R.E() + j] = exp(-kk[j] * th);
        rec[R.DK() + j] = derivs ? dk[j] : 0.0;
        for (int i = 0; i < n; ++i) {
            const double uk = U[i * n + j] / kk[j];
            const double wp = 0.5 * (X[i * n + j] + uk), wm = 0.5 * (X[i * n + j] - uk);
            rec[R.WP() + i + n * j] = wp;
            rec[R.WM() + i + n * j] = wm;
            nr += rw(i) * rmu(i) * (wp * wp - wm * wm);
            if (derivs) {
                const double duk = dU[i * n + j] / kk[j] - U[i * n + j] * dk[j] / lam[j];
                const double dwp = 0.5 * (dX[i * n + j] + duk), dwm = 0.5 * (dX[i * n + j] - duk);
                rec[R.DWP() + i + n * j] = dwp;
                rec[R.DWM() + i + n * j] = dwm;
                dnr += rw(i) * rmu(i) * 2.0 * (wp * dwp - wm * dwm);
            }
        }
        nrm[j] = nr;
        dnrm[j] = dnr;
    }
    KV1T(6)
    // ---- solar pseudo-source: first column of the products with the unpolarised sun (single_scat_st, sktran_do_lpproduct.h:339-384)
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    const double* lpc = g.lpcsz + (size_t(s) * nstr + m) * nstr;
    SKV_PAR(i, N) {
        double qp[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
        const Fun3 fi = fun_mu(g, vg, m, i);
        for (int l = m; l < nstr; ++l) {
            double c4[4];
            load_coef(b, v, nstr, p, l, w, c4);
            const double sg = ((l - m) & 1) ? -1.0 : 1.0;
            const double x[3] = {lpc[l], 0, 0};
            double bx[3], y[3] = {0, 0, 0};
            greek_apply(c4, x, bx);  // B_l D x = B_l x for a pure-intensity vector
            if (S == 1) y[0] = bx[0] * fi.P[l];
            else pi_apply(fi.P[l], fi.R[l], fi.T[l], bx, y);
            for (int s1 = 0; s1 < S; ++s1) {
                qp[s1] += y[s1];
                qm[s1] += sg * y[s1];
            }
        }
        for (int s1 = 0; s1 < S; ++s1) {
            Qp[i * S + s1] = qp[s1] * fm * g.wt[i] * om;
            Qm[i * S + s1] = qm[s1] * fm * g.wt[i] * om;
        }
    }
    team.sync();
    bool allzero = true;
    for (int r = 0; r < n; ++r)
        if (Qp[r] != 0 || Qm[r] != 0) allzero = false;
    // Green's coefficients and multipliers per solution: coef[j][12] = Am Cm, Ap Cp, Am dCm[4], Ap dCp[4], dAm Cm, dAp Cp
    SKV_PAR(j, n) {
        double Ap = 0, Am = 0, dAp = 0, dAm = 0;
        double* cj = coef + 12 * j;
        for (int t = 0; t < 12; ++t) cj[t] = 0;
        if (!allzero) {
            // W+- and their derivatives from the team-shared X, U, dX, dU (contiguous over the solutions; the stored records would be
            // walked with stride n in HBM)
            const double inv_k = 1.0 / kk[j], dk_l = derivs ? dk[j] / lam[j] : 0.0, inv_om = 1.0 / om;
            for (int i = 0; i < n; ++i) {
                const double uk = U[i * n + j] * inv_k;
                const double wp = 0.5 * (X[i * n + j] + uk), wm = 0.5 * (X[i * n + j] - uk);
                Ap += Qp[i] * wp + Qm[i] * wm;
                Am += Qm[i] * wp + Qp[i] * wm;
                if (derivs) {
                    const double duk = dU[i * n + j] * inv_k - U[i * n + j] * dk_l;
                    const double dwp = 0.5 * (dX[i * n + j] + duk), dwm = 0.5 * (dX[i * n + j] - duk);
                    dAp += Qp[i] * dwp + Qp[i] * inv_om * wp + Qm[i] * dwm + Qm[i] * inv_om * wm;
                    dAm += Qm[i] * dwp + Qm[i] * inv_om * wp + Qp[i] * dwm + Qp[i] * inv_om * wm;
                }
            }
            Ap /= nrm[j];
            Am /= nrm[j];
            dAp = dAp / nrm[j] - dnrm[j] * Ap / nrm[j];
            dAm = dAm / nrm[j] - dnrm[j] * Am / nrm[j];
            skd::Mult Cp, Cm;
            skd::green_multipliers(T, th, sc, kk[j], Cp, Cm);
            cj[0] = Am * Cm.v;
            cj[1] = Ap * Cp.v;
            if (derivs) {
                const double dkj = rec[R.DK() + j];
                const double dCp[4] = {Cp.d_tau, Cp.d_k * dkj, T != 0 ? Cp.v / T : 0.0, Cp.d_sec};
                const double dCm[4] = {Cm.d_tau, Cm.d_k * dkj, T != 0 ? Cm.v / T : 0.0, Cm.d_sec};
                for (int t = 0; t < 4; ++t) {
                    cj[2 + t] = Am * dCm[t];
                    cj[6 + t] = Ap * dCp[t];
                }
                cj[10] = dAm * Cm.v;
                cj[11] = dAp * Cp.v;
            }
        }
        rec[R.AP() + j] = Ap;
        rec[R.AM() + j] = Am;
        rec[R.DAP() + j] = derivs ? dAp : 0.0;
        rec[R.DAM() + j] = derivs ? dAm : 0.0;
    }
    team.sync();
    SKV_PAR(i, n) {
        double g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        double dg[4][4];
        for (int t = 0; t < 4; ++t)
            for (int c = 0; c < 4; ++c) dg[t][c] = 0;
        for (int j = 0; j < n; ++j) {
            const double* cj = coef + 12 * j;
            const double wp = rec[R.WP() + i + n * j], wm = rec[R.WM() + i + n * j];
            g0 += cj[0] * wm;
            g1 += cj[0] * wp;
            g2 += cj[1] * wp;
            g3 += cj[1] * wm;
            if (derivs) {
                const double dwp = rec[R.DWP() + i + n * j], dwm = rec[R.DWM() + i + n * j];
                for (int t = 0; t < 4; ++t) {
                    dg[t][0] += cj[2 + t] * wm;
                    dg[t][1] += cj[2 + t] * wp;
                    dg[t][2] += cj[6 + t] * wp;
                    dg[t][3] += cj[6 + t] * wm;
                }
                dg[skd::TH_OM][0] += cj[10] * wm + cj[0] * dwm;
                dg[skd::TH_OM][1] += cj[10] * wp + cj[0] * dwp;
                dg[skd::TH_OM][2] += cj[11] * wp + cj[1] * dwp;
                dg[skd::TH_OM][3] += cj[11] * wm + cj[1] * dwm;
            }
        }
        rec[R.G() + 0 * n + i] = g0;
        rec[R.G() + 1 * n + i] = g1;
        rec[R.G() + 2 * n + i] = g2;
        rec[R.G() + 3 * n + i] = g3;
        for (int t = 0; t < 4; ++t)
            for (int c = 0; c < 4; ++c) rec[R.DG() + (t * 4 + c) * n + i] = dg[t][c];
    }
    KV1T(7)
    // ---- scattering groups: the same perturbation theory along a general direction dc of the layer coefficients,
    //      dS+- = -om M^-1 dE+-(dc) W ;  d(Gamma) X = dS- U + S- (dS+ X) ;  dU = dS+ X + S+ dX ;  dQ = Q(dc)
    //      (the omega derivative above is the direction dc = c / om)
    if (derivs && v.ngroups > 0) {
        const VRec RG(n, v.ngroups);
        double *dEp = Bp, *dEm = Bm, *A1 = Tm, *A2 = V;   // the Cholesky factor, C, dU and dX are no longer needed
        double *dQp = coef, *dQm = coef + n, *dnrm_g = coef + 2 * n, *fcm = coef + 3 * n, *fcp = coef + 4 * n, *dkg = coef + 5 * n;
        for (int gi = 0; gi < v.ngroups; ++gi) {
            team.sync();
            SKV_PAR(blk, N * N) {
                const int i = blk / N, j = blk % N;
                double ep[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, em[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                const Fun3 fi = fun_mu(g, vg, m, i), fj = fun_mu(g, vg, m, j);
                for (int l = m; l < nstr; ++l) {
                    double c4[4];
                    load_dcoef(g, b, v, gi, p, l, w, c4);
                    const double sg = ((l - m) & 1) ? -1.0 : 1.0;
                    if (S == 1) {
                        const double t = c4[0] * fi.P[l] * fj.P[l];
                        ep[0] += 0.5 * (t + sg * t);
                        em[0] += 0.5 * (t - sg * t);
                        continue;
                    }
                    for (int s2 = 0; s2 < 3; ++s2) {
                        double ex[3] = {0, 0, 0}, col[3], bc[3], bdc[3], y0[3], y1[3];
                        ex[s2] = 1;
                        pi_apply(fj.P[l], fj.R[l], fj.T[l], ex, col);
                        greek_apply(c4, col, bc);
                        const double dcol[3] = {col[0], col[1], -col[2]};
                        greek_apply(c4, dcol, bdc);
                        pi_apply(fi.P[l], fi.R[l], fi.T[l], bc, y0);
                        pi_apply(fi.P[l], fi.R[l], fi.T[l], bdc, y1);
                        for (int s1 = 0; s1 < 3; ++s1) {
                            ep[s1 * 3 + s2] += 0.5 * (y0[s1] + sg * y1[s1]);
                            em[s1 * 3 + s2] += 0.5 * (y0[s1] - sg * y1[s1]);
                        }
                    }
                }
                for (int s1 = 0; s1 < S; ++s1)
                    for (int s2 = 0; s2 < S; ++s2) {
                        dEp[(i * S + s1) * n + j * S + s2] = ep[s1 * 3 + s2];
                        dEm[(i * S + s1) * n + j * S + s2] = em[s1 * 3 + s2];
                    }
            }
            team.sync();
            // A1 = dS+ X, A2 = dS- U
            team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * dEp[i * n + r]; }, [&](int r, int j) { return X[r * n + j]; },
                      [&](int i, int j, double val) { A1[i * n + j] = -om / rmu(i) * val; });
            team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * dEm[i * n + r]; }, [&](int r, int j) { return U[r * n + j]; },
                      [&](int i, int j, double val) { A2[i * n + j] = -om / rmu(i) * val; });
            team.sync();
            // d(Gamma) X = A2 + S- A1   (into dEm, which is consumed)
            team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * Em[i * n + r]; }, [&](int r, int j) { return A1[r * n + j]; },
                      [&](int i, int j, double val) { dEm[i * n + j] = A2[i * n + j] + (A1[i * n + j] - om * val) / rmu(i); });
            team.sync();
            double lmax = 0;
            for (int j = 0; j < n; ++j) lmax = fmax(lmax, lam[j]);
            // C (into A2)
            team_gemm(team, n, n, n, [&](int a_, int i) { return rw(i) * rmu(i) * U[i * n + a_]; }, [&](int i, int b_) { return dEm[i * n + b_]; },
                      [&](int a_, int b_, double z) {
                          if (a_ == b_) {
                              dkg[b_] = z / (2 * kk[b_]);
                              A2[a_ * n + b_] = 0;
                          } else {
                              const double den = lam[b_] - lam[a_];
                              A2[a_ * n + b_] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
                          }
                      });
            team.sync();
            // dX = X C  (into dEm)
            team_gemm(team, n, n, n, [&](int i, int r) { return X[i * n + r]; }, [&](int r, int j) { return A2[r * n + j]; },
                      [&](int i, int j, double val) { dEm[i * n + j] = val; });
            team.sync();
            // dU = A1 + S+ dX  (into A2)
            team_gemm(team, n, n, n, [&](int i, int r) { return rw(r) * Ep[i * n + r]; }, [&](int r, int j) { return dEm[r * n + j]; },
                      [&](int i, int j, double val) { A2[i * n + j] = A1[i * n + j] + (dEm[i * n + j] - om * val) / rmu(i); });
            team.sync();
            SKV_PAR(j, n) {
                double dnr = 0;
                rec[RG.GDK(gi) + j] = dkg[j];
                for (int i = 0; i < n; ++i) {
                    const double duk = A2[i * n + j] / kk[j] - U[i * n + j] * dkg[j] / lam[j];
                    const double dwp = 0.5 * (dEm[i * n + j] + duk), dwm = 0.5 * (dEm[i * n + j] - duk);
                    rec[RG.GDWP(gi) + i + n * j] = dwp;
                    rec[RG.GDWM(gi) + i + n * j] = dwm;
                    dnr += rw(i) * rmu(i) * 2.0 * (rec[R.WP() + i + n * j] * dwp - rec[R.WM() + i + n * j] * dwm);
                }
                dnrm_g[j] = dnr;
            }
            SKV_PAR(i, N) {   // dQ+- along the direction
                double qp[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
                const Fun3 fi = fun_mu(g, vg, m, i);
                for (int l = m; l < nstr; ++l) {
                    double c4[4];
                    load_dcoef(g, b, v, gi, p, l, w, c4);
                    const double sg = ((l - m) & 1) ? -1.0 : 1.0;
                    const double x[3] = {lpc[l], 0, 0};
                    double bx[3], y[3] = {0, 0, 0};
                    greek_apply(c4, x, bx);
                    if (S == 1) y[0] = bx[0] * fi.P[l];
                    else pi_apply(fi.P[l], fi.R[l], fi.T[l], bx, y);
                    for (int s1 = 0; s1 < S; ++s1) {
                        qp[s1] += y[s1];
                        qm[s1] += sg * y[s1];
                    }
                }
                for (int s1 = 0; s1 < S; ++s1) {
                    dQp[i * S + s1] = qp[s1] * fm * g.wt[i] * om;
                    dQm[i * S + s1] = qm[s1] * fm * g.wt[i] * om;
                }
            }
            team.sync();
            SKV_PAR(j, n) {
                double dAp = 0, dAm = 0, f_cm = 0, f_cp = 0;
                const double Ap = rec[R.AP() + j], Am = rec[R.AM() + j];
                for (int i = 0; i < n; ++i) {
                    const double wp = rec[R.WP() + i + n * j], wm = rec[R.WM() + i + n * j];
                    const double dwp = rec[RG.GDWP(gi) + i + n * j], dwm = rec[RG.GDWM(gi) + i + n * j];
                    dAp += Qp[i] * dwp + dQp[i] * wp + Qm[i] * dwm + dQm[i] * wm;
                    dAm += Qm[i] * dwp + dQm[i] * wp + Qp[i] * dwm + dQp[i] * wm;
                }
                dAp = dAp / nrm[j] - dnrm_g[j] * Ap / nrm[j];
                dAm = dAm / nrm[j] - dnrm_g[j] * Am / nrm[j];
                skd::Mult Cp, Cm;
                skd::green_multipliers(T, th, sc, kk[j], Cp, Cm);
                f_cm = Am * Cm.d_k * dkg[j] + dAm * Cm.v;
                f_cp = Ap * Cp.d_k * dkg[j] + dAp * Cp.v;
                rec[RG.GDAP(gi) + j] = dAp;
                rec[RG.GDAM(gi) + j] = dAm;
                fcm[j] = f_cm;
                fcp[j] = f_cp;
                // the products A C themselves for the dW terms
                acp[j] = Ap * Cp.v;
                acm[j] = Am * Cm.v;
            }
            team.sync();
            SKV_PAR(i, n) {
                double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                for (int j = 0; j < n; ++j) {
                    const double wp = rec[R.WP() + i + n * j], wm = rec[R.WM() + i + n * j];
                    const double dwp = rec[RG.GDWP(gi) + i + n * j], dwm = rec[RG.GDWM(gi) + i + n * j];
                    d0 += fcm[j] * wm + acm[j] * dwm;
                    d1 += fcm[j] * wp + acm[j] * dwp;
                    d2 += fcp[j] * wp + acp[j] * dwp;
                    d3 += fcp[j] * wm + acp[j] * dwm;
                }
                rec[RG.GDG(gi) + 0 * n + i] = d0;
                rec[RG.GDG(gi) + 1 * n + i] = d1;
                rec[RG.GDG(gi) + 2 * n + i] = d2;
                rec[RG.GDG(gi) + 3 * n + i] = d3;
            }
        }
    }
    KV1T(8)
#if defined(SKV_K1_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    if (team.lane == 0 && w == 0 && p == g.nlyr / 2 && m == 1)
        printf("KV1T n %d lanes %d: build %lld chol %lld LtBL %lld jacobi %lld eigvec+U %lld om-derivs %lld W+norms %lld Q/Green/G %lld groups %lld\n", n, team.width,
               tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], tacc[6], tacc[7], tacc[8]);
#endif
    team.sync();
    if (err && team.lane == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.status[w], err);
#else
        b.status[w] |= err;
#endif
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
// |v| with the row index in the low byte of the mantissa: the maximum of such values is the pivot candidate and names its row
// (a relative perturbation of 2^-44 of the comparison, irrelevant for the choice of a pivot); rows below `first` do not compete
__device__ __forceinline__ double pivot_candidate(double v, int row, int first) {
    const long long bits = (__double_as_longlong(fabs(v)) & ~0xFFll) | row;
    return row >= first ? __longlong_as_double(bits) : -1.0;
}
__device__ __forceinline__ int pivot_row(double cand) { return int(__double_as_longlong(cand) & 0xFF); }
__device__ __forceinline__ bool pivot_usable(double cand) { return __longlong_as_double(__double_as_longlong(cand) & ~0xFFll) > 0.0; }

// In-place Gauss-Jordan inverse with partial pivoting of an M2 x M2 matrix (M2 = 16 T <= 128) by a team of 256 lanes that keeps the
// matrix in registers: element (r, c) lives in lane (r % 16, c % 16), slot (r / 16, c / 16).  Per pivot two barriers: the candidates of
// the pivot search, then the pivot row, the displaced row and the pivot column through `buf` (LDS, double-buffered by pivot parity); every
// lane then updates its T^2 elements with fused multiply-adds; the eliminated column is overwritten by the column of the inverse (zeroed
// before the update, with 1 / pivot standing in for the pivot row's entry).  The pivot row is wave-uniform (held in a scalar register), so
// picking the register slot that holds it is a scalar branch, not a select chain over the slots.  The row exchanges are undone as
// column exchanges when the result is written.  D is read, Dinv written (row-major, LDS or HBM); buf: 7 M2 + 32 doubles of LDS, which may lie
// INSIDE D / Dinv (the block lives in registers between the first and the last barrier of this function).
template <int T> __device__ bool gj_inverse_registers(const double* D, double* Dinv, double* buf) {
    constexpr int MP = 16 * T;
    const int lane = threadIdx.x, ly = lane >> 4, lx = lane & 15;
    double a[T][T];
#pragma unroll
    for (int rb = 0; rb < T; ++rb)
#pragma unroll
        for (int cb = 0; cb < T; ++cb) a[rb][cb] = D[(rb * 16 + ly) * MP + cb * 16 + lx];
    __syncthreads();   // buf may lie inside D (the block is in registers from here on): every lane has read its elements
    double* cand = buf + 6 * MP;                          // [2][16]
    int* ipiv = reinterpret_cast<int*>(buf + 6 * MP + 32);  // [MP]
    int* dst = ipiv + MP;                                  // [MP]
    bool ok = true;
#pragma unroll
    for (int kb = 0; kb < T; ++kb)
        for (int kl = 0; kl < 16 && ok; ++kl) {
            const int cc = kb * 16 + kl, par = cc & 1;
            double* rowp = buf + par * 3 * MP;   // [MP] row that becomes the pivot row (old row piv)
            double* rowc = rowp + MP;            // [MP] old row cc
            double* colf = rowp + 2 * MP;        // [MP] column cc
            double* cnd = cand + par * 16;
            // ---- pivot search over rows >= cc of column cc
            if (lx == kl) {
                double best = -1.0;
#pragma unroll
                for (int rb = 0; rb < T; ++rb) best = fmax(best, pivot_candidate(a[rb][kb], rb * 16 + ly, cc));
                cnd[ly] = best;
            }
            __syncthreads();
            double pvc = cnd[0];
#pragma unroll
            for (int t = 1; t < 16; ++t) pvc = fmax(pvc, cnd[t]);
            if (!pivot_usable(pvc)) { ok = false; break; }
            const int piv = __builtin_amdgcn_readfirstlane(pivot_row(pvc));
            const int prb = piv >> 4, pl = piv & 15;
            if (lane == 0) ipiv[cc] = piv;
            // ---- publish the two rows and the column
#pragma unroll
            for (int rb = 0; rb < T; ++rb)
                if (rb == prb && ly == pl) {
#pragma unroll
                    for (int cb = 0; cb < T; ++cb) rowp[cb * 16 + lx] = a[rb][cb];
                }
            if (ly == kl) {
#pragma unroll
                for (int cb = 0; cb < T; ++cb) rowc[cb * 16 + lx] = a[kb][cb];
            }
            if (lx == kl) {
#pragma unroll
                for (int rb = 0; rb < T; ++rb) {
                    colf[rb * 16 + ly] = a[rb][kb];
                    a[rb][kb] = 0.0;             // the column of the inverse grows here
                }
            }
            __syncthreads();
            // ---- update: every row loses f_r times the new pivot row; then row piv <- old row cc (eliminated), row cc <- pivot row
            const double inv = 1.0 / rowp[cc];
            double pa[T];
#pragma unroll
            for (int cb = 0; cb < T; ++cb) pa[cb] = rowp[cb * 16 + lx] * inv;
            if (lx == kl) pa[kb] = inv;
#pragma unroll
            for (int rb = 0; rb < T; ++rb) {
                const double f = colf[rb * 16 + ly];
#pragma unroll
                for (int cb = 0; cb < T; ++cb) a[rb][cb] -= f * pa[cb];
            }
#pragma unroll
            for (int rb = 0; rb < T; ++rb)
                if (rb == prb && ly == pl) {
                    const double f = colf[cc];   // the displaced row cc carries its own column-cc entry
#pragma unroll
                    for (int cb = 0; cb < T; ++cb) {
                        const double old = (cb == kb && lx == kl) ? 0.0 : rowc[cb * 16 + lx];
                        a[rb][cb] = old - f * pa[cb];
                    }
                }
            if (ly == kl) {
#pragma unroll
                for (int cb = 0; cb < T; ++cb) a[kb][cb] = pa[cb];
            }
        }
    __syncthreads();
    // ---- undo the row exchanges as column exchanges (k descending) while writing out
    if (lane == 0 && ok) {
        for (int c = 0; c < MP; ++c) dst[c] = c;
        for (int k = MP - 1; k >= 0; --k) {
            const int p = ipiv[k];
            if (p != k) { const int t = dst[k]; dst[k] = dst[p]; dst[p] = t; }
        }
        for (int c = 0; c < MP; ++c) ipiv[dst[c]] = c;   // final column of computed column
    }
    __syncthreads();
    int colof[T];      // (buf may lie inside Dinv: the final columns are read before the first element is written)
#pragma unroll
    for (int cb = 0; cb < T; ++cb) colof[cb] = ok ? ipiv[cb * 16 + lx] : 0;
    __syncthreads();
    if (ok) {
#pragma unroll
        for (int rb = 0; rb < T; ++rb)
#pragma unroll
            for (int cb = 0; cb < T; ++cb) Dinv[(rb * 16 + ly) * MP + colof[cb]] = a[rb][cb];
    }
    __syncthreads();
    return ok;
}

// In-place Gauss-Jordan inverse with partial pivoting of an M2 x M2 matrix (M2 = 8 TS) by ONE wavefront that keeps the matrix in
// registers: element (r, c) lives in lane (r % 8) * 8 + (c % 8), slot (r / 8, c / 8).  Pivot search, row exchange and the broadcast of
// the pivot row / column are wave shuffles, so a pivot costs no barrier at all (the 256-lane version above pays two workgroup barriers and
// several dependent LDS round trips per pivot at one wavefront per SIMD).  The row exchanges are undone as column exchanges when the
// result is written.  D is read, Dinv written (row-major); ibuf: 2 M2 ints of LDS / HBM scratch.  Call with the lanes of one wavefront.
template <int TS> __device__ bool gj_inverse_wave(const double* D, double* Dinv, int* ibuf) {
    constexpr int MP = 8 * TS;
    const int lane = threadIdx.x & 63, ly = lane >> 3, lx = lane & 7;
    double a[TS][TS];
#pragma unroll
    for (int rb = 0; rb < TS; ++rb)
#pragma unroll
        for (int cb = 0; cb < TS; ++cb) a[rb][cb] = D[(rb * 8 + ly) * MP + cb * 8 + lx];
    int* ipiv = ibuf;        // [MP]
    int* dst = ibuf + MP;    // [MP] final column of each computed column
    bool ok = true;
#pragma unroll
    for (int kb = 0; kb < TS; ++kb)
        for (int kl = 0; kl < 8 && ok; ++kl) {
            const int cc = kb * 8 + kl;
            // ---- pivot search in column cc (lanes lx == kl hold it), rows >= cc: maximum over the wavefront of the tagged magnitudes
            double best = -1.0;
#pragma unroll
            for (int rb = 0; rb < TS; ++rb) best = fmax(best, pivot_candidate(a[rb][kb], rb * 8 + ly, cc));
            if (lx != kl) best = -1.0;
#pragma unroll
            for (int off = 1; off <= 32; off <<= 1) best = fmax(best, __shfl_xor(best, off));
            if (!pivot_usable(best)) { ok = false; break; }
            const int piv = __builtin_amdgcn_readfirstlane(pivot_row(best));
            if (lane == 0) ipiv[cc] = piv;
            const int prb = piv >> 3, pl = piv & 7;
            // ---- exchange rows cc <-> piv; every lane ends up with the new pivot row's entries of its columns in prow[]
            double prow[TS], mine_p[TS] = {};
#pragma unroll
            for (int rb = 0; rb < TS; ++rb)
                if (rb == prb) {
#pragma unroll
                    for (int cb = 0; cb < TS; ++cb) mine_p[cb] = a[rb][cb];
                }
#pragma unroll
            for (int cb = 0; cb < TS; ++cb) prow[cb] = __shfl(mine_p[cb], (pl << 3) | lx);   // old row piv = new pivot row
            if (piv != cc) {
                double xc[TS];
#pragma unroll
                for (int cb = 0; cb < TS; ++cb) xc[cb] = __shfl(a[kb][cb], (kl << 3) | lx);   // old row cc
                if (ly == kl) {
#pragma unroll
                    for (int cb = 0; cb < TS; ++cb) a[kb][cb] = prow[cb];
                }
#pragma unroll
                for (int rb = 0; rb < TS; ++rb)
                    if (rb == prb && ly == pl) {
#pragma unroll
                        for (int cb = 0; cb < TS; ++cb) a[rb][cb] = xc[cb];
                    }
            }
            const double pv = __shfl(prow[kb], (lane & 56) | kl);
            const double inv = 1.0 / pv;
#pragma unroll
            for (int cb = 0; cb < TS; ++cb) prow[cb] *= inv;
            // ---- eliminate: row r <- row r - f_r (new row cc), column cc -> -f_r / pivot; then row cc <- row / pivot, its column-cc entry 1 / pivot
#pragma unroll
            for (int rb = 0; rb < TS; ++rb) {
                const double f = __shfl(a[rb][kb], (lane & 56) | kl);   // a[r][cc] after the exchange
#pragma unroll
                for (int cb = 0; cb < TS; ++cb) a[rb][cb] -= f * prow[cb];
                if (lx == kl) a[rb][kb] = -f * inv;
            }
            if (ly == kl) {
#pragma unroll
                for (int cb = 0; cb < TS; ++cb) a[kb][cb] = prow[cb];
                if (lx == kl) a[kb][kb] = inv;
            }
        }
    // ---- undo the row exchanges as column exchanges (k descending) while writing out
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0 && ok) {
        for (int c = 0; c < MP; ++c) dst[c] = c;   // dst[c] = computed column that ends at final column c (built as src), inverted below
        for (int k = MP - 1; k >= 0; --k) {
            const int p = ipiv[k];
            if (p != k) { const int t = dst[k]; dst[k] = dst[p]; dst[p] = t; }
        }
        // invert: ipiv is free now, reuse it as final column of computed column
        for (int c = 0; c < MP; ++c) ipiv[dst[c]] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (ok) {
#pragma unroll
        for (int rb = 0; rb < TS; ++rb)
#pragma unroll
            for (int cb = 0; cb < TS; ++cb) Dinv[(rb * 8 + ly) * MP + ipiv[cb * 8 + lx]] = a[rb][cb];
    }
    return ok;
}
#endif

// =================================================================================================
// KV2: block-tridiagonal boundary value problem, factor + solve                 (task = wavelength, one team)
//   same block rows as skd::k2a_bvp with 2n x 2n blocks; team-shared work space: ws (8 n^2 + 10 n doubles: the block and its
//   inverse, vectors) and t1 (2 n^2 doubles), which may live in different memories (LDS / HBM)
// =================================================================================================
//   with a register-resident inversion (GJ != 0) the inverse overwrites the block: `inplace` halves the matrix part
//   `gjalias` (GJ == 2): the exchange buffers of the 256-lane inversion lie inside the block's own storage, which is dead while the block is in
//   registers -- 4 n^2 + 10 n + 64 doubles = 76.3 KB at n = 48: with T1 in HBM / L2 TWO teams share a CU (the inversion is a chain of 96 pivots
//   with two barriers each at one wavefront per SIMD; a second team fills the gaps)
SKD_HD size_t kv2_scratch_main(int n, bool inplace = false, bool gjalias = false) { return (inplace ? 4 : 8) * size_t(n) * n + (gjalias ? 10 : 30) * size_t(n) + 64; }
SKD_HD size_t kv2_scratch_t1(int n) { return 2 * size_t(n) * n; }
SKD_HD size_t kv2_scratch(int n) { return kv2_scratch_main(n) + kv2_scratch_t1(n); }

// GJ selects the block inversion: 0 = in place in the work space (any size, any team), 1 = one wavefront's registers (M2 = 48, team >= 64
// lanes), 2 = the registers of a 256-lane team (M2 = 96); a separate instantiation each, so the small-block kernels do not carry the
// register footprint of the 96 x 96 one (occupancy)
template <int GJ = 0> SKV_TEAM_BODY void kv2_bvp(const Team& team, const Geo& g, const Bufs& b, const VBufs& v, int s, int m, int w, double* ws, double* T1) {
    const int N = g.N, S = v.S, n = v.n, M2 = 2 * n, nw = b.nw, nlyr = g.nlyr;
    const VRec R(n);
    const bool brdf = v.brdf_kind != 0;                          // non-Lambertian surface: every order carries a ground reflection (VSurf)
    const double rho = (m == 0 && !brdf) ? b.albedo[w] : 0.0;   // Lambertian
    const double* sf = brdf ? vsurf_at(v, g, m, w) : nullptr;
    const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
    const double kd = 2.0;
    const int mm = M2 * M2;
    // the register-resident inversions read the whole block before they write the inverse, and the block of the next layer is only built after
    // the product with the previous inverse: block and inverse share their storage then
    constexpr bool inplace = (GJ != 0);
    double *D = ws, *Dinv = inplace ? ws : ws + mm, *r = ws + (inplace ? 1 : 2) * mm, *rprev = r + M2, *xn = rprev + M2, *fcol = xn + M2, *refl = fcol + M2;  // refl: 2n
    double* gjbuf = (GJ == 2) ? ws : refl + 2 * n;  // exchange buffers of the register-resident inversions (GJ == 2: inside the block's storage, see kv2_scratch_main)
#if defined(SKV_K2_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#define KV2T(i) { const long long t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define KV2T(i)
#endif
    bool singular = false;
    for (int p = 0; p < nlyr; ++p) {
        const double* c = vrec_at(v, nlyr, p, w);
        const double* cprev = (p > 0) ? vrec_at(v, nlyr, p - 1, w) : c;
        team.sync();
#ifndef SKV_EXP_SKIP_T1
        if (p > 0) {
            // T1 = Sub_p Dinv_{p-1},  Sub_p rows A = [W+_{p-1} E_{p-1} | W-_{p-1}]   (Dinv still holds layer p - 1)
            // stored transposed, T1[j][i], with the lanes running over i: coalesced reads of W+-_{p-1}, broadcast reads of Dinv
            team_gemm(team, n, M2, M2,
                      [&](int i, int k) { return k < n ? cprev[R.WP() + i + n * k] * cprev[R.E() + k] : cprev[R.WM() + i + n * (k - n)]; },
                      [&](int k, int j) { return Dinv[k * M2 + j]; }, [&](int i, int j, double val) { T1[j * n + i] = val; });
        }
#endif
        KV2T(0)
        if (p == nlyr - 1 && m == 0) {
            // Lambertian reflection sums over the intensity rows of the downwelling field
            SKV_PAR(j, n) {
                double sp = 0, sm = 0;
                for (int t = 0; t < N; ++t) {
                    sp += g.wt[t] * g.mu[t] * c[R.WP() + t * S + n * j];
                    sm += g.wt[t] * g.mu[t] * c[R.WM() + t * S + n * j];
                }
                refl[j] = kd * rho * sp * c[R.E() + j];
                refl[n + j] = kd * rho * sm;
            }
        }
        team.sync();
        SKV_PAR(e, n * n) {
            const int i = e % n, j = e / n;   // lanes over i: coalesced W+-(i, j), consecutive T1[.][i], broadcast W+-(q, j)
            const double wp = c[R.WP() + i + n * j], wm = c[R.WM() + i + n * j], Ej = c[R.E() + j];
            double d00 = -wp, d01 = -wm * Ej, d10 = wm * Ej, d11 = wp;
            if (p == nlyr - 1 && m == 0 && (i % S) == 0 && !brdf) {
                d10 -= refl[j];
                d11 -= refl[n + j];
            }
            if (p == nlyr - 1 && brdf && (i % S) == 0) {
                double sp = 0, sm = 0;
                for (int t = 0; t < N; ++t) {
                    sp += sf[SF.RS(i / S, t)] * c[R.WP() + t * S + n * j];
                    sm += sf[SF.RS(i / S, t)] * c[R.WM() + t * S + n * j];
                }
                d10 -= sp * Ej;
                d11 -= sm;
            }
            D[i * M2 + j] = d00;
            D[i * M2 + n + j] = d01;
            D[(n + i) * M2 + j] = d10;
            D[(n + i) * M2 + n + j] = d11;
        }
#ifndef SKV_EXP_SKIP_T1
        if (p > 0) {
            // rows A += T1[:, n:] [W-_p | W+_p E_p]
            team.sync();
            team_gemm(team, n, M2, n, [&](int i, int k) { return T1[(n + k) * n + i]; },
                      [&](int k, int j) { return j < n ? c[R.WM() + k + n * j] : c[R.WP() + k + n * (j - n)] * c[R.E() + (j - n)]; },
                      [&](int i, int j, double val) { D[i * M2 + j] += val; });
        }
#endif
        SKV_PAR(i, n) {
            double ra = c[R.G() + 0 * n + i], rb = -c[R.G() + 3 * n + i];
            if (p == nlyr - 1) {
                if (m == 0 && (i % S) == 0 && !brdf) {
                    double sg = 0;
                    for (int t = 0; t < N; ++t) sg += g.wt[t] * g.mu[t] * c[R.G() + 2 * n + t * S];
                    // (+ surface emission in the thermal pass, as skd::k2a_bvp)
                    rb += g.csz[s] * rho / kPi * b.T[size_t(nlyr) * nw + w] + (b.surf_em ? b.surf_em[w] : 0.0) + kd * rho * sg;
                }
                if (brdf && (i % S) == 0) {
                    double sg = 0;
                    for (int t = 0; t < N; ++t) sg += sf[SF.RS(i / S, t)] * c[R.G() + 2 * n + t * S];
                    rb += sf[SF.BS(i / S)] * b.T[size_t(nlyr) * nw + w] + sg;
                }
            } else {
                rb += vrec_at(v, nlyr, p + 1, w)[R.G() + 1 * n + i];
            }
            if (p > 0) {
                ra -= cprev[R.G() + 2 * n + i] + sum4(0, M2, [&](int q) { return T1[q * n + i] * rprev[q]; });
            }
            r[i] = ra;
            r[n + i] = rb;
        }
        team.sync();
        KV2T(1)
        // Gauss-Jordan inverse with partial pivoting: register-resident for 256-lane teams and 48 x 48 / 96 x 96 blocks, else in place
        bool inverted = false;
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef SKV_EXP_SKIP_GJ
        if (true) { SKV_PAR(e, mm) Dinv[e] = (e / M2 == e % M2) ? 1.0 : 0.0; inverted = true; } else
#endif
        if constexpr (GJ == 1) {
            // one wavefront inverts the 48 x 48 block in its registers with shuffles (no barrier per pivot), the others wait
            __shared__ int gj_ok;
            if (team.lane < 64) {
                const bool ok = gj_inverse_wave<6>(D, Dinv, reinterpret_cast<int*>(gjbuf));
                if (team.lane == 0) gj_ok = ok ? 1 : 0;
            }
            team.sync();
            if (!gj_ok) singular = true;
            inverted = true;
        } else if constexpr (GJ == 2) {
            // 96 x 96: the block and its inverse in the registers of all 256 lanes
            if (!gj_inverse_registers<6>(D, Dinv, gjbuf)) singular = true;
            inverted = true;
        }
#endif
        if (!inverted) SKV_PAR(e, mm) Dinv[e] = (e / M2 == e % M2) ? 1.0 : 0.0;
        team.sync();
        for (int cc = 0; cc < M2 && !inverted; ++cc) {
            int piv = cc;
            double best = fabs(D[cc * M2 + cc]);
            for (int rr = cc + 1; rr < M2; ++rr) {
                const double a_ = fabs(D[rr * M2 + cc]);
                if (a_ > best) { best = a_; piv = rr; }
            }
            if (!(best > 0)) { singular = true; break; }
            const double inv = 1.0 / D[piv * M2 + cc];
            team.sync();
            SKV_PAR(j, 2 * M2) {  // swap rows cc <-> piv and scale the pivot row (D for j < M2, Dinv for j >= M2)
                double* Mx = (j < M2) ? D : Dinv;
                const int jj = (j < M2) ? j : j - M2;
                const double a_ = Mx[cc * M2 + jj], b_ = Mx[piv * M2 + jj];
                Mx[piv * M2 + jj] = a_;
                Mx[cc * M2 + jj] = b_ * inv;
            }
            team.sync();
            SKV_PAR(rr, M2) fcol[rr] = (rr == cc) ? 0.0 : D[rr * M2 + cc];
            team.sync();
            SKV_PAR(e, 2 * mm) {
                double* Mx = (e < mm) ? D : Dinv;
                const int ee = (e < mm) ? e : e - mm;
                const int rr = ee / M2, j = ee % M2;
                const double f = fcol[rr];
                if (f != 0.0) Mx[ee] -= f * Mx[cc * M2 + j];
            }
            team.sync();
        }
        KV2T(2)
        double* dst = vdinv_at(v, nlyr, p, w);
        double* xs = vxsol_at(v, nlyr, p, w);
        SKV_PAR(e, mm) dst[e] = Dinv[e];
        SKV_PAR(i, M2) {
            rprev[i] = r[i];
            xs[i] = r[i];
        }
    }
    KV2T(3)
    team.sync();
    if (singular && team.lane == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.status[w], skd::ERR_SINGULAR_BVP);
#else
        b.status[w] |= skd::ERR_SINGULAR_BVP;
#endif
    }
    // back substitution:  x_p = Dinv_p (r'_p - Sup_p x_{p+1})
    for (int p = nlyr - 1; p >= 0; --p) {
        double* xs = vxsol_at(v, nlyr, p, w);
        const double* dinv = vdinv_at(v, nlyr, p, w);
        team.sync();
        SKV_PAR(i, M2) {
            double t = xs[i];
            if (p < nlyr - 1 && i >= n) {
                const double* cn = vrec_at(v, nlyr, p + 1, w);
                const int ii = i - n;
                t += sum4(0, n, [&](int j) { return cn[R.WM() + ii + n * j] * xn[j] + cn[R.WP() + ii + n * j] * cn[R.E() + j] * xn[n + j]; });
            }
            r[i] = t;
        }
        team.sync();
        SKV_PAR(i, M2) {
            const double t = sum4(0, M2, [&](int j) { return dinv[i * M2 + j] * r[j]; });
            xn[i] = t;
            xs[i] = t;
        }
    }
    team.sync();
    KV2T(4)
#if defined(SKV_K2_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    if (team.lane == 0 && w == 0 && m == 1)
        printf("KV2T n %d lanes %d: T1 %lld D+update+r %lld inverse %lld store %lld backsub %lld\n", n, team.width, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
}

// Legendre-space source vectors of a layer:  J(out)[s] = sum_l [Pi_l(out) v_l][s],
//   v_l = B_l [ (-1)^(l-m) D cP_l + cQ_l ],  cP_l = sum_j (w_j / 2) Pi_l(mu_j) P_j,  cQ_l likewise with Q_j, where
//   P_r = sum_a W+(r,a) cL_a + W-(r,a) cM_a and Q_r = sum_a W-(r,a) cL_a + W+(r,a) cM_a  (3-vectors per stream)
// vl: [nstr][3] output; pq: scratch of 2 n doubles
SKD_HD void legendre_source(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int m, int p, int w, const double* rec, Scr cL, Scr cM,
                            Scr pq, Scr vl) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr;
    const VRec R(n);
    for (int r = 0; r < n; ++r) {
        double pr = 0, qr = 0;
        for (int a = 0; a < n; ++a) {
            const double wp = rec[R.WP() + r + n * a], wm = rec[R.WM() + r + n * a];
            pr += wp * cL[a] + wm * cM[a];
            qr += wm * cL[a] + wp * cM[a];
        }
        pq[r] = pr * 0.5 * g.wt[r / S];
        pq[n + r] = qr * 0.5 * g.wt[r / S];
    }
    for (int l = 0; l < nstr; ++l) {
        double cp[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
        if (l >= m) {
            for (int j = 0; j < N; ++j) {
                const Fun3 fj = fun_mu(g, vg, m, j);
                if (S == 1) {
                    cp[0] += fj.P[l] * pq[j];
                    cq[0] += fj.P[l] * pq[n + j];
                } else {
                    const double xp[3] = {pq[3 * j], pq[3 * j + 1], pq[3 * j + 2]}, xq[3] = {pq[n + 3 * j], pq[n + 3 * j + 1], pq[n + 3 * j + 2]};
                    double y[3];
                    pi_apply(fj.P[l], fj.R[l], fj.T[l], xp, y);
                    for (int c = 0; c < 3; ++c) cp[c] += y[c];
                    pi_apply(fj.P[l], fj.R[l], fj.T[l], xq, y);
                    for (int c = 0; c < 3; ++c) cq[c] += y[c];
                }
            }
        }
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        const double x[3] = {sg * cp[0] + cq[0], sg * cp[1] + cq[1], -sg * cp[2] + cq[2]};
        double c4[4], y[3];
        load_coef(b, v, nstr, p, l, w, c4);
        greek_apply(c4, x, y);
        for (int c = 0; c < 3; ++c) vl[size_t(l) * 3 + c] = y[c];
    }
}

// =================================================================================================
// KV3: diffuse source table values J_m (3 Stokes components) + convergence bookkeeping   (task = wavelength x layer)
//   reference: do_source_diffuse_storage.cpp:435-695,697-1098 (values only); convergence on the intensity (:1033-1060)
//   scratch: 4 n + 3 nstr doubles per thread
// =================================================================================================
SKD_HD size_t kv3_scratch(int n, int nstr) { return 4 * size_t(n) + 3 * size_t(nstr); }

SKD_HD void kv3_table_values(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int p, int w, Scr scr) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr;
    const VRec R(n);
    const double* c = vrec_at(v, nlyr, p, w);
    const double* xs = vxsol_at(v, nlyr, p, w);
    const double th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    Scr cL = scr.sub(0), cM = scr.sub(n), pq = scr.sub(2 * n), vl = scr.sub(4 * n);
    for (int j = 0; j < n; ++j) {
        skd::PPMult pm;
        skd::pp_multipliers(T, th, sc, c[R.K() + j], g.lfrac[p], pm);
        if (v.derivs) skd::ppm_store(vrec_at(v, nlyr, p, w) + R.PPM() + 14 * j, pm);
        cL[j] = pm.hp * xs[j] + c[R.AP() + j] * pm.Dm;
        cM[j] = pm.hm * xs[n + j] + c[R.AM() + j] * pm.Dp;
    }
    legendre_source(g, vg, b, v, m, p, w, c, cL, cM, pq, vl);
    int unconverged = 0;
    const int t0 = g.pt_start[s * nlyr + p], t1 = g.pt_start[s * nlyr + p + 1];
    for (int t = t0; t < t1; ++t) {
        const int slot = g.pt_slot[t];
        if (skd::conv_at(b, g, slot, w) <= m) continue;
        const Fun3 fa = fun_ang(g, vg, m, g.pt_angle[t]);
        double val[3] = {0, 0, 0};
        for (int l = m; l < nstr; ++l) {
            if (S == 1) {
                val[0] += fa.P[l] * vl[size_t(l) * 3];
            } else {
                const double x[3] = {vl[size_t(l) * 3], vl[size_t(l) * 3 + 1], vl[size_t(l) * 3 + 2]};
                double y[3];
                pi_apply(fa.P[l], fa.R[l], fa.T[l], x, y);
                for (int cidx = 0; cidx < 3; ++cidx) val[cidx] += y[cidx];
            }
        }
        // (the 1/omega of the table and the omega inside the phase sums cancel)
        for (int cidx = 0; cidx < S; ++cidx) vJ_at(v, g, m, slot, w, cidx) = val[cidx];
        if (m >= 2) {
            const double prev = vJ_at(v, g, m - 1, slot, w, 0), prev2 = vJ_at(v, g, m - 2, slot, w, 0);
            if ((fabs(val[0] / prev) < 1e-4 || prev < 1e-10) && (fabs(val[0] / prev2) < 1e-4 || prev2 < 1e-10)) {
                skd::conv_at(b, g, slot, w) = (uint8_t)(m + 1);
            } else {
                unconverged = 1;
            }
        }
    }
    if (p == nlyr - 1 && v.brdf_kind != 0) {
        // non-Lambertian surface: the upwelling reflected intensity at every needed cos-angle grid point k, in every order
        // (accumulate_ground_sources, do_source_diffuse_storage.cpp:435-695): J_m(k) = sum_t RL(k, t) I-_t
        const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
        const double* sf = vsurf_at(v, g, m, w);
        for (int t = g.gpt_start[s]; t < g.gpt_start[s + 1]; ++t) {
            const int k = g.gpt_slot[t] - v.surf_slot0;
            double diffuse = 0;
            for (int i = 0; i < N; ++i) {
                double sv = c[R.G() + 2 * n + i * S];
                for (int j = 0; j < n; ++j)
                    sv += xs[j] * c[R.WP() + i * S + n * j] * c[R.E() + j] + xs[n + j] * c[R.WM() + i * S + n * j];
                diffuse += sf[SF.RL(k, i)] * sv;
            }
            vJ_at(v, g, m, g.gpt_slot[t], w, 0) = diffuse;
            for (int cidx = 1; cidx < S; ++cidx) vJ_at(v, g, m, g.gpt_slot[t], w, cidx) = 0;
        }
    } else if (p == nlyr - 1 && m == 0) {
        // upwelling diffuse reflection of a Lambertian surface (intensity only)
        double diffuse = 0;
        for (int i = 0; i < N; ++i) {
            double sv = c[R.G() + 2 * n + i * S];
            for (int j = 0; j < n; ++j)
                sv += xs[j] * c[R.WP() + i * S + n * j] * c[R.E() + j] + xs[n + j] * c[R.WM() + i * S + n * j];
            diffuse += 2.0 * g.mu[i] * g.wt[i] * sv * b.albedo[w];
        }
        for (int t = g.gpt_start[s]; t < g.gpt_start[s + 1]; ++t) {
            vJ_at(v, g, m, g.gpt_slot[t], w, 0) = diffuse;
            for (int cidx = 1; cidx < S; ++cidx) vJ_at(v, g, m, g.gpt_slot[t], w, cidx) = 0;
        }
    }
    if (unconverged) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.notconv[w], 1);
#else
        b.notconv[w] |= 1;
#endif
    }
}

// The same task by a TEAM (one wavefront per (layer, wavelength) on the device): the lanes run over the solutions, the rows of W+-, the
// Legendre orders and the layer's table points in turn, so that the record is read in unit-stride runs across the lanes -- the per-thread form
// above walks its own 90 KB record (n = 48) with every lane on another cache line and moved 4 x the record's bytes.  Every output element is
// summed in the order of the per-thread form: bit-identical results.   ws: kv3_scratch(n, nstr) doubles of team-shared work space
SKV_TEAM_BODY void kv3_table_values_team(const Team& team, const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int p, int w, double* ws) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr;
    const VRec R(n);
    const double* c = vrec_at(v, nlyr, p, w);
    const double* xs = vxsol_at(v, nlyr, p, w);
    const double th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    double *cL = ws, *cM = ws + n, *pq = ws + 2 * n, *vl = ws + 4 * n;
    SKV_PAR(j, n) {
        skd::PPMult pm;
        skd::pp_multipliers(T, th, sc, c[R.K() + j], g.lfrac[p], pm);
        if (v.derivs) skd::ppm_store(vrec_at(v, nlyr, p, w) + R.PPM() + 14 * j, pm);
        cL[j] = pm.hp * xs[j] + c[R.AP() + j] * pm.Dm;
        cM[j] = pm.hm * xs[n + j] + c[R.AM() + j] * pm.Dp;
    }
    team.sync();
    SKV_PAR(r, n) {   // (legendre_source, first half: lanes over the rows r, W+-(r, a) contiguous in r)
        double pr = 0, qr = 0;
        for (int a = 0; a < n; ++a) {
            const double wp = c[R.WP() + r + n * a], wm = c[R.WM() + r + n * a];
            pr += wp * cL[a] + wm * cM[a];
            qr += wm * cL[a] + wp * cM[a];
        }
        pq[r] = pr * 0.5 * g.wt[r / S];
        pq[n + r] = qr * 0.5 * g.wt[r / S];
    }
    team.sync();
    SKV_PAR(l, nstr) {   // (second half: lanes over the Legendre orders)
        double cp[3] = {0, 0, 0}, cq[3] = {0, 0, 0};
        if (l >= m) {
            for (int j = 0; j < N; ++j) {
                const Fun3 fj = fun_mu(g, vg, m, j);
                if (S == 1) {
                    cp[0] += fj.P[l] * pq[j];
                    cq[0] += fj.P[l] * pq[n + j];
                } else {
                    const double xp[3] = {pq[3 * j], pq[3 * j + 1], pq[3 * j + 2]}, xq[3] = {pq[n + 3 * j], pq[n + 3 * j + 1], pq[n + 3 * j + 2]};
                    double y[3];
                    pi_apply(fj.P[l], fj.R[l], fj.T[l], xp, y);
                    for (int cc = 0; cc < 3; ++cc) cp[cc] += y[cc];
                    pi_apply(fj.P[l], fj.R[l], fj.T[l], xq, y);
                    for (int cc = 0; cc < 3; ++cc) cq[cc] += y[cc];
                }
            }
        }
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        const double x[3] = {sg * cp[0] + cq[0], sg * cp[1] + cq[1], -sg * cp[2] + cq[2]};
        double c4[4], y[3];
        load_coef(b, v, nstr, p, l, w, c4);
        greek_apply(c4, x, y);
        for (int cc = 0; cc < 3; ++cc) vl[size_t(l) * 3 + cc] = y[cc];
    }
    team.sync();
    int unconverged = 0;
    const int t0 = g.pt_start[s * nlyr + p], t1 = g.pt_start[s * nlyr + p + 1];
    SKV_PAR(tt, t1 - t0) {
        const int t = t0 + tt, slot = g.pt_slot[t];
        if (skd::conv_at(b, g, slot, w) <= m) continue;
        const Fun3 fa = fun_ang(g, vg, m, g.pt_angle[t]);
        double val[3] = {0, 0, 0};
        for (int l = m; l < nstr; ++l) {
            if (S == 1) {
                val[0] += fa.P[l] * vl[size_t(l) * 3];
            } else {
                const double x[3] = {vl[size_t(l) * 3], vl[size_t(l) * 3 + 1], vl[size_t(l) * 3 + 2]};
                double y[3];
                pi_apply(fa.P[l], fa.R[l], fa.T[l], x, y);
                for (int cidx = 0; cidx < 3; ++cidx) val[cidx] += y[cidx];
            }
        }
        for (int cidx = 0; cidx < S; ++cidx) vJ_at(v, g, m, slot, w, cidx) = val[cidx];
        if (m >= 2) {
            const double prev = vJ_at(v, g, m - 1, slot, w, 0), prev2 = vJ_at(v, g, m - 2, slot, w, 0);
            if ((fabs(val[0] / prev) < 1e-4 || prev < 1e-10) && (fabs(val[0] / prev2) < 1e-4 || prev2 < 1e-10)) {
                skd::conv_at(b, g, slot, w) = (uint8_t)(m + 1);
            } else {
                unconverged = 1;
            }
        }
    }
    if (p == nlyr - 1 && v.brdf_kind != 0) {
        const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
        const double* sf = vsurf_at(v, g, m, w);
        SKV_PAR(tt, g.gpt_start[s + 1] - g.gpt_start[s]) {
            const int t = g.gpt_start[s] + tt, k = g.gpt_slot[t] - v.surf_slot0;
            double diffuse = 0;
            for (int i = 0; i < N; ++i) {
                double sv = c[R.G() + 2 * n + i * S];
                for (int j = 0; j < n; ++j)
                    sv += xs[j] * c[R.WP() + i * S + n * j] * c[R.E() + j] + xs[n + j] * c[R.WM() + i * S + n * j];
                diffuse += sf[SF.RL(k, i)] * sv;
            }
            vJ_at(v, g, m, g.gpt_slot[t], w, 0) = diffuse;
            for (int cidx = 1; cidx < S; ++cidx) vJ_at(v, g, m, g.gpt_slot[t], w, cidx) = 0;
        }
    } else if (p == nlyr - 1 && m == 0 && team.lane == 0) {
        double diffuse = 0;
        for (int i = 0; i < N; ++i) {
            double sv = c[R.G() + 2 * n + i * S];
            for (int j = 0; j < n; ++j)
                sv += xs[j] * c[R.WP() + i * S + n * j] * c[R.E() + j] + xs[n + j] * c[R.WM() + i * S + n * j];
            diffuse += 2.0 * g.mu[i] * g.wt[i] * sv * b.albedo[w];
        }
        for (int t = g.gpt_start[s]; t < g.gpt_start[s + 1]; ++t) {
            vJ_at(v, g, m, g.gpt_slot[t], w, 0) = diffuse;
            for (int cidx = 1; cidx < S; ++cidx) vJ_at(v, g, m, g.gpt_slot[t], w, cidx) = 0;
        }
    }
    if (unconverged) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.notconv[w], 1);
#else
        b.notconv[w] |= 1;
#endif
    }
}

// =================================================================================================
// KV4: line-of-sight integration of the diffuse table, 3 Stokes components        (task = wavelength x LOS)
//   azimuth factors cos(m azi) for I and Q, sin(m azi) for U (do_source_diffuse_storage.cpp:158-184); scalar attenuation
//   (sourceintegrator.cpp:521-580); Lambertian ground feeds the intensity only (do_source_interpolated_pp.cpp:176-211)
// =================================================================================================
SKD_HD void kv4_los(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int los, int w) {
    const int nw = b.nw, nstr = g.nstr, S = v.S;
    const size_t tid = size_t(w) * g.nlos + los;
    double behind[3] = {0, 0, 0};
    if (g.ray_ground[los]) {
        for (int t = g.g_start[los]; t < g.g_start[los + 1]; ++t) {
            const int slot = g.g_slot[t];
            if (v.brdf_kind != 0) {   // every order with its azimuth factor (the ground source of a BRDF depends on the outgoing direction)
                const int cv = int(skd::conv_at(b, g, slot, w)), mmax = (cv < nstr) ? cv : nstr;
                for (int m = 0; m < mmax; ++m) behind[0] += g.g_w[t] * cos(m * (kPi - g.ray_gsaa[los])) * vJ_at(v, g, m, slot, w, 0);
            } else
            if (skd::conv_at(b, g, slot, w) > 0) behind[0] += g.g_w[t] * vJ_at(v, g, 0, slot, w, 0);  // order 0 only
        }
    }
    const double att_total = b.att_total[size_t(los) * nw + w];
    const size_t nt3 = size_t(nw) * g.nlos * S;
    double prod_upto = 1.0;  // prod_{q' <= q} att_q'
    for (int q = g.ray_qstart[los]; q < g.ray_qstart[los + 1]; ++q) {
        const double att = exp(-b.od[size_t(q) * nw + w]);
        prod_upto *= att;
        double sv[3] = {0, 0, 0};
        if (g.rl_active[q]) {
            const double* cs = g.rl_cosm_t + size_t(q) * nstr;
            const double* sn = vg.rl_sinm_t + size_t(q) * nstr;
            for (int t = g.rl_tab_start[q]; t < g.rl_tab_start[q + 1]; ++t) {
                const int slot = g.tab_slot[t];
                const int cv = int(skd::conv_at(b, g, slot, w));
                const int mmax = (cv < nstr) ? cv : nstr;
                double a[3] = {0, 0, 0};
                for (int m = 0; m < mmax; ++m)
                    for (int c = 0; c < S; ++c) a[c] += (c < 2 ? cs[m] : sn[m]) * vJ_at(v, g, m, slot, w, c);
                for (int c = 0; c < S; ++c) sv[c] += g.tab_w[t] * a[c];
            }
        }
        const double omega = b.omloc[size_t(q) * nw + w];
        if (v.derivs) {
            // LOS-level derivative terms (shell optical depth, local single-scatter albedo), as skd::k4b_los
            const double a_q = (prod_upto > 0) ? att_total / prod_upto : 0.0;
            for (int c = 0; c < S; ++c) {
                const size_t t3 = tid * S + c;
                const double dI_dod = a_q * (-behind[c] * att + omega * att * sv[c]);
                for (int t = g.rl_od_start[q]; t < g.rl_od_start[q + 1]; ++t) v.dnat[size_t(g.od_idx[t]) * nt3 + t3] += g.od_w[t] * dI_dod;
                if (g.rl_active[q]) {
                    const double dI_dom = a_q * (1 - att) * sv[c];
                    for (int t = g.rl_ssa_start[q]; t < g.rl_ssa_start[q + 1]; ++t)
                        v.dnat[size_t(g.ngrid + g.ssa_idx[t]) * nt3 + t3] += g.ssa_w[t] * dI_dom;
                }
            }
        }
        for (int c = 0; c < S; ++c) behind[c] = behind[c] * att + (g.rl_active[q] ? omega * (1 - att) * sv[c] : 0.0);
    }
    for (int c = 0; c < S; ++c) v.radiance[tid * S + c] = behind[c];
}

// =================================================================================================
// Weighting functions of the polarised path: the adjoint machinery of skd::k2c_layer_ops / skd::K2bTask with N -> n and the
// Legendre space widened to NL = C nstr numbers per layer, index (l, c), c < C = 3 (1 for NSTOKES = 1): the source of a table
// point is  J(out)[s] = sum_l [Pi_l(out) v_l][s]  with  v_l = om B_l [ (-1)^(l-m) D tP_l + tM_l ]  linear in (cL, cM) = the layer's
// BVP coefficients times multipliers, so one adjoint sweep per (line of sight, Stokes component) yields the reduced gradient wrt
// (tau, omega, T, sec) of every layer; skd::k4a_chain then maps it to the native derivatives.
// =================================================================================================
struct VOps {
    int n, NL, G;
    SKD_HD VOps(int n_, int NL_, int G_ = 0) : n(n_), NL(NL_), G(G_) {}
    SKD_HD int FP() const { return 0; }                        // [2n][NL]
    SKD_HD int SP() const { return 2 * n * NL; }               // [2n][n]
    SKD_HD int BP() const { return SP() + 2 * n * n; }         // [2n][n]
    SKD_HD int UG() const { return BP() + 2 * n * n; }         // [2n]
    SKD_HD int DI() const { return UG() + 2 * n; }             // [(theta*4 + edge)][n]
    SKD_HD int HD() const { return DI() + 16 * n; }            // [4][NL]
    SKD_HD int GD() const { return HD() + 4 * NL; }            // [4]
    SKD_HD int IPB() const { return GD() + 4; }
    SKD_HD int GDI(int g) const { return IPB() + 1 + g * (4 * n + NL + 1); }   // [4 edges][n] of group g
    SKD_HD int GHD(int g) const { return GDI(g) + 4 * n; }                   // [NL]
    SKD_HD int GGD(int g) const { return GHD(g) + NL; }
    SKD_HD int size() const { return GDI(G); }
};
SKD_HD int legendre_space(const Geo& g, int S) { return (S == 3 ? 3 : 1) * g.nstr; }
SKD_HD double* vops_at(const VBufs& v, const Geo& g, int p, int w) {
    return v.ops + (size_t(w) * g.nlyr + p) * VOps(v.n, legendre_space(g, v.S), v.ngroups).size();
}

// KV2C: adjoint operators of one layer                        (task = wavelength x layer, one team)
//   ws: 2n NL + 4 n + 5 NL n doubles of team-shared work space (F, temporaries, partial sums of the HD rows)
SKD_HD size_t kv2c_scratch(int n, int NL, int G = 0) { return 2 * size_t(n) * NL + 4 * size_t(n) + (5 + size_t(G)) * size_t(NL) * n; }
// the same for a team of `width` lanes: it cuts the solutions into K = width / nstr slices (at most n), not into n
SKD_HD size_t kv2c_scratch_team(int n, int NL, int G, int width, int nstr) {
    const int K = (width / nstr) < 1 ? 1 : ((width / nstr) > n ? n : (width / nstr));
    return 2 * size_t(n) * NL + 4 * size_t(n) + (5 + size_t(G)) * size_t(NL) * K;
}

//   wst (optional, LDS, 4 n^2 doubles): W+, W-, dW+ + W+ / om, dW- + W- / om of the layer staged once with unit-stride loads -- the Legendre sums
//   below then read their (stream, solution) entries as LDS broadcasts instead of global ones (and without the two divisions per entry and lane)
SKV_TEAM_BODY void kv2c_layer_ops(const Team& team, const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int p, int w, double* ws,
                                  double* wst = nullptr) {
#if defined(SKV_K2C_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#define KV2CT(i) { const long long t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; }
#else
#define KV2CT(i)
#endif
    const int N = g.N, S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr;
    const VRec R(n);
    const VOps O(n, NL, v.ngroups);
    const double* c = vrec_at(v, nlyr, p, w);
    const double* dinv = vdinv_at(v, nlyr, p, w);
    const double* xs = vxsol_at(v, nlyr, p, w);
    double* o = vops_at(v, g, p, w);
    const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w];
    const bool last = (p == nlyr - 1);
    const bool brdf = v.brdf_kind != 0;
    const double rho = (m == 0 && !brdf) ? b.albedo[w] : 0.0;
    double *F = ws, *tmp = ws + M2 * NL, *ipbv = tmp + M2, *part = ipbv + 2 * n;   // F [2n][NL], tmp [2n], ipbv [n], part [K][5][NL]
    (void)s;
    // ---- F = [UL; UM] and the direct-dependence rows HD: a lane takes one Legendre order and one of K slices of the solutions
    // (plane-parallel post-processing supplies u_p directly: F, FP and HD are not needed)
    const bool table = !g.pp_mode;
    const int K = (team.width / nstr) < 1 ? 1 : ((team.width / nstr) > n ? n : (team.width / nstr)), per_k = (n + K - 1) / K;
    const bool staged = wst != nullptr && table;
    if (staged) {
        const int nn2 = n * n;
        const double inv_om = 1.0 / om;
        SKV_PAR(e, nn2) {
            const double wp = c[R.WP() + e], wm = c[R.WM() + e];
            wst[e] = wp;
            wst[nn2 + e] = wm;
            wst[2 * nn2 + e] = c[R.DWP() + e] + wp * inv_om;
            wst[3 * nn2 + e] = c[R.DWM() + e] + wm * inv_om;
        }
        team.sync();
    }
    if (table) SKV_PAR(item, nstr * K) {
        const int l = item % nstr, kslice = item / nstr;
        const int a_lo = kslice * per_k, a_hi = (a_lo + per_k < n) ? a_lo + per_k : n;
        const int NG = v.ngroups, NP = 5 + NG;
        const VRec RG(n, NG);
        double hd[4][3], val[3] = {0, 0, 0};
        for (int t = 0; t < 4; ++t)
            for (int cc = 0; cc < 3; ++cc) hd[t][cc] = 0;
        for (int gi = 0; gi < NG; ++gi)
            for (int cc = 0; cc < C; ++cc) part[(kslice * NP + 5 + gi) * NL + l * C + cc] = 0;
        double c4[4];
        load_coef(b, v, nstr, p, l, w, c4);
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        for (int a_ = a_lo; a_ < a_hi; ++a_) {
            double yp[3] = {0, 0, 0}, ym[3] = {0, 0, 0}, dyp[3] = {0, 0, 0}, dym[3] = {0, 0, 0};
            double xPk[3] = {0, 0, 0}, xMk[3] = {0, 0, 0};
            if (l >= m) {
                double tP[3] = {0, 0, 0}, tM[3] = {0, 0, 0}, dtP[3] = {0, 0, 0}, dtM[3] = {0, 0, 0};
                for (int j = 0; j < N; ++j) {
                    const Fun3 fj = fun_mu(g, vg, m, j);
                    const double hw = 0.5 * g.wt[j];
                    double wp[3] = {0, 0, 0}, wm[3] = {0, 0, 0}, dwp[3] = {0, 0, 0}, dwm[3] = {0, 0, 0};
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int r = j * S + s2;
                        if (staged) {
                            const int e = r + n * a_, nn2 = n * n;
                            wp[s2] = wst[e];
                            wm[s2] = wst[nn2 + e];
                            dwp[s2] = wst[2 * nn2 + e];
                            dwm[s2] = wst[3 * nn2 + e];
                            continue;
                        }
                        wp[s2] = c[R.WP() + r + n * a_];
                        wm[s2] = c[R.WM() + r + n * a_];
                        dwp[s2] = c[R.DWP() + r + n * a_] + wp[s2] / om;
                        dwm[s2] = c[R.DWM() + r + n * a_] + wm[s2] / om;
                    }
                    double y[3];
                    if (S == 1) {
                        tP[0] += hw * fj.P[l] * wp[0]; tM[0] += hw * fj.P[l] * wm[0];
                        dtP[0] += hw * fj.P[l] * dwp[0]; dtM[0] += hw * fj.P[l] * dwm[0];
                    } else {
                        pi_apply(fj.P[l], fj.R[l], fj.T[l], wp, y);  for (int cc = 0; cc < 3; ++cc) tP[cc] += hw * y[cc];
                        pi_apply(fj.P[l], fj.R[l], fj.T[l], wm, y);  for (int cc = 0; cc < 3; ++cc) tM[cc] += hw * y[cc];
                        pi_apply(fj.P[l], fj.R[l], fj.T[l], dwp, y); for (int cc = 0; cc < 3; ++cc) dtP[cc] += hw * y[cc];
                        pi_apply(fj.P[l], fj.R[l], fj.T[l], dwm, y); for (int cc = 0; cc < 3; ++cc) dtM[cc] += hw * y[cc];
                    }
                }
                const double xP[3] = {sg * tP[0] + tM[0], sg * tP[1] + tM[1], -sg * tP[2] + tM[2]};
                const double xM[3] = {sg * tM[0] + tP[0], sg * tM[1] + tP[1], -sg * tM[2] + tP[2]};
                const double dxP[3] = {sg * dtP[0] + dtM[0], sg * dtP[1] + dtM[1], -sg * dtP[2] + dtM[2]};
                const double dxM[3] = {sg * dtM[0] + dtP[0], sg * dtM[1] + dtP[1], -sg * dtM[2] + dtP[2]};
                for (int cc = 0; cc < 3; ++cc) { xPk[cc] = xP[cc]; xMk[cc] = xM[cc]; }
                greek_apply(c4, xP, yp);
                greek_apply(c4, xM, ym);
                greek_apply(c4, dxP, dyp);
                greek_apply(c4, dxM, dym);
                for (int cc = 0; cc < 3; ++cc) { yp[cc] *= om; ym[cc] *= om; dyp[cc] *= om; dym[cc] *= om; }
            }
            skd::PPMult pm;
            skd::ppm_load(c + R.PPM() + 14 * a_, pm);
            const double L = xs[a_], Mv = xs[n + a_], dk = c[R.DK() + a_];
            const double Ap = c[R.AP() + a_], Am = c[R.AM() + a_], dAp = c[R.DAP() + a_], dAm = c[R.DAM() + a_];
            const double tL = pm.hp * L + Ap * pm.Dm, tM_ = pm.hm * Mv + Am * pm.Dp;
            if (l >= m)
                for (int gi = 0; gi < NG; ++gi) {
                    // direction derivative of (yp, ym): om [ B_l (sg D dtP + dtM) + dB_l (sg D tP + tM) ] with dW of the group
                    double gtP[3] = {0, 0, 0}, gtM[3] = {0, 0, 0};
                    for (int j = 0; j < N; ++j) {
                        const Fun3 fj = fun_mu(g, vg, m, j);
                        const double hw = 0.5 * g.wt[j];
                        double dwp[3] = {0, 0, 0}, dwm[3] = {0, 0, 0}, y[3];
                        for (int s2 = 0; s2 < S; ++s2) {
                            dwp[s2] = c[RG.GDWP(gi) + j * S + s2 + n * a_];
                            dwm[s2] = c[RG.GDWM(gi) + j * S + s2 + n * a_];
                        }
                        if (S == 1) { gtP[0] += hw * fj.P[l] * dwp[0]; gtM[0] += hw * fj.P[l] * dwm[0]; }
                        else {
                            pi_apply(fj.P[l], fj.R[l], fj.T[l], dwp, y); for (int cc = 0; cc < 3; ++cc) gtP[cc] += hw * y[cc];
                            pi_apply(fj.P[l], fj.R[l], fj.T[l], dwm, y); for (int cc = 0; cc < 3; ++cc) gtM[cc] += hw * y[cc];
                        }
                    }
                    const double gxP[3] = {sg * gtP[0] + gtM[0], sg * gtP[1] + gtM[1], -sg * gtP[2] + gtM[2]};
                    const double gxM[3] = {sg * gtM[0] + gtP[0], sg * gtM[1] + gtP[1], -sg * gtM[2] + gtP[2]};
                    double dc4[4], gyp[3], gym[3], eyp[3], eym[3];
                    load_dcoef(g, b, v, gi, p, l, w, dc4);
                    greek_apply(c4, gxP, gyp);
                    greek_apply(c4, gxM, gym);
                    greek_apply(dc4, xPk, eyp);
                    greek_apply(dc4, xMk, eym);
                    const double dkg = c[RG.GDK(gi) + a_], dApg = c[RG.GDAP(gi) + a_], dAmg = c[RG.GDAM(gi) + a_];
                    for (int cc = 0; cc < C; ++cc) {
                        const double dy_p = om * (gyp[cc] + eyp[cc]), dy_m = om * (gym[cc] + eym[cc]);
                        part[(kslice * NP + 5 + gi) * NL + l * C + cc] +=
                            dy_p * tL + dy_m * tM_ + yp[cc] * (pm.hp_k * dkg * L + dApg * pm.Dm + Ap * pm.Dm_k * dkg) +
                            ym[cc] * (pm.hm_k * dkg * Mv + dAmg * pm.Dp + Am * pm.Dp_k * dkg);
                    }
                }
            for (int cc = 0; cc < C; ++cc) {
                F[a_ * NL + l * C + cc] = yp[cc] * pm.hp / om;
                F[(n + a_) * NL + l * C + cc] = ym[cc] * pm.hm / om;
                val[cc] += yp[cc] * tL + ym[cc] * tM_;
                hd[skd::TH_TAU][cc] += yp[cc] * (pm.hp_tau * L + Ap * pm.Dm_tau) + ym[cc] * (pm.hm_tau * Mv + Am * pm.Dp_tau);
                hd[skd::TH_SEC][cc] += yp[cc] * Ap * pm.Dm_sec + ym[cc] * Am * pm.Dp_sec;
                hd[skd::TH_T][cc] += (T != 0) ? (yp[cc] * Ap * pm.Dm + ym[cc] * Am * pm.Dp) / T : 0.0;
                hd[skd::TH_OM][cc] += dyp[cc] * tL + dym[cc] * tM_ + yp[cc] * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) +
                                      ym[cc] * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
            }
        }
        for (int cc = 0; cc < C; ++cc) {
            for (int t = 0; t < 4; ++t) part[(kslice * NP + t) * NL + l * C + cc] = hd[t][cc];
            part[(kslice * NP + 4) * NL + l * C + cc] = val[cc];
        }
    }
    team.sync();
    if (table) SKV_PAR(lam_, NL) {
        const int NP = 5 + v.ngroups;
        double hd[4] = {0, 0, 0, 0}, val = 0;
        for (int k = 0; k < K; ++k) {
            for (int t = 0; t < 4; ++t) hd[t] += part[(k * NP + t) * NL + lam_];
            val += part[(k * NP + 4) * NL + lam_];
        }
        for (int t = 0; t < 4; ++t) hd[t] /= om;
        hd[skd::TH_OM] -= val / (om * om);
        for (int t = 0; t < 4; ++t) o[O.HD() + t * NL + lam_] = hd[t];
        for (int gi = 0; gi < v.ngroups; ++gi) {
            double h = 0;
            for (int k = 0; k < K; ++k) h += part[(k * NP + 5 + gi) * NL + lam_];
            o[O.GHD(gi) + lam_] = h / om;
        }
    }
    team.sync();
    KV2CT(0)
    // ---- FP = Dinv^T F
    if (table)
        team_gemm(team, M2, NL, M2, [&](int i, int j) { return dinv[j * M2 + i]; }, [&](int j, int lam_) { return F[j * NL + lam_]; },
                  [&](int i, int lam_, double val) { o[O.FP() + i * NL + lam_] = val; });
    KV2CT(1)
    // ---- SP = Dinv^T SupT, BP = Dinv^T SubT  (SupT[j][i] = W-(i,j), SupT[n+j][i] = W+(i,j) E_j ; SubT[j][i] = W+(i,j) E_j, SubT[n+j][i] = W-(i,j))
    team_gemm(team, M2, M2, M2, [&](int i, int j) { return dinv[j * M2 + i]; },
              [&](int j, int cc) {   // columns 0 .. n-1: SupT, n .. 2n-1: SubT
                  const int cidx = cc < n ? cc : cc - n, jj = j < n ? j : j - n;
                  const bool minus = (cc < n) == (j < n);   // SupT rows j < n and SubT rows j >= n carry W-
                  if (staged) return minus ? wst[n * n + cidx + n * jj] : wst[cidx + n * jj] * c[R.E() + jj];
                  const double wp = c[R.WP() + cidx + n * jj], wm = c[R.WM() + cidx + n * jj], E = c[R.E() + jj];
                  return minus ? wm : wp * E;
              },
              [&](int i, int cc, double val) {
                  if (cc < n) o[O.SP() + i * n + cc] = val;
                  else o[O.BP() + i * n + (cc - n)] = val;
              });
    KV2CT(2)
    // ---- edge-intensity derivatives at fixed (L, M).  Without scattering groups a lane takes one row and a chunk of the solutions, the
    //      partial sums meet in the (now free) storage of the HD partial sums; otherwise one row per lane.
    int CHE = (v.ngroups == 0) ? team.width / n : 1;
    if (CHE > 8) CHE = 8;
    while (CHE > 1 && size_t(CHE) * 9 * n > size_t(K) * (5 + v.ngroups) * NL) --CHE;
    if (CHE >= 2) {
        const int perj = (n + CHE - 1) / CHE;
        SKV_PAR(item, n * CHE) {
            const int i = item % n, ch = item / n, j0 = ch * perj, j1 = (j0 + perj < n) ? j0 + perj : n;
            double d_tau[4] = {0, 0, 0, 0}, d_om[4] = {0, 0, 0, 0}, ipb_i = 0;
            for (int j = j0; j < j1; ++j) {
                const double wp = c[R.WP() + i + n * j], wm = c[R.WM() + i + n * j], dwp = c[R.DWP() + i + n * j], dwm = c[R.DWM() + i + n * j];
                const double E = c[R.E() + j], L = xs[j], Mv = xs[n + j];
                const double dE_tau = -c[R.K() + j] * E, dE_om = -c[R.DK() + j] * th * E;
                d_tau[0] += wm * dE_tau * Mv;
                d_tau[1] += wp * dE_tau * Mv;
                d_tau[2] += wp * dE_tau * L;
                d_tau[3] += wm * dE_tau * L;
                d_om[0] += dwp * L + dwm * E * Mv + wm * dE_om * Mv;
                d_om[1] += dwm * L + dwp * E * Mv + wp * dE_om * Mv;
                d_om[2] += dwp * E * L + wp * dE_om * L + dwm * Mv;
                d_om[3] += dwm * E * L + wm * dE_om * L + dwp * Mv;
                ipb_i += wp * E * L + wm * Mv;
            }
            double* pc = part + size_t(ch) * 9 * n;
            for (int e = 0; e < 4; ++e) {
                pc[e * n + i] = d_tau[e];
                pc[(4 + e) * n + i] = d_om[e];
            }
            pc[8 * n + i] = ipb_i;
        }
        team.sync();
        SKV_PAR(i, n) {
            double sums[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int ch = 0; ch < CHE; ++ch)
                for (int e = 0; e < 9; ++e) sums[e] += part[(size_t(ch) * 9 + e) * n + i];
            const double ipb_i = c[R.G() + 2 * n + i] + sums[8];
            ipbv[i] = ((i % S) == 0) ? g.wt[i / S] * g.mu[i / S] * ipb_i : 0.0;
            for (int e = 0; e < 4; ++e) {
                o[O.DI() + (skd::TH_TAU * 4 + e) * n + i] = c[R.DG() + (skd::TH_TAU * 4 + e) * n + i] + sums[e];
                o[O.DI() + (skd::TH_OM * 4 + e) * n + i] = c[R.DG() + (skd::TH_OM * 4 + e) * n + i] + sums[4 + e];
                o[O.DI() + (skd::TH_T * 4 + e) * n + i] = c[R.DG() + (skd::TH_T * 4 + e) * n + i];
                o[O.DI() + (skd::TH_SEC * 4 + e) * n + i] = c[R.DG() + (skd::TH_SEC * 4 + e) * n + i];
            }
        }
    } else
    SKV_PAR(i, n) {
        double d_tau[4] = {0, 0, 0, 0}, d_om[4] = {0, 0, 0, 0}, ipb_i = c[R.G() + 2 * n + i];
        for (int j = 0; j < n; ++j) {
            const double wp = c[R.WP() + i + n * j], wm = c[R.WM() + i + n * j], dwp = c[R.DWP() + i + n * j], dwm = c[R.DWM() + i + n * j];
            const double E = c[R.E() + j], L = xs[j], Mv = xs[n + j];
            const double dE_tau = -c[R.K() + j] * E, dE_om = -c[R.DK() + j] * th * E;
            d_tau[0] += wm * dE_tau * Mv;
            d_tau[1] += wp * dE_tau * Mv;
            d_tau[2] += wp * dE_tau * L;
            d_tau[3] += wm * dE_tau * L;
            d_om[0] += dwp * L + dwm * E * Mv + wm * dE_om * Mv;
            d_om[1] += dwm * L + dwp * E * Mv + wp * dE_om * Mv;
            d_om[2] += dwp * E * L + wp * dE_om * L + dwm * Mv;
            d_om[3] += dwm * E * L + wm * dE_om * L + dwp * Mv;
            ipb_i += wp * E * L + wm * Mv;
        }
        ipbv[i] = ((i % S) == 0) ? g.wt[i / S] * g.mu[i / S] * ipb_i : 0.0;
        for (int gi = 0; gi < v.ngroups; ++gi) {
            const VRec RG(n, v.ngroups);
            double dg[4] = {0, 0, 0, 0};
            for (int j = 0; j < n; ++j) {
                const double wp = c[R.WP() + i + n * j], wm = c[R.WM() + i + n * j];
                const double dwp = c[RG.GDWP(gi) + i + n * j], dwm = c[RG.GDWM(gi) + i + n * j];
                const double E = c[R.E() + j], L = xs[j], Mv = xs[n + j];
                const double dE = -c[RG.GDK(gi) + j] * th * E;
                dg[0] += dwp * L + dwm * E * Mv + wm * dE * Mv;
                dg[1] += dwm * L + dwp * E * Mv + wp * dE * Mv;
                dg[2] += dwp * E * L + wp * dE * L + dwm * Mv;
                dg[3] += dwm * E * L + wm * dE * L + dwp * Mv;
            }
            for (int e = 0; e < 4; ++e) o[O.GDI(gi) + e * n + i] = c[RG.GDG(gi) + e * n + i] + dg[e];
        }
        for (int e = 0; e < 4; ++e) {
            o[O.DI() + (skd::TH_TAU * 4 + e) * n + i] = c[R.DG() + (skd::TH_TAU * 4 + e) * n + i] + d_tau[e];
            o[O.DI() + (skd::TH_OM * 4 + e) * n + i] = c[R.DG() + (skd::TH_OM * 4 + e) * n + i] + d_om[e];
            o[O.DI() + (skd::TH_T * 4 + e) * n + i] = c[R.DG() + (skd::TH_T * 4 + e) * n + i];
            o[O.DI() + (skd::TH_SEC * 4 + e) * n + i] = c[R.DG() + (skd::TH_SEC * 4 + e) * n + i];
        }
    }
    KV2CT(3)
    // ground part of u (order 0, last layer): tmp = [2 rho sp_j E_j ; 2 rho sm_j] over the intensity rows
    SKV_PAR(j, n) {
        double sp = 0, sm = 0;
        if (last && m == 0)
            for (int i = 0; i < N; ++i) {
                sp += g.mu[i] * g.wt[i] * c[R.WP() + i * S + n * j];
                sm += g.mu[i] * g.wt[i] * c[R.WM() + i * S + n * j];
            }
        tmp[j] = 2.0 * rho * sp * c[R.E() + j];
        tmp[n + j] = 2.0 * rho * sm;
    }
    team.sync();
    SKV_PAR(i, M2) o[O.UG() + i] = sum4(0, M2, [&](int j) { return dinv[j * M2 + i] * tmp[j]; });
    if (brdf && last) {
        // non-Lambertian ground: per incoming stream t the adjoint seed Dinv^T e_t, e_t = d I-_t(ground) / d(L, M) = [W+(t, j) E_j ; W-(t, j)], and
        // I-_t at the ground itself (what the derivatives wrt the BRDF arguments multiply)
        double* so = v.surf_ops + size_t(w) * N * (M2 + 1);
        SKV_PAR(item, N * M2) {
            const int t = item / M2, i = item % M2;
            so[t * (M2 + 1) + i] = sum4(0, M2, [&](int j) {
                const double e = j < n ? c[R.WP() + t * S + n * j] * c[R.E() + j] : c[R.WM() + t * S + n * (j - n)];
                return dinv[j * M2 + i] * e;
            });
        }
        SKV_PAR(t, N) {
            double val = c[R.G() + 2 * n + t * S];
            for (int j = 0; j < n; ++j) val += c[R.WP() + t * S + n * j] * c[R.E() + j] * xs[j] + c[R.WM() + t * S + n * j] * xs[n + j];
            so[t * (M2 + 1) + M2] = val;
        }
    }
    if (team.lane == 0) {
        double ipb = 0;
        for (int i = 0; i < n; ++i) ipb += ipbv[i];
        o[O.IPB()] = ipb;
        for (int t = 0; t < 4; ++t) {
            double sacc = 0;
            if (last && m == 0)
                for (int i = 0; i < N; ++i) sacc += g.wt[i] * g.mu[i] * o[O.DI() + (t * 4 + 2) * n + i * S];
            o[O.GD() + t] = 2.0 * rho * sacc;
        }
        for (int gi = 0; gi < v.ngroups; ++gi) {
            double sacc = 0;
            if (last && m == 0)
                for (int i = 0; i < N; ++i) sacc += g.wt[i] * g.mu[i] * o[O.GDI(gi) + 2 * n + i * S];
            o[O.GGD(gi)] = 2.0 * rho * sacc;
        }
    }
    team.sync();
    KV2CT(4)
#if defined(SKV_K2C_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    // phase shares measured (32 / 16 streams): Legendre sums + F / HD rows 55 / 46 %, FP 13 / 10 %, SP / BP 18 / 14 %, edge derivatives 9 / 25 %.
    // (the Legendre sums as one fused team_gemm with the result staged in HBM were measured slower: 198 -> 224 ms, 110 -> 128 ms)
    if (team.lane == 0 && w == 0 && p == g.nlyr / 2 && m == 1)
        printf("KV2CT n %d lanes %d: F/HD %lld FP %lld SP/BP %lld edge %lld ground/UG %lld\n", n, team.width, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
}

// out[i] (+)= sum_{k in [k0, K)} A[i * ld + k] * x[k], i < M: eight outputs at a time, so that one load of x[k] (thread-strided scratch) feeds
// eight fused multiply-adds whose A operands are the same for every lane of a wavefront (broadcast loads)
template <class Out> SKD_HD void matvec8(const double* A, int ld, int M, int k0, int K, const Scr& x, double sign, Out out) {
    for (int i0 = 0; i0 < M; i0 += 8) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int cnt = (M - i0 < 8) ? M - i0 : 8;
        if (cnt == 8) {
            for (int k = k0; k < K; ++k) {
                const double xv = x[k];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += A[(i0 + u) * ld + k] * xv;
            }
        } else {
            for (int k = k0; k < K; ++k) {
                const double xv = x[k];
                for (int u = 0; u < cnt; ++u) acc[u] += A[(i0 + u) * ld + k] * xv;
            }
        }
        for (int u = 0; u < cnt; ++u) out(i0 + u, sign * acc[u]);
    }
}

// backward sweep  z_p = y_p - BP zA_{p+1}  and reduced-gradient contraction of one (line of sight, Stokes component); the forward
// sweep has left y in v.z.  table_weights: the Legendre-space HD term applies (spherical table; false for plane parallel)
// reduced-gradient contraction of one (line of sight, Stokes component) from the adjoint vectors z left in v.z
// (BRDF = false: an instantiation without the non-Lambertian ground, for the tuned team kernel of Lambertian solves -- its register count is its occupancy)
template <bool BRDF = true> SKD_HD void adjoint_gradient_column(const Geo& g, const Bufs& b, const VBufs& v, int s, int m, int w, size_t tid, size_t nt, double gw,
                                    bool table_weights) {
    const int S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr;
    const VOps O(n, NL, v.ngroups);
    for (int p = 0; p < nlyr; ++p) {
        const double* ops = vops_at(v, g, p, w);
        const bool last = (p == nlyr - 1);
        double gth[4] = {0, 0, 0, 0};
        double zBsum = 0;
        for (int i = 0; i < n; ++i) {
            const double zA = v.z[(size_t(p) * M2 + i) * nt + tid], zB = v.z[(size_t(p) * M2 + n + i) * nt + tid];
            const double zAn = last ? 0.0 : v.z[(size_t(p + 1) * M2 + i) * nt + tid];
            const double zBp = (p > 0) ? v.z[(size_t(p - 1) * M2 + n + i) * nt + tid] : 0.0;
            if (last && m == 0 && (i % S) == 0) zBsum += zB;
            for (int t = 0; t < 4; ++t)
                gth[t] += zA * ops[O.DI() + (t * 4 + 0) * n + i] + zBp * ops[O.DI() + (t * 4 + 1) * n + i] -
                          zAn * ops[O.DI() + (t * 4 + 2) * n + i] - zB * ops[O.DI() + (t * 4 + 3) * n + i];
        }
        if (table_weights)
            for (int l = m * C; l < NL; ++l) {
                const double q = v.qlbuf[(size_t(p) * NL + l) * nt + tid];
                for (int t = 0; t < 4; ++t) gth[t] += ops[O.HD() + t * NL + l] * q;
            }
        if (last && m == 0 && (!BRDF || v.brdf_kind == 0)) {
            for (int t = 0; t < 4; ++t) gth[t] += (zBsum + gw) * ops[O.GD() + t];
            const double rho = b.albedo[w], ipb = ops[O.IPB()];
            v.grad[size_t(4 * nlyr) * nt + tid] += gw * 2.0 * ipb + zBsum * (2.0 * ipb + g.csz[s] / kPi * b.T[size_t(nlyr) * nw + w]);
            // solar: d / d(beam transmission at the ground); thermal pass: d / d(surface emission) (as skd::K2bTask::gradient)
            v.grad[size_t(4 * nlyr + 1) * nt + tid] += b.thermal_pass ? zBsum : zBsum * g.csz[s] / kPi * rho;
        }
        if constexpr (BRDF) if (last && v.brdf_kind != 0) {
            // non-Lambertian ground (every order): the adjoint weight of incoming stream t is c_t = sum_i zB_i RS(i, t) + gw RL(los, t)
            const int N = g.N, los = int((tid / S) % g.nlos), na = v.brdf_nargs;
            const VSurf SF(N, v.surf_nlos, na);
            const double* sf = vsurf_at(v, g, m, w);
            const double* so = v.surf_ops + size_t(w) * N * (M2 + 1);
            const double Tg = b.T[size_t(nlyr) * nw + w];
            double ga[3] = {0, 0, 0}, gT = 0;
            double cg[kMaxSurfStreams], dcg[3][kMaxSurfStreams];
            ground_los_weights(g, b, v, sf, 0, s, m, los, w, gw, cg);
            for (int a = 0; a < na; ++a) ground_los_weights(g, b, v, sf, 1 + a, s, m, los, w, gw, dcg[a]);
            for (int t = 0; t < N; ++t) {
                double ct = cg[t], da[3] = {0, 0, 0};
                for (int a = 0; a < na; ++a) da[a] = dcg[a][t];
                for (int i = 0; i < N; ++i) {
                    const double zbi = v.z[(size_t(p) * M2 + n + i * S) * nt + tid];
                    ct += zbi * sf[SF.RS(i, t)];
                    for (int a = 0; a < na; ++a) da[a] += zbi * sf[size_t(1 + a) * SF.NP() + SF.RS(i, t)];
                }
                for (int th4 = 0; th4 < 4; ++th4) gth[th4] += ct * ops[O.DI() + (th4 * 4 + 2) * n + t * S];
                for (int a = 0; a < na; ++a) ga[a] += da[a] * so[t * (M2 + 1) + M2];
            }
            for (int i = 0; i < N; ++i) {
                const double zbi = v.z[(size_t(p) * M2 + n + i * S) * nt + tid];
                gT += zbi * sf[SF.BS(i)];
                for (int a = 0; a < na; ++a) ga[a] += zbi * sf[size_t(1 + a) * SF.NP() + SF.BS(i)] * Tg;
            }
            for (int a = 0; a < na; ++a) v.grad[surf_grad_row(g, v, a) * nt + tid] += ga[a];
            v.grad[size_t(4 * nlyr + 1) * nt + tid] += gT;
        }
        for (int t = 0; t < 4; ++t) v.grad[size_t(t * nlyr + p) * nt + tid] += gth[t];
        // scattering groups: the same contraction with the group's DI / HD / GD rows
        double ggb[BRDF ? kMaxGroupsWithBrdf : 1];   // non-Lambertian ground: sum_t c_t (d I-_t / d group at fixed L, M), the counterpart of (zBsum + gw) GGD
        for (int gi = 0; gi < (BRDF ? kMaxGroupsWithBrdf : 1); ++gi) ggb[gi] = 0;
        if constexpr (BRDF) if (last && v.brdf_kind != 0 && v.ngroups > 0) {
            const int N = g.N, los = int((tid / S) % g.nlos);
            const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
            const double* sf = vsurf_at(v, g, m, w);
            double cg[kMaxSurfStreams];
            ground_los_weights(g, b, v, sf, 0, s, m, los, w, gw, cg);
            for (int t = 0; t < N; ++t) {
                double ct = cg[t];
                for (int i = 0; i < N; ++i) ct += v.z[(size_t(p) * M2 + n + i * S) * nt + tid] * sf[SF.RS(i, t)];
                for (int gi = 0; gi < v.ngroups && gi < kMaxGroupsWithBrdf; ++gi) ggb[gi] += ct * ops[O.GDI(gi) + 2 * n + t * S];
            }
        }
        for (int gi = 0; gi < v.ngroups; ++gi) {
            double gg = 0;
            for (int i = 0; i < n; ++i) {
                const double zA = v.z[(size_t(p) * M2 + i) * nt + tid], zB = v.z[(size_t(p) * M2 + n + i) * nt + tid];
                const double zAn = last ? 0.0 : v.z[(size_t(p + 1) * M2 + i) * nt + tid];
                const double zBp = (p > 0) ? v.z[(size_t(p - 1) * M2 + n + i) * nt + tid] : 0.0;
                gg += zA * ops[O.GDI(gi) + 0 * n + i] + zBp * ops[O.GDI(gi) + 1 * n + i] - zAn * ops[O.GDI(gi) + 2 * n + i] -
                      zB * ops[O.GDI(gi) + 3 * n + i];
            }
            if (table_weights)
                for (int l = m * C; l < NL; ++l) gg += ops[O.GHD(gi) + l] * v.qlbuf[(size_t(p) * NL + l) * nt + tid];
            if (last && m == 0 && (!BRDF || v.brdf_kind == 0)) gg += (zBsum + gw) * ops[O.GGD(gi)];
            if constexpr (BRDF) if (last && v.brdf_kind != 0 && gi < kMaxGroupsWithBrdf) gg += ggb[gi];
            v.grad[size_t(4 * nlyr + 2 + gi * nlyr + p) * nt + tid] += gg;
        }
    }
}

SKD_HD void adjoint_backward_gradient(const Geo& g, const Bufs& b, const VBufs& v, int s, int m, int w, size_t tid, size_t nt, double gw,
                                      bool table_weights, Scr y, Scr carry) {
    const int S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr;
    const VOps O(n, NL, v.ngroups);
    // ---- backward sweep  z_p = y_p - BP zA_{p+1}
    for (int i = 0; i < n; ++i) carry[i] = 0;
    for (int p = nlyr - 1; p >= 0; --p) {
        const double* ops = vops_at(v, g, p, w);
        for (int i = 0; i < M2; ++i) y[i] = v.z[(size_t(p) * M2 + i) * nt + tid];
        if (p < nlyr - 1) matvec8(ops + O.BP(), n, M2, 0, n, carry, -1.0, [&](int i, double t) { y[i] += t; });
        for (int i = 0; i < M2; ++i) v.z[(size_t(p) * M2 + i) * nt + tid] = y[i];
        for (int i = 0; i < n; ++i) carry[i] = y[i];
    }
    adjoint_gradient_column(g, b, v, s, m, w, tid, nt, gw, table_weights);
}

// KV2B: adjoint sweep of one (line of sight, Stokes component) for azimuth order m   (task = wavelength x LOS x Stokes)
//   scratch: NL + 3 n doubles per thread
SKD_HD size_t kv2b_scratch(int n, int NL) { return size_t(NL) + 3 * size_t(n); }

SKD_HD void kv2b_adjoint(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int los, int sout, int w, Scr scr) {
    const int S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr;
    const VOps O(n, NL, v.ngroups);
    const size_t tid = (size_t(w) * g.nlos + los) * S + sout, nt = size_t(nw) * g.nlos * S;
    Scr ql = scr.sub(0), y = scr.sub(NL), carry = scr.sub(NL + M2);
    double gw = 0;  // the ground feeds the intensity only
    const bool brdf = v.brdf_kind != 0;
    double cg[kMaxSurfStreams];
    if (sout == 0 && m == 0 && g.ray_ground[los] && !brdf) {
        for (int t = g.g_start[los]; t < g.g_start[los + 1]; ++t)
            if (g.g_sza[t] == s && skd::conv_at(b, g, g.g_slot[t], w) > m) gw += g.g_w[t];
        gw *= b.att_total[size_t(los) * nw + w];
    }
    if (brdf) {   // non-Lambertian ground: every order; the ray's ground weights over the incoming streams (ground_los_weights)
        if (sout == 0 && g.ray_ground[los]) gw = b.att_total[size_t(los) * nw + w];
        ground_los_weights(g, b, v, vsurf_at(v, g, m, w), 0, s, m, los, w, gw, cg);
    }
    for (int i = 0; i < n; ++i) carry[i] = 0;
    // ---- forward sweep  y_p = FP ql_p + SP yB_{p-1} (+ gw UG)
    for (int p = 0; p < nlyr; ++p) {
        const double* ops = vops_at(v, g, p, w);
        for (int l = 0; l < NL; ++l) ql[l] = 0;
        const int base = (los * g.nsza + s) * nlyr + p;
        for (int t = g.lp_start[base]; t < g.lp_start[base + 1]; ++t) {
            if (skd::conv_at(b, g, g.lp_slot[t], w) <= m) continue;
            const int q = g.lp_q[t];
            const double az = (sout < 2) ? g.rl_cosm[size_t(m) * g.nq_total + q] : vg.rl_sinm_t[size_t(q) * nstr + m];
            const double cw = b.srcw[size_t(q) * nw + w] * g.lp_w[t] * az;
            if (cw == 0.0) continue;
            const Fun3 fa = fun_ang(g, vg, m, g.lp_angle[t]);
            for (int l = m; l < nstr; ++l) {
                if (S == 1) { ql[l] += cw * fa.P[l]; continue; }
                // row `sout` of Pi_l(angle)
                if (sout == 0) ql[l * 3] += cw * fa.P[l];
                else if (sout == 1) { ql[l * 3 + 1] += cw * fa.R[l]; ql[l * 3 + 2] -= cw * fa.T[l]; }
                else { ql[l * 3 + 1] -= cw * fa.T[l]; ql[l * 3 + 2] += cw * fa.R[l]; }
            }
        }
        for (int l = 0; l < NL; ++l) v.qlbuf[(size_t(p) * NL + l) * nt + tid] = ql[l];
        for (int i = 0; i < M2; ++i) y[i] = (p == nlyr - 1) ? gw * ops[O.UG() + i] : 0.0;
        if (brdf && p == nlyr - 1 && gw != 0.0) {
            const double* so = v.surf_ops + size_t(w) * g.N * (M2 + 1);
            for (int i = 0; i < M2; ++i) {
                double acc = 0;
                for (int t = 0; t < g.N; ++t) acc += cg[t] * so[t * (M2 + 1) + i];
                y[i] = acc;
            }
        }
        matvec8(ops + O.FP(), NL, M2, m * C, NL, ql, 1.0, [&](int i, double t) { y[i] += t; });
        matvec8(ops + O.SP(), n, M2, 0, n, carry, 1.0, [&](int i, double t) { y[i] += t; });
        for (int i = 0; i < M2; ++i) v.z[(size_t(p) * M2 + i) * nt + tid] = y[i];
        for (int i = 0; i < n; ++i) carry[i] = y[n + i];
    }
    adjoint_backward_gradient(g, b, v, s, m, w, tid, nt, gw, true, y, carry);
}

// KV2B (team form): the adjoint sweeps of ALL (line of sight, Stokes component) columns of one wavelength as matrix products
//   Y_p = FP QL_p + SP Y_{p-1}[n:] (+ gw UG),   Z_p = Y_p - BP Z_{p+1}[:n]      (M2 x NC, NC = nlos * S columns)
// on the matrix cores (team_gemm), then the reduced-gradient contraction per column.   (task = wavelength, one team)
//   ws: (NL + 3 n + 1) NC doubles of team-shared work space: QL [NL][NC], Y [2n][NC], carry [n][NC], gw [NC]
SKD_HD size_t kv2bt_scratch(int n, int NL, int NC) { return (size_t(NL) + 3 * size_t(n) + 1) * size_t(NC); }

template <bool BRDF = true> SKV_TEAM_BODY void kv2b_team(const Team& team, const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int w, double* ws) {
    const int S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr, NC = g.nlos * S;
    const VOps O(n, NL, v.ngroups);
    const size_t nt = size_t(nw) * NC, t0 = size_t(w) * NC;
    double *QL = ws, *Y = QL + size_t(NL) * NC, *Cc = Y + size_t(M2) * NC, *gwv = Cc + size_t(n) * NC;
    SKV_PAR(col, NC) {
        const int los = col / S, sout = col % S;
        double gw = 0;  // the ground feeds the intensity only
        if (sout == 0 && m == 0 && g.ray_ground[los] && (!BRDF || v.brdf_kind == 0)) {
            for (int t = g.g_start[los]; t < g.g_start[los + 1]; ++t)
                if (g.g_sza[t] == s && skd::conv_at(b, g, g.g_slot[t], w) > m) gw += g.g_w[t];
            gw *= b.att_total[size_t(los) * nw + w];
        }
        if constexpr (BRDF) if (v.brdf_kind != 0 && sout == 0 && g.ray_ground[los]) gw = b.att_total[size_t(los) * nw + w];   // (every order; weighed stream by stream below)
        gwv[col] = gw;
        for (int i = 0; i < n; ++i) Cc[size_t(i) * NC + col] = 0;
    }
    // ---- forward sweep
    for (int p = 0; p < nlyr; ++p) {
        const double* ops = vops_at(v, g, p, w);
        const bool last = (p == nlyr - 1);
        team.sync();
        {   // Legendre-space weights of the column's line of sight (row `sout` of Pi_l at the table angles): a lane takes one column and a
            // chunk of the Legendre orders, accumulates in registers (eight orders at a time) and writes every entry once
            const int LC = (team.width / NC) < 1 ? 1 : ((team.width / NC) > nstr ? nstr : (team.width / NC)), per = (nstr + LC - 1) / LC;
            SKV_PAR(item, NC * LC) {
                const int col = item % NC, ch = item / NC, los = col / S, sout = col % S;
                const int l0 = ch * per, l1 = (l0 + per < nstr) ? l0 + per : nstr;
                const int base = (los * g.nsza + s) * nlyr + p;
                for (int lb = l0; lb < l1; lb += 4) {
                    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
                    for (int t = g.lp_start[base]; t < g.lp_start[base + 1]; ++t) {
                        if (skd::conv_at(b, g, g.lp_slot[t], w) <= m) continue;
                        const int q = g.lp_q[t];
                        const double az = (sout < 2) ? g.rl_cosm[size_t(m) * g.nq_total + q] : vg.rl_sinm_t[size_t(q) * nstr + m];
                        const double cw = b.srcw[size_t(q) * nw + w] * g.lp_w[t] * az;
                        if (cw == 0.0) continue;
                        const Fun3 fa = fun_ang(g, vg, m, g.lp_angle[t]);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int l = lb + u;
                            if (l >= l1 || l < m) continue;
                            if (S == 1 || sout == 0) a0[u] += cw * fa.P[l];
                            else if (sout == 1) { a1[u] += cw * fa.R[l]; a2[u] -= cw * fa.T[l]; }
                            else { a1[u] -= cw * fa.T[l]; a2[u] += cw * fa.R[l]; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int l = lb + u;
                        if (l >= l1) continue;
                        QL[size_t(l * C) * NC + col] = a0[u];
                        v.qlbuf[(size_t(p) * NL + l * C) * nt + t0 + col] = a0[u];
                        if (C == 3) {
                            QL[size_t(l * C + 1) * NC + col] = a1[u];
                            QL[size_t(l * C + 2) * NC + col] = a2[u];
                            v.qlbuf[(size_t(p) * NL + l * C + 1) * nt + t0 + col] = a1[u];
                            v.qlbuf[(size_t(p) * NL + l * C + 2) * nt + t0 + col] = a2[u];
                        }
                    }
                }
            }
        }
        team.sync();
        team_gemm(team, M2, NC, NL, [&](int i, int k) { return ops[O.FP() + i * NL + k]; }, [&](int k, int col) { return QL[size_t(k) * NC + col]; },
                  [&](int i, int col, double val) {
                      double seed = last ? gwv[col] * ops[O.UG() + i] : 0.0;
                      if constexpr (BRDF) if (last && v.brdf_kind != 0 && gwv[col] != 0.0) {   // non-Lambertian ground: sum_t c_t Dinv^T e_t (few columns: the rays that end on the ground)
                          double cg[kMaxSurfStreams];
                          ground_los_weights(g, b, v, vsurf_at(v, g, m, w), 0, s, m, col / S, w, gwv[col], cg);
                          const double* so = v.surf_ops + size_t(w) * g.N * (M2 + 1);
                          seed = 0;
                          for (int t = 0; t < g.N; ++t) seed += cg[t] * so[t * (M2 + 1) + i];
                      }
                      Y[size_t(i) * NC + col] = val + seed;
                  });
        team.sync();
        team_gemm(team, M2, NC, n, [&](int i, int k) { return ops[O.SP() + i * n + k]; }, [&](int k, int col) { return Cc[size_t(k) * NC + col]; },
                  [&](int i, int col, double val) { Y[size_t(i) * NC + col] += val; });
        team.sync();
        SKV_PAR(e, M2 * NC) {
            const int i = e / NC, col = e % NC;
            const double yv = Y[e];
            v.z[(size_t(p) * M2 + i) * nt + t0 + col] = yv;
            if (i >= n) Cc[size_t(i - n) * NC + col] = yv;
        }
    }
    // ---- backward sweep
    team.sync();
    SKV_PAR(e, n * NC) Cc[e] = 0;
    for (int p = nlyr - 1; p >= 0; --p) {
        const double* ops = vops_at(v, g, p, w);
        team.sync();
        SKV_PAR(e, M2 * NC) Y[e] = v.z[(size_t(p) * M2 + e / NC) * nt + t0 + e % NC];
        team.sync();
        if (p < nlyr - 1) {
            team_gemm(team, M2, NC, n, [&](int i, int k) { return ops[O.BP() + i * n + k]; }, [&](int k, int col) { return Cc[size_t(k) * NC + col]; },
                      [&](int i, int col, double val) { Y[size_t(i) * NC + col] -= val; });
            team.sync();
        }
        SKV_PAR(e, M2 * NC) {
            const int i = e / NC, col = e % NC;
            const double zv = Y[e];
            v.z[(size_t(p) * M2 + i) * nt + t0 + col] = zv;
            if (i < n) Cc[size_t(i) * NC + col] = zv;
        }
    }
    team.sync();
    // ---- reduced gradient: a lane takes one column and a chunk of the contraction indices (solutions, Legendre entries), the partial
    //      sums meet in the free QL storage; the sweep states are read once.  (More than four scattering groups: one column per lane.)
    const int NG = v.ngroups, NR = 4 + NG;
    int CH = team.width / NC;
    if (CH > 8) CH = 8;
    while (CH > 1 && CH * (NR + 1) > NL) --CH;
    if (NG > 4 || CH < 2 || (BRDF && v.brdf_kind != 0)) {   // (a non-Lambertian ground weighs the ground rows stream by stream: the per-column form carries it)
        SKV_PAR(col, NC) adjoint_gradient_column<BRDF>(g, b, v, s, m, w, t0 + col, nt, gwv[col], true);
        team.sync();
        return;
    }
    double* part = QL;   // [CH][NR + 1][NC]
    const int peri = (n + CH - 1) / CH, perl = (NL + CH - 1) / CH;
    for (int p = 0; p < nlyr; ++p) {
        const double* ops = vops_at(v, g, p, w);
        const bool last = (p == nlyr - 1);
        SKV_PAR(item, NC * CH) {
            const int col = item % NC, ch = item / NC;
            const size_t tid = t0 + col;
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zs = 0;
            const int i0 = ch * peri, i1 = (i0 + peri < n) ? i0 + peri : n;
            for (int i = i0; i < i1; ++i) {
                const double zA = v.z[(size_t(p) * M2 + i) * nt + tid], zB = v.z[(size_t(p) * M2 + n + i) * nt + tid];
                const double zAn = last ? 0.0 : v.z[(size_t(p + 1) * M2 + i) * nt + tid];
                const double zBp = (p > 0) ? v.z[(size_t(p - 1) * M2 + n + i) * nt + tid] : 0.0;
                if (last && m == 0 && (i % S) == 0) zs += zB;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] += zA * ops[O.DI() + (t * 4 + 0) * n + i] + zBp * ops[O.DI() + (t * 4 + 1) * n + i] -
                              zAn * ops[O.DI() + (t * 4 + 2) * n + i] - zB * ops[O.DI() + (t * 4 + 3) * n + i];
#pragma unroll
                for (int gi = 0; gi < 4; ++gi)
                    if (gi < NG)
                        acc[4 + gi] += zA * ops[O.GDI(gi) + 0 * n + i] + zBp * ops[O.GDI(gi) + 1 * n + i] - zAn * ops[O.GDI(gi) + 2 * n + i] -
                                       zB * ops[O.GDI(gi) + 3 * n + i];
            }
            const int lf = m * C, l0 = (ch * perl > lf) ? ch * perl : lf, l1 = (ch * perl + perl < NL) ? ch * perl + perl : NL;
            for (int l = l0; l < l1; ++l) {
                const double q = v.qlbuf[(size_t(p) * NL + l) * nt + tid];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] += ops[O.HD() + t * NL + l] * q;
#pragma unroll
                for (int gi = 0; gi < 4; ++gi)
                    if (gi < NG) acc[4 + gi] += ops[O.GHD(gi) + l] * q;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < NR) part[(size_t(ch) * (NR + 1) + r) * NC + col] = acc[r];
            part[(size_t(ch) * (NR + 1) + NR) * NC + col] = zs;
        }
        team.sync();
        SKV_PAR(item, NC * NR) {
            const int col = item % NC, r = item / NC;
            const size_t tid = t0 + col;
            double a_ = 0, zs = 0;
            for (int ch = 0; ch < CH; ++ch) {
                a_ += part[(size_t(ch) * (NR + 1) + r) * NC + col];
                zs += part[(size_t(ch) * (NR + 1) + NR) * NC + col];
            }
            if (last && m == 0) a_ += (zs + gwv[col]) * ops[(r < 4) ? O.GD() + r : O.GGD(r - 4)];
            v.grad[(r < 4 ? size_t(r * nlyr + p) : size_t(4 * nlyr + 2 + (r - 4) * nlyr + p)) * nt + tid] += a_;
            if (r == 0 && last && m == 0) {
                const double rho = b.albedo[w], ipb = ops[O.IPB()], gw = gwv[col];
                v.grad[size_t(4 * nlyr) * nt + tid] += gw * 2.0 * ipb + zs * (2.0 * ipb + g.csz[s] / kPi * b.T[size_t(nlyr) * nw + w]);
                v.grad[size_t(4 * nlyr + 1) * nt + tid] += b.thermal_pass ? zs : zs * g.csz[s] / kPi * rho;
            }
        }
        team.sync();
    }
}

// =================================================================================================
// KVPP: plane-parallel line of sight for azimuth order m, 3 Stokes components      (task = wavelength x LOS)
//   reference: OpticalLayer::integrate_source sktran_do_opticallayer.cpp:93-555, computeReflectedIntensities
//   sktran_do_layerarray.cpp:5-288, DOSourcePlaneParallelPostProcessing::calculate do_source_planeparallel.cpp:60-218
//   With v.derivs the same adjoint machinery as the spherical mode runs per Stokes component (u_p = dI/d(L_p, M_p) supplied
//   directly, as skd::kpp_los): the operators of kv2c_layer_ops must be up to date.   scratch: (21 + 8 G) n doubles per thread
// =================================================================================================
SKD_HD size_t kvpp_scratch(int n, int G = 0) { return (21 + 8 * size_t(G)) * size_t(n); }

SKD_HD void kvpp_los(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int los, int w, Scr scr) {
    const int N = g.N, S = v.S, n = v.n, M2 = 2 * n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr, C = (S == 3) ? 3 : 1, NL = C * nstr;
    const VRec R(n);
    const VOps O(n, NL, v.ngroups);
    const size_t tid = size_t(w) * g.nlos + los, nt3 = size_t(nw) * g.nlos * S;
    const double mu = g.pp_mu[los];
    const Fun3 fl = fun_los(g, vg, m, los);
    const double* lpc = g.lpcsz + (size_t(s) * nstr + m) * nstr;
    const int first = g.pp_first[los];
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    const bool derivs = v.derivs != 0;
    Scr yp = scr.sub(0), ym = scr.sub(3 * size_t(n)), dyp = scr.sub(6 * size_t(n)), dym = scr.sub(9 * size_t(n)), lrow = scr.sub(12 * size_t(n)),
        carry = scr.sub(14 * size_t(n)), u = scr.sub(17 * size_t(n)), yv = scr.sub(19 * size_t(n)),  // yp .. dym: [S][n]; carry: [S][n]
        gy = scr.sub(21 * size_t(n));  // per scattering group: d yp, d ym [S][n] each (6 n), d lrow (2 n)
    const int NG = derivs ? v.ngroups : 0;
    const VRec RG(n, v.ngroups);
    const double azf[3] = {cos(m * g.pp_azi[los]), cos(m * g.pp_azi[los]), sin(m * g.pp_azi[los])};
    double att = 1.0;  // attenuation to the observer
    double total[3] = {0, 0, 0};
    if (derivs)
        for (int e = 0; e < S * n; ++e) carry[e] = 0;
    for (int p = 0; p < nlyr; ++p) {
        double contrib[3] = {0, 0, 0};
        const bool in = (p >= first);
        const double lf = (p == first) ? g.pp_lf[los] : 1.0;
        const double* c = vrec_at(v, nlyr, p, w);
        const double* xs = vxsol_at(v, nlyr, p, w);
        const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
        const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
        double Q[3] = {0, 0, 0};
        skd::PPBeam Eb{0, 0, 0, 0};
        if (in) {
            // Y+-[s][a] = sum_r lp[s][r] W+-(r,a) + lm[s][r] W-+(r,a);  lm = (om w_j / 2) [Pi(out) B Pi(mu_j)], lp the negated product
            for (int e = 0; e < S * n; ++e) { yp[e] = 0; ym[e] = 0; dyp[e] = 0; dym[e] = 0; }
            for (int gi = 0; gi < NG; ++gi)
                for (int e = 0; e < 6 * n; ++e) gy[size_t(gi) * 8 * n + e] = 0;
            double Qg[4][3];
            for (int gi = 0; gi < 4; ++gi)
                for (int cc = 0; cc < 3; ++cc) Qg[gi][cc] = 0;
            for (int l = m; l < nstr; ++l) {
                double c4[4];
                load_coef(b, v, nstr, p, l, w, c4);
                const double sg = ((l - m) & 1) ? -1.0 : 1.0;
                for (int s1 = 0; s1 < S; ++s1) {
                    // row s1 of Pi_l(out) B_l  (B_l and Pi_l symmetric)
                    double ex[3] = {0, 0, 0}, prow[3], brow[3];
                    ex[s1] = 1;
                    pi_apply(fl.P[l], fl.R[l], fl.T[l], ex, prow);
                    greek_apply(c4, prow, brow);
                    for (int j = 0; j < N; ++j) {
                        const Fun3 fj = fun_mu(g, vg, m, j);
                        const double wj = 0.5 * om * g.wt[j];
                        if (S == 1) {
                            const double t = brow[0] * fj.P[l] * wj;
                            lrow[j] = t;
                            lrow[n + j] = sg * t;
                        } else {
                            double y0[3], y1[3];
                            const double bd[3] = {brow[0], brow[1], -brow[2]};
                            pi_apply(fj.P[l], fj.R[l], fj.T[l], brow, y0);
                            pi_apply(fj.P[l], fj.R[l], fj.T[l], bd, y1);
                            for (int s2 = 0; s2 < 3; ++s2) {
                                lrow[j * 3 + s2] = y0[s2] * wj;
                                lrow[n + j * 3 + s2] = sg * y1[s2] * wj;
                            }
                        }
                    }
                    for (int a = 0; a < n; ++a) {
                        double tp = 0, tm = 0, dtp = 0, dtm = 0;
                        for (int r = 0; r < n; ++r) {
                            const double wp = c[R.WP() + r + n * a], wm = c[R.WM() + r + n * a];
                            tp += lrow[n + r] * wp + lrow[r] * wm;
                            tm += lrow[n + r] * wm + lrow[r] * wp;
                            if (derivs) {
                                const double dwp = c[R.DWP() + r + n * a], dwm = c[R.DWM() + r + n * a];
                                dtp += lrow[n + r] * dwp + lrow[r] * dwm;
                                dtm += lrow[n + r] * dwm + lrow[r] * dwp;
                            }
                        }
                        yp[size_t(s1) * n + a] += tp;
                        ym[size_t(s1) * n + a] += tm;
                        dyp[size_t(s1) * n + a] += dtp + tp / om;
                        dym[size_t(s1) * n + a] += dtm + tm / om;
                    }
                    // solar single-scatter term: first column of the negated product with the sun direction
                    if (g.include_ss || b.thermal_pass) Q[s1] += sg * brow[0] * lpc[l] * fm * om;   // (thermal pass: the direct emission term, as skd::kpp_los)
                    // scattering groups: d(lrow) W + lrow dW_g  (direction dc of the coefficients)
                    for (int gi = 0; gi < NG && gi < 4; ++gi) {
                        Scr gyp = gy.sub(size_t(gi) * 8 * n), gym = gy.sub(size_t(gi) * 8 * n + 3 * size_t(n)), dl = gy.sub(size_t(gi) * 8 * n + 6 * size_t(n));
                        double dc4[4], dbrow[3];
                        load_dcoef(g, b, v, gi, p, l, w, dc4);
                        greek_apply(dc4, prow, dbrow);
                        for (int j = 0; j < N; ++j) {
                            const Fun3 fj = fun_mu(g, vg, m, j);
                            const double wj = 0.5 * om * g.wt[j];
                            if (S == 1) {
                                const double t = dbrow[0] * fj.P[l] * wj;
                                dl[j] = t;
                                dl[n + j] = sg * t;
                            } else {
                                double y0[3], y1[3];
                                const double bd[3] = {dbrow[0], dbrow[1], -dbrow[2]};
                                pi_apply(fj.P[l], fj.R[l], fj.T[l], dbrow, y0);
                                pi_apply(fj.P[l], fj.R[l], fj.T[l], bd, y1);
                                for (int s2 = 0; s2 < 3; ++s2) {
                                    dl[j * 3 + s2] = y0[s2] * wj;
                                    dl[n + j * 3 + s2] = sg * y1[s2] * wj;
                                }
                            }
                        }
                        for (int a = 0; a < n; ++a) {
                            double tp = 0, tm = 0;
                            for (int r = 0; r < n; ++r) {
                                const double wp = c[R.WP() + r + n * a], wm = c[R.WM() + r + n * a];
                                const double dwp = c[RG.GDWP(gi) + r + n * a], dwm = c[RG.GDWM(gi) + r + n * a];
                                tp += dl[n + r] * wp + dl[r] * wm + lrow[n + r] * dwp + lrow[r] * dwm;
                                tm += dl[n + r] * wm + dl[r] * wp + lrow[n + r] * dwm + lrow[r] * dwp;
                            }
                            gyp[size_t(s1) * n + a] += tp;
                            gym[size_t(s1) * n + a] += tm;
                        }
                        if (g.include_ss || b.thermal_pass) Qg[gi][s1] += sg * dbrow[0] * lpc[l] * fm * om;
                    }
                }
            }
            skd::pp_beam_multiplier(mu, lf, th, T, sc, Eb);
            // group rows of the layer source (no explicit 1 / omega here: the source is not divided by omega in this mode)
            for (int gi = 0; gi < NG && gi < 4; ++gi) {
                Scr gyp = gy.sub(size_t(gi) * 8 * n), gym = gy.sub(size_t(gi) * 8 * n + 3 * size_t(n));
                for (int s1 = 0; s1 < S; ++s1) {
                    double dSg = Qg[gi][s1] * Eb.v;
                    for (int a = 0; a < n; ++a) {
                        skd::PPLosMult pm;
                        skd::pp_los_multipliers(mu, lf, th, T, sc, c[R.K() + a], Eb, pm);
                        const double L = xs[a], Mv = xs[n + a], Ap = c[R.AP() + a], Am = c[R.AM() + a];
                        const double dk = c[RG.GDK(gi) + a], dAp = c[RG.GDAP(gi) + a], dAm = c[RG.GDAM(gi) + a];
                        const double y_p = yp[size_t(s1) * n + a], y_m = ym[size_t(s1) * n + a];
                        dSg += gyp[size_t(s1) * n + a] * (pm.hp * L + Ap * pm.Dm) + gym[size_t(s1) * n + a] * (pm.hm * Mv + Am * pm.Dp) +
                               y_p * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) + y_m * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
                    }
                    v.grad[size_t(4 * nlyr + 2 + gi * nlyr + p) * nt3 + tid * S + s1] += att * azf[s1] * dSg;
                }
            }
        }
        for (int s1 = 0; s1 < S; ++s1) {
            const double A = att * azf[s1];
            double Ssrc = Q[s1] * Eb.v;
            double dS[4] = {Q[s1] * Eb.d_th, Q[s1] / om * Eb.v, Q[s1] * Eb.d_T, Q[s1] * Eb.d_sec};
            for (int i = 0; i < M2; ++i) u[i] = 0;
            if (in)
                for (int a = 0; a < n; ++a) {
                    skd::PPLosMult pm;
                    skd::pp_los_multipliers(mu, lf, th, T, sc, c[R.K() + a], Eb, pm);
                    const double L = xs[a], Mv = xs[n + a], Ap = c[R.AP() + a], Am = c[R.AM() + a];
                    const double y_p = yp[size_t(s1) * n + a], y_m = ym[size_t(s1) * n + a];
                    Ssrc += y_p * (pm.hp * L + Ap * pm.Dm) + y_m * (pm.hm * Mv + Am * pm.Dp);
                    if (derivs) {
                        const double dk = c[R.DK() + a], dAp = c[R.DAP() + a], dAm = c[R.DAM() + a];
                        dS[skd::TH_TAU] += y_p * L * pm.hp_th + y_m * Mv * pm.hm_th + Ap * y_p * pm.Dm_th + Am * y_m * pm.Dp_th;
                        dS[skd::TH_T] += Ap * y_p * pm.Dm_T + Am * y_m * pm.Dp_T;
                        dS[skd::TH_SEC] += Ap * y_p * pm.Dm_sec + Am * y_m * pm.Dp_sec;
                        dS[skd::TH_OM] += dyp[size_t(s1) * n + a] * (pm.hp * L + Ap * pm.Dm) + dym[size_t(s1) * n + a] * (pm.hm * Mv + Am * pm.Dp) +
                                          y_p * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) +
                                          y_m * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
                        u[a] = A * y_p * pm.hp;
                        u[n + a] = A * y_m * pm.hm;
                    }
                }
            contrib[s1] = in ? A * Ssrc : 0.0;
            total[s1] += contrib[s1];
            if (derivs) {
                const size_t t3 = tid * S + s1;
                if (in)
                    for (int t = 0; t < 4; ++t) v.grad[size_t(t * nlyr + p) * nt3 + t3] += A * dS[t];
                v.qlbuf[(size_t(p) * NL) * nt3 + t3] = contrib[s1];
                // forward substitution of A^T z = u:  y_p = Dinv^T u_p + SP yB_{p-1}
                const double* dinv = vdinv_at(v, nlyr, p, w);
                const double* ops = vops_at(v, g, p, w);
                for (int i = 0; i < M2; ++i) {
                    double acc = 0;
                    for (int j = 0; j < M2; ++j) acc += dinv[size_t(j) * M2 + i] * u[j];
                    for (int cidx = 0; cidx < n; ++cidx) acc += ops[O.SP() + i * n + cidx] * carry[size_t(s1) * n + cidx];
                    yv[i] = acc;
                }
                for (int i = 0; i < M2; ++i) v.z[(size_t(p) * M2 + i) * nt3 + t3] = yv[i];
                if (p < nlyr - 1)
                    for (int i = 0; i < n; ++i) carry[size_t(s1) * n + i] = yv[n + i];
            }
        }
        if (in) att *= exp(-th * lf / mu);
    }
    // Lambertian ground seen through the whole column (order 0, intensity only)
    double ground = 0, gw = 0;
    if (v.brdf_kind != 0) {
        // non-Lambertian ground: I_los = sum_t RL(los, t) I-_t + BL(los) T_ground in every order (computeReflectedIntensities with the
        // los_stream / los_solar expansions, sktran_do_layerarray.cpp:5-288)
        const double* c = vrec_at(v, nlyr, nlyr - 1, w);
        const double* xs = vxsol_at(v, nlyr, nlyr - 1, w);
        const VSurf SF(N, v.surf_nlos, v.brdf_nargs);
        const double* sf = vsurf_at(v, g, m, w);
        const double Tg = b.T[size_t(nlyr) * nw + w];
        double refl = 0;
        for (int t = 0; t < N; ++t) {
            double val = c[R.G() + 2 * n + t * S];
            for (int j = 0; j < n; ++j) val += c[R.WP() + t * S + n * j] * c[R.E() + j] * xs[j] + c[R.WM() + t * S + n * j] * xs[n + j];
            refl += sf[SF.RL(los, t)] * val;
        }
        const double A = att * azf[0];
        ground = A * (refl + (g.include_ss ? sf[SF.BL(los)] * Tg : 0.0));
        total[0] += ground;
        if (derivs) {
            gw = A;
            const size_t t3 = tid * S;
            const double* so = v.surf_ops + size_t(w) * N * (M2 + 1);
            for (int i = 0; i < M2; ++i) {
                double acc = 0;
                for (int t = 0; t < N; ++t) acc += sf[SF.RL(los, t)] * so[t * (M2 + 1) + i];
                v.z[(size_t(nlyr - 1) * M2 + i) * nt3 + t3] += gw * acc;
            }
            if (g.include_ss) {   // direct-beam part (the diffuse part goes through gw in the gradient contraction)
                for (int a = 0; a < v.brdf_nargs; ++a) v.grad[surf_grad_row(g, v, a) * nt3 + t3] += A * sf[size_t(1 + a) * SF.NP() + SF.BL(los)] * Tg;
                v.grad[size_t(4 * nlyr + 1) * nt3 + t3] += A * sf[SF.BL(los)];
            }
        }
    } else if (m == 0) {
        const double* c = vrec_at(v, nlyr, nlyr - 1, w);
        const double* xs = vxsol_at(v, nlyr, nlyr - 1, w);
        const double rho = b.albedo[w];
        double ipb = 0;
        for (int i = 0; i < N; ++i) {
            double val = c[R.G() + 2 * n + i * S];
            for (int j = 0; j < n; ++j) val += c[R.WP() + i * S + n * j] * c[R.E() + j] * xs[j] + c[R.WM() + i * S + n * j] * xs[n + j];
            ipb += g.wt[i] * g.mu[i] * val;
        }
        const double direct = g.include_ss ? g.csz[s] / kPi * b.T[size_t(nlyr) * nw + w] : 0.0;
        const double A = att * azf[0];
        // thermal pass: the surface emission joins the reflected intensity where the direct bounce does (sktran_do_layerarray.cpp:255-266)
        ground = A * (rho * (2.0 * ipb + direct) + ((g.include_ss && b.surf_em) ? b.surf_em[w] : 0.0));
        total[0] += ground;
        if (derivs) {
            gw = A;
            const size_t t3 = tid * S;
            const double* ops = vops_at(v, g, nlyr - 1, w);
            for (int i = 0; i < M2; ++i) v.z[(size_t(nlyr - 1) * M2 + i) * nt3 + t3] += gw * ops[O.UG() + i];
            // direct-beam part of the reflected intensity (the diffuse part goes through gw in the gradient contraction)
            v.grad[size_t(4 * nlyr) * nt3 + t3] += A * direct;
            v.grad[size_t(4 * nlyr + 1) * nt3 + t3] += b.thermal_pass ? ((g.include_ss && b.surf_em) ? A : 0.0) : (g.include_ss ? A * rho * g.csz[s] / kPi : 0.0);
        }
    }
    for (int s1 = 0; s1 < S; ++s1) v.radiance[tid * S + s1] += total[s1];
    if (!derivs) return;
    for (int s1 = 0; s1 < S; ++s1) {
        const size_t t3 = tid * S + s1;
        // attenuation of everything below a layer: dI/dtau_q = -(lf_q / mu) * (radiance entering the layer from below)
        double below = (s1 == 0) ? ground : 0.0;
        for (int p = nlyr - 1; p >= first; --p) {
            const double lf = (p == first) ? g.pp_lf[los] : 1.0;
            v.grad[size_t(skd::TH_TAU * nlyr + p) * nt3 + t3] -= lf / mu * below;
            below += v.qlbuf[(size_t(p) * NL) * nt3 + t3];
        }
        adjoint_backward_gradient(g, b, v, s, m, w, t3, nt3, (s1 == 0) ? gw : 0.0, false, yv, carry.sub(size_t(s1) * n));
    }
}

// Scattering-group gradients of the layers -> native derivatives [2 ngrid + 1 + g ngrid + q]: layer p feeds its stencil points q
// with weight w_pq omega_q k_q / (layer scattering extinction)  (sktran_do_layerarray.cpp:502-511,768-775).  Clears the rows
// (one call per solar zenith index).                                      (task = wavelength x LOS x Stokes)
SKD_HD void kv4g_chain(const Geo& g, const Bufs& b, const VBufs& v, int w, size_t tid, size_t nt) {
    const int nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    for (int gi = 0; gi < v.ngroups; ++gi)
        for (int p = 0; p < nlyr; ++p) {
            double& gr = v.grad[size_t(4 * nlyr + 2 + gi * nlyr + p) * nt + tid];
            const double scat = b.om[size_t(p) * nw + w] * b.kbar[size_t(p) * nw + w];
            for (int t = 0; t < 2; ++t) {
                const int q = g.wq_idx[2 * p + t];
                if (q < 0) continue;
                v.dnat[size_t(2 * ngrid + 1 + gi * ngrid + q) * nt + tid] += gr * g.wq_w[2 * p + t] * ssa[q] * ext[q] / scat;
            }
            gr = 0;
        }
}

// Gradient rows of the BRDF arguments beyond the first -> native derivative rows (argument 0 rides the albedo row through skd::k4a_chain).
// Clears the rows (one call per solar zenith index).                      (task = wavelength x LOS x Stokes)
SKD_HD void kv4s_chain(const Geo& g, const VBufs& v, size_t tid, size_t nt) {
    for (int a = 1; a < v.brdf_nargs; ++a) {
        double& gr = v.grad[surf_grad_row(g, v, a) * nt + tid];
        v.dnat[surf_dnat_row(g, v, a) * nt + tid] += gr;
        gr = 0;
    }
}

// (I, Q, U) of one line of sight in the configured Stokes basis:  Q' = C Q - S U,  U' = S Q + C U
SKD_HD void stokes_rotate(const VGeo& vg, int los, const double* in, double* out, int S) {
    out[0] = in[0];
    if (S == 3) {
        out[1] = vg.stokes_c[los] * in[1] - vg.stokes_s[los] * in[2];
        out[2] = vg.stokes_s[los] * in[1] + vg.stokes_c[los] * in[2];
    }
}

}  // namespace skv
