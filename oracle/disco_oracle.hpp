// TEST INFRASTRUCTURE ONLY -- CPU oracle for the DISCO (discrete ordinates) hot path.
//
// A plain C++ restatement of the reference's per-wavelength DISCO solver (NSTOKES = 1 and the polarised
// NSTOKES = 3 form: n = 3 N rows per hemisphere, index r = 3*stream + stokes) with its dense forward-mode linearization, used ONLY by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg.  The product (HIP) path never includes this file.
//
// Parity status: PINNED.  Checked against the reference's own known-answer tests
//   cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp:84-880      (plane parallel, 1e-8)
//   tests/engine/test_1d_solver_regression.py:112-237             (spherical limb+nadir, 5e-7)
//   cpp/lib/tests/sktran_disco/legacy/test_vector.cpp:81-314      (NSTOKES=3: scalar equivalence, Siewert, Coulson)
// see tests/test_oracle_golden.py.
//
// Reference files each stage follows (relative to /root/reference):
//   layer optics + derivative wiring   cpp/lib/sktran_disco/sktran_do_layerarray.cpp:332-477,487-657,729-757
//   SSA dither                         cpp/lib/sktran_disco/sktran_do_opticallayer.cpp:37-40
//   beam transmission / secants        cpp/lib/sktran_disco/sktran_do_layerarray.cpp:890-979
//   S+/S- assembly                     cpp/include/sktran_disco/sktran_do_lpproduct.h:22-263 (3x3 Greek-matrix products :38-78)
//   homogeneous solution               cpp/lib/sktran_disco/sktran_do_rte.cpp:382-553
//   its linearization                  cpp/lib/sktran_disco/sktran_do_rte.cpp:197-298
//   Green's function particular sol.   cpp/lib/sktran_disco/sktran_do_rte.cpp:555-580,902-1332
//   boundary value problem             cpp/lib/sktran_disco/sktran_do_rte.cpp:1620-1790,1897-2294,
//                                      cpp/include/sktran_disco/sktran_do_rte.h:116-345,
//                                      cpp/include/sktran_disco/sktran_do_linalg.h:209-295
//   plane-parallel post-processing     cpp/lib/sktran_disco/sktran_do_opticallayer.cpp:93-555,784-938,
//                                      cpp/lib/sktran_disco/sktran_do_layerarray.cpp:5-288,
//                                      cpp/lib/sktran_disco/source_term/do_source_planeparallel.cpp:19-218
//   spherical diffuse table            cpp/lib/sktran_disco/source_term/do_source_diffuse_storage.cpp:435-1098,
//                                      cpp/include/sktran_disco/sktran_do_postprocessing.h:19-188
//   LOS integration                    cpp/lib/sourceintegrator/sourceintegrator.cpp:521-580,
//                                      cpp/lib/sktran_disco/source_term/do_source_interpolated_pp.cpp:95-211
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../sasktran2_amd/csrc/host/geometry.hpp"
#include "linalg.hpp"

namespace oracle {

constexpr double kPi = 3.14159265358979323846;
constexpr double kGreensEps = 1e-4;   // SKTRAN_DO_GREENS_EPS, sktran_do_types.h:11
constexpr double kSSADither = 1e-9;   // sktran_do_specs.h:104
constexpr double kMinShell = 1e-4;    // MINIMUM_SHELL_SIZE_M

struct AtmosphereView {
    int ngrid = 0, nwavel = 0, nleg = 0;  // nleg = number of Legendre orders
    int nstokes = 1;                     // 3: four coefficients (a1, a2, a3, b1) per order, order fastest after the coefficient
    const double* extinction = nullptr;  // [ngrid, nwavel] column-major (location fastest)
    const double* ssa = nullptr;         // [ngrid, nwavel]
    const double* leg = nullptr;         // [nleg (* 4), ngrid, nwavel]
    const double* solar_irradiance = nullptr;  // [nwavel] (nullptr -> 1)
    const double* albedo = nullptr;      // [nwavel] Lambertian
    // non-Lambertian surfaces (cpp/include/sasktran2/atmosphere/surface.h:151-352): 0 Lambertian (albedo above), 1 SnowKokhanovsky, 2 MODIS;
    // brdf_args [nargs, nwavel] column-major (argument fastest, surface.h:366-372); plane-parallel / pseudo-spherical post-processing only
    int brdf_kind = 0, brdf_nargs = 1;
    const double* brdf_args = nullptr;
    const double* delta_f = nullptr;     // [ngrid, nwavel] delta-M truncation fraction of the (already scaled) storage, or null
    // EmissionSource::discrete_ordinates (plane-parallel post-processing only, radiances only): thermal source inside the DO equations
    const double* emission = nullptr;          // [ngrid, nwavel] emission_source of the storage, or null
    const double* surface_emission = nullptr;  // [nwavel], or null
    bool derivatives = false;
    double k(int q, int w) const { return extinction[size_t(w) * ngrid + q]; }
    double om(int q, int w) const { return ssa[size_t(w) * ngrid + q]; }
    double beta(int l, int q, int w, int c = 0) const {
        if (l >= nleg) return 0.0;
        return nstokes == 1 ? leg[(size_t(w) * ngrid + q) * nleg + l] : leg[((size_t(w) * ngrid + q) * nleg + l) * 4 + c];
    }
};

// native derivative vector of one (wavelength, los): [0,ngrid) d/dk, [ngrid,2ngrid) d/dssa, [2ngrid] d/dalbedo
struct Result {
    std::vector<double> radiance;  // [nlos][nstokes]
    std::vector<double> deriv;     // [nlos][nstokes][nnative]
    std::vector<double> los_od;    // [nlos]
    int num_azimuth_used = 0;
    std::vector<double> flux;      // [flux observer][3]: upwelling, downwelling, actinic (plane-parallel post-processing only)
};

// ---- BRDFs of the reference (cpp/include/sasktran2/atmosphere/surface.h): value and derivative wrt argument k of the (0, 0) element;
//      mu_in / mu_out cosines of the incoming / outgoing zenith angle, phi = azimuth difference, 0 = forward scattering
struct BrdfKernels { double k_vol, k_geo; };
inline BrdfKernels modis_kernels(double mu_in, double mu_out, double phi) {   // surface.h:236-281
    const double csza = mu_in, cvza = mu_out, ssza = std::sqrt(1 - csza * csza), svza = std::sqrt(1 - cvza * cvza);
    const double tsza = ssza / csza, tvza = svza / cvza, craa = -std::cos(phi), sraa = std::sin(phi);
    const double csa = std::max(-1.0, std::min(1.0, csza * cvza + ssza * svza * craa)), sa = std::acos(csa), ssa_ = std::sin(sa);
    BrdfKernels K;
    K.k_vol = ((0.5 * kPi - sa) * csa + ssa_) / (csza + cvza) - 0.25 * kPi;                       // Ross-thick
    const double d2 = tsza * tsza + tvza * tvza - 2 * tsza * tvza * craa;
    const double ct = std::max(-1.0, std::min(1.0, 2 * std::sqrt(d2 + tsza * tsza * tvza * tvza * sraa * sraa) * csza * cvza / (csza + cvza)));
    const double t = std::acos(ct), st = std::sin(t);
    const double o = (t - st * ct) * (csza + cvza) / (kPi * csza * cvza);
    K.k_geo = o - (csza + cvza - 0.5 * (1 + csa)) / (csza * cvza);                                  // Li-sparse-R
    return K;
}
inline void kokhanovsky_parts(double mu_in, double mu_out, double phi, double& r0, double& kk) {   // surface.h:151-204
    const double mus = mu_in, muv = mu_out, ss = std::sqrt(1 - mus * mus), sv = std::sqrt(1 - muv * muv);
    const double cost = std::max(-1.0, std::min(1.0, -mus * muv + ss * sv * std::cos(phi)));
    const double theta = std::acos(cost) * 180.0 / kPi;
    const double pth = 11.1 * std::exp(-0.087 * theta) + 1.1 * std::exp(-0.014 * theta);
    r0 = (1.247 + 1.186 * (mus + muv) + 5.157 * mus * muv + pth) / (4.0 * (mus + muv));
    kk = (3.0 / 7.0) * (1.0 + 2.0 * mus) * (3.0 / 7.0) * (1.0 + 2.0 * muv);
}
inline double brdf_value(int kind, double mu_in, double mu_out, double phi, const double* args) {
    if (kind == 1) {
        double r0, kk;
        kokhanovsky_parts(mu_in, mu_out, phi, r0, kk);
        const double alpha = std::sqrt(4 * kPi * args[0]);
        return r0 * std::exp(-alpha * kk / r0) / kPi;
    }
    if (kind == 2) {
        const BrdfKernels K = modis_kernels(mu_in, mu_out, phi);
        return (args[0] + args[1] * K.k_vol + args[2] * K.k_geo) / kPi;
    }
    // kind 3 (oracle only): the functional BRDF of the reference's "BRDF surface" known answer, cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp:806-845
    // (abs(cos(caz_diff)) * abs(1 - abs(abs(out) - abs(in))) in the legacy convention rho = pi * brdf), scaled by args[0]
    if (kind == 3) return args[0] * std::abs(std::cos(phi)) * std::abs(1 - std::abs(std::abs(mu_out) - std::abs(mu_in))) / kPi;
    return args[0] / kPi;
}
inline double brdf_deriv(int kind, int k, double mu_in, double mu_out, double phi, const double* args) {
    if (kind == 1) {   // surface.h:206-229 with d_args = 1
        double r0, kk;
        kokhanovsky_parts(mu_in, mu_out, phi, r0, kk);
        const double alpha = std::sqrt(4 * kPi * args[0]), factor = -kk / r0;
        return r0 * std::exp(alpha * factor) * (2 / alpha * factor);
    }
    if (kind == 2) {
        if (k == 0) return 1.0 / kPi;
        const BrdfKernels K = modis_kernels(mu_in, mu_out, phi);
        return (k == 1 ? K.k_vol : K.k_geo) / kPi;
    }
    if (kind == 3) return std::abs(std::cos(phi)) * std::abs(1 - std::abs(std::abs(mu_out) - std::abs(mu_in))) / kPi;   // d / d(scale)
    return 1.0 / kPi;
}

class Solver {
    const skgeo::DiscoGeometry& G;
    const int nstr, N, S, n, nlyr, ngrid;  // n = N * S rows per hemisphere, row r = stream (r / S), Stokes component (r % S)
    int ND = 0;       // DO-internal derivative count
    int nnative = 0;  // native derivative count
    bool include_ss;  // single scatter inside DO (plane parallel only)
    // thermal source S(x) = b0 exp(-b1 x) per layer (sktran_do_layerarray.cpp:340-475), its Green's-function coefficients, surface emission
    bool thermal = false;
    std::vector<double> th_b0, th_b1, Ap_th, Am_th;
    double surf_emission = 0;

    // ---- per wavelength layer state
    std::vector<double> tau, ssa, totext, scatext, od_ceil, od_floor, a;  // a[(p*nstr+l)*4 + {a1,a2,a3,b1}]
    std::vector<double> sec, dsec, trans, dtrans;                         // dsec[p*ND+d], dtrans[(lev)*ND+d]
    // derivative wiring: DO deriv d -> list of (native idx, weight*extinction)
    struct Wire { int native; double w; };
    std::vector<std::vector<Wire>> wires;
    std::vector<int> d_layer, d_kind;  // kind: 0 od, 1 ssa, 2 albedo
    std::vector<int> lstart, nloc;
    double albedo = 0, F0 = 1;
    // ---- surface: azimuth expansion of the BRDF at the stream / sun / line-of-sight directions (SurfaceStorage::compute_expansion,
    //      cpp/include/sktran_disco/sktran_do_surface.h:49-94: two 512-point Gauss-Legendre panels over the azimuth) and of its derivative wrt
    //      the selected argument.  Lambertian: albedo for m = 0 (the special case at :53-57)
    int brdf_kind = 0, brdf_nargs = 1, surf_max_order = 1;
    std::vector<double> brdf_arg;                       // arguments of this wavelength
    int brdf_darg = 0;
    std::vector<double> qx, qw;                         // positive nodes / weights of the 512-point rule
    std::vector<double> smp, dsmp;                      // [pair][4 * 256] samples of the BRDF (its derivative) over the azimuth
    std::vector<double> rss, rsun, rlos, rlossun, drss, drsun, drlos, drlossun;   // order m: [N out][N in], [N], [nlos][N], [nlos]
    // spherical post-processing: the ground source is evaluated at the cos-angle grid points a line of sight interpolates between
    // (accumulate_ground_sources, do_source_diffuse_storage.cpp:435-695): expansion at (grid angle, stream) for the needed angles of this sza
    std::vector<int> gnd_aidx;
    std::vector<double> rgnd, drgnd;                    // [needed angle][N]
    int npair() const { return N * N + N + int(G.pp_los.size()) * (N + 1) + int(gnd_aidx.size()) * N; }
    void pair_angles(int pr, double& mu_out, double& mu_in) const {
        const int nl_ = int(G.pp_los.size());
        if (pr < N * N) { mu_out = G.mu[pr / N]; mu_in = G.mu[pr % N]; return; }
        pr -= N * N;
        if (pr < N) { mu_out = G.mu[pr]; mu_in = G.csz[cur_sza]; return; }
        pr -= N;
        if (pr < nl_ * N) { mu_out = G.pp_los[pr / N].coszenith; mu_in = G.mu[pr % N]; return; }
        pr -= nl_ * N;
        if (pr < nl_) { mu_out = G.pp_los[pr].coszenith; mu_in = G.csz[cur_sza]; return; }
        pr -= nl_;
        mu_out = G.cos_angles[gnd_aidx[pr / N]]; mu_in = G.mu[pr % N];
    }
    void sample_surface() {
        gnd_aidx.clear();
        if (G.spherical_pp)
            for (int aidx = 0; aidx < G.nangle; ++aidx)
                if (ground_slot[size_t(cur_sza) * G.nangle + aidx] >= 0) gnd_aidx.push_back(aidx);
        if (brdf_kind == 0) return;
        if (qx.empty()) {
            std::vector<double> x, w;
            skgeo::gauss_legendre(512, x, w);
            for (size_t i = 0; i < x.size(); ++i)
                if (x[i] > 0) { qx.push_back(x[i]); qw.push_back(w[i]); }
        }
        const int np = npair(), nq = int(qx.size());
        smp.assign(size_t(np) * 4 * nq, 0.0);
        dsmp.assign(smp.size(), 0.0);
        for (int pr = 0; pr < np; ++pr) {
            double mo, mi;
            pair_angles(pr, mo, mi);
            for (int i = 0; i < nq; ++i) {
                const double a[4] = {0.5 * qx[i] + 0.5, -0.5 * qx[i] + 0.5, 0.5 * qx[i] - 0.5, -0.5 * qx[i] - 0.5};
                for (int t = 0; t < 4; ++t) {
                    smp[(size_t(pr) * nq + i) * 4 + t] = brdf_value(brdf_kind, mi, mo, kPi * a[t], brdf_arg.data());
                    dsmp[(size_t(pr) * nq + i) * 4 + t] = brdf_deriv(brdf_kind, brdf_darg, mi, mo, kPi * a[t], brdf_arg.data());
                }
            }
        }
    }
    void expand_surface(int m) {
        const int nl_ = int(G.pp_los.size());
        rss.assign(size_t(N) * N, 0.0); rsun.assign(N, 0.0); rlos.assign(size_t(nl_) * N, 0.0); rlossun.assign(nl_, 0.0);
        drss.assign(rss.size(), 0.0); drsun.assign(N, 0.0); drlos.assign(rlos.size(), 0.0); drlossun.assign(nl_, 0.0);
        rgnd.assign(gnd_aidx.size() * N, 0.0); drgnd.assign(rgnd.size(), 0.0);
        if (m >= surf_max_order) return;
        auto put = [&](int pr, double v, double dv) {
            if (pr < N * N) { rss[pr] = v; drss[pr] = dv; return; }
            pr -= N * N;
            if (pr < N) { rsun[pr] = v; drsun[pr] = dv; return; }
            pr -= N;
            if (pr < nl_ * N) { rlos[pr] = v; drlos[pr] = dv; return; }
            pr -= nl_ * N;
            if (pr < nl_) { rlossun[pr] = v; drlossun[pr] = dv; return; }
            pr -= nl_;
            rgnd[pr] = v; drgnd[pr] = dv;
        };
        const int np = npair();
        if (brdf_kind == 0) {
            for (int pr = 0; pr < np; ++pr) put(pr, albedo, 1.0);
            return;
        }
        const int nq = int(qx.size());
        for (int pr = 0; pr < np; ++pr) {
            double r = 0, dr = 0;
            for (int i = 0; i < nq; ++i) {
                const double a[4] = {0.5 * qx[i] + 0.5, -0.5 * qx[i] + 0.5, 0.5 * qx[i] - 0.5, -0.5 * qx[i] - 0.5};
                for (int t = 0; t < 4; ++t) {
                    const double cw = 0.5 * qw[i] * std::cos(m * kPi * a[t]);
                    r += cw * smp[(size_t(pr) * nq + i) * 4 + t];
                    dr += cw * dsmp[(size_t(pr) * nq + i) * 4 + t];
                }
            }
            const double f = 0.5 * kPi * (m == 0 ? 1.0 : 2.0);
            put(pr, r * f, dr * f);
        }
    }
    // ---- generalized spherical functions packed as [..][l][{P,R,T}]
    std::vector<double> f_mu, f_csz, f_ang, f_los;
    // ---- per azimuth-order solution
    std::vector<double> eigk, deigk, Wp, Wm, dWp, dWm, Ap, Am, dAp, dAm;
    std::vector<double> Gpt, Gpb, Gmt, Gmb, dGpt, dGpb, dGmt, dGmb;
    std::vector<double> Lc, Mc, dLc, dMc;
    std::vector<double> Splus, Sminus, dSplus, dSminus, Gam, dGam;
    oracle_la::RealEigen eigsolver;
    oracle_la::FullPivLU fullpiv;
    oracle_la::BandMatrix bvp;
    std::vector<double> rhs;  // [M][1+ND]
    // ---- spherical table
    std::vector<int> point_slot;   // [sza][alt][angle] -> slot or -1
    std::vector<int> ground_slot;  // [sza][angle]
    int npoint = 0, nground = 0;
    std::vector<double> tab_val;    // [m][slot][S]  (interior slots then ground slots)
    std::vector<double> tab_deriv;  // [m][slot][S][nnative]
    std::vector<uint8_t> converged; // [m][slot] (interior only)
    std::vector<uint8_t> filled;    // [m][slot] value/derivative row written for this wavelength
    bool all_converged = false;

    inline int nl(int p) const { return nloc[p]; }
    size_t hidx(int p, int i, int j) const { return (size_t(p) * n + j) * n + i; }          // W(row i, solution j)
    size_t dhidx(int p, int k, int i, int j) const { return ((size_t(p) * 3 + k) * n + j) * n + i; }
    double rw(int r) const { return G.wt[r / S]; }
    double rmu(int r) const { return G.mu[r / S]; }
    // D = diag(1, 1, -1) per Stokes triple (stokes_negation_factor, sktran_do_polarization_types.h:17-25)
    double neg(int r) const { return (r % S) >= 2 ? -1.0 : 1.0; }
    const double* fmu(int m, int i) const { return &f_mu[(size_t(m) * nstr + i) * nstr * 3]; }
    const double* fcsz(int s, int m) const { return &f_csz[(size_t(s) * nstr + m) * nstr * 3]; }
    const double* fang(int m, int a_) const { return &f_ang[(size_t(m) * G.nangle + a_) * nstr * 3]; }
    const double* flos(int j, int m) const { return &f_los[(size_t(j) * nstr + m) * nstr * 3]; }
    const double* coef(int p, int l) const { return &a[(size_t(p) * nstr + l) * 4]; }

    static void pack(const std::vector<double>& P, const std::vector<double>& R, const std::vector<double>& T, std::vector<double>& out) {
        out.assign(P.size() * 3, 0.0);
        for (size_t i = 0; i < P.size(); ++i) {
            out[3 * i] = P[i];
            if (!R.empty()) { out[3 * i + 1] = R[i]; out[3 * i + 2] = T[i]; }
        }
    }

    // One Legendre order of the (S x S, row-major) product  Pi_l(out) B_l Pi_l(in)  (negation: B_l -> (-1)^(l-m) B_l D)
    //   lp_triple_product, sktran_do_lpproduct.h:22-79
    void triple(bool negation, int l, int m, const double* c4, const double* lp1, const double* lp2, double* out) const {
        double up = 1, lo = 1;
        if (negation) {
            lo = -1;
            if ((l - m) % 2 != 0) { up = -up; lo = -lo; }
        }
        const double P1 = lp1[0], P2 = lp2[0];
        if (S == 1) { out[0] += P1 * P2 * c4[0] * up; return; }
        const double R1 = lp1[1], T1 = lp1[2], R2 = lp2[1], T2 = lp2[2];
        const double a1 = c4[0], a2 = c4[1], a3 = c4[2], b1 = c4[3];
        out[0] += P1 * P2 * a1 * up;
        out[1] += -1.0 * P1 * R2 * b1 * up;
        out[2] += 1.0 * P1 * T2 * b1 * up;
        out[3] += -1.0 * R1 * P2 * b1 * up;
        out[4] += R1 * R2 * a2 * up + T1 * T2 * a3 * lo;
        out[5] += -R1 * T2 * a2 * up - T1 * R2 * a3 * lo;
        out[6] += T1 * P2 * b1 * up;
        out[7] += -T1 * R2 * a2 * up - R1 * T2 * a3 * lo;
        out[8] += T1 * T2 * a2 * up + R1 * R2 * a3 * lo;
    }
    // sum over l >= m for layer p with function tables lp1/lp2 ([l][3])
    void triple_sum(bool negation, int m, int p, const double* lp1, const double* lp2, double* out) const {
        for (int i = 0; i < S * S; ++i) out[i] = 0;
        for (int l = m; l < nstr; ++l) triple(negation, l, m, coef(p, l), lp1 + 3 * l, lp2 + 3 * l, out);
    }

  public:
    Solver(const skgeo::DiscoGeometry& g, bool derivatives, bool ss_in_do)
        : G(g), nstr(g.nstr), N(g.nstr / 2), S(g.nstokes), n(g.nstr / 2 * g.nstokes), nlyr(g.nlyr), ngrid(g.ngrid), include_ss(ss_in_do) {
        if (S != 1 && S != 3) throw std::invalid_argument("NSTOKES must be 1 or 3");
        nnative = derivatives ? 2 * ngrid + 1 : 0;
        lstart.assign(nlyr, 0);
        nloc.assign(nlyr, 0);
        if (derivatives) {
            for (int p = 0; p < nlyr; ++p) {
                lstart[p] = ND;
                nloc[p] = (p == nlyr - 1) ? 3 : 2;
                for (int k = 0; k < nloc[p]; ++k) { d_layer.push_back(p); d_kind.push_back(k); }
                ND += nloc[p];
            }
        }
        pack(G.lp_mu, G.lp_mu_R, G.lp_mu_T, f_mu);
        pack(G.lp_csz, {}, {}, f_csz);  // only P(mu_0) enters (unpolarised sun: first column of the product)
        pack(G.lp_angle, G.lp_angle_R, G.lp_angle_T, f_ang);
        pack(G.lp_los, G.lp_los_R, G.lp_los_T, f_los);
        wires.resize(ND);
        tau.resize(nlyr); ssa.resize(nlyr); totext.resize(nlyr); scatext.resize(nlyr);
        od_ceil.resize(nlyr); od_floor.resize(nlyr); a.resize(size_t(nlyr) * nstr * 4);
        sec.resize(nlyr); dsec.resize(size_t(nlyr) * ND); trans.resize(nlyr + 1); dtrans.resize(size_t(nlyr + 1) * ND);
        eigk.resize(size_t(nlyr) * n); deigk.resize(size_t(nlyr) * 3 * n);
        Wp.resize(size_t(nlyr) * n * n); Wm.resize(Wp.size());
        dWp.resize(size_t(nlyr) * 3 * n * n); dWm.resize(dWp.size());
        Ap.resize(size_t(nlyr) * n); Am.resize(Ap.size()); dAp.resize(size_t(nlyr) * 3 * n); dAm.resize(dAp.size());
        Gpt.resize(size_t(nlyr) * n); Gpb.resize(Gpt.size()); Gmt.resize(Gpt.size()); Gmb.resize(Gpt.size());
        dGpt.resize(size_t(nlyr) * ND * n); dGpb.resize(dGpt.size()); dGmt.resize(dGpt.size()); dGmb.resize(dGpt.size());
        Lc.resize(size_t(nlyr) * n); Mc.resize(Lc.size()); dLc.resize(size_t(nlyr) * ND * n); dMc.resize(dLc.size());
        Splus.resize(size_t(n) * n); Sminus.resize(Splus.size()); dSplus.resize(3 * Splus.size()); dSminus.resize(dSplus.size());
        Gam.resize(Splus.size()); dGam.resize(dSplus.size());
        bvp.resize(2 * n * nlyr, 3 * n - 1, 3 * n - 1);
        rhs.resize(size_t(2 * n) * nlyr * (1 + ND));
        if (G.spherical_pp) {
            point_slot.assign(G.need_point.size(), -1);
            for (size_t i = 0; i < G.need_point.size(); ++i)
                if (G.need_point[i]) point_slot[i] = npoint++;
            ground_slot.assign(G.need_ground.size(), -1);
            for (size_t i = 0; i < G.need_ground.size(); ++i)
                if (G.need_ground[i]) ground_slot[i] = npoint + nground++;
            tab_val.assign(size_t(nstr) * (npoint + nground) * S, 0.0);
            tab_deriv.assign(size_t(nstr) * (npoint + nground) * S * nnative, 0.0);
            converged.assign(size_t(nstr) * npoint, 0);
            filled.assign(size_t(nstr) * (npoint + nground), 0);
        }
    }
    int num_native() const { return nnative; }
    int num_stokes() const { return S; }

  private:
    double* dG(std::vector<double>& g, int p, int d) { return &g[(size_t(p) * ND + d) * n]; }
    double* dLM(std::vector<double>& g, int p, int d) { return &g[(size_t(p) * ND + d) * n]; }

    // ------------------------------------------------------------------ layer optics
    void configure_layers(const AtmosphereView& atm, int w, int szaidx) {
        if (atm.nstokes != S) throw std::invalid_argument("atmosphere and geometry disagree on NSTOKES");
        double ceiling_depth = 0, floor_depth = 0;
        for (int p = 0; p < nlyr; ++p) {
            double dh = G.ceil_h[p] - G.floor_h[p];
            double od = 0, s = 0;
            double* ap = &a[size_t(p) * nstr * 4];
            for (int l = 0; l < nstr * 4; ++l) ap[l] = 0;
            for (int q = 0; q < ngrid; ++q) {
                double wt = G.W(p, q);
                if (wt > 0) {
                    double kext = atm.k(q, w), kscat = atm.om(q, w) * kext;
                    od += kext * wt;
                    s += kscat * wt;
                    // delta-M: the multiple-scatter expansion drops the truncated forward peak (sktran_do_layerarray.cpp:395-405);
                    // the a's carry the f subtraction, b1 does not and is stored with its sign flipped (:405-418)
                    const double f = atm.delta_f ? atm.delta_f[size_t(w) * atm.ngrid + q] : 0.0;
                    for (int l = 0; l < nstr; ++l) {
                        ap[4 * l] += wt * kscat * (atm.beta(l, q, w, 0) - (2 * l + 1) * f / (1 - f));
                        if (S == 3) {
                            ap[4 * l + 1] += wt * kscat * (atm.beta(l, q, w, 1) - f * (2 * l + 1) / (1 - f));
                            ap[4 * l + 2] += wt * kscat * (atm.beta(l, q, w, 2) - f * (2 * l + 1) / (1 - f));
                            ap[4 * l + 3] += -wt * kscat * atm.beta(l, q, w, 3);
                        }
                    }
                }
            }
            if (s > 0) for (int l = 0; l < nstr * 4; ++l) ap[l] /= s;
            else ap[0] = 0;
            s /= od;
            od *= dh;
            floor_depth += od;
            double te = od / dh;
            double se = std::max(te * s, te * kSSADither);
            totext[p] = te;
            scatext[p] = se;
            double m_ssa = se / te;
            if (1 - m_ssa < kSSADither) m_ssa = 1 - kSSADither;
            ssa[p] = m_ssa;
            od_ceil[p] = ceiling_depth;
            od_floor[p] = floor_depth;
            tau[p] = floor_depth - ceiling_depth;
            ceiling_depth = floor_depth;
        }
        thermal = atm.emission != nullptr;
        if (thermal) {
            if (ND) throw std::invalid_argument("the oracle's thermal source is not linearised");
            if (G.spherical_pp) throw std::invalid_argument("the discrete-ordinates thermal source needs plane-parallel post-processing");
            th_b0.assign(nlyr, 0.0); th_b1.assign(nlyr, 0.0); Ap_th.assign(size_t(nlyr) * n, 0.0); Am_th.assign(size_t(nlyr) * n, 0.0);
            surf_emission = atm.surface_emission ? atm.surface_emission[w] : 0.0;
            for (int p = 0; p < nlyr; ++p) {
                int min_q = -1, max_q = -1;
                for (int q = 0; q < ngrid; ++q)
                    if (G.W(p, q) > 0) { if (min_q < 0) min_q = q; max_q = q; }
                const double top = max_q >= 0 ? atm.emission[size_t(w) * ngrid + max_q] : 0.0;
                const double bot = (min_q >= 0 && min_q != max_q) ? atm.emission[size_t(w) * ngrid + min_q] : top;
                th_b0[p] = top;
                if (tau[p] > 1e-10 && top > 1e-30 && bot > 1e-30 && std::abs(top - bot) > 1e-15 * std::max(top, bot)) th_b1[p] = std::log(top / bot) / tau[p];
            }
        }
        albedo = atm.albedo ? atm.albedo[w] : 0.0;
        brdf_kind = atm.brdf_kind;
        brdf_nargs = atm.brdf_nargs;
        brdf_darg = surface_deriv_arg;
        surf_max_order = (brdf_kind == 0) ? 1 : nstr;   // BRDF::max_azimuthal_order: 1 for Lambertian, unbounded otherwise (surface.h:88-90,142)
        if (brdf_kind != 0) {
            if (!atm.brdf_args) throw std::invalid_argument("oracle: brdf_args missing");
            brdf_arg.assign(atm.brdf_args + size_t(w) * brdf_nargs, atm.brdf_args + size_t(w + 1) * brdf_nargs);
        }
        F0 = atm.solar_irradiance ? atm.solar_irradiance[w] : 1.0;
        // derivative wiring
        for (int d = 0; d < ND; ++d) {
            auto& wr = wires[d];
            wr.clear();
            int p = d_layer[d];
            double dh = G.ceil_h[p] - G.floor_h[p];
            if (d_kind[d] == 0) {
                for (int q = 0; q < ngrid; ++q)
                    if (G.W(p, q) > 0) wr.push_back({q, G.W(p, q) * dh});
            } else if (d_kind[d] == 1) {
                for (int q = 0; q < ngrid; ++q)
                    if (G.W(p, q) > 0) {
                        wr.push_back({q + ngrid, G.W(p, q) * (atm.k(q, w) / totext[p])});
                        wr.push_back({q, G.W(p, q) * ((atm.om(q, w) - ssa[p]) / totext[p])});
                    }
            } else {
                wr.push_back({2 * ngrid, 1.0});
            }
        }
        // beam transmission
        trans[0] = 0;
        for (int d = 0; d < ND; ++d) dtrans[d] = 0;
        for (int p = 0; p < nlyr; ++p) {
            double* dt = &dtrans[size_t(p + 1) * ND];
            double* dtp = &dtrans[size_t(p) * ND];
            double* ds = &dsec[size_t(p) * ND];
            trans[p + 1] = 0;
            for (int d = 0; d < ND; ++d) dt[d] = 0;
            if (G.chap(szaidx, p, 0) < 0) {
                trans[p + 1] = -1.0;
                sec[p] = 1.0 / G.csz[szaidx];
                for (int d = 0; d < ND; ++d) ds[d] = 0;
                continue;
            }
            for (int r = 0; r < nlyr; ++r) {
                double c = G.chap(szaidx, p, r);
                trans[p + 1] += c * tau[r];
                if (ND) dt[lstart[r]] += c;  // thickness derivative is 1 for the od-derivative of layer r
            }
            sec[p] = (trans[p + 1] - trans[p]) / tau[p];
            for (int d = 0; d < ND; ++d) ds[d] = (dt[d] - dtp[d]) / tau[p];
            if (ND) ds[lstart[p]] -= 1.0 / tau[p] * sec[p];
        }
        trans[0] = F0;
        for (int p = 0; p < nlyr; ++p) {
            double* dt = &dtrans[size_t(p + 1) * ND];
            if (trans[p + 1] < 0) {
                trans[p + 1] = 0.0;
                sec[p] = 1e10;
            } else {
                trans[p + 1] = std::exp(-trans[p + 1]) * F0;
            }
            for (int d = 0; d < ND; ++d) dt[d] = -dt[d] * trans[p + 1];
        }
    }

    // local derivative seeds of layer p: d_optical_depth, d_SSA, d_albedo for local index k
    static double d_od(int k) { return k == 0 ? 1.0 : 0.0; }
    static double d_ssa(int k) { return k == 1 ? 1.0 : 0.0; }
    static double d_alb(int k) { return k == 2 ? 1.0 : 0.0; }

    // ------------------------------------------------------------------ homogeneous solution
    void solve_homogeneous(int m, int p) {
        const int nk = nl(p);
        const double om = ssa[p];
        double spb[9], etab[9];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                triple_sum(false, m, p, fmu(m, i), fmu(m, j), spb);
                triple_sum(true, m, p, fmu(m, i), fmu(m, j), etab);
                const double fac = -0.5 * G.wt[j] / G.mu[i];
                for (int s1 = 0; s1 < S; ++s1)
                    for (int s2 = 0; s2 < S; ++s2) {
                        const int r = i * S + s1, c = j * S + s2;
                        double sp = spb[s1 * S + s2] * fac, eta = etab[s1 * S + s2] * fac;
                        for (int k = 0; k < nk; ++k) {
                            // d_legendre = 0 for the od/ssa/albedo derivatives
                            double dsp = sp * d_ssa(k), deta = eta * d_ssa(k);
                            dSminus[(size_t(k) * n + r) * n + c] = dsp - deta;
                            dSplus[(size_t(k) * n + r) * n + c] = dsp + deta;
                        }
                        sp *= om;
                        eta *= om;
                        if (r == c) sp += 1 / G.mu[i];
                        Sminus[size_t(r) * n + c] = sp - eta;
                        Splus[size_t(r) * n + c] = sp + eta;
                    }
            }
        // eigmtx = S- S+
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int k = 0; k < n; ++k) s += Sminus[size_t(i) * n + k] * Splus[size_t(k) * n + j];
                Gam[size_t(i) * n + j] = s;
            }
        for (int k = 0; k < nk; ++k) {
            const double* dsm = &dSminus[size_t(k) * n * n];
            const double* dsp = &dSplus[size_t(k) * n * n];
            double* dg = &dGam[size_t(k) * n * n];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0;
                    for (int r = 0; r < n; ++r) s += dsm[size_t(i) * n + r] * Splus[size_t(r) * n + j] + Sminus[size_t(i) * n + r] * dsp[size_t(r) * n + j];
                    dg[size_t(i) * n + j] = s;
                }
        }
        std::vector<double> wr(n), wi(n), X(size_t(n) * n), MXm(size_t(n) * n);
        if (!eigsolver.compute(n, Gam.data(), wr.data(), wi.data(), X.data())) {
            // fallback used by the reference: diagonal entries / identity vectors
            for (int i = 0; i < n; ++i) {
                wr[i] = Gam[size_t(i) * n + i];
                for (int j = 0; j < n; ++j) X[size_t(i) * n + j] = (i == j);
            }
        }
        for (int j = 0; j < n; ++j) {
            double nrm = 0;
            for (int i = 0; i < n; ++i) nrm += X[size_t(i) * n + j] * X[size_t(i) * n + j];
            nrm = std::sqrt(nrm);
            for (int i = 0; i < n; ++i) X[size_t(i) * n + j] /= nrm;
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0;
                for (int r = 0; r < n; ++r) s += Splus[size_t(i) * n + r] * X[size_t(r) * n + j];
                MXm[size_t(i) * n + j] = s;
            }
        double* kk = &eigk[size_t(p) * n];
        for (int j = 0; j < n; ++j) {
            if (wr[j] <= 0 && S == 1) throw std::runtime_error("An homogeneous solution was found to be imaginary");
            kk[j] = std::sqrt(std::abs(wr[j]));
            for (int i = 0; i < n; ++i) {
                Wp[hidx(p, i, j)] = 0.5 * (X[size_t(i) * n + j] + MXm[size_t(i) * n + j] / kk[j]);
                Wm[hidx(p, i, j)] = 0.5 * (X[size_t(i) * n + j] - MXm[size_t(i) * n + j] / kk[j]);
            }
        }
        if (nk == 0) return;
        // linearization: bordered (n+1) systems; partial pivoting for NSTOKES = 1, complete pivoting for 3 (sktran_do_rte.cpp:249-263)
        const int n1 = n + 1;
        std::vector<double> lhs(size_t(n1) * n1), sol(n1);
        std::vector<int> piv(n1);
        for (int j = 0; j < n; ++j) {
            bool factored = false;
            for (int k = 0; k < nk; ++k) {
                const double* dg = &dGam[size_t(k) * n * n];
                bool zero = true;
                for (int i = 0; i < n; ++i) {
                    double s = 0;
                    for (int r = 0; r < n; ++r) s += dg[size_t(i) * n + r] * X[size_t(r) * n + j];
                    sol[i] = -s;
                    if (sol[i] != 0.0) zero = false;
                }
                sol[n] = 0;
                // the reference checks the whole RHS block; with od/albedo columns identically zero this is per column here
                if (zero) {
                    for (int i = 0; i <= n; ++i) sol[i] = 0;
                } else {
                    if (!factored) {
                        for (int r = 0; r < n; ++r) {
                            for (int cidx = 0; cidx < n; ++cidx) lhs[size_t(r) * n1 + cidx] = Gam[size_t(r) * n + cidx];
                            lhs[size_t(r) * n1 + r] -= wr[j];
                            lhs[size_t(r) * n1 + n] = -2 * kk[j] * X[size_t(r) * n + j];
                        }
                        for (int cidx = 0; cidx < n; ++cidx) lhs[size_t(n) * n1 + cidx] = X[size_t(cidx) * n + j];
                        lhs[size_t(n) * n1 + n] = 0;
                        if (S == 1) oracle_la::lu_factor(n1, lhs.data(), piv.data());
                        else fullpiv.compute(n1, lhs.data());
                        factored = true;
                    }
                    if (S == 1) oracle_la::lu_solve(n1, lhs.data(), piv.data(), sol.data());
                    else fullpiv.solve(sol.data());
                }
                double dk = sol[n];
                deigk[(size_t(p) * 3 + k) * n + j] = dk;
                const double* dsp = &dSplus[size_t(k) * n * n];
                for (int i = 0; i < n; ++i) {
                    double t1 = 0, t2 = 0, t3 = 0;
                    for (int r = 0; r < n; ++r) {
                        t1 += Splus[size_t(i) * n + r] * X[size_t(r) * n + j];
                        t2 += dsp[size_t(i) * n + r] * X[size_t(r) * n + j];
                        t3 += Splus[size_t(i) * n + r] * sol[r];
                    }
                    double first = 0.5 * ((-dk / wr[j]) * t1 + (1 / kk[j]) * t2);
                    dWp[dhidx(p, k, i, j)] = first + 0.5 * (sol[i] + t3 / kk[j]);
                    dWm[dhidx(p, k, i, j)] = -first + 0.5 * (sol[i] - t3 / kk[j]);
                }
            }
        }
    }

    // ------------------------------------------------------------------ Green's function particular solution
    void solve_particular(int m, int p, int szaidx) {
        const int nk = nl(p), ls = lstart[p];
        const double om = ssa[p];
        std::vector<double> Qp(n), Qm(n), dQp(size_t(3) * n), dQm(size_t(3) * n);
        bool allzero = true;
        double bneg[9], bpos[9];
        for (int i = 0; i < N; ++i) {
            // first column of the product with the unpolarised sun (single_scat_st, sktran_do_lpproduct.h:339-384)
            triple_sum(true, m, p, fmu(m, i), fcsz(szaidx, m), bneg);
            triple_sum(false, m, p, fmu(m, i), fcsz(szaidx, m), bpos);
            const double factor = (2 - (m == 0 ? 1 : 0)) * (1.0 / (4.0 * kPi)) * G.wt[i];
            for (int s1 = 0; s1 < S; ++s1) {
                const int r = i * S + s1;
                double sneg = bneg[s1 * S] * factor, spos = bpos[s1 * S] * factor;
                for (int k = 0; k < nk; ++k) {
                    dQm[size_t(k) * n + r] = sneg * d_ssa(k);
                    dQp[size_t(k) * n + r] = spos * d_ssa(k);
                    if (dQm[size_t(k) * n + r] != 0 || dQp[size_t(k) * n + r] != 0) allzero = false;
                }
                Qm[r] = sneg * om;
                Qp[r] = spos * om;
                if (Qm[r] != 0 || Qp[r] != 0) allzero = false;
            }
        }
        double* gpt = &Gpt[size_t(p) * n]; double* gpb = &Gpb[size_t(p) * n];
        double* gmt = &Gmt[size_t(p) * n]; double* gmb = &Gmb[size_t(p) * n];
        for (int i = 0; i < n; ++i) { gpt[i] = gpb[i] = gmt[i] = gmb[i] = 0; Ap[size_t(p) * n + i] = Am[size_t(p) * n + i] = 0; }
        for (int k = 0; k < 3; ++k) for (int i = 0; i < n; ++i) dAp[(size_t(p) * 3 + k) * n + i] = dAm[(size_t(p) * 3 + k) * n + i] = 0;
        if (ND) {
            std::memset(dG(dGpt, p, 0), 0, sizeof(double) * ND * n);
            std::memset(dG(dGpb, p, 0), 0, sizeof(double) * ND * n);
            std::memset(dG(dGmt, p, 0), 0, sizeof(double) * ND * n);
            std::memset(dG(dGmb, p, 0), 0, sizeof(double) * ND * n);
        }
        if (thermal && m == 0) thermal_particular(p);
        if (allzero) return;
        const double T = trans[p], sc = sec[p], th = tau[p];
        const double* dT = &dtrans[size_t(p) * ND];
        const double* dsc = &dsec[size_t(p) * ND];
        const double exp_ts = std::exp(-th * sc);
        std::vector<double> dCp(ND), dCm(ND);
        for (int i = 0; i < n; ++i) {  // solution index
            double norm = 0, Apv = 0, Amv = 0;
            double dnorm[3] = {0, 0, 0}, dApv[3] = {0, 0, 0}, dAmv[3] = {0, 0, 0};
            for (int j = 0; j < n; ++j) {
                double hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                norm += rw(j) * rmu(j) * (hp * hp - hm * hm);
                Apv += Qp[j] * hp + Qm[j] * hm;
                Amv += Qm[j] * hp + Qp[j] * hm;
                for (int k = 0; k < nk; ++k) {
                    double dhp = dWp[dhidx(p, k, j, i)], dhm = dWm[dhidx(p, k, j, i)];
                    dnorm[k] += rw(j) * rmu(j) * (2.0 * dhp * hp - 2.0 * dhm * hm);
                    dApv[k] += Qp[j] * dhp + dQp[size_t(k) * n + j] * hp + Qm[j] * dhm + dQm[size_t(k) * n + j] * hm;
                    dAmv[k] += Qm[j] * dhp + dQm[size_t(k) * n + j] * hp + Qp[j] * dhm + dQp[size_t(k) * n + j] * hm;
                }
            }
            Apv /= norm;
            Amv /= norm;
            for (int k = 0; k < nk; ++k) {
                dApv[k] = dApv[k] / norm - dnorm[k] * Apv / norm;
                dAmv[k] = dAmv[k] / norm - dnorm[k] * Amv / norm;
                dAp[(size_t(p) * 3 + k) * n + i] = dApv[k];
                dAm[(size_t(p) * 3 + k) * n + i] = dAmv[k];
            }
            Ap[size_t(p) * n + i] = Apv;
            Am[size_t(p) * n + i] = Amv;
            const double ev = eigk[size_t(p) * n + i];
            const double exp_te = std::exp(-th * ev);
            double Cp, Cm;
            if (std::abs(sc - ev) > kGreensEps) {
                double den = sc - ev;
                Cp = T * (exp_te - exp_ts) / den;
                for (int d = 0; d < ND; ++d)
                    dCp[d] = (exp_te - exp_ts) / den * dT[d] + dsc[d] * (T * th * exp_ts - Cp) / den;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dCp[ls + k] += dk / den * (Cp - T * th * exp_te);
                    dCp[ls + k] -= d_od(k) * T / den * (ev * exp_te - sc * exp_ts);
                }
            } else {
                Cp = T * exp_te * th * (1 - th / 2 * (sc - ev));
                for (int d = 0; d < ND; ++d)
                    dCp[d] = exp_te * th * (1 - th / 2 * (sc - ev)) * dT[d] + -1.0 * dsc[d] * th / 2.0 * th * T * exp_te;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dCp[ls + k] += dk * Cp * -1.0 * th;
                    dCp[ls + k] += dk * th / 2.0 * th * T * exp_te;
                    dCp[ls + k] += d_od(k) * T * exp_te * (1 - th * (sc - ev));
                    dCp[ls + k] += -1.0 * ev * Cp * d_od(k);
                }
            }
            if (std::abs(sc + ev) > kGreensEps) {
                double den = sc + ev;
                double exp_sum = std::exp(-th * (sc + ev));
                Cm = T * (1 - exp_ts * exp_te) / den;
                for (int d = 0; d < ND; ++d)
                    dCm[d] = (1 - exp_ts * exp_te) / den * dT[d] + dsc[d] / den * (T * th * exp_sum - Cm);
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dCm[ls + k] += dk / den * (T * th * exp_sum - Cm);
                    dCm[ls + k] += d_od(k) * T * exp_sum;
                }
            } else {
                Cm = T * th * (1 - th / 2 * (sc + ev));
                for (int d = 0; d < ND; ++d)
                    dCm[d] = dT[d] * th * (1 - th / 2 * (sc + ev)) + dsc[d] * -1.0 * th / 2 * th * T;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dCm[ls + k] += dk * T * th * th / 2 * -1.0;
                    dCm[ls + k] += d_od(k) * T * (1 - th * (sc + ev));
                }
            }
            for (int j = 0; j < n; ++j) {
                const double ng = neg(j);  // G- carries D (sktran_do_rte.cpp:1239-1251)
                double hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                gpt[j] += Amv * Cm * hm;
                gmt[j] += ng * Amv * Cm * hp;
                gpb[j] += Apv * Cp * hp;
                gmb[j] += ng * Apv * Cp * hm;
                for (int d = 0; d < ND; ++d) {
                    dG(dGpt, p, d)[j] += Amv * hm * dCm[d];
                    dG(dGmt, p, d)[j] += ng * Amv * hp * dCm[d];
                    dG(dGpb, p, d)[j] += Apv * hp * dCp[d];
                    dG(dGmb, p, d)[j] += ng * Apv * hm * dCp[d];
                }
                for (int k = 0; k < nk; ++k) {
                    double dhp = dWp[dhidx(p, k, j, i)], dhm = dWm[dhidx(p, k, j, i)];
                    dG(dGpt, p, ls + k)[j] += dAmv[k] * Cm * hm + Amv * Cm * dhm;
                    dG(dGmt, p, ls + k)[j] += ng * (dAmv[k] * Cm * hp + Amv * Cm * dhp);
                    dG(dGpb, p, ls + k)[j] += dApv[k] * Cp * hp + Apv * Cp * dhp;
                    dG(dGmb, p, ls + k)[j] += ng * (dApv[k] * Cp * hm + Apv * Cp * dhm);
                }
            }
        }
    }

    // thermal particular solution (RTESolver::solveParticularGreenThermal, sktran_do_rte.cpp:1335-1618): the solar one with Q+ = Q- = w (1 - ssa)
    // on the intensity rows, b0 for the beam transmission and b1 for the average secant; added to the G vectors (values only)
    void thermal_particular(int p) {
        const double b0 = th_b0[p], b1 = th_b1[p], th = tau[p];
        const double exp_tb = std::exp(-th * b1);
        double* gpt = &Gpt[size_t(p) * n]; double* gpb = &Gpb[size_t(p) * n];
        double* gmt = &Gmt[size_t(p) * n]; double* gmb = &Gmb[size_t(p) * n];
        for (int i = 0; i < n; ++i) {
            double norm = 0, Apv = 0, Amv = 0;
            for (int j = 0; j < n; j += S) {
                const double Q = G.wt[j / S] * (1.0 - ssa[p]);
                const double hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                norm += G.wt[j / S] * G.mu[j / S] * (hp * hp - hm * hm);
                Apv += Q * hp + Q * hm;
                Amv += Q * hp + Q * hm;
            }
            Apv /= norm;
            Amv /= norm;
            Ap_th[size_t(p) * n + i] = Apv;
            Am_th[size_t(p) * n + i] = Amv;
            const double ev = eigk[size_t(p) * n + i], exp_te = std::exp(-th * ev);
            double Cp, Cm;
            if (std::abs(b1 - ev) > kGreensEps) Cp = b0 * (exp_te - exp_tb) / (b1 - ev);
            else Cp = b0 * exp_te * th * (1 - th / 2 * (b1 - ev));
            if (std::abs(b1 + ev) > kGreensEps) Cm = b0 * (1 - exp_tb * exp_te) / (b1 + ev);
            else Cm = b0 * th * (1 - th / 2 * (b1 + ev));
            for (int j = 0; j < n; j += S) {
                const double ng = neg(j), hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                gpt[j] += Amv * Cm * hm;
                gmt[j] += ng * Amv * Cm * hp;
                gpb[j] += Apv * Cp * hp;
                gmb[j] += ng * Apv * Cp * hm;
            }
        }
    }

    // ------------------------------------------------------------------ boundary value problem
    // the surface reflects the intensity only, into the intensity only (v_plus/v_minus, sktran_do_rte.h:116-150: "TODO: Polarized surface");
    // rss / rsun / rlos / rlossun hold the order's expansion (expand_surface), zero from the BRDF's max_azimuthal_order on
    bool refl_row(int r) const { return (r % S) == 0; }

    void solve_bvp(int m) {
        const int n2 = 2 * n, M = n2 * nlyr, nr = 1 + ND;
        bvp.set_zero();
        std::fill(rhs.begin(), rhs.end(), 0.0);
        auto E = [&](int p, int j) { return std::exp(-std::abs(eigk[size_t(p) * n + j]) * tau[p]); };
        const bool refl = (m < surf_max_order);
        const double kd = (m == 0) ? 2.0 : 1.0;
        // TOA
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) {
                bvp.at(i, j) = Wp[hidx(0, i, j)];
                bvp.at(i, j + n) = Wm[hidx(0, i, j)] * E(0, j);
            }
            rhs[size_t(i) * nr] = -Gpt[i];
            for (int d = 0; d < ND; ++d) rhs[size_t(i) * nr + 1 + d] = -dG(dGpt, 0, d)[i];
        }
        // continuity (the "minus" rows carry D, sktran_do_rte.cpp:1971-1990)
        for (int p = 1; p < nlyr; ++p) {
            int r0 = n + (p - 1) * n2, cu = (p - 1) * n2, cl = p * n2;
            for (int i = 0; i < n; ++i) {
                const double ng = neg(i);
                for (int j = 0; j < n; ++j) {
                    double eu = E(p - 1, j), el = E(p, j);
                    bvp.at(r0 + i + n, cu + j) = Wp[hidx(p - 1, i, j)] * eu;
                    bvp.at(r0 + i + n, cl + j) = -Wp[hidx(p, i, j)];
                    bvp.at(r0 + i, cu + j) = ng * Wm[hidx(p - 1, i, j)] * eu;
                    bvp.at(r0 + i, cl + j) = -ng * Wm[hidx(p, i, j)];
                    bvp.at(r0 + i + n, cu + n + j) = Wm[hidx(p - 1, i, j)];
                    bvp.at(r0 + i + n, cl + n + j) = -Wm[hidx(p, i, j)] * el;
                    bvp.at(r0 + i, cu + n + j) = ng * Wp[hidx(p - 1, i, j)];
                    bvp.at(r0 + i, cl + n + j) = -ng * Wp[hidx(p, i, j)] * el;
                }
                rhs[size_t(r0 + i) * nr] = -Gmb[size_t(p - 1) * n + i] + Gmt[size_t(p) * n + i];
                rhs[size_t(r0 + i + n) * nr] = -Gpb[size_t(p - 1) * n + i] + Gpt[size_t(p) * n + i];
                for (int d = 0; d < ND; ++d) {
                    rhs[size_t(r0 + i) * nr + 1 + d] = dG(dGmt, p, d)[i] - dG(dGmb, p - 1, d)[i];
                    rhs[size_t(r0 + i + n) * nr + 1 + d] = dG(dGpt, p, d)[i] - dG(dGpb, p - 1, d)[i];
                }
            }
        }
        // ground
        {
            const int p = nlyr - 1, r0 = M - n, c0 = p * n2;
            for (int i = 0; i < n; ++i) {
                const double ng = neg(i);
                const bool rr = refl && refl_row(i);
                for (int j = 0; j < n; ++j) {
                    double vm = Wm[hidx(p, i, j)], vp = Wp[hidx(p, i, j)];
                    if (rr)
                        for (int s = 0; s < N; ++s) {
                            vm -= kd * rss[size_t(i / S) * N + s] * G.wt[s] * G.mu[s] * Wp[hidx(p, s * S, j)];
                            vp -= kd * rss[size_t(i / S) * N + s] * G.wt[s] * G.mu[s] * Wm[hidx(p, s * S, j)];
                        }
                    bvp.at(r0 + i, c0 + j) = ng * vm * E(p, j);
                    bvp.at(r0 + i, c0 + n + j) = ng * vp;
                }
                double direct = (thermal && m == 0) ? surf_emission : 0.0, um = Gmb[size_t(p) * n + i];   // ground_direct_sun, sktran_do_rte.h:222-245
                if (rr) {
                    direct += G.csz[cur_sza] * rsun[i / S] / kPi * trans[nlyr];
                    for (int s = 0; s < N; ++s) um -= kd * rss[size_t(i / S) * N + s] * G.wt[s] * G.mu[s] * Gpb[size_t(p) * n + s * S];
                }
                rhs[size_t(r0 + i) * nr] = direct - um;
                for (int d = 0; d < ND; ++d) {
                    double dalb = (d_layer[d] == p) ? d_alb(d_kind[d]) : 0.0;
                    double ddir = 0, dum = dG(dGmb, p, d)[i];
                    if (rr) {
                        ddir = G.csz[cur_sza] * rsun[i / S] / kPi * dtrans[size_t(nlyr) * ND + d] + G.csz[cur_sza] * dalb * drsun[i / S] / kPi * trans[nlyr];
                        for (int s = 0; s < N; ++s) {
                            dum -= kd * rss[size_t(i / S) * N + s] * G.wt[s] * G.mu[s] * dG(dGpb, p, d)[s * S];
                            dum -= kd * dalb * drss[size_t(i / S) * N + s] * G.wt[s] * G.mu[s] * Gpb[size_t(p) * n + s * S];
                        }
                    }
                    rhs[size_t(r0 + i) * nr + 1 + d] = ddir - dum;
                }
            }
        }
        int info = bvp.factor();
        if (info != 0) throw std::runtime_error("dgbsv couldn't solve BVP since the coefficient matrix was singular");
        // value column first (column 0), derivative columns need x
        std::vector<double> x(M);
        for (int r = 0; r < M; ++r) x[r] = rhs[size_t(r) * nr];
        bvp.solve(x.data(), 1);
        for (int p = 0; p < nlyr; ++p)
            for (int j = 0; j < n; ++j) {
                Lc[size_t(p) * n + j] = x[size_t(p) * n2 + j];
                Mc[size_t(p) * n + j] = x[size_t(p) * n2 + n + j];
            }
        if (ND == 0) return;
        // rhs_d -= dA_d x   (dA_d only touches the columns of the derivative's layer)
        for (int d = 0; d < ND; ++d) {
            const int p = d_layer[d], k = d_kind[d], c0 = p * n2;
            const double* xl = &x[c0];
            const double* xm = &x[c0 + n];
            auto dE = [&](int j) {
                return -(deigk[(size_t(p) * 3 + k) * n + j] * tau[p] + eigk[size_t(p) * n + j] * d_od(k)) * E(p, j);
            };
            // boundary above layer p
            if (p == 0) {
                for (int i = 0; i < n; ++i) {
                    double s = 0;
                    for (int j = 0; j < n; ++j)
                        s += dWp[dhidx(p, k, i, j)] * xl[j] + (dWm[dhidx(p, k, i, j)] * E(p, j) + Wm[hidx(p, i, j)] * dE(j)) * xm[j];
                    rhs[size_t(i) * nr + 1 + d] -= s;
                }
            } else {
                int r0 = n + (p - 1) * n2;
                for (int i = 0; i < n; ++i) {
                    double s1 = 0, s0 = 0;
                    for (int j = 0; j < n; ++j) {
                        s1 += -dWp[dhidx(p, k, i, j)] * xl[j] + (-dWm[dhidx(p, k, i, j)] * E(p, j) - Wm[hidx(p, i, j)] * dE(j)) * xm[j];
                        s0 += -dWm[dhidx(p, k, i, j)] * xl[j] - (dWp[dhidx(p, k, i, j)] * E(p, j) + Wp[hidx(p, i, j)] * dE(j)) * xm[j];
                    }
                    rhs[size_t(r0 + i + n) * nr + 1 + d] -= s1;
                    rhs[size_t(r0 + i) * nr + 1 + d] -= neg(i) * s0;
                }
            }
            // boundary below layer p
            if (p == nlyr - 1) {
                int r0 = M - n;
                for (int i = 0; i < n; ++i) {
                    const bool rr = refl && refl_row(i);
                    double s = 0;
                    for (int j = 0; j < n; ++j) {
                        double vm = Wm[hidx(p, i, j)], vp_d = dWp[dhidx(p, k, i, j)], vm_d = dWm[dhidx(p, k, i, j)];
                        if (rr)
                            for (int t = 0; t < N; ++t) {
                                double f = kd * G.wt[t] * G.mu[t];
                                const double r_it = rss[size_t(i / S) * N + t], dr_it = d_alb(k) * drss[size_t(i / S) * N + t];
                                vm -= f * r_it * Wp[hidx(p, t * S, j)];
                                vm_d -= f * r_it * dWp[dhidx(p, k, t * S, j)] + f * dr_it * Wp[hidx(p, t * S, j)];
                                vp_d -= f * r_it * dWm[dhidx(p, k, t * S, j)] + f * dr_it * Wm[hidx(p, t * S, j)];
                            }
                        s += (vm_d * E(p, j) + vm * dE(j)) * xl[j] + vp_d * xm[j];
                    }
                    rhs[size_t(r0 + i) * nr + 1 + d] -= neg(i) * s;
                }
            } else {
                int r0 = n + p * n2;
                for (int i = 0; i < n; ++i) {
                    double s1 = 0, s0 = 0;
                    for (int j = 0; j < n; ++j) {
                        s1 += (dWp[dhidx(p, k, i, j)] * E(p, j) + Wp[hidx(p, i, j)] * dE(j)) * xl[j] + dWm[dhidx(p, k, i, j)] * xm[j];
                        s0 += (dWm[dhidx(p, k, i, j)] * E(p, j) + Wm[hidx(p, i, j)] * dE(j)) * xl[j] + dWp[dhidx(p, k, i, j)] * xm[j];
                    }
                    rhs[size_t(r0 + i + n) * nr + 1 + d] -= s1;
                    rhs[size_t(r0 + i) * nr + 1 + d] -= neg(i) * s0;
                }
            }
        }
        bvp.solve(rhs.data(), nr);  // column 0 is re-solved (cheap) and ignored
        for (int p = 0; p < nlyr; ++p)
            for (int d = 0; d < ND; ++d)
                for (int j = 0; j < n; ++j) {
                    dLM(dLc, p, d)[j] = rhs[size_t(p * n2 + j) * nr + 1 + d];
                    dLM(dMc, p, d)[j] = rhs[size_t(p * n2 + n + j) * nr + 1 + d];
                }
    }

    int cur_sza = 0;

    void solve_order(int m, int szaidx) {
        cur_sza = szaidx;
        std::fill(dWp.begin(), dWp.end(), 0.0);
        std::fill(dWm.begin(), dWm.end(), 0.0);
        std::fill(deigk.begin(), deigk.end(), 0.0);
        for (int p = 0; p < nlyr; ++p) {
            solve_homogeneous(m, p);
            solve_particular(m, p, szaidx);
        }
        solve_bvp(m);
    }

    // Legendre sums at an outgoing direction with tabulated functions lp_out ([l][3])  (post-processing call-site naming):
    //   lm[s][r] ("lpsum_minus") = 0.5 w_j ssa [sum_l Pi_l(out) B_l Pi_l(mu_j)](s, s'),  r = j*S + s'
    //   lp[s][r] ("lpsum_plus")  = the negated product (scat_phase_f, sktran_do_lpproduct.h:265-337)
    void lpsums(int m, int p, const double* lp_out, double* lp, double* lm, double* dlp, double* dlm) {
        double b0[9], b1[9];
        for (int j = 0; j < N; ++j) {
            triple_sum(false, m, p, lp_out, fmu(m, j), b0);
            triple_sum(true, m, p, lp_out, fmu(m, j), b1);
            for (int s1 = 0; s1 < S; ++s1)
                for (int s2 = 0; s2 < S; ++s2) {
                    const int r = j * S + s2;
                    const double s0 = b0[s1 * S + s2], s1v = b1[s1 * S + s2];
                    for (int k = 0; k < nl(p); ++k) {
                        dlm[(size_t(k) * S + s1) * n + r] = s0 * d_ssa(k) * 0.5 * G.wt[j];
                        dlp[(size_t(k) * S + s1) * n + r] = s1v * d_ssa(k) * 0.5 * G.wt[j];
                    }
                    lm[size_t(s1) * n + r] = s0 * 0.5 * ssa[p] * G.wt[j];
                    lp[size_t(s1) * n + r] = s1v * 0.5 * ssa[p] * G.wt[j];
                }
        }
    }

    // Y+-[s][i] = sum_r lp[s][r] W+-(r, i) + lm[s][r] W-+(r, i) and their layer derivatives (sktran_do_opticallayer.cpp:196-262)
    void ymatrices(int p, const double* lp, const double* lm, const double* dlp, const double* dlm, double* Yp, double* Ym, double* dYp, double* dYm) {
        const int nk = nl(p);
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < n; ++i) {
                double yp = 0, ym = 0;
                const double* lps = lp + size_t(s) * n;
                const double* lms = lm + size_t(s) * n;
                for (int r = 0; r < n; ++r) {
                    yp += lps[r] * Wp[hidx(p, r, i)] + lms[r] * Wm[hidx(p, r, i)];
                    ym += lps[r] * Wm[hidx(p, r, i)] + lms[r] * Wp[hidx(p, r, i)];
                }
                Yp[size_t(s) * n + i] = yp;
                Ym[size_t(s) * n + i] = ym;
                for (int k = 0; k < nk; ++k) {
                    const double* dlps = dlp + (size_t(k) * S + s) * n;
                    const double* dlms = dlm + (size_t(k) * S + s) * n;
                    double a1 = 0, a2 = 0;
                    for (int r = 0; r < n; ++r) {
                        a1 += dlps[r] * Wp[hidx(p, r, i)] + lps[r] * dWp[dhidx(p, k, r, i)] + dlms[r] * Wm[hidx(p, r, i)] + lms[r] * dWm[dhidx(p, k, r, i)];
                        a2 += dlps[r] * Wm[hidx(p, r, i)] + lps[r] * dWm[dhidx(p, k, r, i)] + dlms[r] * Wp[hidx(p, r, i)] + lms[r] * dWp[dhidx(p, k, r, i)];
                    }
                    dYp[(size_t(k) * S + s) * n + i] = a1;
                    dYm[(size_t(k) * S + s) * n + i] = a2;
                }
            }
    }

    void map_to_native(const double* do_deriv, double scale, double* native, int stride = 1) const {
        for (int d = 0; d < ND; ++d) {
            double v = do_deriv[size_t(d) * stride] * scale;
            if (v == 0.0) continue;
            for (auto& wr : wires[d]) native[wr.native] += wr.w * v;
        }
    }

    // ------------------------------------------------------------------ plane-parallel post-processing
    void pp_postprocess(int m, int szaidx, Result& res, std::vector<double>& rad, std::vector<double>& drad);
    // ------------------------------------------------------------------ spherical table
    void accumulate_ground(int m, int szaidx);
    void accumulate_sources(int m, int szaidx);
    void integrate_los(const AtmosphereView& atm, int w, Result& res);

    void compute_flux(Result& res);

  public:
    void solve(const AtmosphereView& atm, int w, Result& res);
    // plane-parallel post-processing: > 0 solves exactly this many azimuth orders (Config::num_do_forced_azimuth, do_source_planeparallel.cpp:55-60)
    int forced_azimuth = 0;
    // flux observers (altitudes in m above the surface), plane-parallel post-processing only
    std::vector<double> flux_altitudes;
    // the BRDF argument the native surface derivative [2 ngrid] is taken for (the caller runs one pass per argument)
    int surface_deriv_arg = 0;
};

// ======================================================================================
// Fluxes through horizontal surfaces at the flux observers from the order-0 solution
// (DOSourcePlaneParallelPostProcessing::compute_flux, cpp/lib/sktran_disco/source_term/do_source_planeparallel.cpp:221-566; the sampled
// multipliers h+- and D+- of cpp/include/sktran_disco/sktran_do_postprocessing.h:19-330, values only).
inline void Solver::compute_flux(Result& res) {
    if (S != 1) throw std::invalid_argument("flux observers: NSTOKES = 1 only");
    res.flux.assign(flux_altitudes.size() * 3, 0.0);
    for (size_t o = 0; o < flux_altitudes.size(); ++o) {
        const double alt = flux_altitudes[o];
        // OpticalLayerArray::layerAtAltitude: the first layer from the top whose floor is not above the observer
        int p = nlyr - 1;
        for (int r = 0; r < nlyr; ++r)
            if (G.floor_h[r] <= alt) { p = r; break; }
        const double lf = (G.ceil_h[p] - alt) / (G.ceil_h[p] - G.floor_h[p]);
        const double th = tau[p], x = lf * th, T = trans[p], sc = sec[p];
        double up = 0, dn = 0, up_a = 0, dn_a = 0;
        for (int j = 0; j < n; ++j) {
            const double ev = eigk[size_t(p) * n + j];
            const double Hp = std::exp(-ev * th * lf), Hm = std::exp(-ev * th * (1 - lf));
            const double Dm = T * (std::exp(-x * ev) - std::exp(-x * sc)) / (sc - ev);
            const double Dp = T * (std::exp(-x * sc) - std::exp(-th * sc) * std::exp(-(th - x) * ev)) / (ev + sc);
            double plus = Lc[size_t(p) * n + j] * Hp + Ap[size_t(p) * n + j] * Dm;
            double minus = Mc[size_t(p) * n + j] * Hm + Am[size_t(p) * n + j] * Dp;
            if (thermal) {
                const double b0 = th_b0[p], b1 = th_b1[p];
                const double Dm_t = b0 * (std::exp(-x * ev) - std::exp(-x * b1)) / (b1 - ev);
                const double Dp_t = b0 * (std::exp(-x * b1) - std::exp(-th * b1) * std::exp(-(th - x) * ev)) / (ev + b1);
                plus += Ap_th[size_t(p) * n + j] * Dm_t;
                minus += Am_th[size_t(p) * n + j] * Dp_t;
            }
            for (int l = 0; l < N; ++l) {
                const double q = G.wt[l] * 2.0 * kPi, hp = Wp[hidx(p, l, j)], hm = Wm[hidx(p, l, j)];
                const double d = minus * hm + plus * hp, u = minus * hp + plus * hm;
                dn += q * G.mu[l] * d;
                up += q * G.mu[l] * u;
                dn_a += q * d;
                up_a += q * u;
            }
        }
        if (include_ss) {   // the direct beam at the layer's ceiling, as the reference adds it
            dn += T * G.csz[cur_sza];
            dn_a += T;
        }
        res.flux[o * 3 + 0] = up;
        res.flux[o * 3 + 1] = dn;
        res.flux[o * 3 + 2] = up_a + dn_a;
    }
}

inline void Solver::pp_postprocess(int m, int szaidx, Result& res, std::vector<double>& rad, std::vector<double>& drad) {
    (void)szaidx;
    const int p_bot = nlyr - 1;
    // per-Stokes derivative blocks are stored [ND][S]
    std::vector<double> comp(S), comp_d(size_t(ND) * S), integ_d(size_t(ND) * S), strm_d(ND), lp(size_t(S) * n), lm(size_t(S) * n), dlp(size_t(3) * S * n), dlm(size_t(3) * S * n);
    std::vector<double> Yp(size_t(S) * n), Ym(size_t(S) * n), dYp(size_t(3) * S * n), dYm(size_t(3) * S * n), dEf(ND), dDp(ND), dDm(ND);
    for (size_t j = 0; j < G.pp_los.size(); ++j) {
        const auto& los = G.pp_los[j];
        const double mu = los.coszenith;
        double obs_od = 0.0;
        if (los.observer_altitude >= 0) {
            // opticalDepthAt(altitude)
            int pl = nlyr - 1;
            for (int p = 0; p < nlyr; ++p)
                if (los.observer_altitude <= G.ceil_h[p] && los.observer_altitude >= G.floor_h[p]) { pl = p; break; }
            if (los.observer_altitude > G.ceil_h[0]) pl = 0;
            if (los.observer_altitude > G.ceil_h[pl]) obs_od = 0.0;
            else {
                double f = 1 - (G.ceil_h[pl] - los.observer_altitude) / (G.ceil_h[pl] - G.floor_h[pl]);
                obs_od = od_floor[pl] - tau[pl] * f;
            }
        }
        // ground reflection (computeReflectedIntensities): Lambertian, intensity component only
        std::fill(comp.begin(), comp.end(), 0.0);
        std::fill(comp_d.begin(), comp_d.end(), 0.0);
        if (m < surf_max_order) {
            const int p = p_bot, ls = ND ? lstart[p] : 0, nk = nl(p);
            double diffuse = 0;
            std::vector<double> diffuse_d(ND, 0.0);
            for (int is = 0; is < N; ++is) {
                const int i = is * S;
                const double r = rlos[j * N + is], dr = drlos[j * N + is];   // los_stream expansion of this line of sight
                double sc_v = Gpb[size_t(p) * n + i];
                for (int d = 0; d < ND; ++d) strm_d[d] = dG(dGpb, p, d)[i];
                double factor = (m == 0 ? 2.0 : 1.0) * G.mu[is] * G.wt[is];
                for (int jj = 0; jj < n; ++jj) {
                    double st = std::exp(-std::abs(eigk[size_t(p) * n + jj]) * tau[p]);
                    double hp = Wp[hidx(p, i, jj)], hm = Wm[hidx(p, i, jj)];
                    sc_v += Lc[size_t(p) * n + jj] * hp * st + Mc[size_t(p) * n + jj] * hm;
                    for (int d = 0; d < ND; ++d)
                        strm_d[d] += dLM(dLc, p, d)[jj] * hp * st + dLM(dMc, p, d)[jj] * hm;
                    for (int k = 0; k < nk; ++k) {
                        double dst = -(deigk[(size_t(p) * 3 + k) * n + jj] * tau[p] + eigk[size_t(p) * n + jj] * d_od(k)) * st;
                        strm_d[ls + k] += Lc[size_t(p) * n + jj] * dWp[dhidx(p, k, i, jj)] * st + Lc[size_t(p) * n + jj] * hp * dst;
                        strm_d[ls + k] += Mc[size_t(p) * n + jj] * dWm[dhidx(p, k, i, jj)];
                    }
                }
                diffuse += factor * sc_v * r;
                for (int d = 0; d < ND; ++d) diffuse_d[d] += factor * strm_d[d] * r;
                for (int k = 0; k < nk; ++k) diffuse_d[ls + k] += factor * sc_v * d_alb(k) * dr;
            }
            if (include_ss) {
                double direct = G.csz[cur_sza] / kPi * trans[nlyr] * rlossun[j];
                comp[0] = direct + diffuse;
                if (thermal && m == 0) comp[0] += surf_emission;   // sktran_do_layerarray.cpp:255-266
                for (int d = 0; d < ND; ++d) comp_d[size_t(d) * S] = G.csz[cur_sza] / kPi * dtrans[size_t(nlyr) * ND + d] * rlossun[j] + diffuse_d[d];
                for (int k = 0; k < nk; ++k) comp_d[size_t(ls + k) * S] += G.csz[cur_sza] / kPi * trans[nlyr] * d_alb(k) * drlossun[j];
            } else {
                comp[0] = diffuse;
                for (int d = 0; d < ND; ++d) comp_d[size_t(d) * S] = diffuse_d[d];
            }
        }
        // upward through the layers
        for (int p = nlyr - 1; p >= 0; --p) {
            const double entry = od_floor[p], exitod = od_ceil[p];
            if (entry < obs_od) continue;
            double layerfraction = 1;
            if (exitod < obs_od) layerfraction = (entry - obs_od) / (entry - exitod);
            const int nk = nl(p), ls = ND ? lstart[p] : 0;
            double att = std::exp(-1.0 * tau[p] * layerfraction / mu);
            for (int s = 0; s < S; ++s) comp[s] *= att;
            for (size_t d = 0; d < comp_d.size(); ++d) comp_d[d] *= att;
            for (int k = 0; k < nk; ++k)
                for (int s = 0; s < S; ++s) comp_d[size_t(ls + k) * S + s] += -layerfraction / mu * comp[s] * d_od(k);

            // integrate_source(m, mu, obsod, lp_mu, ...)
            double exit_od = od_ceil[p];
            double upper = (obs_od < exit_od) ? exit_od : obs_od;
            double x = upper - exit_od;
            const double T = trans[p], sc = sec[p], th = tau[p];
            const double* dT = &dtrans[size_t(p) * ND];
            const double* dsc = &dsec[size_t(p) * ND];
            lpsums(m, p, flos(int(j), m), lp.data(), lm.data(), dlp.data(), dlm.data());
            double Q[3] = {0, 0, 0}, dQ[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // dQ[k][s]
            if (include_ss) {
                double b[9];
                triple_sum(true, m, p, flos(int(j), m), fcsz(cur_sza, m), b);
                const double factor = (2 - (m == 0 ? 1 : 0)) * (1.0 / (4.0 * kPi));
                for (int s = 0; s < S; ++s) {
                    double sneg = b[s * S] * factor;
                    for (int k = 0; k < nk; ++k) dQ[k][s] = sneg * d_ssa(k);
                    Q[s] = sneg * ssa[p];
                }
            }
            ymatrices(p, lp.data(), lm.data(), dlp.data(), dlm.data(), Yp.data(), Ym.data(), dYp.data(), dYm.data());
            // E(mu, x, thickness, transmission)
            const double lf = 1 - x / th;
            double Ef;
            {
                double e1 = std::exp(-x * sc), e2 = std::exp(-1.0 * th * sc) * std::exp(-1.0 * (th - x) / mu);
                double den = 1.0 + mu * sc;
                Ef = T / den * (e1 - e2);
                for (int d = 0; d < ND; ++d)
                    dEf[d] = dT[d] / den * (e1 - e2) + dsc[d] * ((T / den * e2 * th) - (Ef / den * mu) + (T / den * e1 * (-1.0 * x)));
                for (int k = 0; k < nk; ++k)
                    dEf[ls + k] += T / den * (e2 * d_od(k) * (sc + lf / mu) - e1 * d_od(k) * (1 - lf) * sc);
            }
            double Ef_th = 0;   // E_thermal (sktran_do_opticallayer.cpp:941-975): E with b0 for the transmission and b1 for the secant
            if (thermal && m == 0) {
                const double b0 = th_b0[p], b1 = th_b1[p];
                Ef_th = b0 / (1.0 + mu * b1) * (std::exp(-x * b1) - std::exp(-1.0 * th * b1) * std::exp(-1.0 * (th - x) / mu));
            }
            double J[3] = {0, 0, 0}, V[3] = {0, 0, 0};
            std::fill(integ_d.begin(), integ_d.end(), 0.0);
            for (int i = 0; i < n; ++i) {
                const double ev = eigk[size_t(p) * n + i];
                double hp, hm, dhp[3] = {0, 0, 0}, dhm[3] = {0, 0, 0};
                {
                    double den = 1.0 + mu * ev;
                    if (std::abs(den) > 0.0001) {
                        double e1 = std::exp(-x * ev), e2 = std::exp(-1.0 * th * ev) * std::exp(-(th - x) / mu);
                        hp = (e1 - e2) / den;
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * n + i];
                            dhp[k] = (-1.0 * e1 * (dk * x + d_od(k) * (1 - lf) * ev) + e2 * ((ev + lf / mu) * d_od(k) + th * dk)) / den;
                            dhp[k] -= 1.0 / den * hp * mu * dk;
                        }
                    } else {
                        double e1 = std::exp(-x * ev);
                        hp = (th - x) / mu * e1 * (1 - (th - x) * (ev + 1 / mu));
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * n + i];
                            dhp[k] = -1.0 * x * dk * hp + 1 / mu * e1 * d_od(k) * (1 - (th - x) * (ev + 1 / mu)) + 1 / mu * e1 * (th - x) * -1.0 * (ev + 1 / mu) * d_od(k);
                        }
                    }
                    double den2 = 1.0 - mu * ev;
                    if (std::abs(den2) > 0.0001) {
                        double e1 = std::exp(-1.0 * ev * (th - x)), e2 = std::exp(-1 / mu * (th - x));
                        hm = (e1 - e2) / den2;
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * n + i];
                            dhm[k] = (-1.0 * e1 * (dk * (th - x) + d_od(k) * lf * ev) + e2 * d_od(k) * lf / mu) / den2;
                            dhm[k] += 1.0 / den2 * hm * mu * dk;
                        }
                    } else {
                        double e1 = std::exp(-1.0 * ev * (th - x));
                        hm = e1 * (th - x) / mu * (1 - (th - x) * (ev - 1 / mu));
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * n + i];
                            dhm[k] = -1.0 * (ev * d_od(k) + th * dk) * hm + e1 * d_od(k) / mu * (1 - (th - x) * (ev - 1 / mu)) + e1 * (th - x) / mu * -1.0 * d_od(k) * (ev - 1 / mu);
                        }
                    }
                }
                const double Lv = Lc[size_t(p) * n + i], Mv = Mc[size_t(p) * n + i];
                double expfactor = std::exp(-1 * th * sc);
                double Dp = (-1 * T * expfactor * hm + Ef) / (sc + ev);
                double Dm = (T * hp - Ef) / (sc - ev);
                for (int d = 0; d < ND; ++d) {
                    dDp[d] = dsc[d] * (T * hm * th * expfactor - Dp) / (sc + ev) + dEf[d] / (sc + ev) + dT[d] * (-hm * expfactor) / (sc + ev);
                    dDm[d] = dsc[d] * (-Dm) / (sc - ev) + dT[d] * hp / (sc - ev) + dEf[d] * -1.0 / (sc - ev);
                }
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dDp[ls + k] += dk * (-Dp) / (sc + ev) + dhm[k] * (-T * expfactor) / (sc + ev) + d_od(k) * (T * sc * hm * expfactor) / (sc + ev);
                    dDm[ls + k] += dhp[k] * T / (sc - ev) + dk * Dm / (sc - ev);
                }
                const double Apv = Ap[size_t(p) * n + i], Amv = Am[size_t(p) * n + i];
                double Dm_th = 0, Dp_th = 0;
                if (thermal && m == 0) {   // sktran_do_opticallayer.cpp:421-431
                    const double b0 = th_b0[p], b1 = th_b1[p];
                    Dp_th = (-1 * b0 * std::exp(-th * b1) * hm + Ef_th) / (b1 + ev);
                    Dm_th = (b0 * hp - Ef_th) / (b1 - ev);
                }
                for (int s = 0; s < S; ++s) {
                    const double yp = Yp[size_t(s) * n + i], ym = Ym[size_t(s) * n + i];
                    J[s] += yp * hp * Lv + ym * hm * Mv;
                    V[s] += Apv * yp * Dm + Amv * ym * Dp;
                    if (thermal && m == 0) V[s] += Ap_th[size_t(p) * n + i] * yp * Dm_th + Am_th[size_t(p) * n + i] * ym * Dp_th;
                    for (int k = 0; k < nk; ++k) {
                        const double dyp = dYp[(size_t(k) * S + s) * n + i], dym = dYm[(size_t(k) * S + s) * n + i];
                        double& t = integ_d[size_t(ls + k) * S + s];
                        t += dyp * hp * Lv + dym * hm * Mv;
                        t += yp * dhp[k] * Lv + ym * dhm[k] * Mv;
                        t += Apv * dyp * Dm + Amv * dym * Dp;
                        t += dAp[(size_t(p) * 3 + k) * n + i] * yp * Dm + dAm[(size_t(p) * 3 + k) * n + i] * ym * Dp;
                    }
                    for (int d = 0; d < ND; ++d)
                        integ_d[size_t(d) * S + s] += yp * hp * dLM(dLc, p, d)[i] + ym * hm * dLM(dMc, p, d)[i] + Apv * yp * dDm[d] + Amv * ym * dDp[d];
                }
            }
            for (int s = 0; s < S; ++s) {
                double integral = J[s] + V[s] + Q[s] * Ef;
                if (thermal && m == 0 && s == 0) integral += Ef_th * (1 - ssa[p]);   // sktran_do_opticallayer.cpp:513-521
                for (int k = 0; k < nk; ++k) integ_d[size_t(ls + k) * S + s] += dQ[k][s] * Ef;
                for (int d = 0; d < ND; ++d) integ_d[size_t(d) * S + s] += Q[s] * dEf[d];
                comp[s] += integral;
                for (int d = 0; d < ND; ++d) comp_d[size_t(d) * S + s] += integ_d[size_t(d) * S + s];
            }
        }
        // azimuth expansion: cos for I and Q, sin for U (do_source_planeparallel.cpp:154-190)
        for (int s = 0; s < S; ++s) {
            double azf = s < 2 ? std::cos(m * los.azimuth) : std::sin(m * los.azimuth);
            rad[j * S + s] += comp[s] * azf;
            if (ND) map_to_native(comp_d.data() + s, azf, &drad[(j * S + s) * nnative], S);
        }
    }
    (void)res;
}

// ======================================================================================
inline void Solver::accumulate_ground(int m, int szaidx) {
    if (m >= surf_max_order) return;   // (m >= max_azimuthal_order: 1 for the Lambertian surface, do_source_diffuse_storage.cpp:445-447)
    const int p = nlyr - 1, ls = ND ? lstart[p] : 0, nk = nl(p);
    const size_t nslot = npoint + nground;
    std::vector<double> strm_d(ND), diffuse_d(ND);
    for (int aidx = 0; aidx < G.nangle; ++aidx) {
        int slot = ground_slot[size_t(szaidx) * G.nangle + aidx];
        if (slot < 0) continue;
        double diffuse = 0;
        std::fill(diffuse_d.begin(), diffuse_d.end(), 0.0);
        for (int is = 0; is < N; ++is) {
            const int i = is * S;  // intensity rows only ("TODO ... might be fine for scalar surface reflection", :478-486)
            // compute_expansion(m, mu_in = stream, mu_out = grid angle) (:474-477); the Lambertian special case is the albedo
            const size_t gk = size_t(std::find(gnd_aidx.begin(), gnd_aidx.end(), aidx) - gnd_aidx.begin());
            const double r = rgnd[gk * N + is], dr = drgnd[gk * N + is];
            double sc_v = Gpb[size_t(p) * n + i];
            for (int d = 0; d < ND; ++d) strm_d[d] = dG(dGpb, p, d)[i];
            for (int jj = 0; jj < n; ++jj) {
                double st = std::exp(-std::abs(eigk[size_t(p) * n + jj]) * tau[p]);
                double hp = Wp[hidx(p, i, jj)], hm = Wm[hidx(p, i, jj)];
                sc_v += Lc[size_t(p) * n + jj] * hp * st + Mc[size_t(p) * n + jj] * hm;
                for (int d = 0; d < ND; ++d) strm_d[d] += dLM(dLc, p, d)[jj] * hp * st + dLM(dMc, p, d)[jj] * hm;
                for (int k = 0; k < nk; ++k) {
                    double dst = -(deigk[(size_t(p) * 3 + k) * n + jj] * tau[p] + eigk[size_t(p) * n + jj] * d_od(k)) * st;
                    strm_d[ls + k] += Lc[size_t(p) * n + jj] * hp * dst + Lc[size_t(p) * n + jj] * dWp[dhidx(p, k, i, jj)] * st;
                    strm_d[ls + k] += Mc[size_t(p) * n + jj] * dWm[dhidx(p, k, i, jj)];
                }
            }
            double factor = (m == 0 ? 2.0 : 1.0) * G.mu[is] * G.wt[is];
            diffuse += factor * sc_v * r;
            for (int d = 0; d < ND; ++d) diffuse_d[d] += factor * strm_d[d] * r;
            for (int k = 0; k < nk; ++k) diffuse_d[ls + k] += factor * sc_v * d_alb(k) * dr;
        }
        double* tv = &tab_val[(size_t(m) * nslot + slot) * S];
        for (int s = 0; s < S; ++s) tv[s] = 0;
        tv[0] = diffuse;
        filled[size_t(m) * nslot + slot] = 1;
        if (ND) {
            double* nat = &tab_deriv[(size_t(m) * nslot + slot) * S * nnative];
            std::fill(nat, nat + size_t(S) * nnative, 0.0);
            map_to_native(diffuse_d.data(), 1.0, nat);
        }
    }
}

inline void Solver::accumulate_sources(int m, int szaidx) {
    const size_t nslot = npoint + nground;
    if (m >= 2) all_converged = true;
    accumulate_ground(m, szaidx);
    std::vector<double> lp(size_t(S) * n), lm(size_t(S) * n), dlp(size_t(3) * S * n), dlm(size_t(3) * S * n);
    std::vector<double> Yp(size_t(S) * n), Ym(size_t(S) * n), dYp(size_t(3) * S * n), dYm(size_t(3) * S * n), temp(ND);
    std::vector<double> hpv(n), hmv(n), dhp(3 * n), dhm(3 * n), Dmv(n), Dpv(n), dDm(size_t(n) * ND), dDp(size_t(n) * ND);
    for (int lidx = 0; lidx < nlyr; ++lidx) {
        const int p = nlyr - lidx - 1;
        const int nk = nl(p), ls = ND ? lstart[p] : 0;
        const double altitude = (G.ceil_h[p] + G.floor_h[p]) / 2.0;
        const double lf = (G.ceil_h[p] - altitude) / (G.ceil_h[p] - G.floor_h[p]);
        const double th = tau[p], T = trans[p], sc = sec[p];
        const double* dT = &dtrans[size_t(p) * ND];
        const double* dsc = &dsec[size_t(p) * ND];
        const double x = th * lf;
        for (int i = 0; i < n; ++i) {
            const double ev = eigk[size_t(p) * n + i];
            hpv[i] = std::exp(-1.0 * ev * th * lf);
            hmv[i] = std::exp(-ev * th * (1 - lf));
            for (int k = 0; k < nk; ++k) {
                double dk = deigk[(size_t(p) * 3 + k) * n + i];
                dhp[size_t(k) * n + i] = -hpv[i] * (dk * th * lf + d_od(k) * ev * lf);
                dhm[size_t(k) * n + i] = -hmv[i] * (dk * th * (1 - lf) + d_od(k) * ev * (1 - lf));
            }
            // d_minus_sampled
            {
                double denom = sc - ev;
                double v = std::exp(-x * ev) - std::exp(-x * sc);
                double* dd = &dDm[size_t(i) * ND];
                for (int d = 0; d < ND; ++d) dd[d] = dT[d] * v + dsc[d] * T * (-v / denom + std::exp(-x * sc) * x);
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dd[ls + k] += dk * T * (-x * std::exp(-x * ev) + v / denom);
                    dd[ls + k] += lf * d_od(k) * T * (std::exp(-x * sc) * sc - std::exp(-x * ev) * ev);
                }
                Dmv[i] = v * T / denom;
                for (int d = 0; d < ND; ++d) dd[d] *= 1 / denom;
            }
            // d_plus_sampled
            {
                double denom = ev + sc;
                double ef2 = std::exp(-th * sc) * std::exp(-(th - x) * ev);
                double v = std::exp(-x * sc) - ef2;
                double* dd = &dDp[size_t(i) * ND];
                for (int d = 0; d < ND; ++d) dd[d] = dT[d] * v + dsc[d] * T * (-v / denom + -x * std::exp(-x * sc) + th * ef2);
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * n + i];
                    dd[ls + k] += dk * T * (-v / denom + ef2 * (th - x));
                    dd[ls + k] += d_od(k) * T * (-std::exp(-x * sc) * lf * sc + ef2 * sc + ef2 * (ev - lf * ev));
                }
                Dpv[i] = v * T / denom;
                for (int d = 0; d < ND; ++d) dd[d] *= 1 / denom;
            }
        }
        for (int aidx = 0; aidx < G.nangle; ++aidx) {
            int slot = point_slot[(size_t(szaidx) * nlyr + lidx) * G.nangle + aidx];
            if (slot < 0 || converged[size_t(m) * npoint + slot]) continue;
            lpsums(m, p, fang(m, aidx), lp.data(), lm.data(), dlp.data(), dlm.data());
            ymatrices(p, lp.data(), lm.data(), dlp.data(), dlm.data(), Yp.data(), Ym.data(), dYp.data(), dYm.data());
            for (int s = 0; s < S; ++s) {
                std::fill(temp.begin(), temp.end(), 0.0);
                double val = 0;
                for (int i = 0; i < n; ++i) {
                    const double yp = Yp[size_t(s) * n + i], ym = Ym[size_t(s) * n + i];
                    const double Lv = Lc[size_t(p) * n + i], Mv = Mc[size_t(p) * n + i];
                    const double Apv = Ap[size_t(p) * n + i], Amv = Am[size_t(p) * n + i];
                    val += yp * hpv[i] * Lv + ym * hmv[i] * Mv;
                    val += Apv * yp * Dmv[i] + Amv * ym * Dpv[i];
                    const double c1 = yp * hpv[i], c2 = ym * hmv[i], c3 = Apv * yp, c4 = Amv * ym;
                    for (int d = 0; d < ND; ++d)
                        temp[d] += dLM(dLc, p, d)[i] * c1 + dLM(dMc, p, d)[i] * c2 + dDm[size_t(i) * ND + d] * c3 + dDp[size_t(i) * ND + d] * c4;
                    for (int k = 0; k < nk; ++k) {
                        const double dyp = dYp[(size_t(k) * S + s) * n + i], dym = dYm[(size_t(k) * S + s) * n + i];
                        temp[ls + k] += dyp * hpv[i] * Lv + dym * hmv[i] * Mv;
                        temp[ls + k] += dhp[size_t(k) * n + i] * yp * Lv + dhm[size_t(k) * n + i] * ym * Mv;
                        temp[ls + k] += dyp * Dmv[i] * Apv + dym * Dpv[i] * Amv;
                        temp[ls + k] += dAp[(size_t(p) * 3 + k) * n + i] * yp * Dmv[i] + dAm[(size_t(p) * 3 + k) * n + i] * ym * Dpv[i];
                    }
                }
                val /= ssa[p];
                tab_val[(size_t(m) * nslot + slot) * S + s] = val;
                filled[size_t(m) * nslot + slot] = 1;
                // convergence is judged on the intensity only (do_source_diffuse_storage.cpp:1033-1060)
                if (m >= 2 && s == 0) {
                    double prev = tab_val[(size_t(m - 1) * nslot + slot) * S], prev2 = tab_val[(size_t(m - 2) * nslot + slot) * S];
                    if ((std::abs(val / prev) < 1e-4 || prev < 1e-10) && (std::abs(val / prev2) < 1e-4 || prev2 < 1e-10)) {
                        for (int azi = m + 1; azi < nstr; ++azi) converged[size_t(azi) * npoint + slot] = 1;
                    } else {
                        all_converged = false;
                    }
                }
                if (ND) {
                    for (int k = 0; k < nk; ++k) temp[ls + k] -= d_ssa(k) * val;
                    double* nat = &tab_deriv[((size_t(m) * nslot + slot) * S + s) * nnative];
                    std::fill(nat, nat + nnative, 0.0);
                    map_to_native(temp.data(), 1.0 / ssa[p], nat);
                }
            }
        }
    }
}

inline void Solver::integrate_los(const AtmosphereView& atm, int w, Result& res) {
    const size_t nslot = npoint + nground;
    std::vector<double> I(S), sv(S);
    for (int r = 0; r < G.nlos; ++r) {
        const auto& ray = G.rays[r];
        std::fill(I.begin(), I.end(), 0.0);
        double* dI = nnative ? &res.deriv[size_t(r) * S * nnative] : nullptr;  // [S][nnative]
        double od_total = 0;
        if (ray.ground_is_hit) {
            // Lambertian ground: intensity only (end_of_ray_source, do_source_interpolated_pp.cpp:176-211)
            for (int m = 0; m < nstr; ++m) {
                double azf = std::cos(m * (kPi - ray.ground_saa));
                for (auto& e : ray.ground) {
                    int slot = ground_slot[size_t(e.sza) * G.nangle + e.angle];
                    if (!filled[size_t(m) * nslot + slot]) continue;
                    I[0] += e.w * azf * tab_val[(size_t(m) * nslot + slot) * S];
                    if (nnative) {
                        const double* nat = &tab_deriv[(size_t(m) * nslot + slot) * S * nnative];
                        for (int q = 0; q < nnative; ++q) dI[q] += e.w * azf * nat[q];
                    }
                }
            }
        }
        for (auto& L : ray.layers) {
            double od = 0;
            for (size_t t = 0; t < L.od.idx.size(); ++t) od += L.od.w[t] * atm.k(L.od.idx[t], w);
            od_total += od;
            double att = std::exp(-od);
            if (nnative) {
                for (int s = 0; s < S; ++s)
                    for (size_t t = 0; t < L.od.idx.size(); ++t) dI[size_t(s) * nnative + L.od.idx[t]] -= L.od.w[t] * I[s];
                for (size_t q = 0; q < size_t(S) * nnative; ++q) dI[q] *= att;
            }
            for (int s = 0; s < S; ++s) I[s] *= att;
            if (L.layer_distance < kMinShell) continue;
            double omega = 0;
            for (size_t t = 0; t < L.ssa_interp.idx.size(); ++t) omega += atm.om(L.ssa_interp.idx[t], w) * L.ssa_interp.w[t];
            double sf = 1 - att;
            // azimuth weights: cos for I and Q, sin for U (do_source_diffuse_storage.cpp:158-184)
            std::fill(sv.begin(), sv.end(), 0.0);
            for (int m = 0; m < nstr; ++m)
                for (int s = 0; s < S; ++s) {
                    double azf = s < 2 ? std::cos(m * L.azi) : std::sin(m * L.azi);
                    for (auto& e : L.table) {
                        int slot = point_slot[(size_t(e.sza) * nlyr + e.alt) * G.nangle + e.angle];
                        sv[s] += e.w * azf * tab_val[(size_t(m) * nslot + slot) * S + s];
                    }
                }
            for (int s = 0; s < S; ++s) I[s] += omega * sf * sv[s];
            if (nnative) {
                for (int s = 0; s < S; ++s) {
                    double* dIs = dI + size_t(s) * nnative;
                    for (size_t t = 0; t < L.od.idx.size(); ++t) dIs[L.od.idx[t]] += L.od.w[t] * (1 - sf) * sv[s] * omega;
                    for (size_t t = 0; t < L.ssa_interp.idx.size(); ++t) dIs[ngrid + L.ssa_interp.idx[t]] += L.ssa_interp.w[t] * sf * sv[s];
                    for (int m = 0; m < nstr; ++m) {
                        double azf = s < 2 ? std::cos(m * L.azi) : std::sin(m * L.azi);
                        for (auto& e : L.table) {
                            int slot = point_slot[(size_t(e.sza) * nlyr + e.alt) * G.nangle + e.angle];
                            if (!filled[size_t(m) * nslot + slot]) continue;
                            const double* nat = &tab_deriv[((size_t(m) * nslot + slot) * S + s) * nnative];
                            double f = omega * sf * e.w * azf;
                            for (int q = 0; q < nnative; ++q) dIs[q] += f * nat[q];
                        }
                    }
                }
            }
        }
        for (int s = 0; s < S; ++s) res.radiance[size_t(r) * S + s] = I[s];
        res.los_od[r] = od_total;
    }
}

inline void Solver::solve(const AtmosphereView& atm, int w, Result& res) {
    res.radiance.assign(size_t(G.nlos) * S, 0.0);
    res.deriv.assign(size_t(G.nlos) * S * nnative, 0.0);
    res.los_od.assign(G.nlos, 0.0);
    res.num_azimuth_used = 0;
    if (G.spherical_pp) {
        std::fill(tab_val.begin(), tab_val.end(), 0.0);
        std::fill(filled.begin(), filled.end(), 0);
        std::fill(converged.begin(), converged.end(), 0);
        for (int s = 0; s < G.nsza; ++s) {
            configure_layers(atm, w, s);
            cur_sza = s;   // (the surface samples are taken for this solar zenith index: the sun's direction enters the stream-sun pairs)
            sample_surface();
            all_converged = false;
            for (int m = 0; m < nstr; ++m) {
                expand_surface(m);
                solve_order(m, s);
                accumulate_sources(m, s);
                res.num_azimuth_used = std::max(res.num_azimuth_used, m + 1);
                if (all_converged) break;
            }
        }
        integrate_los(atm, w, res);
    } else {
        configure_layers(atm, w, 0);
        cur_sza = 0;
        sample_surface();
        const int num_azi = forced_azimuth > 0 ? forced_azimuth : nstr;
        for (int m = 0; m < num_azi; ++m) {
            expand_surface(m);
            solve_order(m, 0);
            if (m == 0 && !flux_altitudes.empty()) compute_flux(res);
            pp_postprocess(m, 0, res, res.radiance, res.deriv);
            res.num_azimuth_used = m + 1;
        }
    }
}

}  // namespace oracle
