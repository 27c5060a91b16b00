// TEST INFRASTRUCTURE ONLY -- CPU oracle for the DISCO (discrete ordinates) hot path.
//
// A plain C++ restatement of the reference's per-wavelength scalar (NSTOKES=1) DISCO solver
// with its dense forward-mode linearization, used ONLY by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg.  The product (HIP) path never includes this file.
//
// Parity status: PINNED.  Checked against the reference's own known-answer tests
//   cpp/lib/tests/sktran_disco/legacy/test_scalar.cpp:84-880      (plane parallel, 1e-8)
//   tests/engine/test_1d_solver_regression.py:112-237             (spherical limb+nadir, 5e-7)
// see tests/test_oracle_golden.py.
//
// Reference files each stage follows (relative to /root/reference):
//   layer optics + derivative wiring   cpp/lib/sktran_disco/sktran_do_layerarray.cpp:332-477,487-657,729-757
//   SSA dither                         cpp/lib/sktran_disco/sktran_do_opticallayer.cpp:37-40
//   beam transmission / secants        cpp/lib/sktran_disco/sktran_do_layerarray.cpp:890-979
//   S+/S- assembly                     cpp/include/sktran_disco/sktran_do_lpproduct.h:165-263
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
    int ngrid = 0, nwavel = 0, nleg = 0;
    const double* extinction = nullptr;  // [ngrid, nwavel] column-major (location fastest)
    const double* ssa = nullptr;         // [ngrid, nwavel]
    const double* leg = nullptr;         // [nleg, ngrid, nwavel]
    const double* solar_irradiance = nullptr;  // [nwavel] (nullptr -> 1)
    const double* albedo = nullptr;      // [nwavel] Lambertian
    const double* delta_f = nullptr;     // [ngrid, nwavel] delta-M truncation fraction of the (already scaled) storage, or null
    bool derivatives = false;
    double k(int q, int w) const { return extinction[size_t(w) * ngrid + q]; }
    double om(int q, int w) const { return ssa[size_t(w) * ngrid + q]; }
    double beta(int l, int q, int w) const { return l < nleg ? leg[(size_t(w) * ngrid + q) * nleg + l] : 0.0; }
};

// native derivative vector of one (wavelength, los): [0,ngrid) d/dk, [ngrid,2ngrid) d/dssa, [2ngrid] d/dalbedo
struct Result {
    std::vector<double> radiance;  // [nlos]
    std::vector<double> deriv;     // [nlos][nnative]
    std::vector<double> los_od;    // [nlos]
    int num_azimuth_used = 0;
};

class Solver {
    const skgeo::DiscoGeometry& G;
    const int nstr, N, nlyr, ngrid;
    int ND = 0;       // DO-internal derivative count
    int nnative = 0;  // native derivative count
    bool include_ss;  // single scatter inside DO (plane parallel only)

    // ---- per wavelength layer state
    std::vector<double> tau, ssa, totext, scatext, od_ceil, od_floor, a;  // a[p*nstr+l]
    std::vector<double> sec, dsec, trans, dtrans;                         // dsec[p*ND+d], dtrans[(lev)*ND+d]
    // derivative wiring: DO deriv d -> list of (native idx, weight*extinction)
    struct Wire { int native; double w; };
    std::vector<std::vector<Wire>> wires;
    std::vector<int> d_layer, d_kind;  // kind: 0 od, 1 ssa, 2 albedo
    std::vector<int> lstart, nloc;
    double albedo = 0, F0 = 1;
    // ---- per azimuth-order solution
    std::vector<double> eigk, deigk, Wp, Wm, dWp, dWm, Ap, Am, dAp, dAm;
    std::vector<double> Gpt, Gpb, Gmt, Gmb, dGpt, dGpb, dGmt, dGmb;
    std::vector<double> Lc, Mc, dLc, dMc;
    std::vector<double> Splus, Sminus, dSplus, dSminus, Gam, dGam;
    oracle_la::RealEigen eigsolver;
    oracle_la::BandMatrix bvp;
    std::vector<double> rhs;  // [M][1+ND]
    // ---- spherical table
    std::vector<int> point_slot;   // [sza][alt][angle] -> slot or -1
    std::vector<int> ground_slot;  // [sza][angle]
    int npoint = 0, nground = 0;
    std::vector<double> tab_val;    // [m][slot]  (interior slots then ground slots)
    std::vector<double> tab_deriv;  // [m][slot][nnative]
    std::vector<uint8_t> converged; // [m][slot] (interior only)
    std::vector<uint8_t> filled;    // [m][slot] value/derivative row written for this wavelength
    bool all_converged = false;

    inline int nl(int p) const { return nloc[p]; }
    size_t hidx(int p, int i, int j) const { return (size_t(p) * N + j) * N + i; }          // W(i, solution j)
    size_t dhidx(int p, int k, int i, int j) const { return ((size_t(p) * 3 + k) * N + j) * N + i; }

  public:
    Solver(const skgeo::DiscoGeometry& g, bool derivatives, bool ss_in_do)
        : G(g), nstr(g.nstr), N(g.nstr / 2), nlyr(g.nlyr), ngrid(g.ngrid), include_ss(ss_in_do) {
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
        wires.resize(ND);
        tau.resize(nlyr); ssa.resize(nlyr); totext.resize(nlyr); scatext.resize(nlyr);
        od_ceil.resize(nlyr); od_floor.resize(nlyr); a.resize(size_t(nlyr) * nstr);
        sec.resize(nlyr); dsec.resize(size_t(nlyr) * ND); trans.resize(nlyr + 1); dtrans.resize(size_t(nlyr + 1) * ND);
        eigk.resize(size_t(nlyr) * N); deigk.resize(size_t(nlyr) * 3 * N);
        Wp.resize(size_t(nlyr) * N * N); Wm.resize(Wp.size());
        dWp.resize(size_t(nlyr) * 3 * N * N); dWm.resize(dWp.size());
        Ap.resize(size_t(nlyr) * N); Am.resize(Ap.size()); dAp.resize(size_t(nlyr) * 3 * N); dAm.resize(dAp.size());
        Gpt.resize(size_t(nlyr) * N); Gpb.resize(Gpt.size()); Gmt.resize(Gpt.size()); Gmb.resize(Gpt.size());
        dGpt.resize(size_t(nlyr) * ND * N); dGpb.resize(dGpt.size()); dGmt.resize(dGpt.size()); dGmb.resize(dGpt.size());
        Lc.resize(size_t(nlyr) * N); Mc.resize(Lc.size()); dLc.resize(size_t(nlyr) * ND * N); dMc.resize(dLc.size());
        Splus.resize(size_t(N) * N); Sminus.resize(Splus.size()); dSplus.resize(3 * Splus.size()); dSminus.resize(dSplus.size());
        Gam.resize(Splus.size()); dGam.resize(dSplus.size());
        bvp.resize(nstr * nlyr, 3 * N - 1, 3 * N - 1);
        rhs.resize(size_t(nstr) * nlyr * (1 + ND));
        if (G.spherical_pp) {
            point_slot.assign(G.need_point.size(), -1);
            for (size_t i = 0; i < G.need_point.size(); ++i)
                if (G.need_point[i]) point_slot[i] = npoint++;
            ground_slot.assign(G.need_ground.size(), -1);
            for (size_t i = 0; i < G.need_ground.size(); ++i)
                if (G.need_ground[i]) ground_slot[i] = npoint + nground++;
            tab_val.assign(size_t(nstr) * (npoint + nground), 0.0);
            tab_deriv.assign(size_t(nstr) * (npoint + nground) * nnative, 0.0);
            converged.assign(size_t(nstr) * npoint, 0);
            filled.assign(size_t(nstr) * (npoint + nground), 0);
        }
    }
    int num_native() const { return nnative; }

  private:
    double* dG(std::vector<double>& g, int p, int d) { return &g[(size_t(p) * ND + d) * N]; }
    double* dLM(std::vector<double>& g, int p, int d) { return &g[(size_t(p) * ND + d) * N]; }

    // ------------------------------------------------------------------ layer optics
    void configure_layers(const AtmosphereView& atm, int w, int szaidx) {
        double ceiling_depth = 0, floor_depth = 0;
        for (int p = 0; p < nlyr; ++p) {
            double dh = G.ceil_h[p] - G.floor_h[p];
            double od = 0, s = 0;
            double* ap = &a[size_t(p) * nstr];
            for (int l = 0; l < nstr; ++l) ap[l] = 0;
            for (int q = 0; q < ngrid; ++q) {
                double wt = G.W(p, q);
                if (wt > 0) {
                    double kext = atm.k(q, w), kscat = atm.om(q, w) * kext;
                    od += kext * wt;
                    s += kscat * wt;
                    // delta-M: the multiple-scatter expansion drops the truncated forward peak (sktran_do_layerarray.cpp:395-405)
                    const double f = atm.delta_f ? atm.delta_f[size_t(w) * atm.ngrid + q] : 0.0;
                    for (int l = 0; l < nstr; ++l) ap[l] += wt * kscat * (atm.beta(l, q, w) - (2 * l + 1) * f / (1 - f));
                }
            }
            if (s > 0) for (int l = 0; l < nstr; ++l) ap[l] /= s;
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
        albedo = atm.albedo ? atm.albedo[w] : 0.0;
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
        const double* ap = &a[size_t(p) * nstr];
        const int nk = nl(p);
        const double om = ssa[p];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double sp = 0, eta = 0;
                for (int l = m; l < nstr; ++l) {
                    double t = G.lpmu(m, i, l) * G.lpmu(m, j, l) * ap[l];
                    sp += t;
                    eta += ((l - m) % 2 != 0) ? -t : t;
                }
                double fac = -0.5 * G.wt[j] / G.mu[i];
                sp *= fac;
                eta *= fac;
                for (int k = 0; k < nk; ++k) {
                    // d_legendre = 0 for the od/ssa/albedo derivatives
                    double dsp = sp * d_ssa(k), deta = eta * d_ssa(k);
                    dSminus[(size_t(k) * N + i) * N + j] = dsp - deta;
                    dSplus[(size_t(k) * N + i) * N + j] = dsp + deta;
                }
                sp *= om;
                eta *= om;
                if (i == j) sp += 1 / G.mu[i];
                Sminus[size_t(i) * N + j] = sp - eta;
                Splus[size_t(i) * N + j] = sp + eta;
            }
        // eigmtx = S- S+
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int k = 0; k < N; ++k) s += Sminus[size_t(i) * N + k] * Splus[size_t(k) * N + j];
                Gam[size_t(i) * N + j] = s;
            }
        for (int k = 0; k < nk; ++k) {
            const double* dsm = &dSminus[size_t(k) * N * N];
            const double* dsp = &dSplus[size_t(k) * N * N];
            double* dg = &dGam[size_t(k) * N * N];
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    double s = 0;
                    for (int r = 0; r < N; ++r) s += dsm[size_t(i) * N + r] * Splus[size_t(r) * N + j] + Sminus[size_t(i) * N + r] * dsp[size_t(r) * N + j];
                    dg[size_t(i) * N + j] = s;
                }
        }
        std::vector<double> wr(N), wi(N), X(size_t(N) * N), MXm(size_t(N) * N);
        if (!eigsolver.compute(N, Gam.data(), wr.data(), wi.data(), X.data())) {
            // fallback used by the reference: diagonal entries / identity vectors
            for (int i = 0; i < N; ++i) {
                wr[i] = Gam[size_t(i) * N + i];
                for (int j = 0; j < N; ++j) X[size_t(i) * N + j] = (i == j);
            }
        }
        for (int j = 0; j < N; ++j) {
            double nrm = 0;
            for (int i = 0; i < N; ++i) nrm += X[size_t(i) * N + j] * X[size_t(i) * N + j];
            nrm = std::sqrt(nrm);
            for (int i = 0; i < N; ++i) X[size_t(i) * N + j] /= nrm;
        }
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double s = 0;
                for (int r = 0; r < N; ++r) s += Splus[size_t(i) * N + r] * X[size_t(r) * N + j];
                MXm[size_t(i) * N + j] = s;
            }
        double* kk = &eigk[size_t(p) * N];
        for (int j = 0; j < N; ++j) {
            if (wr[j] <= 0) throw std::runtime_error("An homogeneous solution was found to be imaginary");
            kk[j] = std::sqrt(std::abs(wr[j]));
            for (int i = 0; i < N; ++i) {
                Wp[hidx(p, i, j)] = 0.5 * (X[size_t(i) * N + j] + MXm[size_t(i) * N + j] / kk[j]);
                Wm[hidx(p, i, j)] = 0.5 * (X[size_t(i) * N + j] - MXm[size_t(i) * N + j] / kk[j]);
            }
        }
        if (nk == 0) return;
        // linearization: bordered (N+1) systems
        const int n1 = N + 1;
        std::vector<double> lhs(size_t(n1) * n1), sol(n1);
        std::vector<int> piv(n1);
        for (int j = 0; j < N; ++j) {
            bool factored = false;
            for (int k = 0; k < nk; ++k) {
                const double* dg = &dGam[size_t(k) * N * N];
                bool zero = true;
                for (int i = 0; i < N; ++i) {
                    double s = 0;
                    for (int r = 0; r < N; ++r) s += dg[size_t(i) * N + r] * X[size_t(r) * N + j];
                    sol[i] = -s;
                    if (sol[i] != 0.0) zero = false;
                }
                sol[N] = 0;
                // the reference checks the whole RHS block; with od/albedo columns identically zero this is per column here
                if (zero) {
                    for (int i = 0; i <= N; ++i) sol[i] = 0;
                } else {
                    if (!factored) {
                        for (int r = 0; r < N; ++r) {
                            for (int cidx = 0; cidx < N; ++cidx) lhs[size_t(r) * n1 + cidx] = Gam[size_t(r) * N + cidx];
                            lhs[size_t(r) * n1 + r] -= wr[j];
                            lhs[size_t(r) * n1 + N] = -2 * kk[j] * X[size_t(r) * N + j];
                        }
                        for (int cidx = 0; cidx < N; ++cidx) lhs[size_t(N) * n1 + cidx] = X[size_t(cidx) * N + j];
                        lhs[size_t(N) * n1 + N] = 0;
                        oracle_la::lu_factor(n1, lhs.data(), piv.data());
                        factored = true;
                    }
                    oracle_la::lu_solve(n1, lhs.data(), piv.data(), sol.data());
                }
                double dk = sol[N];
                deigk[(size_t(p) * 3 + k) * N + j] = dk;
                const double* dsp = &dSplus[size_t(k) * N * N];
                for (int i = 0; i < N; ++i) {
                    double t1 = 0, t2 = 0, t3 = 0;
                    for (int r = 0; r < N; ++r) {
                        t1 += Splus[size_t(i) * N + r] * X[size_t(r) * N + j];
                        t2 += dsp[size_t(i) * N + r] * X[size_t(r) * N + j];
                        t3 += Splus[size_t(i) * N + r] * sol[r];
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
        const double* ap = &a[size_t(p) * nstr];
        const double om = ssa[p];
        std::vector<double> Qp(N), Qm(N), dQp(size_t(3) * N), dQm(size_t(3) * N);
        bool allzero = true;
        for (int i = 0; i < N; ++i) {
            double sneg = 0, spos = 0;
            for (int l = m; l < nstr; ++l) {
                double t = ap[l] * G.lpmu(m, i, l) * G.lpcsz(szaidx, m, l);
                spos += t;
                sneg += ((l - m) % 2 != 0) ? -t : t;
            }
            double factor = (2 - (m == 0 ? 1 : 0)) * (1.0 / (4.0 * kPi)) * G.wt[i];
            sneg *= factor;
            spos *= factor;
            for (int k = 0; k < nk; ++k) {
                dQm[size_t(k) * N + i] = sneg * d_ssa(k);
                dQp[size_t(k) * N + i] = spos * d_ssa(k);
                if (dQm[size_t(k) * N + i] != 0 || dQp[size_t(k) * N + i] != 0) allzero = false;
            }
            Qm[i] = sneg * om;
            Qp[i] = spos * om;
            if (Qm[i] != 0 || Qp[i] != 0) allzero = false;
        }
        double* gpt = &Gpt[size_t(p) * N]; double* gpb = &Gpb[size_t(p) * N];
        double* gmt = &Gmt[size_t(p) * N]; double* gmb = &Gmb[size_t(p) * N];
        for (int i = 0; i < N; ++i) { gpt[i] = gpb[i] = gmt[i] = gmb[i] = 0; Ap[size_t(p) * N + i] = Am[size_t(p) * N + i] = 0; }
        for (int k = 0; k < 3; ++k) for (int i = 0; i < N; ++i) dAp[(size_t(p) * 3 + k) * N + i] = dAm[(size_t(p) * 3 + k) * N + i] = 0;
        if (ND) {
            std::memset(dG(dGpt, p, 0), 0, sizeof(double) * ND * N);
            std::memset(dG(dGpb, p, 0), 0, sizeof(double) * ND * N);
            std::memset(dG(dGmt, p, 0), 0, sizeof(double) * ND * N);
            std::memset(dG(dGmb, p, 0), 0, sizeof(double) * ND * N);
        }
        if (allzero) return;
        const double T = trans[p], sc = sec[p], th = tau[p];
        const double* dT = &dtrans[size_t(p) * ND];
        const double* dsc = &dsec[size_t(p) * ND];
        const double exp_ts = std::exp(-th * sc);
        std::vector<double> dCp(ND), dCm(ND);
        for (int i = 0; i < N; ++i) {  // solution index
            double norm = 0, Apv = 0, Amv = 0;
            double dnorm[3] = {0, 0, 0}, dApv[3] = {0, 0, 0}, dAmv[3] = {0, 0, 0};
            for (int j = 0; j < N; ++j) {
                double hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                norm += G.wt[j] * G.mu[j] * (hp * hp - hm * hm);
                Apv += Qp[j] * hp + Qm[j] * hm;
                Amv += Qm[j] * hp + Qp[j] * hm;
                for (int k = 0; k < nk; ++k) {
                    double dhp = dWp[dhidx(p, k, j, i)], dhm = dWm[dhidx(p, k, j, i)];
                    dnorm[k] += G.wt[j] * G.mu[j] * (2.0 * dhp * hp - 2.0 * dhm * hm);
                    dApv[k] += Qp[j] * dhp + dQp[size_t(k) * N + j] * hp + Qm[j] * dhm + dQm[size_t(k) * N + j] * hm;
                    dAmv[k] += Qm[j] * dhp + dQm[size_t(k) * N + j] * hp + Qp[j] * dhm + dQp[size_t(k) * N + j] * hm;
                }
            }
            Apv /= norm;
            Amv /= norm;
            for (int k = 0; k < nk; ++k) {
                dApv[k] = dApv[k] / norm - dnorm[k] * Apv / norm;
                dAmv[k] = dAmv[k] / norm - dnorm[k] * Amv / norm;
                dAp[(size_t(p) * 3 + k) * N + i] = dApv[k];
                dAm[(size_t(p) * 3 + k) * N + i] = dAmv[k];
            }
            Ap[size_t(p) * N + i] = Apv;
            Am[size_t(p) * N + i] = Amv;
            const double ev = eigk[size_t(p) * N + i];
            const double exp_te = std::exp(-th * ev);
            double Cp, Cm;
            if (std::abs(sc - ev) > kGreensEps) {
                double den = sc - ev;
                Cp = T * (exp_te - exp_ts) / den;
                for (int d = 0; d < ND; ++d)
                    dCp[d] = (exp_te - exp_ts) / den * dT[d] + dsc[d] * (T * th * exp_ts - Cp) / den;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
                    dCp[ls + k] += dk / den * (Cp - T * th * exp_te);
                    dCp[ls + k] -= d_od(k) * T / den * (ev * exp_te - sc * exp_ts);
                }
            } else {
                Cp = T * exp_te * th * (1 - th / 2 * (sc - ev));
                for (int d = 0; d < ND; ++d)
                    dCp[d] = exp_te * th * (1 - th / 2 * (sc - ev)) * dT[d] + -1.0 * dsc[d] * th / 2.0 * th * T * exp_te;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
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
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
                    dCm[ls + k] += dk / den * (T * th * exp_sum - Cm);
                    dCm[ls + k] += d_od(k) * T * exp_sum;
                }
            } else {
                Cm = T * th * (1 - th / 2 * (sc + ev));
                for (int d = 0; d < ND; ++d)
                    dCm[d] = dT[d] * th * (1 - th / 2 * (sc + ev)) + dsc[d] * -1.0 * th / 2 * th * T;
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
                    dCm[ls + k] += dk * T * th * th / 2 * -1.0;
                    dCm[ls + k] += d_od(k) * T * (1 - th * (sc + ev));
                }
            }
            for (int j = 0; j < N; ++j) {
                double hp = Wp[hidx(p, j, i)], hm = Wm[hidx(p, j, i)];
                gpt[j] += Amv * Cm * hm;
                gmt[j] += Amv * Cm * hp;
                gpb[j] += Apv * Cp * hp;
                gmb[j] += Apv * Cp * hm;
                for (int d = 0; d < ND; ++d) {
                    dG(dGpt, p, d)[j] += Amv * hm * dCm[d];
                    dG(dGmt, p, d)[j] += Amv * hp * dCm[d];
                    dG(dGpb, p, d)[j] += Apv * hp * dCp[d];
                    dG(dGmb, p, d)[j] += Apv * hm * dCp[d];
                }
                for (int k = 0; k < nk; ++k) {
                    double dhp = dWp[dhidx(p, k, j, i)], dhm = dWm[dhidx(p, k, j, i)];
                    dG(dGpt, p, ls + k)[j] += dAmv[k] * Cm * hm + Amv * Cm * dhm;
                    dG(dGmt, p, ls + k)[j] += dAmv[k] * Cm * hp + Amv * Cm * dhp;
                    dG(dGpb, p, ls + k)[j] += dApv[k] * Cp * hp + Apv * Cp * dhp;
                    dG(dGmb, p, ls + k)[j] += dApv[k] * Cp * hm + Apv * Cp * dhm;
                }
            }
        }
    }

    // ------------------------------------------------------------------ boundary value problem
    double rho(int m) const { return m == 0 ? albedo : 0.0; }  // Lambertian, max_azimuthal_order == 1

    void solve_bvp(int m) {
        const int M = nstr * nlyr, n2 = nstr, nr = 1 + ND;
        bvp.set_zero();
        std::fill(rhs.begin(), rhs.end(), 0.0);
        auto E = [&](int p, int j) { return std::exp(-std::abs(eigk[size_t(p) * N + j]) * tau[p]); };
        const bool refl = (m < 1);
        const double kd = (m == 0) ? 2.0 : 1.0;
        // TOA
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) {
                bvp.at(i, j) = Wp[hidx(0, i, j)];
                bvp.at(i, j + N) = Wm[hidx(0, i, j)] * E(0, j);
            }
            rhs[size_t(i) * nr] = -Gpt[i];
            for (int d = 0; d < ND; ++d) rhs[size_t(i) * nr + 1 + d] = -dG(dGpt, 0, d)[i];
        }
        // continuity
        for (int p = 1; p < nlyr; ++p) {
            int r0 = N + (p - 1) * n2, cu = (p - 1) * n2, cl = p * n2;
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) {
                    double eu = E(p - 1, j), el = E(p, j);
                    bvp.at(r0 + i + N, cu + j) = Wp[hidx(p - 1, i, j)] * eu;
                    bvp.at(r0 + i + N, cl + j) = -Wp[hidx(p, i, j)];
                    bvp.at(r0 + i, cu + j) = Wm[hidx(p - 1, i, j)] * eu;
                    bvp.at(r0 + i, cl + j) = -Wm[hidx(p, i, j)];
                    bvp.at(r0 + i + N, cu + N + j) = Wm[hidx(p - 1, i, j)];
                    bvp.at(r0 + i + N, cl + N + j) = -Wm[hidx(p, i, j)] * el;
                    bvp.at(r0 + i, cu + N + j) = Wp[hidx(p - 1, i, j)];
                    bvp.at(r0 + i, cl + N + j) = -Wp[hidx(p, i, j)] * el;
                }
                rhs[size_t(r0 + i) * nr] = -Gmb[size_t(p - 1) * N + i] + Gmt[size_t(p) * N + i];
                rhs[size_t(r0 + i + N) * nr] = -Gpb[size_t(p - 1) * N + i] + Gpt[size_t(p) * N + i];
                for (int d = 0; d < ND; ++d) {
                    rhs[size_t(r0 + i) * nr + 1 + d] = dG(dGmt, p, d)[i] - dG(dGmb, p - 1, d)[i];
                    rhs[size_t(r0 + i + N) * nr + 1 + d] = dG(dGpt, p, d)[i] - dG(dGpb, p - 1, d)[i];
                }
            }
        }
        // ground
        {
            const int p = nlyr - 1, r0 = M - N, c0 = p * n2;
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) {
                    double vm = Wm[hidx(p, i, j)], vp = Wp[hidx(p, i, j)];
                    if (refl)
                        for (int s = 0; s < N; ++s) {
                            vm -= kd * rho(m) * G.wt[s] * G.mu[s] * Wp[hidx(p, s, j)];
                            vp -= kd * rho(m) * G.wt[s] * G.mu[s] * Wm[hidx(p, s, j)];
                        }
                    bvp.at(r0 + i, c0 + j) = vm * E(p, j);
                    bvp.at(r0 + i, c0 + N + j) = vp;
                }
                double direct = 0, um = Gmb[size_t(p) * N + i];
                if (refl) {
                    direct = G.csz[cur_sza] * rho(m) / kPi * trans[nlyr];
                    for (int s = 0; s < N; ++s) um -= kd * rho(m) * G.wt[s] * G.mu[s] * Gpb[size_t(p) * N + s];
                }
                rhs[size_t(r0 + i) * nr] = direct - um;
                for (int d = 0; d < ND; ++d) {
                    double dalb = (d_layer[d] == p) ? d_alb(d_kind[d]) : 0.0;
                    double ddir = 0, dum = dG(dGmb, p, d)[i];
                    if (refl) {
                        ddir = G.csz[cur_sza] * rho(m) / kPi * dtrans[size_t(nlyr) * ND + d] + G.csz[cur_sza] * dalb / kPi * trans[nlyr];
                        for (int s = 0; s < N; ++s) {
                            dum -= kd * rho(m) * G.wt[s] * G.mu[s] * dG(dGpb, p, d)[s];
                            dum -= kd * dalb * G.wt[s] * G.mu[s] * Gpb[size_t(p) * N + s];
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
            for (int j = 0; j < N; ++j) {
                Lc[size_t(p) * N + j] = x[size_t(p) * n2 + j];
                Mc[size_t(p) * N + j] = x[size_t(p) * n2 + N + j];
            }
        if (ND == 0) return;
        // rhs_d -= dA_d x   (dA_d only touches the columns of the derivative's layer)
        for (int d = 0; d < ND; ++d) {
            const int p = d_layer[d], k = d_kind[d], c0 = p * n2;
            const double* xl = &x[c0];
            const double* xm = &x[c0 + N];
            auto dE = [&](int j) {
                return -(deigk[(size_t(p) * 3 + k) * N + j] * tau[p] + eigk[size_t(p) * N + j] * d_od(k)) * E(p, j);
            };
            // boundary above layer p
            if (p == 0) {
                for (int i = 0; i < N; ++i) {
                    double s = 0;
                    for (int j = 0; j < N; ++j)
                        s += dWp[dhidx(p, k, i, j)] * xl[j] + (dWm[dhidx(p, k, i, j)] * E(p, j) + Wm[hidx(p, i, j)] * dE(j)) * xm[j];
                    rhs[size_t(i) * nr + 1 + d] -= s;
                }
            } else {
                int r0 = N + (p - 1) * n2;
                for (int i = 0; i < N; ++i) {
                    double s1 = 0, s0 = 0;
                    for (int j = 0; j < N; ++j) {
                        s1 += -dWp[dhidx(p, k, i, j)] * xl[j] + (-dWm[dhidx(p, k, i, j)] * E(p, j) - Wm[hidx(p, i, j)] * dE(j)) * xm[j];
                        s0 += -dWm[dhidx(p, k, i, j)] * xl[j] - (dWp[dhidx(p, k, i, j)] * E(p, j) + Wp[hidx(p, i, j)] * dE(j)) * xm[j];
                    }
                    rhs[size_t(r0 + i + N) * nr + 1 + d] -= s1;
                    rhs[size_t(r0 + i) * nr + 1 + d] -= s0;
                }
            }
            // boundary below layer p
            if (p == nlyr - 1) {
                int r0 = M - N;
                for (int i = 0; i < N; ++i) {
                    double s = 0;
                    for (int j = 0; j < N; ++j) {
                        double vm = Wm[hidx(p, i, j)], vp_d = dWp[dhidx(p, k, i, j)], vm_d = dWm[dhidx(p, k, i, j)];
                        if (refl)
                            for (int t = 0; t < N; ++t) {
                                double f = kd * G.wt[t] * G.mu[t];
                                vm -= f * rho(m) * Wp[hidx(p, t, j)];
                                vm_d -= f * rho(m) * dWp[dhidx(p, k, t, j)] + f * d_alb(k) * Wp[hidx(p, t, j)];
                                vp_d -= f * rho(m) * dWm[dhidx(p, k, t, j)] + f * d_alb(k) * Wm[hidx(p, t, j)];
                            }
                        s += (vm_d * E(p, j) + vm * dE(j)) * xl[j] + vp_d * xm[j];
                    }
                    rhs[size_t(r0 + i) * nr + 1 + d] -= s;
                }
            } else {
                int r0 = N + p * n2;
                for (int i = 0; i < N; ++i) {
                    double s1 = 0, s0 = 0;
                    for (int j = 0; j < N; ++j) {
                        s1 += (dWp[dhidx(p, k, i, j)] * E(p, j) + Wp[hidx(p, i, j)] * dE(j)) * xl[j] + dWm[dhidx(p, k, i, j)] * xm[j];
                        s0 += (dWm[dhidx(p, k, i, j)] * E(p, j) + Wm[hidx(p, i, j)] * dE(j)) * xl[j] + dWp[dhidx(p, k, i, j)] * xm[j];
                    }
                    rhs[size_t(r0 + i + N) * nr + 1 + d] -= s1;
                    rhs[size_t(r0 + i) * nr + 1 + d] -= s0;
                }
            }
        }
        bvp.solve(rhs.data(), nr);  // column 0 is re-solved (cheap) and ignored
        for (int p = 0; p < nlyr; ++p)
            for (int d = 0; d < ND; ++d)
                for (int j = 0; j < N; ++j) {
                    dLM(dLc, p, d)[j] = rhs[size_t(p * n2 + j) * nr + 1 + d];
                    dLM(dMc, p, d)[j] = rhs[size_t(p * n2 + N + j) * nr + 1 + d];
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

    // legendre sums at an outgoing direction with tabulated lp_out[l]:  (post-processing call-site naming)
    //   lm[j] ("lpsum_minus") = 0.5 w_j ssa sum_l a_l P_l(out) P_l(mu_j)
    //   lp[j] ("lpsum_plus")  = same with (-1)^(l-m)
    void lpsums(int m, int p, const double* lp_out, double* lp, double* lm, double* dlp, double* dlm) {
        const double* ap = &a[size_t(p) * nstr];
        for (int j = 0; j < N; ++j) {
            double s0 = 0, s1 = 0;
            for (int l = m; l < nstr; ++l) {
                double t = ap[l] * lp_out[l] * G.lpmu(m, j, l);
                s0 += t;
                s1 += ((l - m) % 2 != 0) ? -t : t;
            }
            for (int k = 0; k < nl(p); ++k) {
                dlm[size_t(k) * N + j] = s0 * d_ssa(k) * 0.5 * G.wt[j];
                dlp[size_t(k) * N + j] = s1 * d_ssa(k) * 0.5 * G.wt[j];
            }
            lm[j] = s0 * 0.5 * ssa[p] * G.wt[j];
            lp[j] = s1 * 0.5 * ssa[p] * G.wt[j];
        }
    }

    void map_to_native(const double* do_deriv, double scale, double* native) const {
        for (int d = 0; d < ND; ++d) {
            double v = do_deriv[d] * scale;
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

  public:
    void solve(const AtmosphereView& atm, int w, Result& res);
};

// ======================================================================================
inline void Solver::pp_postprocess(int m, int szaidx, Result& res, std::vector<double>& rad, std::vector<double>& drad) {
    (void)szaidx;
    const int p_bot = nlyr - 1;
    std::vector<double> comp_d(ND), integ_d(ND), strm_d(ND), lp(N), lm(N), dlp(3 * N), dlm(3 * N);
    std::vector<double> Yp(N), Ym(N), dYp(3 * N), dYm(3 * N), dEf(ND), dDp(ND), dDm(ND);
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
        // ground reflection (computeReflectedIntensities)
        double comp = 0;
        std::fill(comp_d.begin(), comp_d.end(), 0.0);
        if (m < 1) {
            const int p = p_bot, ls = ND ? lstart[p] : 0, nk = nl(p);
            double diffuse = 0;
            std::vector<double> diffuse_d(ND, 0.0);
            for (int i = 0; i < N; ++i) {
                double r = rho(m);
                double sc_v = Gpb[size_t(p) * N + i];
                for (int d = 0; d < ND; ++d) strm_d[d] = dG(dGpb, p, d)[i];
                double factor = (m == 0 ? 2.0 : 1.0) * G.mu[i] * G.wt[i];
                for (int jj = 0; jj < N; ++jj) {
                    double st = std::exp(-std::abs(eigk[size_t(p) * N + jj]) * tau[p]);
                    double hp = Wp[hidx(p, i, jj)], hm = Wm[hidx(p, i, jj)];
                    sc_v += Lc[size_t(p) * N + jj] * hp * st + Mc[size_t(p) * N + jj] * hm;
                    for (int d = 0; d < ND; ++d)
                        strm_d[d] += dLM(dLc, p, d)[jj] * hp * st + dLM(dMc, p, d)[jj] * hm;
                    for (int k = 0; k < nk; ++k) {
                        double dst = -(deigk[(size_t(p) * 3 + k) * N + jj] * tau[p] + eigk[size_t(p) * N + jj] * d_od(k)) * st;
                        strm_d[ls + k] += Lc[size_t(p) * N + jj] * dWp[dhidx(p, k, i, jj)] * st + Lc[size_t(p) * N + jj] * hp * dst;
                        strm_d[ls + k] += Mc[size_t(p) * N + jj] * dWm[dhidx(p, k, i, jj)];
                    }
                }
                diffuse += factor * sc_v * r;
                for (int d = 0; d < ND; ++d) diffuse_d[d] += factor * strm_d[d] * r;
                for (int k = 0; k < nk; ++k) diffuse_d[ls + k] += factor * sc_v * d_alb(k);
            }
            if (include_ss) {
                double direct = G.csz[cur_sza] / kPi * trans[nlyr] * rho(m);
                comp = direct + diffuse;
                for (int d = 0; d < ND; ++d) comp_d[d] = G.csz[cur_sza] / kPi * dtrans[size_t(nlyr) * ND + d] * rho(m) + diffuse_d[d];
                for (int k = 0; k < nk; ++k) comp_d[ls + k] += G.csz[cur_sza] / kPi * trans[nlyr] * d_alb(k);
            } else {
                comp = diffuse;
                comp_d = diffuse_d;
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
            comp *= att;
            for (int d = 0; d < ND; ++d) comp_d[d] *= att;
            for (int k = 0; k < nk; ++k) comp_d[ls + k] += -layerfraction / mu * comp * d_od(k);

            // integrate_source(m, mu, obsod, lp_mu, ...)
            double exit_od = od_ceil[p];
            double upper = (obs_od < exit_od) ? exit_od : obs_od;
            double x = upper - exit_od;
            const double T = trans[p], sc = sec[p], th = tau[p];
            const double* dT = &dtrans[size_t(p) * ND];
            const double* dsc = &dsec[size_t(p) * ND];
            lpsums(m, p, &G.lp_los[(j * nstr + m) * nstr], lp.data(), lm.data(), dlp.data(), dlm.data());
            double Q = 0, dQ[3] = {0, 0, 0};
            if (include_ss) {
                double sneg = 0;
                const double* ap = &a[size_t(p) * nstr];
                for (int l = m; l < nstr; ++l) {
                    double t = ap[l] * G.lplos(j, m, l) * G.lpcsz(cur_sza, m, l);
                    sneg += ((l - m) % 2 != 0) ? -t : t;
                }
                sneg *= (2 - (m == 0 ? 1 : 0)) * (1.0 / (4.0 * kPi));
                for (int k = 0; k < nk; ++k) dQ[k] = sneg * d_ssa(k);
                Q = sneg * ssa[p];
            }
            for (int i = 0; i < N; ++i) {
                double yp = 0, ym = 0;
                for (int r = 0; r < N; ++r) {
                    yp += lp[r] * Wp[hidx(p, r, i)] + lm[r] * Wm[hidx(p, r, i)];
                    ym += lp[r] * Wm[hidx(p, r, i)] + lm[r] * Wp[hidx(p, r, i)];
                }
                Yp[i] = yp;
                Ym[i] = ym;
                for (int k = 0; k < nk; ++k) {
                    double a1 = 0, a2 = 0;
                    for (int r = 0; r < N; ++r) {
                        a1 += dlp[size_t(k) * N + r] * Wp[hidx(p, r, i)] + lp[r] * dWp[dhidx(p, k, r, i)] + dlm[size_t(k) * N + r] * Wm[hidx(p, r, i)] + lm[r] * dWm[dhidx(p, k, r, i)];
                        a2 += dlp[size_t(k) * N + r] * Wm[hidx(p, r, i)] + lp[r] * dWm[dhidx(p, k, r, i)] + dlm[size_t(k) * N + r] * Wp[hidx(p, r, i)] + lm[r] * dWp[dhidx(p, k, r, i)];
                    }
                    dYp[size_t(k) * N + i] = a1;
                    dYm[size_t(k) * N + i] = a2;
                }
            }
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
            double J = 0, V = 0;
            std::fill(integ_d.begin(), integ_d.end(), 0.0);
            for (int i = 0; i < N; ++i) {
                const double ev = eigk[size_t(p) * N + i];
                double hp, hm, dhp[3] = {0, 0, 0}, dhm[3] = {0, 0, 0};
                {
                    double den = 1.0 + mu * ev;
                    if (std::abs(den) > 0.0001) {
                        double e1 = std::exp(-x * ev), e2 = std::exp(-1.0 * th * ev) * std::exp(-(th - x) / mu);
                        hp = (e1 - e2) / den;
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * N + i];
                            dhp[k] = (-1.0 * e1 * (dk * x + d_od(k) * (1 - lf) * ev) + e2 * ((ev + lf / mu) * d_od(k) + th * dk)) / den;
                            dhp[k] -= 1.0 / den * hp * mu * dk;
                        }
                    } else {
                        double e1 = std::exp(-x * ev);
                        hp = (th - x) / mu * e1 * (1 - (th - x) * (ev + 1 / mu));
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * N + i];
                            dhp[k] = -1.0 * x * dk * hp + 1 / mu * e1 * d_od(k) * (1 - (th - x) * (ev + 1 / mu)) + 1 / mu * e1 * (th - x) * -1.0 * (ev + 1 / mu) * d_od(k);
                        }
                    }
                    double den2 = 1.0 - mu * ev;
                    if (std::abs(den2) > 0.0001) {
                        double e1 = std::exp(-1.0 * ev * (th - x)), e2 = std::exp(-1 / mu * (th - x));
                        hm = (e1 - e2) / den2;
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * N + i];
                            dhm[k] = (-1.0 * e1 * (dk * (th - x) + d_od(k) * lf * ev) + e2 * d_od(k) * lf / mu) / den2;
                            dhm[k] += 1.0 / den2 * hm * mu * dk;
                        }
                    } else {
                        double e1 = std::exp(-1.0 * ev * (th - x));
                        hm = e1 * (th - x) / mu * (1 - (th - x) * (ev - 1 / mu));
                        for (int k = 0; k < nk; ++k) {
                            double dk = deigk[(size_t(p) * 3 + k) * N + i];
                            dhm[k] = -1.0 * (ev * d_od(k) + th * dk) * hm + e1 * d_od(k) / mu * (1 - (th - x) * (ev - 1 / mu)) + e1 * (th - x) / mu * -1.0 * d_od(k) * (ev - 1 / mu);
                        }
                    }
                }
                const double Lv = Lc[size_t(p) * N + i], Mv = Mc[size_t(p) * N + i];
                J += Yp[i] * hp * Lv + Ym[i] * hm * Mv;
                for (int k = 0; k < nk; ++k) {
                    integ_d[ls + k] += dYp[size_t(k) * N + i] * hp * Lv + dYm[size_t(k) * N + i] * hm * Mv;
                    integ_d[ls + k] += Yp[i] * dhp[k] * Lv + Ym[i] * dhm[k] * Mv;
                }
                for (int d = 0; d < ND; ++d)
                    integ_d[d] += Yp[i] * hp * dLM(dLc, p, d)[i] + Ym[i] * hm * dLM(dMc, p, d)[i];
                double expfactor = std::exp(-1 * th * sc);
                double Dp = (-1 * T * expfactor * hm + Ef) / (sc + ev);
                double Dm = (T * hp - Ef) / (sc - ev);
                for (int d = 0; d < ND; ++d) {
                    dDp[d] = dsc[d] * (T * hm * th * expfactor - Dp) / (sc + ev) + dEf[d] / (sc + ev) + dT[d] * (-hm * expfactor) / (sc + ev);
                    dDm[d] = dsc[d] * (-Dm) / (sc - ev) + dT[d] * hp / (sc - ev) + dEf[d] * -1.0 / (sc - ev);
                }
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
                    dDp[ls + k] += dk * (-Dp) / (sc + ev) + dhm[k] * (-T * expfactor) / (sc + ev) + d_od(k) * (T * sc * hm * expfactor) / (sc + ev);
                    dDm[ls + k] += dhp[k] * T / (sc - ev) + dk * Dm / (sc - ev);
                }
                const double Apv = Ap[size_t(p) * N + i], Amv = Am[size_t(p) * N + i];
                V += Apv * Yp[i] * Dm + Amv * Ym[i] * Dp;
                for (int k = 0; k < nk; ++k) {
                    integ_d[ls + k] += Apv * dYp[size_t(k) * N + i] * Dm + Amv * dYm[size_t(k) * N + i] * Dp;
                    integ_d[ls + k] += dAp[(size_t(p) * 3 + k) * N + i] * Yp[i] * Dm + dAm[(size_t(p) * 3 + k) * N + i] * Ym[i] * Dp;
                }
                for (int d = 0; d < ND; ++d) integ_d[d] += Apv * Yp[i] * dDm[d] + Amv * Ym[i] * dDp[d];
            }
            double integral = J + V + Q * Ef;
            for (int k = 0; k < nk; ++k) integ_d[ls + k] += dQ[k] * Ef;
            for (int d = 0; d < ND; ++d) integ_d[d] += Q * dEf[d];
            comp += integral;
            for (int d = 0; d < ND; ++d) comp_d[d] += integ_d[d];
        }
        double azf = std::cos(m * los.azimuth);
        rad[j] += comp * azf;
        if (ND) map_to_native(comp_d.data(), azf, &drad[j * nnative]);
    }
    (void)res;
}

// ======================================================================================
inline void Solver::accumulate_ground(int m, int szaidx) {
    if (m >= 1) return;
    const int p = nlyr - 1, ls = ND ? lstart[p] : 0, nk = nl(p);
    const size_t nslot = npoint + nground;
    std::vector<double> strm_d(ND), diffuse_d(ND);
    for (int aidx = 0; aidx < G.nangle; ++aidx) {
        int slot = ground_slot[size_t(szaidx) * G.nangle + aidx];
        if (slot < 0) continue;
        double diffuse = 0;
        std::fill(diffuse_d.begin(), diffuse_d.end(), 0.0);
        for (int i = 0; i < N; ++i) {
            double r = albedo;  // Lambertian compute_expansion special case
            double sc_v = Gpb[size_t(p) * N + i];
            for (int d = 0; d < ND; ++d) strm_d[d] = dG(dGpb, p, d)[i];
            for (int jj = 0; jj < N; ++jj) {
                double st = std::exp(-std::abs(eigk[size_t(p) * N + jj]) * tau[p]);
                double hp = Wp[hidx(p, i, jj)], hm = Wm[hidx(p, i, jj)];
                sc_v += Lc[size_t(p) * N + jj] * hp * st + Mc[size_t(p) * N + jj] * hm;
                for (int d = 0; d < ND; ++d) strm_d[d] += dLM(dLc, p, d)[jj] * hp * st + dLM(dMc, p, d)[jj] * hm;
                for (int k = 0; k < nk; ++k) {
                    double dst = -(deigk[(size_t(p) * 3 + k) * N + jj] * tau[p] + eigk[size_t(p) * N + jj] * d_od(k)) * st;
                    strm_d[ls + k] += Lc[size_t(p) * N + jj] * hp * dst + Lc[size_t(p) * N + jj] * dWp[dhidx(p, k, i, jj)] * st;
                    strm_d[ls + k] += Mc[size_t(p) * N + jj] * dWm[dhidx(p, k, i, jj)];
                }
            }
            double factor = (m == 0 ? 2.0 : 1.0) * G.mu[i] * G.wt[i];
            diffuse += factor * sc_v * r;
            for (int d = 0; d < ND; ++d) diffuse_d[d] += factor * strm_d[d] * r;
            for (int k = 0; k < nk; ++k) diffuse_d[ls + k] += factor * sc_v * d_alb(k);
        }
        tab_val[size_t(m) * nslot + slot] = diffuse;
        filled[size_t(m) * nslot + slot] = 1;
        if (ND) {
            double* nat = &tab_deriv[(size_t(m) * nslot + slot) * nnative];
            std::fill(nat, nat + nnative, 0.0);
            map_to_native(diffuse_d.data(), 1.0, nat);
        }
    }
}

inline void Solver::accumulate_sources(int m, int szaidx) {
    const size_t nslot = npoint + nground;
    if (m >= 2) all_converged = true;
    accumulate_ground(m, szaidx);
    std::vector<double> lp(N), lm(N), dlp(3 * N), dlm(3 * N), Yp(N), Ym(N), dYp(3 * N), dYm(3 * N), temp(ND);
    std::vector<double> hpv(N), hmv(N), dhp(3 * N), dhm(3 * N), Dmv(N), Dpv(N), dDm(size_t(N) * ND), dDp(size_t(N) * ND);
    for (int lidx = 0; lidx < nlyr; ++lidx) {
        const int p = nlyr - lidx - 1;
        const int nk = nl(p), ls = ND ? lstart[p] : 0;
        const double altitude = (G.ceil_h[p] + G.floor_h[p]) / 2.0;
        const double lf = (G.ceil_h[p] - altitude) / (G.ceil_h[p] - G.floor_h[p]);
        const double th = tau[p], T = trans[p], sc = sec[p];
        const double* dT = &dtrans[size_t(p) * ND];
        const double* dsc = &dsec[size_t(p) * ND];
        const double x = th * lf;
        for (int i = 0; i < N; ++i) {
            const double ev = eigk[size_t(p) * N + i];
            hpv[i] = std::exp(-1.0 * ev * th * lf);
            hmv[i] = std::exp(-ev * th * (1 - lf));
            for (int k = 0; k < nk; ++k) {
                double dk = deigk[(size_t(p) * 3 + k) * N + i];
                dhp[size_t(k) * N + i] = -hpv[i] * (dk * th * lf + d_od(k) * ev * lf);
                dhm[size_t(k) * N + i] = -hmv[i] * (dk * th * (1 - lf) + d_od(k) * ev * (1 - lf));
            }
            // d_minus_sampled
            {
                double denom = sc - ev;
                double v = std::exp(-x * ev) - std::exp(-x * sc);
                double* dd = &dDm[size_t(i) * ND];
                for (int d = 0; d < ND; ++d) dd[d] = dT[d] * v + dsc[d] * T * (-v / denom + std::exp(-x * sc) * x);
                for (int k = 0; k < nk; ++k) {
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
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
                    double dk = deigk[(size_t(p) * 3 + k) * N + i];
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
            std::fill(temp.begin(), temp.end(), 0.0);
            lpsums(m, p, &G.lp_angle[(size_t(m) * G.nangle + aidx) * nstr], lp.data(), lm.data(), dlp.data(), dlm.data());
            for (int i = 0; i < N; ++i) {
                double yp = 0, ym = 0;
                for (int r = 0; r < N; ++r) {
                    yp += lp[r] * Wp[hidx(p, r, i)] + lm[r] * Wm[hidx(p, r, i)];
                    ym += lp[r] * Wm[hidx(p, r, i)] + lm[r] * Wp[hidx(p, r, i)];
                }
                Yp[i] = yp;
                Ym[i] = ym;
                for (int k = 0; k < nk; ++k) {
                    double a1 = 0, a2 = 0;
                    for (int r = 0; r < N; ++r) {
                        a1 += dlp[size_t(k) * N + r] * Wp[hidx(p, r, i)] + lp[r] * dWp[dhidx(p, k, r, i)] + dlm[size_t(k) * N + r] * Wm[hidx(p, r, i)] + lm[r] * dWm[dhidx(p, k, r, i)];
                        a2 += dlp[size_t(k) * N + r] * Wm[hidx(p, r, i)] + lp[r] * dWm[dhidx(p, k, r, i)] + dlm[size_t(k) * N + r] * Wp[hidx(p, r, i)] + lm[r] * dWp[dhidx(p, k, r, i)];
                    }
                    dYp[size_t(k) * N + i] = a1;
                    dYm[size_t(k) * N + i] = a2;
                }
            }
            double val = 0;
            for (int i = 0; i < N; ++i) {
                const double Lv = Lc[size_t(p) * N + i], Mv = Mc[size_t(p) * N + i];
                const double Apv = Ap[size_t(p) * N + i], Amv = Am[size_t(p) * N + i];
                val += Yp[i] * hpv[i] * Lv + Ym[i] * hmv[i] * Mv;
                val += Apv * Yp[i] * Dmv[i] + Amv * Ym[i] * Dpv[i];
                const double c1 = Yp[i] * hpv[i], c2 = Ym[i] * hmv[i], c3 = Apv * Yp[i], c4 = Amv * Ym[i];
                const double* dl = nullptr; (void)dl;
                for (int d = 0; d < ND; ++d)
                    temp[d] += dLM(dLc, p, d)[i] * c1 + dLM(dMc, p, d)[i] * c2 + dDm[size_t(i) * ND + d] * c3 + dDp[size_t(i) * ND + d] * c4;
                for (int k = 0; k < nk; ++k) {
                    temp[ls + k] += dYp[size_t(k) * N + i] * hpv[i] * Lv + dYm[size_t(k) * N + i] * hmv[i] * Mv;
                    temp[ls + k] += dhp[size_t(k) * N + i] * Yp[i] * Lv + dhm[size_t(k) * N + i] * Ym[i] * Mv;
                    temp[ls + k] += dYp[size_t(k) * N + i] * Dmv[i] * Apv + dYm[size_t(k) * N + i] * Dpv[i] * Amv;
                    temp[ls + k] += dAp[(size_t(p) * 3 + k) * N + i] * Yp[i] * Dmv[i] + dAm[(size_t(p) * 3 + k) * N + i] * Ym[i] * Dpv[i];
                }
            }
            val /= ssa[p];
            tab_val[size_t(m) * nslot + slot] = val;
            filled[size_t(m) * nslot + slot] = 1;
            if (m >= 2) {
                double prev = tab_val[size_t(m - 1) * nslot + slot], prev2 = tab_val[size_t(m - 2) * nslot + slot];
                if ((std::abs(val / prev) < 1e-4 || prev < 1e-10) && (std::abs(val / prev2) < 1e-4 || prev2 < 1e-10)) {
                    for (int azi = m + 1; azi < nstr; ++azi) converged[size_t(azi) * npoint + slot] = 1;
                } else {
                    all_converged = false;
                }
            }
            if (ND) {
                for (int k = 0; k < nk; ++k) temp[ls + k] -= d_ssa(k) * val;
                double* nat = &tab_deriv[(size_t(m) * nslot + slot) * nnative];
                std::fill(nat, nat + nnative, 0.0);
                map_to_native(temp.data(), 1.0 / ssa[p], nat);
            }
        }
    }
}

inline void Solver::integrate_los(const AtmosphereView& atm, int w, Result& res) {
    const size_t nslot = npoint + nground;
    for (int r = 0; r < G.nlos; ++r) {
        const auto& ray = G.rays[r];
        double I = 0;
        double* dI = nnative ? &res.deriv[size_t(r) * nnative] : nullptr;
        double od_total = 0;
        if (ray.ground_is_hit) {
            for (int m = 0; m < nstr; ++m) {
                double azf = std::cos(m * (kPi - ray.ground_saa));
                for (auto& e : ray.ground) {
                    int slot = ground_slot[size_t(e.sza) * G.nangle + e.angle];
                    if (!filled[size_t(m) * nslot + slot]) continue;
                    I += e.w * azf * tab_val[size_t(m) * nslot + slot];
                    if (nnative) {
                        const double* nat = &tab_deriv[(size_t(m) * nslot + slot) * nnative];
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
                for (size_t t = 0; t < L.od.idx.size(); ++t) dI[L.od.idx[t]] -= L.od.w[t] * I;
                for (int q = 0; q < nnative; ++q) dI[q] *= att;
            }
            I *= att;
            if (L.layer_distance < kMinShell) continue;
            double omega = 0;
            for (size_t t = 0; t < L.ssa_interp.idx.size(); ++t) omega += atm.om(L.ssa_interp.idx[t], w) * L.ssa_interp.w[t];
            double sf = 1 - att;
            double sv = 0;
            for (int m = 0; m < nstr; ++m) {
                double azf = std::cos(m * L.azi);
                for (auto& e : L.table) {
                    int slot = point_slot[(size_t(e.sza) * nlyr + e.alt) * G.nangle + e.angle];
                    sv += e.w * azf * tab_val[size_t(m) * nslot + slot];
                }
            }
            I += omega * sf * sv;
            if (nnative) {
                for (size_t t = 0; t < L.od.idx.size(); ++t) dI[L.od.idx[t]] += L.od.w[t] * (1 - sf) * sv * omega;
                for (size_t t = 0; t < L.ssa_interp.idx.size(); ++t) dI[ngrid + L.ssa_interp.idx[t]] += L.ssa_interp.w[t] * sf * sv;
                for (int m = 0; m < nstr; ++m) {
                    double azf = std::cos(m * L.azi);
                    for (auto& e : L.table) {
                        int slot = point_slot[(size_t(e.sza) * nlyr + e.alt) * G.nangle + e.angle];
                        if (!filled[size_t(m) * nslot + slot]) continue;
                        const double* nat = &tab_deriv[(size_t(m) * nslot + slot) * nnative];
                        double f = omega * sf * e.w * azf;
                        for (int q = 0; q < nnative; ++q) dI[q] += f * nat[q];
                    }
                }
            }
        }
        res.radiance[r] = I;
        res.los_od[r] = od_total;
    }
}

inline void Solver::solve(const AtmosphereView& atm, int w, Result& res) {
    res.radiance.assign(G.nlos, 0.0);
    res.deriv.assign(size_t(G.nlos) * nnative, 0.0);
    res.los_od.assign(G.nlos, 0.0);
    res.num_azimuth_used = 0;
    if (G.spherical_pp) {
        std::fill(tab_val.begin(), tab_val.end(), 0.0);
        std::fill(filled.begin(), filled.end(), 0);
        std::fill(converged.begin(), converged.end(), 0);
        for (int s = 0; s < G.nsza; ++s) {
            configure_layers(atm, w, s);
            all_converged = false;
            for (int m = 0; m < nstr; ++m) {
                solve_order(m, s);
                accumulate_sources(m, s);
                res.num_azimuth_used = std::max(res.num_azimuth_used, m + 1);
                if (all_converged) break;
            }
        }
        integrate_los(atm, w, res);
    } else {
        configure_layers(atm, w, 0);
        for (int m = 0; m < nstr; ++m) {
            solve_order(m, 0);
            pp_postprocess(m, 0, res, res.radiance, res.deriv);
            res.num_azimuth_used = m + 1;
        }
    }
}

}  // namespace oracle
