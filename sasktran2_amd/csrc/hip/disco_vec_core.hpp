// DISCO hot path for MI355X -- polarised (NSTOKES = 3) radiances, per-task device math, first correct form.
//
// Same formulation as disco_core.hpp (symmetrised eigenproblem + Jacobi, Green's function particular solution,
// block-tridiagonal boundary value problem, Legendre-space source table) written for a run-time matrix size
// n = NSTOKES * N (row r = NSTOKES * stream + Stokes component) and one GPU thread per task, with every work matrix in a
// thread-strided HBM scratch ([element][thread], so the lanes of a wavefront touch consecutive doubles).  It is the
// device-side specification the matrix-core kernels of the scalar path will be generalised against; the C ABI uses it for
// NSTOKES = 3 radiances (weighting functions of the polarised solver are not built yet).
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
#include "disco_core.hpp"

namespace skv {

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
};

struct VBufs {
    int S, n;               // n = S * N
    const double* leg;      // [w][q][nleg][4]  (a1, a2, a3, b1 as stored by the caller, cpp/include/c_api/atmosphere.h:77-83)
    int nleg;
    double* a4;             // [p][l][4][w]  layer Greek coefficients (b1 with the sign flip of sktran_do_layerarray.cpp:418)
    double* rec;            // [w][p][REC]   REC = 2 n^2 + 8 n : K, E, WP, WM, AP, AM, G[4]
    double* dinv;           // [w][p][4 n^2]
    double* xsol;           // [w][p][2 n]
    double* J;              // [w][slot][m][S]
    double* radiance;       // [w*nlos+los][S]
    double* scratch;        // [element][thread] work matrices of the running kernel
    size_t scratch_stride;  // threads of the running kernel
};

struct VRec {
    int n;
    SKD_HD explicit VRec(int n_) : n(n_) {}
    SKD_HD int K() const { return 0; }
    SKD_HD int E() const { return n; }
    SKD_HD int WP() const { return 2 * n; }
    SKD_HD int WM() const { return 2 * n + n * n; }
    SKD_HD int AP() const { return 2 * n + 2 * n * n; }
    SKD_HD int AM() const { return 3 * n + 2 * n * n; }
    SKD_HD int G() const { return 4 * n + 2 * n * n; }   // 0 G+top, 1 G-top, 2 G+bottom, 3 G-bottom (no D factor)
    SKD_HD int size() const { return 8 * n + 2 * n * n; }
};
SKD_HD double* vrec_at(const VBufs& v, int nlyr, int p, int w) { return v.rec + (size_t(w) * nlyr + p) * VRec(v.n).size(); }
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
    }
}

SKD_HD void load_coef(const Bufs& b, const VBufs& v, int nstr, int p, int l, int w, double* c4) {
    const double* a = v.a4 + (size_t(p) * nstr + l) * 4 * b.nw + w;
    for (int c = 0; c < 4; ++c) c4[c] = a[size_t(c) * b.nw];
}

// =================================================================================================
// KV1: homogeneous + particular solution of one layer for azimuth order m    (task = wavelength x layer)
//   scratch: 8 n^2 + 6 n doubles per thread
// =================================================================================================
SKD_HD size_t kv1_scratch(int n) { return 8 * size_t(n) * n + 6 * size_t(n); }

SKD_HD void kv1_layer_solution(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int p, int w, Scr scr) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr, nw = b.nw;
    const VRec R(n);
    double* rec = vrec_at(v, g.nlyr, p, w);
    const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
    const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    const size_t nn = size_t(n) * n;
    Scr Ep = scr.sub(0), Em = scr.sub(nn), Bp = scr.sub(2 * nn), Bm = scr.sub(3 * nn), Tm = scr.sub(4 * nn), V = scr.sub(5 * nn),
        X = scr.sub(6 * nn), U = scr.sub(7 * nn);
    Scr lam = scr.sub(8 * nn), kk = scr.sub(8 * nn + n), nrm = scr.sub(8 * nn + 2 * n), Qp = scr.sub(8 * nn + 3 * n),
        Qm = scr.sub(8 * nn + 4 * n), tmp = scr.sub(8 * nn + 5 * n);
    auto rw = [&](int r) { return g.wt[r / S]; };
    auto rmu = [&](int r) { return g.mu[r / S]; };
    // ---- E+ = (plain + negated) / 2, E- = (plain - negated) / 2 of the Greek-matrix products, block by block
    for (int e = 0; e < int(nn); ++e) { Ep[e] = 0; Em[e] = 0; }
    for (int l = m; l < nstr; ++l) {
        double c4[4];
        load_coef(b, v, nstr, p, l, w, c4);
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        for (int i = 0; i < N; ++i) {
            const Fun3 fi = fun_mu(g, vg, m, i);
            for (int j = 0; j < N; ++j) {
                const Fun3 fj = fun_mu(g, vg, m, j);
                if (S == 1) {
                    const double t = c4[0] * fi.P[l] * fj.P[l];
                    Ep[size_t(i) * n + j] += 0.5 * (t + sg * t);
                    Em[size_t(i) * n + j] += 0.5 * (t - sg * t);
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
                        const size_t e = size_t(i * 3 + s1) * n + (j * 3 + s2);
                        Ep[e] += 0.5 * (y0[s1] + sg * y1[s1]);
                        Em[e] += 0.5 * (y0[s1] - sg * y1[s1]);
                    }
                }
            }
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            const double dlt = (r == c) ? 1.0 / rmu(r) : 0.0;
            const double sq = sqrt(rw(r) / rmu(r)) * sqrt(rw(c) / rmu(c));
            Bp[size_t(r) * n + c] = dlt - sq * om * Ep[size_t(r) * n + c];
            Bm[size_t(r) * n + c] = dlt - sq * om * Em[size_t(r) * n + c];
        }
    int err = 0;
    // ---- Cholesky B+ = L L^T (lower, in place)
    for (int j = 0; j < n; ++j) {
        double d = Bp[size_t(j) * n + j];
        for (int k = 0; k < j; ++k) d -= Bp[size_t(j) * n + k] * Bp[size_t(j) * n + k];
        if (!(d > 0)) { err |= skd::ERR_CHOLESKY; d = 1.0; }
        d = sqrt(d);
        Bp[size_t(j) * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = Bp[size_t(i) * n + j];
            for (int k = 0; k < j; ++k) t -= Bp[size_t(i) * n + k] * Bp[size_t(j) * n + k];
            Bp[size_t(i) * n + j] = t / d;
        }
        for (int i = 0; i < j; ++i) Bp[size_t(i) * n + j] = 0;
    }
    // ---- As = L^T B- L (into Bm)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = 0;
            for (int r = j; r < n; ++r) t += Bm[size_t(i) * n + r] * Bp[size_t(r) * n + j];
            Tm[size_t(i) * n + j] = t;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0;
            for (int r = i; r < n; ++r) t += Bp[size_t(r) * n + i] * Tm[size_t(r) * n + j];
            Bm[size_t(i) * n + j] = t;
            Bm[size_t(j) * n + i] = t;
        }
    // ---- cyclic Jacobi on As, eigenvectors -> V
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[size_t(i) * n + j] = (i == j) ? 1.0 : 0.0;
    bool converged = false;
    for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        double off = 0, dia = 0;
        for (int i = 0; i < n; ++i) {
            dia += Bm[size_t(i) * n + i] * Bm[size_t(i) * n + i];
            for (int j = i + 1; j < n; ++j) off += Bm[size_t(i) * n + j] * Bm[size_t(i) * n + j];
        }
        if (off <= 1e-30 * dia || off == 0) { converged = true; break; }
        for (int pp = 0; pp < n - 1; ++pp)
            for (int q = pp + 1; q < n; ++q) {
                const double apq = Bm[size_t(pp) * n + q];
                if (apq == 0.0) continue;
                const double app = Bm[size_t(pp) * n + pp], aqq = Bm[size_t(q) * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                Bm[size_t(pp) * n + pp] = app - t * apq;
                Bm[size_t(q) * n + q] = aqq + t * apq;
                Bm[size_t(pp) * n + q] = 0;
                Bm[size_t(q) * n + pp] = 0;
                for (int r = 0; r < n; ++r) {
                    if (r != pp && r != q) {
                        const double arp = Bm[size_t(r) * n + pp], arq = Bm[size_t(r) * n + q];
                        const double np_ = c * arp - sn * arq, nq_ = sn * arp + c * arq;
                        Bm[size_t(r) * n + pp] = np_;
                        Bm[size_t(pp) * n + r] = np_;
                        Bm[size_t(r) * n + q] = nq_;
                        Bm[size_t(q) * n + r] = nq_;
                    }
                    const double vrp = V[size_t(r) * n + pp], vrq = V[size_t(r) * n + q];
                    V[size_t(r) * n + pp] = c * vrp - sn * vrq;
                    V[size_t(r) * n + q] = sn * vrp + c * vrq;
                }
            }
    }
    if (!converged) err |= skd::ERR_EIGEN;
    for (int j = 0; j < n; ++j) {
        double l_ = Bm[size_t(j) * n + j];
        if (!(l_ > 0)) { err |= skd::ERR_EIGEN; l_ = fabs(l_) + 1e-300; }
        lam[j] = l_;
        kk[j] = sqrt(l_);
    }
    // ---- X = T^-1 L^-T V ;  U = S+ X with S+ = M^-1 (I - om E+ W)
    for (int j = 0; j < n; ++j)
        for (int i = n - 1; i >= 0; --i) {
            double t = V[size_t(i) * n + j];
            for (int r = i + 1; r < n; ++r) t -= Bp[size_t(r) * n + i] * X[size_t(r) * n + j];
            X[size_t(i) * n + j] = t / Bp[size_t(i) * n + i];
        }
    for (int i = 0; i < n; ++i) {
        const double tt = sqrt(rw(i) * rmu(i));
        for (int j = 0; j < n; ++j) X[size_t(i) * n + j] /= tt;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = X[size_t(i) * n + j] / rmu(i);
            for (int r = 0; r < n; ++r) t -= om * rw(r) * Ep[size_t(i) * n + r] / rmu(i) * X[size_t(r) * n + j];
            U[size_t(i) * n + j] = t;
        }
    // ---- W+-, norms
    for (int j = 0; j < n; ++j) {
        double nr = 0;
        rec[R.K() + j] = kk[j];
        rec[R.E() + j] = exp(-kk[j] * th);
        for (int i = 0; i < n; ++i) {
            const double uk = U[size_t(i) * n + j] / kk[j];
            const double wp = 0.5 * (X[size_t(i) * n + j] + uk), wm = 0.5 * (X[size_t(i) * n + j] - uk);
            rec[R.WP() + i + n * j] = wp;
            rec[R.WM() + i + n * j] = wm;
            nr += rw(i) * rmu(i) * (wp * wp - wm * wm);
        }
        nrm[j] = nr;
    }
    // ---- solar pseudo-source: first column of the products with the unpolarised sun (single_scat_st, sktran_do_lpproduct.h:339-384)
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    const double* lpc = g.lpcsz + (size_t(s) * nstr + m) * nstr;
    bool allzero = true;
    for (int r = 0; r < n; ++r) { Qp[r] = 0; Qm[r] = 0; }
    for (int l = m; l < nstr; ++l) {
        double c4[4];
        load_coef(b, v, nstr, p, l, w, c4);
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        const double x[3] = {lpc[l], 0, 0};
        double bx[3];
        greek_apply(c4, x, bx);  // B_l D x = B_l x for a pure-intensity vector
        for (int i = 0; i < N; ++i) {
            const Fun3 fi = fun_mu(g, vg, m, i);
            if (S == 1) {
                Qp[i] += bx[0] * fi.P[l];
                Qm[i] += sg * bx[0] * fi.P[l];
            } else {
                double y[3];
                pi_apply(fi.P[l], fi.R[l], fi.T[l], bx, y);
                for (int s1 = 0; s1 < 3; ++s1) {
                    Qp[i * 3 + s1] += y[s1];
                    Qm[i * 3 + s1] += sg * y[s1];
                }
            }
        }
    }
    for (int r = 0; r < n; ++r) {
        Qp[r] *= fm * rw(r) * om;
        Qm[r] *= fm * rw(r) * om;
        if (Qp[r] != 0 || Qm[r] != 0) allzero = false;
    }
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < n; ++i) rec[R.G() + c * n + i] = 0;
    for (int j = 0; j < n; ++j) {
        double Ap = 0, Am = 0;
        if (!allzero) {
            for (int i = 0; i < n; ++i) {
                const double wp = rec[R.WP() + i + n * j], wm = rec[R.WM() + i + n * j];
                Ap += Qp[i] * wp + Qm[i] * wm;
                Am += Qm[i] * wp + Qp[i] * wm;
            }
            Ap /= nrm[j];
            Am /= nrm[j];
            skd::Mult Cp, Cm;
            skd::green_multipliers(T, th, sc, kk[j], Cp, Cm);
            for (int i = 0; i < n; ++i) {
                const double wp = rec[R.WP() + i + n * j], wm = rec[R.WM() + i + n * j];
                rec[R.G() + 0 * n + i] += Am * Cm.v * wm;
                rec[R.G() + 1 * n + i] += Am * Cm.v * wp;
                rec[R.G() + 2 * n + i] += Ap * Cp.v * wp;
                rec[R.G() + 3 * n + i] += Ap * Cp.v * wm;
            }
        }
        rec[R.AP() + j] = Ap;
        rec[R.AM() + j] = Am;
    }
    (void)tmp;
    if (err) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.status[w], err);
#else
        b.status[w] |= err;
#endif
    }
}

// =================================================================================================
// KV2: block-tridiagonal boundary value problem, factor + solve                 (task = wavelength)
//   same block rows as skd::k2a_bvp with 2n x 2n blocks; scratch: 10 n^2 + 8 n doubles per thread
// =================================================================================================
SKD_HD size_t kv2_scratch(int n) { return 10 * size_t(n) * n + 8 * size_t(n); }

SKD_HD void kv2_bvp(const Geo& g, const Bufs& b, const VBufs& v, int s, int m, int w, Scr scr) {
    const int N = g.N, S = v.S, n = v.n, M2 = 2 * n, nw = b.nw, nlyr = g.nlyr;
    const VRec R(n);
    const double rho = (m == 0) ? b.albedo[w] : 0.0;  // Lambertian
    const double kd = 2.0;
    const size_t mm = size_t(M2) * M2;
    Scr D = scr.sub(0), Dinv = scr.sub(mm), T1 = scr.sub(2 * mm), r = scr.sub(2 * mm + size_t(n) * M2), rprev = scr.sub(2 * mm + size_t(n) * M2 + M2),
        E = scr.sub(2 * mm + size_t(n) * M2 + 2 * M2), Eprev = scr.sub(2 * mm + size_t(n) * M2 + 2 * M2 + n), xn = scr.sub(2 * mm + size_t(n) * M2 + 2 * M2 + 2 * n);
    for (int p = 0; p < nlyr; ++p) {
        const double* c = vrec_at(v, nlyr, p, w);
        for (int j = 0; j < n; ++j) E[j] = c[R.E() + j];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double wp = c[R.WP() + i + n * j], wm = c[R.WM() + i + n * j];
                D[size_t(i) * M2 + j] = -wp;
                D[size_t(i) * M2 + n + j] = -wm * E[j];
                D[size_t(n + i) * M2 + j] = wm * E[j];
                D[size_t(n + i) * M2 + n + j] = wp;
            }
        for (int i = 0; i < n; ++i) {
            r[i] = c[R.G() + 0 * n + i];
            r[n + i] = -c[R.G() + 3 * n + i];
        }
        if (p == nlyr - 1) {
            if (m == 0) {
                // Lambertian reflection: intensity rows only, fed by the intensity rows of the downwelling field
                for (int j = 0; j < n; ++j) {
                    double sp = 0, sm = 0;
                    for (int t = 0; t < N; ++t) {
                        sp += g.wt[t] * g.mu[t] * c[R.WP() + t * S + n * j];
                        sm += g.wt[t] * g.mu[t] * c[R.WM() + t * S + n * j];
                    }
                    for (int i = 0; i < N; ++i) {
                        D[size_t(n + i * S) * M2 + j] -= kd * rho * sp * E[j];
                        D[size_t(n + i * S) * M2 + n + j] -= kd * rho * sm;
                    }
                }
                double sg = 0;
                for (int t = 0; t < N; ++t) sg += g.wt[t] * g.mu[t] * c[R.G() + 2 * n + t * S];
                const double direct = g.csz[s] * rho / kPi * b.T[size_t(nlyr) * nw + w];
                for (int i = 0; i < N; ++i) r[n + i * S] += direct + kd * rho * sg;
            }
        } else {
            const double* cn = vrec_at(v, nlyr, p + 1, w);
            for (int i = 0; i < n; ++i) r[n + i] += cn[R.G() + 1 * n + i];
        }
        if (p > 0) {
            const double* cp = vrec_at(v, nlyr, p - 1, w);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < M2; ++j) {
                    double t = 0;
                    for (int q = 0; q < n; ++q)
                        t += cp[R.WP() + i + n * q] * Eprev[q] * Dinv[size_t(q) * M2 + j] + cp[R.WM() + i + n * q] * Dinv[size_t(n + q) * M2 + j];
                    T1[size_t(i) * M2 + j] = t;
                }
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < n; ++j) {
                    double t1 = 0, t2 = 0;
                    for (int q = 0; q < n; ++q) {
                        t1 += T1[size_t(i) * M2 + n + q] * c[R.WM() + q + n * j];
                        t2 += T1[size_t(i) * M2 + n + q] * c[R.WP() + q + n * j];
                    }
                    D[size_t(i) * M2 + j] += t1;
                    D[size_t(i) * M2 + n + j] += t2 * E[j];
                }
                double t = 0;
                for (int q = 0; q < M2; ++q) t += T1[size_t(i) * M2 + q] * rprev[q];
                r[i] -= cp[R.G() + 2 * n + i] + t;
            }
        }
        // Gauss-Jordan inverse with partial pivoting
        bool singular = false;
        for (int i = 0; i < M2; ++i)
            for (int j = 0; j < M2; ++j) Dinv[size_t(i) * M2 + j] = (i == j) ? 1.0 : 0.0;
        for (int cc = 0; cc < M2; ++cc) {
            int piv = cc;
            double best = fabs(D[size_t(cc) * M2 + cc]);
            for (int rr = cc + 1; rr < M2; ++rr)
                if (fabs(D[size_t(rr) * M2 + cc]) > best) { best = fabs(D[size_t(rr) * M2 + cc]); piv = rr; }
            if (!(best > 0)) { singular = true; break; }
            if (piv != cc)
                for (int j = 0; j < M2; ++j) {
                    double t = D[size_t(cc) * M2 + j]; D[size_t(cc) * M2 + j] = D[size_t(piv) * M2 + j]; D[size_t(piv) * M2 + j] = t;
                    t = Dinv[size_t(cc) * M2 + j]; Dinv[size_t(cc) * M2 + j] = Dinv[size_t(piv) * M2 + j]; Dinv[size_t(piv) * M2 + j] = t;
                }
            const double inv = 1.0 / D[size_t(cc) * M2 + cc];
            for (int j = 0; j < M2; ++j) { D[size_t(cc) * M2 + j] *= inv; Dinv[size_t(cc) * M2 + j] *= inv; }
            for (int rr = 0; rr < M2; ++rr) {
                if (rr == cc) continue;
                const double f = D[size_t(rr) * M2 + cc];
                if (f == 0.0) continue;
                for (int j = 0; j < M2; ++j) {
                    D[size_t(rr) * M2 + j] -= f * D[size_t(cc) * M2 + j];
                    Dinv[size_t(rr) * M2 + j] -= f * Dinv[size_t(cc) * M2 + j];
                }
            }
        }
        if (singular) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicOr(&b.status[w], skd::ERR_SINGULAR_BVP);
#else
            b.status[w] |= skd::ERR_SINGULAR_BVP;
#endif
        }
        double* dst = vdinv_at(v, nlyr, p, w);
        double* xs = vxsol_at(v, nlyr, p, w);
        for (int i = 0; i < M2; ++i) {
            rprev[i] = r[i];
            xs[i] = r[i];
            for (int j = 0; j < M2; ++j) dst[size_t(i) * M2 + j] = Dinv[size_t(i) * M2 + j];
        }
        for (int j = 0; j < n; ++j) Eprev[j] = E[j];
    }
    // back substitution:  x_p = Dinv_p (r'_p - Sup_p x_{p+1})
    for (int p = nlyr - 1; p >= 0; --p) {
        double* xs = vxsol_at(v, nlyr, p, w);
        const double* dinv = vdinv_at(v, nlyr, p, w);
        for (int i = 0; i < M2; ++i) r[i] = xs[i];
        if (p < nlyr - 1) {
            const double* cn = vrec_at(v, nlyr, p + 1, w);
            for (int i = 0; i < n; ++i) {
                double t = 0;
                for (int j = 0; j < n; ++j) t += cn[R.WM() + i + n * j] * xn[j] + cn[R.WP() + i + n * j] * cn[R.E() + j] * xn[n + j];
                r[n + i] += t;
            }
        }
        for (int i = 0; i < M2; ++i) {
            double t = 0;
            for (int j = 0; j < M2; ++j) t += dinv[size_t(i) * M2 + j] * r[j];
            D[i] = t;
        }
        for (int i = 0; i < M2; ++i) {
            xn[i] = D[i];
            xs[i] = D[i];
        }
    }
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
    if (p == nlyr - 1 && m == 0) {
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
            if (skd::conv_at(b, g, slot, w) > 0) behind[0] += g.g_w[t] * vJ_at(v, g, 0, slot, w, 0);  // order 0 only
        }
    }
    for (int q = g.ray_qstart[los]; q < g.ray_qstart[los + 1]; ++q) {
        const double att = exp(-b.od[size_t(q) * nw + w]);
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
        for (int c = 0; c < S; ++c) behind[c] = behind[c] * att + (g.rl_active[q] ? omega * (1 - att) * sv[c] : 0.0);
    }
    for (int c = 0; c < S; ++c) v.radiance[tid * S + c] = behind[c];
}

// =================================================================================================
// KVPP: plane-parallel line of sight for azimuth order m, 3 Stokes components      (task = wavelength x LOS)
//   reference: OpticalLayer::integrate_source sktran_do_opticallayer.cpp:93-555, computeReflectedIntensities
//   sktran_do_layerarray.cpp:5-288, DOSourcePlaneParallelPostProcessing::calculate do_source_planeparallel.cpp:60-218
//   scratch: 7 n doubles per thread
// =================================================================================================
SKD_HD size_t kvpp_scratch(int n) { return 7 * size_t(n); }

SKD_HD void kvpp_los(const Geo& g, const VGeo& vg, const Bufs& b, const VBufs& v, int s, int m, int los, int w, Scr scr) {
    const int N = g.N, S = v.S, n = v.n, nstr = g.nstr, nw = b.nw, nlyr = g.nlyr;
    const VRec R(n);
    const size_t tid = size_t(w) * g.nlos + los;
    const double mu = g.pp_mu[los];
    const Fun3 fl = fun_los(g, vg, m, los);
    const double* lpc = g.lpcsz + (size_t(s) * nstr + m) * nstr;
    const int first = g.pp_first[los];
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    Scr yp = scr.sub(0), ym = scr.sub(3 * size_t(n)), lrow = scr.sub(6 * size_t(n));  // yp/ym: [S][n]
    double att = 1.0;  // attenuation to the observer
    double total[3] = {0, 0, 0};
    for (int p = first; p < nlyr; ++p) {
        const double lf = (p == first) ? g.pp_lf[los] : 1.0;
        const double* c = vrec_at(v, nlyr, p, w);
        const double* xs = vxsol_at(v, nlyr, p, w);
        const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
        const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
        // Y+-[s][a] = sum_r lp[s][r] W+-(r,a) + lm[s][r] W-+(r,a);  lm = (om w_j / 2) [Pi(out) B Pi(mu_j)], lp the negated product
        for (int e = 0; e < S * n; ++e) { yp[e] = 0; ym[e] = 0; }
        double Q[3] = {0, 0, 0};
        for (int l = m; l < nstr; ++l) {
            double c4[4];
            load_coef(b, v, nstr, p, l, w, c4);
            const double sg = ((l - m) & 1) ? -1.0 : 1.0;
            for (int s1 = 0; s1 < S; ++s1) {
                // row s1 of Pi_l(out) B_l  (B_l symmetric: the row is B_l applied to row s1 of Pi_l(out))
                double ex[3] = {0, 0, 0}, prow[3], brow[3];
                ex[s1] = 1;
                pi_apply(fl.P[l], fl.R[l], fl.T[l], ex, prow);  // Pi symmetric: row s1 = column s1
                greek_apply(c4, prow, brow);
                // lm row over r = (j, s2):  (brow . Pi_l(mu_j)[:, s2]) ;  lp row: ((brow D) . Pi_l(mu_j)[:, s2]) (-1)^(l-m)
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
                    double tp = 0, tm = 0;
                    for (int r = 0; r < n; ++r) {
                        const double wp = c[R.WP() + r + n * a], wm = c[R.WM() + r + n * a];
                        tp += lrow[n + r] * wp + lrow[r] * wm;
                        tm += lrow[n + r] * wm + lrow[r] * wp;
                    }
                    yp[size_t(s1) * n + a] += tp;
                    ym[size_t(s1) * n + a] += tm;
                }
                // solar single-scatter term: first column of the negated product with the sun direction
                if (g.include_ss) Q[s1] += sg * brow[0] * lpc[l] * fm * om;
            }
        }
        skd::PPBeam Eb;
        skd::pp_beam_multiplier(mu, lf, th, T, sc, Eb);
        double Ssrc[3] = {Q[0] * Eb.v, Q[1] * Eb.v, Q[2] * Eb.v};
        for (int a = 0; a < n; ++a) {
            skd::PPLosMult pm;
            skd::pp_los_multipliers(mu, lf, th, T, sc, c[R.K() + a], Eb, pm);
            const double L = xs[a], Mv = xs[n + a], Ap = c[R.AP() + a], Am = c[R.AM() + a];
            for (int s1 = 0; s1 < S; ++s1)
                Ssrc[s1] += yp[size_t(s1) * n + a] * (pm.hp * L + Ap * pm.Dm) + ym[size_t(s1) * n + a] * (pm.hm * Mv + Am * pm.Dp);
        }
        for (int s1 = 0; s1 < S; ++s1) total[s1] += att * Ssrc[s1];
        att *= exp(-th * lf / mu);
    }
    if (m == 0) {
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
        total[0] += att * rho * (2.0 * ipb + direct);
    }
    const double cz = cos(m * g.pp_azi[los]), sz = sin(m * g.pp_azi[los]);
    for (int s1 = 0; s1 < S; ++s1) v.radiance[tid * S + s1] += total[s1] * (s1 < 2 ? cz : sz);
}

}  // namespace skv
