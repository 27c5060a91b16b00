// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into the product library.
//
// Dense linear algebra the reference's DISCO path takes from third-party code that is
// not vendored under /root/reference:
//   * real non-symmetric eigen-decomposition: Eigen's RealSchur/EigenSolver behind
//     cpp/include/sasktran2/math/real_eigensolver.h:70 (call site sktran_do_rte.cpp:437-489),
//     or LAPACK dgeev (sktran_do_rte.cpp:503-511).  Restated here as the classical
//     Householder-Hessenberg + Francis double-shift QR (EISPACK orthes/hqr2).
//   * general band LU with partial pivoting + triangular solves: LAPACK dgbsv/dgbtrs
//     (call sites sktran_do_rte.cpp:1676-1682,1755-1760; unblocked variant
//     cpp/lib/sktran_disco/sktran_do_banded_lu.cpp:8-121).  Restated as dgbtf2/dgbtrs.
//   * small dense LU with partial pivoting: Eigen::PartialPivLU (sktran_do_rte.cpp:255-256), and with
//     complete pivoting: Eigen::FullPivLU (sktran_do_rte.cpp:258-263, NSTOKES = 3).
#pragma once
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <vector>

namespace oracle_la {

// ----------------------------------------------------------------------------------------
// Real eigen-decomposition of a general n x n matrix (row-major input a[i*n+j]).
// Outputs wr/wi (eigenvalues) and V (row-major; column j = "pseudo eigenvector" j, real
// eigenvalues give real eigenvectors).  Returns false when QR does not converge.
// ----------------------------------------------------------------------------------------
class RealEigen {
    int n = 0;
    std::vector<double> H, V, ort, d, e;
    double& h(int i, int j) { return H[size_t(i) * n + j]; }
    double& v(int i, int j) { return V[size_t(i) * n + j]; }

    void orthes() {
        int low = 0, high = n - 1;
        for (int m = low + 1; m <= high - 1; m++) {
            double scale = 0.0;
            for (int i = m; i <= high; i++) scale += std::abs(h(i, m - 1));
            if (scale != 0.0) {
                double hh = 0.0;
                for (int i = high; i >= m; i--) {
                    ort[i] = h(i, m - 1) / scale;
                    hh += ort[i] * ort[i];
                }
                double g = std::sqrt(hh);
                if (ort[m] > 0) g = -g;
                hh -= ort[m] * g;
                ort[m] -= g;
                for (int j = m; j < n; j++) {
                    double f = 0.0;
                    for (int i = high; i >= m; i--) f += ort[i] * h(i, j);
                    f /= hh;
                    for (int i = m; i <= high; i++) h(i, j) -= f * ort[i];
                }
                for (int i = 0; i <= high; i++) {
                    double f = 0.0;
                    for (int j = high; j >= m; j--) f += ort[j] * h(i, j);
                    f /= hh;
                    for (int j = m; j <= high; j++) h(i, j) -= f * ort[j];
                }
                ort[m] = scale * ort[m];
                h(m, m - 1) = scale * g;
            }
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) v(i, j) = (i == j ? 1.0 : 0.0);
        for (int m = high - 1; m >= low + 1; m--) {
            if (h(m, m - 1) != 0.0) {
                for (int i = m + 1; i <= high; i++) ort[i] = h(i, m - 1);
                for (int j = m; j <= high; j++) {
                    double g = 0.0;
                    for (int i = m; i <= high; i++) g += ort[i] * v(i, j);
                    g = (g / ort[m]) / h(m, m - 1);
                    for (int i = m; i <= high; i++) v(i, j) += g * ort[i];
                }
            }
        }
    }

    static void cdiv(double xr, double xi, double yr, double yi, double& cr, double& ci) {
        double r, dd;
        if (std::abs(yr) > std::abs(yi)) {
            r = yi / yr;
            dd = yr + r * yi;
            cr = (xr + r * xi) / dd;
            ci = (xi - r * xr) / dd;
        } else {
            r = yr / yi;
            dd = yi + r * yr;
            cr = (r * xr + xi) / dd;
            ci = (r * xi - xr) / dd;
        }
    }

    bool hqr2() {
        int nn = n;
        int nidx = nn - 1;
        int low = 0, high = nn - 1;
        double eps = std::pow(2.0, -52.0);
        double exshift = 0.0;
        double p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
        double norm = 0.0;
        for (int i = 0; i < nn; i++) {
            if (i < low || i > high) { d[i] = h(i, i); e[i] = 0.0; }
            for (int j = std::max(i - 1, 0); j < nn; j++) norm += std::abs(h(i, j));
        }
        int iter = 0, total_iter = 0;
        while (nidx >= low) {
            int l = nidx;
            while (l > low) {
                s = std::abs(h(l - 1, l - 1)) + std::abs(h(l, l));
                if (s == 0.0) s = norm;
                if (std::abs(h(l, l - 1)) < eps * s) break;
                l--;
            }
            if (l == nidx) {
                h(nidx, nidx) += exshift;
                d[nidx] = h(nidx, nidx);
                e[nidx] = 0.0;
                nidx--;
                iter = 0;
            } else if (l == nidx - 1) {
                w = h(nidx, nidx - 1) * h(nidx - 1, nidx);
                p = (h(nidx - 1, nidx - 1) - h(nidx, nidx)) / 2.0;
                q = p * p + w;
                z = std::sqrt(std::abs(q));
                h(nidx, nidx) += exshift;
                h(nidx - 1, nidx - 1) += exshift;
                x = h(nidx, nidx);
                if (q >= 0) {
                    z = (p >= 0) ? p + z : p - z;
                    d[nidx - 1] = x + z;
                    d[nidx] = d[nidx - 1];
                    if (z != 0.0) d[nidx] = x - w / z;
                    e[nidx - 1] = 0.0;
                    e[nidx] = 0.0;
                    x = h(nidx, nidx - 1);
                    s = std::abs(x) + std::abs(z);
                    p = x / s;
                    q = z / s;
                    r = std::sqrt(p * p + q * q);
                    p /= r;
                    q /= r;
                    for (int j = nidx - 1; j < nn; j++) {
                        z = h(nidx - 1, j);
                        h(nidx - 1, j) = q * z + p * h(nidx, j);
                        h(nidx, j) = q * h(nidx, j) - p * z;
                    }
                    for (int i = 0; i <= nidx; i++) {
                        z = h(i, nidx - 1);
                        h(i, nidx - 1) = q * z + p * h(i, nidx);
                        h(i, nidx) = q * h(i, nidx) - p * z;
                    }
                    for (int i = low; i <= high; i++) {
                        z = v(i, nidx - 1);
                        v(i, nidx - 1) = q * z + p * v(i, nidx);
                        v(i, nidx) = q * v(i, nidx) - p * z;
                    }
                } else {
                    d[nidx - 1] = x + p;
                    d[nidx] = x + p;
                    e[nidx - 1] = z;
                    e[nidx] = -z;
                }
                nidx -= 2;
                iter = 0;
            } else {
                x = h(nidx, nidx);
                y = 0.0;
                w = 0.0;
                if (l < nidx) {
                    y = h(nidx - 1, nidx - 1);
                    w = h(nidx, nidx - 1) * h(nidx - 1, nidx);
                }
                if (iter == 10) {
                    exshift += x;
                    for (int i = low; i <= nidx; i++) h(i, i) -= x;
                    s = std::abs(h(nidx, nidx - 1)) + std::abs(h(nidx - 1, nidx - 2));
                    x = y = 0.75 * s;
                    w = -0.4375 * s * s;
                }
                if (iter == 30) {
                    s = (y - x) / 2.0;
                    s = s * s + w;
                    if (s > 0) {
                        s = std::sqrt(s);
                        if (y < x) s = -s;
                        s = x - w / ((y - x) / 2.0 + s);
                        for (int i = low; i <= nidx; i++) h(i, i) -= s;
                        exshift += s;
                        x = y = w = 0.964;
                    }
                }
                iter++;
                if (++total_iter > 60 * nn) return false;
                int m = nidx - 2;
                while (m >= l) {
                    z = h(m, m);
                    r = x - z;
                    s = y - z;
                    p = (r * s - w) / h(m + 1, m) + h(m, m + 1);
                    q = h(m + 1, m + 1) - z - r - s;
                    r = h(m + 2, m + 1);
                    s = std::abs(p) + std::abs(q) + std::abs(r);
                    p /= s;
                    q /= s;
                    r /= s;
                    if (m == l) break;
                    if (std::abs(h(m, m - 1)) * (std::abs(q) + std::abs(r)) <
                        eps * (std::abs(p) * (std::abs(h(m - 1, m - 1)) + std::abs(z) + std::abs(h(m + 1, m + 1)))))
                        break;
                    m--;
                }
                for (int i = m + 2; i <= nidx; i++) {
                    h(i, i - 2) = 0.0;
                    if (i > m + 2) h(i, i - 3) = 0.0;
                }
                for (int k = m; k <= nidx - 1; k++) {
                    bool notlast = (k != nidx - 1);
                    if (k != m) {
                        p = h(k, k - 1);
                        q = h(k + 1, k - 1);
                        r = (notlast ? h(k + 2, k - 1) : 0.0);
                        x = std::abs(p) + std::abs(q) + std::abs(r);
                        if (x == 0.0) continue;
                        p /= x;
                        q /= x;
                        r /= x;
                    }
                    s = std::sqrt(p * p + q * q + r * r);
                    if (p < 0) s = -s;
                    if (s != 0) {
                        if (k != m) h(k, k - 1) = -s * x;
                        else if (l != m) h(k, k - 1) = -h(k, k - 1);
                        p += s;
                        x = p / s;
                        y = q / s;
                        z = r / s;
                        q /= p;
                        r /= p;
                        for (int j = k; j < nn; j++) {
                            p = h(k, j) + q * h(k + 1, j);
                            if (notlast) {
                                p += r * h(k + 2, j);
                                h(k + 2, j) -= p * z;
                            }
                            h(k, j) -= p * x;
                            h(k + 1, j) -= p * y;
                        }
                        for (int i = 0; i <= std::min(nidx, k + 3); i++) {
                            p = x * h(i, k) + y * h(i, k + 1);
                            if (notlast) {
                                p += z * h(i, k + 2);
                                h(i, k + 2) -= p * r;
                            }
                            h(i, k) -= p;
                            h(i, k + 1) -= p * q;
                        }
                        for (int i = low; i <= high; i++) {
                            p = x * v(i, k) + y * v(i, k + 1);
                            if (notlast) {
                                p += z * v(i, k + 2);
                                v(i, k + 2) -= p * r;
                            }
                            v(i, k) -= p;
                            v(i, k + 1) -= p * q;
                        }
                    }
                }
            }
        }
        if (norm == 0.0) return true;
        // back-substitute for the vectors of the upper triangular form
        for (nidx = nn - 1; nidx >= 0; nidx--) {
            p = d[nidx];
            q = e[nidx];
            if (q == 0) {
                int l = nidx;
                h(nidx, nidx) = 1.0;
                for (int i = nidx - 1; i >= 0; i--) {
                    w = h(i, i) - p;
                    r = 0.0;
                    for (int j = l; j <= nidx; j++) r += h(i, j) * h(j, nidx);
                    if (e[i] < 0.0) {
                        z = w;
                        s = r;
                    } else {
                        l = i;
                        if (e[i] == 0.0) {
                            if (w != 0.0) h(i, nidx) = -r / w;
                            else h(i, nidx) = -r / (eps * norm);
                        } else {
                            x = h(i, i + 1);
                            y = h(i + 1, i);
                            q = (d[i] - p) * (d[i] - p) + e[i] * e[i];
                            t = (x * s - z * r) / q;
                            h(i, nidx) = t;
                            if (std::abs(x) > std::abs(z)) h(i + 1, nidx) = (-r - w * t) / x;
                            else h(i + 1, nidx) = (-s - y * t) / z;
                        }
                        t = std::abs(h(i, nidx));
                        if ((eps * t) * t > 1)
                            for (int j = i; j <= nidx; j++) h(j, nidx) /= t;
                    }
                }
            } else if (q < 0) {
                int l = nidx - 1;
                double cr, ci;
                if (std::abs(h(nidx, nidx - 1)) > std::abs(h(nidx - 1, nidx))) {
                    h(nidx - 1, nidx - 1) = q / h(nidx, nidx - 1);
                    h(nidx - 1, nidx) = -(h(nidx, nidx) - p) / h(nidx, nidx - 1);
                } else {
                    cdiv(0.0, -h(nidx - 1, nidx), h(nidx - 1, nidx - 1) - p, q, cr, ci);
                    h(nidx - 1, nidx - 1) = cr;
                    h(nidx - 1, nidx) = ci;
                }
                h(nidx, nidx - 1) = 0.0;
                h(nidx, nidx) = 1.0;
                for (int i = nidx - 2; i >= 0; i--) {
                    double ra = 0.0, sa = 0.0, vr, vi;
                    for (int j = l; j <= nidx; j++) {
                        ra += h(i, j) * h(j, nidx - 1);
                        sa += h(i, j) * h(j, nidx);
                    }
                    w = h(i, i) - p;
                    if (e[i] < 0.0) {
                        z = w;
                        r = ra;
                        s = sa;
                    } else {
                        l = i;
                        if (e[i] == 0) {
                            cdiv(-ra, -sa, w, q, cr, ci);
                            h(i, nidx - 1) = cr;
                            h(i, nidx) = ci;
                        } else {
                            x = h(i, i + 1);
                            y = h(i + 1, i);
                            vr = (d[i] - p) * (d[i] - p) + e[i] * e[i] - q * q;
                            vi = (d[i] - p) * 2.0 * q;
                            if (vr == 0.0 && vi == 0.0)
                                vr = eps * norm * (std::abs(w) + std::abs(q) + std::abs(x) + std::abs(y) + std::abs(z));
                            cdiv(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, cr, ci);
                            h(i, nidx - 1) = cr;
                            h(i, nidx) = ci;
                            if (std::abs(x) > (std::abs(z) + std::abs(q))) {
                                h(i + 1, nidx - 1) = (-ra - w * h(i, nidx - 1) + q * h(i, nidx)) / x;
                                h(i + 1, nidx) = (-sa - w * h(i, nidx) - q * h(i, nidx - 1)) / x;
                            } else {
                                cdiv(-r - y * h(i, nidx - 1), -s - y * h(i, nidx), z, q, cr, ci);
                                h(i + 1, nidx - 1) = cr;
                                h(i + 1, nidx) = ci;
                            }
                        }
                        t = std::max(std::abs(h(i, nidx - 1)), std::abs(h(i, nidx)));
                        if ((eps * t) * t > 1)
                            for (int j = i; j <= nidx; j++) {
                                h(j, nidx - 1) /= t;
                                h(j, nidx) /= t;
                            }
                    }
                }
            }
        }
        for (int i = 0; i < nn; i++)
            if (i < low || i > high)
                for (int j = i; j < nn; j++) v(i, j) = h(i, j);
        for (int j = nn - 1; j >= low; j--) {
            for (int i = low; i <= high; i++) {
                z = 0.0;
                for (int k = low; k <= std::min(j, high); k++) z += v(i, k) * h(k, j);
                v(i, j) = z;
            }
        }
        return true;
    }

  public:
    // a: row-major n x n.  wr, wi: n.  vec: row-major n x n.
    bool compute(int n_, const double* a, double* wr, double* wi, double* vec) {
        n = n_;
        H.assign(a, a + size_t(n) * n);
        V.assign(size_t(n) * n, 0.0);
        ort.assign(n, 0.0);
        d.assign(n, 0.0);
        e.assign(n, 0.0);
        orthes();
        if (!hqr2()) return false;
        for (int i = 0; i < n; ++i) {
            wr[i] = d[i];
            wi[i] = e[i];
        }
        std::copy(V.begin(), V.end(), vec);
        return true;
    }
};

// ----------------------------------------------------------------------------------------
// Dense LU with partial pivoting (row-major, in place).  Returns false when singular.
// ----------------------------------------------------------------------------------------
inline bool lu_factor(int n, double* a, int* piv) {
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::abs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::abs(a[i * n + k]) > best) { best = std::abs(a[i * n + k]); p = i; }
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
        if (a[k * n + k] == 0.0) return false;
        double inv = 1.0 / a[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            double l = a[i * n + k] * inv;
            a[i * n + k] = l;
            if (l != 0.0)
                for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
        }
    }
    return true;
}
inline void lu_solve(int n, const double* a, const int* piv, double* b) {
    // rows of L were interchanged retroactively during factorization (LAPACK dgetrf convention),
    // so all interchanges are applied to b before the unit-lower solve
    for (int k = 0; k < n; ++k)
        if (piv[k] != k) std::swap(b[k], b[piv[k]]);
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i < n; ++i) b[i] -= a[i * n + k] * b[k];
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= a[k * n + j] * b[j];
        b[k] = s / a[k * n + k];
    }
}

// ----------------------------------------------------------------------------------------
// Dense LU with complete pivoting and rank detection: Eigen::FullPivLU, which the reference uses for the bordered
// eigen-derivative systems of the polarised solver (sktran_do_rte.cpp:258-263).  As in Eigen, pivots below
// eps * n * |largest pivot| count as zero and the corresponding unknowns of a solve are set to zero.
// ----------------------------------------------------------------------------------------
struct FullPivLU {
    int n = 0, rank = 0;
    std::vector<double> lu;
    std::vector<int> rowperm, colperm;  // row k of LU is row rowperm[k] of A, column k is column colperm[k] of A
    void compute(int n_, const double* a) {
        n = n_;
        lu.assign(a, a + size_t(n) * n);
        rowperm.resize(n);
        colperm.resize(n);
        for (int i = 0; i < n; ++i) rowperm[i] = colperm[i] = i;
        double maxpivot = 0;
        std::vector<double> piv(n, 0.0);
        for (int k = 0; k < n; ++k) {
            int pr = k, pc = k;
            double best = 0;
            for (int i = k; i < n; ++i)
                for (int j = k; j < n; ++j)
                    if (std::abs(lu[size_t(i) * n + j]) > best) { best = std::abs(lu[size_t(i) * n + j]); pr = i; pc = j; }
            if (best == 0.0) break;
            maxpivot = std::max(maxpivot, best);
            if (pr != k) {
                for (int j = 0; j < n; ++j) std::swap(lu[size_t(k) * n + j], lu[size_t(pr) * n + j]);
                std::swap(rowperm[k], rowperm[pr]);
            }
            if (pc != k) {
                for (int i = 0; i < n; ++i) std::swap(lu[size_t(i) * n + k], lu[size_t(i) * n + pc]);
                std::swap(colperm[k], colperm[pc]);
            }
            piv[k] = lu[size_t(k) * n + k];
            const double inv = 1.0 / piv[k];
            for (int i = k + 1; i < n; ++i) {
                double l = lu[size_t(i) * n + k] * inv;
                lu[size_t(i) * n + k] = l;
                if (l != 0.0)
                    for (int j = k + 1; j < n; ++j) lu[size_t(i) * n + j] -= l * lu[size_t(k) * n + j];
            }
        }
        const double thresh = 2.220446049250313e-16 * n * maxpivot;
        rank = 0;
        for (int k = 0; k < n; ++k)
            if (std::abs(piv[k]) > thresh) ++rank;
    }
    void solve(double* b) const {
        std::vector<double> c(n);
        for (int k = 0; k < n; ++k) c[k] = b[rowperm[k]];
        for (int k = 0; k < n; ++k)
            for (int i = k + 1; i < n; ++i) c[i] -= lu[size_t(i) * n + k] * c[k];
        for (int k = rank - 1; k >= 0; --k) {
            double s = c[k];
            for (int j = k + 1; j < rank; ++j) s -= lu[size_t(k) * n + j] * c[j];
            c[k] = s / lu[size_t(k) * n + k];
        }
        for (int k = 0; k < n; ++k) b[colperm[k]] = k < rank ? c[k] : 0.0;
    }
};

// ----------------------------------------------------------------------------------------
// General band matrix, LAPACK column-major band storage: A(i,j) -> ab[(kl+ku+i-j) + j*ldab],
// ldab = 2*kl+ku+1.  dgbtf2-style factorization and dgbtrs-style solve for RHS stored
// row-major as b[row*nrhs + col] (inner loop over right-hand sides).
// ----------------------------------------------------------------------------------------
struct BandMatrix {
    int n = 0, kl = 0, ku = 0, ldab = 0;
    std::vector<double> ab;
    std::vector<int> ipiv;
    void resize(int n_, int kl_, int ku_) {
        n = n_; kl = kl_; ku = ku_; ldab = 2 * kl + ku + 1;
        ab.assign(size_t(ldab) * n, 0.0);
        ipiv.assign(n, 0);
    }
    void set_zero() { std::fill(ab.begin(), ab.end(), 0.0); }
    double& at(int i, int j) { return ab[size_t(kl + ku + i - j) + size_t(j) * ldab]; }

    // returns 0 on success, j+1 if U(j,j) is exactly zero
    int factor() {
        const int kv = ku + kl;
        int info = 0, ju = 0;
        for (int j = 0; j < n; ++j) {
            int km = std::min(kl, n - 1 - j);
            int jp = 0;
            double best = std::abs(at(j, j));
            for (int i = 1; i <= km; ++i)
                if (std::abs(at(j + i, j)) > best) { best = std::abs(at(j + i, j)); jp = i; }
            ipiv[j] = j + jp;
            if (at(j + jp, j) != 0.0) {
                ju = std::max(ju, std::min(j + ku + jp, n - 1));
                if (jp != 0)
                    for (int c = j; c <= ju; ++c) std::swap(at(j, c), at(j + jp, c));
                if (km > 0) {
                    double inv = 1.0 / at(j, j);
                    for (int i = 1; i <= km; ++i) at(j + i, j) *= inv;
                    for (int c = j + 1; c <= ju; ++c) {
                        double u = at(j, c);
                        if (u != 0.0)
                            for (int i = 1; i <= km; ++i) at(j + i, c) -= at(j + i, j) * u;
                    }
                }
            } else if (info == 0) {
                info = j + 1;
            }
        }
        (void)kv;
        return info;
    }

    void solve(double* b, int nrhs) const {
        const int kv = ku + kl;
        auto A = [&](int i, int j) { return ab[size_t(kl + ku + i - j) + size_t(j) * ldab]; };
        for (int j = 0; j < n - 1; ++j) {
            int lm = std::min(kl, n - 1 - j);
            int l = ipiv[j];
            double* bj = b + size_t(j) * nrhs;
            if (l != j) {
                double* bl = b + size_t(l) * nrhs;
                for (int c = 0; c < nrhs; ++c) std::swap(bj[c], bl[c]);
            }
            for (int i = 1; i <= lm; ++i) {
                double m = A(j + i, j);
                if (m == 0.0) continue;
                double* bi = b + size_t(j + i) * nrhs;
                for (int c = 0; c < nrhs; ++c) bi[c] -= m * bj[c];
            }
        }
        for (int j = n - 1; j >= 0; --j) {
            double* bj = b + size_t(j) * nrhs;
            double inv = 1.0 / A(j, j);
            for (int c = 0; c < nrhs; ++c) bj[c] *= inv;
            int i0 = std::max(0, j - kv);
            for (int i = i0; i < j; ++i) {
                double u = A(i, j);
                if (u == 0.0) continue;
                double* bi = b + size_t(i) * nrhs;
                for (int c = 0; c < nrhs; ++c) bi[c] -= u * bj[c];
            }
        }
    }
};

}  // namespace oracle_la
