// Two-stream source (multiple_scatter_source = twostream, emission_source = twostream): the per-wavelength arithmetic.
//
// Every function here is `TS_HD` (host + device): the HIP kernels of ts_engine.hip call them with one thread per wavelength (lane =
// wavelength, scratch laid out [item][lane] so that every access of a wavefront is one contiguous 512-byte row), and
// tests/emulate_twostream.cpp runs the same bodies on the CPU.
//
// What is computed (reference: cpp/lib/sktran_disco/cpp_twostream_source.cpp, behaviour restated):
//   solve_thread   layer optics from the level profiles, slant solar transmission / thermal profile, the two-stream homogeneous and
//                  particular solutions of every layer for azimuth orders 0 and 1, and the LU sweep of the five-diagonal boundary-value
//                  problem (:1923-2122, :1861-1899)
//   view_forward_thread   the radiance of one line of sight: plane parallel (:2550-2668) or the spherical path integration over traced
//                         segments (:1196-1233, :2160-2234); with weighting functions also dF/dc and the forward half of the transposed solve
//   view_adjoint_thread   the weighting functions of that line of sight.
// Weighting functions are NOT a translation of the reference's hand-written reverse sweep.  The radiance is F(p, c) with A(p) c = r(p)
// (p: the five numbers per layer -- optical depth, single-scatter albedo, first phase moment, source amplitude and decay rate -- c: the
// 2n homogeneous amplitudes).  One transposed five-diagonal solve gives lambda = A^-T dF/dc, and the gradient with respect to p is the
// p-derivative of the Lagrangian F(p, c) + lambda . (r(p) - A(p) c) at fixed (c, lambda), evaluated layer by layer in forward-mode dual
// numbers with five directions (D5).  The chain from p to the level profiles is written out by hand (:2839-2921 gives the same result).
#pragma once
#include <cmath>
#include <cstddef>
#include <type_traits>

#if defined(__HIPCC__)
#define TS_HD __host__ __device__ inline
#define TS_HD_COLD __host__ __device__ __attribute__((noinline))   // rarely taken series branches: kept out of the callers' register budget
#else
#define TS_HD inline
#define TS_HD_COLD inline
#endif

namespace skt {

constexpr double kPi = 3.14159265358979323846;
constexpr double kFourPi = 4.0 * kPi;
constexpr double kThermalMinOd = 1.0e-10;        // THERMAL_MIN_OPTICAL_DEPTH, cpp_twostream_source.cpp:21
constexpr double kThermalMinEmission = 1.0e-30;  // THERMAL_MIN_EMISSION, :22
constexpr double kSsaCap = 1.0 - 1.0e-9;         // clamp of the layer single-scatter albedo, :1944-1946

struct Geo {
    int nlyr, ngrid, nlos, ncol, spherical, max_seg;
    double mu;
    const double *thickness, *csz, *chapman;         // [nlyr], [ncol], [ncol][boundary][layer]
    const double *chapman_t, *chap_uniform;          // [ncol][layer][boundary]; [ncol] the one factor of a plane-parallel column (layers above the
                                                     // boundary), 0 for a ray-traced one
    const double *view_cos, *view_azimuth;           // plane parallel, [nlos]
    const int *ray_start, *ground_hit, *ground_col;  // spherical
    const int* view_cols;                            // [nlos][2] first / last column a line of sight refers to
    const double* ground_colw;
    const int* seg_layer;
    const double *seg_frac, *seg_cos, *seg_azimuth;
    const int* seg_col;
    const double* seg_colw;
    const int *od_start, *od_level;
    const double* od_weight;
};

struct Bufs {
    int nw, W, w0, nw_total, nleg, derivs, accumulate;
    // inputs in the reference storage layout (location fastest); delta_f, irradiance, emission, surface_emission may be null
    const double *ext, *ssa, *leg, *delta_f, *albedo, *irradiance, *emission, *surface_emission;
    // outputs: radiance [nw_total][nlos]; wf_ext / wf_ssa / wf_b1 / wf_em [ngrid][nw_total][nlos]; wf_alb / wf_sem [nw_total][nlos]
    double *radiance, *wf_ext, *wf_ssa, *wf_b1, *wf_alb, *wf_em, *wf_sem;
    double* los_od;   // [nw_total][nlos] optical depth along the line of sight, or null
    // scratch, every item a row of W lanes
    double* lay;    // [3][n]                   od, ssa, b1 of the layers
    double* lev;    // [4][n + 1]               extinction, ssa, b1, emission of the levels (top first): the strided inputs are read once
    double* colv;   // [ncol][2n + 1]           source amplitude at the n + 1 boundaries (solar: transmission) and decay rate of the n layers
    double* lu;     // [ncol][naz][5][2n]       c (solution), alpha, beta, gamma, 1 / mu of the LU sweep
    double* lam;    // [nlos][ncol][naz][2n]    dF/dc, then lambda
    double* g5;     // [nlos][ncol][n][5]       gradient with respect to the layer parameters
    double* gsum;   // [nlos][5][n]             od / ssa / b1 gradients summed over the columns; thermal: amplitude / rate gradients
    double* vstate; // [nlos][2]                weight of the surface term in the radiance, optical depth along the line of sight
    double* path;   // [nlos][3][max_seg]       spherical: transmission, source and incoming radiance of every segment
    double* glev;   // [nlos][ngrid]            spherical: level-extinction gradient of the path transmissions
};

// ---------------------------------------------------------------------------------------------------------------------------
// dual numbers: value and the derivatives with respect to (od, ssa, b1, amplitude, rate) of one layer
// ---------------------------------------------------------------------------------------------------------------------------
struct D5 {
    double v;
    double d[5];
};
TS_HD D5 mkd(double v) { return {v, {0, 0, 0, 0, 0}}; }
TS_HD D5 seed(double v, int dir) {
    D5 r = mkd(v);
    r.d[dir] = 1.0;
    return r;
}
TS_HD D5 operator+(const D5& a, const D5& b) { D5 r; r.v = a.v + b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
TS_HD D5 operator-(const D5& a, const D5& b) { D5 r; r.v = a.v - b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
TS_HD D5 operator*(const D5& a, const D5& b) { D5 r; r.v = a.v * b.v; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
TS_HD D5 operator/(const D5& a, const D5& b) {
    D5 r;
    const double ib = 1.0 / b.v;
    r.v = a.v * ib;
    for (int i = 0; i < 5; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
TS_HD D5 operator-(const D5& a) { D5 r; r.v = -a.v; for (int i = 0; i < 5; ++i) r.d[i] = -a.d[i]; return r; }
TS_HD D5 operator+(const D5& a, double b) { D5 r = a; r.v += b; return r; }
TS_HD D5 operator+(double a, const D5& b) { D5 r = b; r.v += a; return r; }
TS_HD D5 operator-(const D5& a, double b) { D5 r = a; r.v -= b; return r; }
TS_HD D5 operator-(double a, const D5& b) { D5 r; r.v = a - b.v; for (int i = 0; i < 5; ++i) r.d[i] = -b.d[i]; return r; }
TS_HD D5 operator*(const D5& a, double b) { D5 r; r.v = a.v * b; for (int i = 0; i < 5; ++i) r.d[i] = a.d[i] * b; return r; }
TS_HD D5 operator*(double a, const D5& b) { return b * a; }
TS_HD D5 operator/(const D5& a, double b) { return a * (1.0 / b); }
TS_HD D5 operator/(double a, const D5& b) {
    D5 r;
    const double ib = 1.0 / b.v;
    r.v = a * ib;
    for (int i = 0; i < 5; ++i) r.d[i] = -r.v * b.d[i] * ib;
    return r;
}
TS_HD D5& operator+=(D5& a, const D5& b) { a = a + b; return a; }

TS_HD double val(double x) { return x; }
TS_HD double val(const D5& x) { return x.v; }
TS_HD double dpart(double, int) { return 0.0; }
TS_HD double dpart(const D5& x, int i) { return x.d[i]; }
TS_HD double t_exp(double x) { return exp(x); }
TS_HD D5 t_exp(const D5& x) { D5 r; r.v = exp(x.v); for (int i = 0; i < 5; ++i) r.d[i] = r.v * x.d[i]; return r; }
TS_HD double t_sqrt(double x) { return sqrt(x); }
TS_HD D5 t_sqrt(const D5& x) { D5 r; r.v = sqrt(x.v); const double h = 0.5 / r.v; for (int i = 0; i < 5; ++i) r.d[i] = h * x.d[i]; return r; }
template <class S> TS_HD S cst(double v);
template <> TS_HD double cst<double>(double v) { return v; }
template <> TS_HD D5 cst<D5>(double v) { return mkd(v); }
// a function of three arguments with known partial derivatives
TS_HD double fn3(double v, double, double, double, double, double, double) { return v; }
TS_HD D5 fn3(double v, const D5& a, double da, const D5& b, double db, const D5& c, double dc) {
    D5 r;
    r.v = v;
    for (int i = 0; i < 5; ++i) r.d[i] = da * a.d[i] + db * b.d[i] + dc * c.d[i];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// removable singularities (cpp_twostream_source.cpp:30-128)
// ---------------------------------------------------------------------------------------------------------------------------
struct Stable { double value, da, db, dt; };

// integral of x^order exp(-rate x) over [0, thickness]
TS_HD double exp_moment(int order, double rate, double thickness) {
    if (thickness == 0.0) return 0.0;
    const double x = rate * thickness;
    double unit;
    if (fabs(x) < 0.5) {
        double term = 1.0;
        unit = 0.0;
        for (int j = 0; j < 40; ++j) {
            const double c = term / double(order + j + 1);
            unit += c;
            if (fabs(c) <= 2.220446049250313e-16 * fmax(fabs(unit), 1.0)) break;
            term *= -x / double(j + 1);
        }
    } else {
        const double e = exp(-x);
        unit = -expm1(-x) / x;
        for (int j = 1; j <= order; ++j) unit = (double(j) * unit - e) / x;
    }
    double p = thickness;
    for (int j = 0; j < order; ++j) p *= thickness;
    return p * unit;
}
// (exp(-a t) - exp(-b t)) / (b - a)
TS_HD_COLD Stable exp_difference(double a, double b, double t) {
    const double delta = b - a;
    if (fabs(delta * t) > 1.0e-5) {
        const double ea = exp(-a * t), eb = exp(-b * t), num = ea - eb, inv = 1.0 / delta, inv2 = inv * inv;
        return {num * inv, (-t * ea * delta + num) * inv2, (t * eb * delta - num) * inv2, (-a * ea + b * eb) * inv};
    }
    const double mid = 0.5 * (a + b), half = 0.5 * delta, u = half * t, u2 = u * u;
    const double sinhc = 1.0 + u2 * (1.0 / 6.0 + u2 * (1.0 / 120.0 + u2 / 5040.0));
    const double dsinhc = u * (1.0 / 3.0 + u2 * (1.0 / 30.0 + u2 / 840.0));
    const double em = exp(-mid * t), value = t * em * sinhc;
    const double dmid = -t * value, dhalf = t * t * em * dsinhc;
    return {value, 0.5 * (dmid - dhalf), 0.5 * (dmid + dhalf), em * ((1.0 - mid * t) * sinhc + t * half * dsinhc)};
}
// integral over [0, t] of exp_difference(a, b, x)
TS_HD_COLD Stable integrated_exp_difference(double a, double b, double t) {
    const double delta = b - a;
    if (fabs(delta * t) > 1.0e-4) {
        const Stable ga = exp_difference(0.0, a, t), gb = exp_difference(0.0, b, t);
        const double num = ga.value - gb.value, inv = 1.0 / delta, inv2 = inv * inv;
        return {num * inv, (ga.db * delta + num) * inv2, (-gb.db * delta - num) * inv2, exp_difference(a, b, t).value};
    }
    const double mid = 0.5 * (a + b), half = 0.5 * delta, h2 = half * half;
    const double m1 = exp_moment(1, mid, t), m2 = exp_moment(2, mid, t), m3 = exp_moment(3, mid, t), m4 = exp_moment(4, mid, t);
    const double dmid = -m2 - h2 * m4 / 6.0, dhalf = half * m3 / 3.0;
    return {m1 + h2 * m3 / 6.0, 0.5 * (dmid - dhalf), 0.5 * (dmid + dhalf), exp_difference(a, b, t).value};
}
template <class S> TS_HD S stable_exp_difference(const S& a, const S& b, const S& t) {
    const Stable s = exp_difference(val(a), val(b), val(t));
    return fn3(s.value, a, s.da, b, s.db, t, s.dt);
}
template <class S> TS_HD S stable_integrated_exp_difference(const S& a, const S& b, const S& t) {
    const Stable s = integrated_exp_difference(val(a), val(b), val(t));
    return fn3(s.value, a, s.da, b, s.db, t, s.dt);
}
// (exp_a - exp_b) / (b - a) given exp_a = exp(-a t), exp_b = exp(-b t), with the limit b -> a (:551-581)
template <class S> TS_HD S exp_difference_ratio(const S& exp_a, const S& exp_b, const S& a, const S& b, const S& t) {
    if (fabs((val(b) - val(a)) * val(t)) <= 1.0e-5) return stable_exp_difference(a, b, t);
    return (exp_a - exp_b) / (b - a);
}

// ---------------------------------------------------------------------------------------------------------------------------
// one layer, one azimuth order
// ---------------------------------------------------------------------------------------------------------------------------
template <class S> struct LayerP { S od, ssa, b1, amp, rate; };   // amp / rate: solar transmission at the layer top and slant secant,
                                                                  // or thermal emission at the layer top and its decay rate
TS_HD LayerP<double> make_p(double od, double ssa, double b1, double amp, double rate, double) { return {od, ssa, b1, amp, rate}; }
TS_HD LayerP<D5> make_p(double od, double ssa, double b1, double amp, double rate, const D5&) {
    return {seed(od, 0), seed(ssa, 1), seed(b1, 2), seed(amp, 3), seed(rate, 4)};
}

template <class S> struct LayerSol {
    S k, xp, xm, omega;   // eigenvalue, the two components of the eigenvector, exp(-k od)
    S upper, lower;       // particular-solution amplitudes: solar (a+, a-), thermal (a_t, a_t)
    S expo;               // exp(-rate od)
};

// homogeneous and particular solution of a layer (forward_explicit_layers, :2048-2096)
template <class S, bool SOLAR> TS_HD LayerSol<S> layer_solution(int az, const LayerP<S>& p, double mu, double csz, double angular) {
    LayerSol<S> h;
    S d, s;
    if (az == 0) {
        d = p.ssa * p.b1 * mu - 1.0 / mu;
        s = (p.ssa - 1.0) / mu;
    } else {
        d = cst<S>(-1.0 / mu);
        s = (p.ssa * p.b1 * (1.0 - mu * mu) - 2.0) / (2.0 * mu);
    }
    h.k = t_sqrt(s * d);
    const S s_over_k = s / h.k;
    h.xp = 0.5 * (1.0 - s_over_k);
    h.xm = 0.5 * (1.0 + s_over_k);
    h.omega = t_exp(-(h.k * p.od));
    h.expo = t_exp(-(p.rate * p.od));
    const S norm = mu * (h.xp * h.xp - h.xm * h.xm);
    if (SOLAR) {
        S qp, qm;
        if (az == 0) {
            qp = p.ssa * (1.0 + p.b1 * (csz * mu)) / kFourPi;
            qm = p.ssa * (1.0 - p.b1 * (csz * mu)) / kFourPi;
        } else {
            qp = p.ssa * p.b1 * (angular / kFourPi);
            qm = qp;
        }
        h.upper = (qp * h.xp + qm * h.xm) / norm;
        h.lower = (qm * h.xp + qp * h.xm) / norm;
    } else {
        h.upper = (1.0 - p.ssa) * (h.xp + h.xm) / norm;
        h.lower = h.upper;
    }
    return h;
}

// The four particular-solution values at the layer boundaries (:2097-2118): g+ / g- at the top (t) and bottom (b)
template <class S> struct Boundary { S gpt, gpb, gmt, gmb; };
template <class S> TS_HD Boundary<S> boundary_particular(const LayerP<S>& p, const LayerSol<S>& h) {
    const S cp = p.amp * exp_difference_ratio(h.omega, h.expo, h.k, p.rate, p.od);
    const S cm = p.amp * exp_difference_ratio(cst<S>(1.0), h.omega * h.expo, cst<S>(0.0), p.rate + h.k, p.od);
    Boundary<S> g;
    g.gpt = h.lower * cm * h.xm;
    g.gpb = h.upper * cp * h.xp;
    g.gmt = h.lower * cm * h.xp;
    g.gmb = h.upper * cp * h.xm;
    return g;
}

// phase-function sums of the two streams toward a propagation cosine (explicit_lpsum, :2145-2158)
template <class S, bool SOLAR> TS_HD void lpsum(int az, double phase_mu, double phase_sine, const LayerP<S>& p, S& lp, S& lm) {
    if (az == 0) {
        lp = 0.5 * p.ssa * (1.0 - p.b1 * phase_mu);
        lm = 0.5 * p.ssa * (1.0 + p.b1 * phase_mu);
    } else if (SOLAR) {
        lp = p.ssa * p.b1 * phase_sine;
        lm = lp;
    } else {
        lp = cst<S>(0.0);
        lm = lp;
    }
}

// layer-integrated source toward a plane-parallel viewing cosine, one azimuth order: c0 U + c1 V + P (explicit_plane_view, :2598-2640).
// `beam` = exp(-od / vc), `source_integral` the attenuated integral of the driving term (see plane_source_integral)
template <class S, bool SOLAR>
TS_HD void plane_source_terms(int az, const LayerP<S>& p, const LayerSol<S>& h, double vc, double mu, bool fast, const S& beam,
                              const S& source_integral, S& U, S& V, S& P) {
    const double inv = 1.0 / vc, phase_mu = vc * mu;
    const double phase_sine = 0.25 * sqrt(fmax(0.0, (1.0 - vc * vc) * (1.0 - mu * mu)));
    S lp, lm;
    lpsum<S, SOLAR>(az, phase_mu, phase_sine, p, lp, lm);
    const S yp = lp * h.xp + lm * h.xm, ym = lp * h.xm + lm * h.xp;
    S hm, hp, dp_ratio, dm_ratio;
    if (fast) {
        hm = (h.omega - beam) / (1.0 - h.k * vc);
        hp = (1.0 - h.omega * beam) / (1.0 + h.k * vc);
        dp_ratio = (source_integral - h.expo * hm) / (p.rate + h.k);
        dm_ratio = (hp - source_integral) / (p.rate - h.k);
    } else {
        const S tinv = cst<S>(inv), zero = cst<S>(0.0);
        hm = inv * stable_exp_difference(h.k, tinv, p.od);
        hp = inv * stable_exp_difference(zero, h.k + tinv, p.od);
        dp_ratio = inv * stable_integrated_exp_difference(p.rate + h.k, p.rate + tinv, p.od);
        dm_ratio = inv * stable_integrated_exp_difference(h.k + tinv, p.rate + tinv, p.od);
    }
    U = yp * hp;
    V = ym * hm;
    P = h.upper * yp * (p.amp * dm_ratio) + h.lower * ym * (p.amp * dp_ratio);
}
template <class S> TS_HD S plane_source_integral(const LayerP<S>& p, const S& expo, const S& beam, double vc, bool fast) {
    if (fast) return (1.0 - expo * beam) / (1.0 + p.rate * vc);
    return (1.0 / vc) * stable_exp_difference(cst<S>(0.0), p.rate + cst<S>(1.0 / vc), p.od);
}
// no exponent difference of this layer near a removable singularity (plane_source_nonresonant, :1843-1859)
TS_HD bool plane_nonresonant(double rate, double k, double od, double inv) {
    const double diff[5] = {rate + inv, inv - k, k + inv, rate + k, rate - k};
    for (int i = 0; i < 5; ++i)
        if (fabs(diff[i] * od) <= 1.0e-5) return false;
    return true;
}

// source function inside a layer, a fraction of its optical depth below its top, one azimuth order: c0 U + c1 V + P (:2160-2234)
template <class S, bool SOLAR>
TS_HD void local_source_terms(int az, const LayerP<S>& p, const LayerSol<S>& h, double frac_top, double cosine, double mu, S& U, S& V, S& P) {
    const double phase_mu = cosine * mu;
    const double phase_sine = 0.25 * sqrt(fmax(0.0, (1.0 - cosine * cosine) * (1.0 - mu * mu)));
    const double frac_bottom = 1.0 - frac_top;
    S lp, lm;
    lpsum<S, SOLAR>(az, phase_mu, phase_sine, p, lp, lm);
    const S yp = lp * h.xp + lm * h.xm, ym = lp * h.xm + lm * h.xp;
    const S top_e = t_exp(-(h.k * p.od * frac_top)), bottom_e = t_exp(-(h.k * p.od * frac_bottom));
    const S beam = t_exp(-(p.rate * p.od * frac_top));
    const S plus = p.amp * exp_difference_ratio(top_e, beam, h.k, p.rate, p.od * frac_top);
    const S minus = p.amp * frac_bottom *
                    exp_difference_ratio(beam, h.expo * bottom_e, p.rate * frac_top, p.rate + h.k * frac_bottom, p.od);
    U = yp * top_e;
    V = ym * bottom_e;
    P = h.upper * yp * plus + h.lower * ym * minus;
}
// the emission itself seen along the path (thermal only): amp exp(-rate od frac) (1 - ssa)
template <class S> TS_HD S local_emission(const LayerP<S>& p, double frac_top) {
    return p.amp * t_exp(-(p.rate * p.od * frac_top)) * (1.0 - p.ssa);
}

// ---------------------------------------------------------------------------------------------------------------------------
// scratch addressing (every item a row of b.W lanes)
// ---------------------------------------------------------------------------------------------------------------------------
TS_HD int naz_of(bool solar) { return solar ? 2 : 1; }
TS_HD double& lay_at(const Bufs& b, const Geo& g, int which, int l, int w) { return b.lay[(size_t(which) * g.nlyr + l) * b.W + w]; }
TS_HD double& lev_at(const Bufs& b, const Geo& g, int which, int t, int w) { return b.lev[(size_t(which) * g.ngrid + t) * b.W + w]; }
TS_HD double& amp_at(const Bufs& b, const Geo& g, int col, int boundary, int w) { return b.colv[(size_t(col) * (2 * g.nlyr + 1) + boundary) * b.W + w]; }
TS_HD double& rate_at(const Bufs& b, const Geo& g, int col, int l, int w) { return b.colv[(size_t(col) * (2 * g.nlyr + 1) + g.nlyr + 1 + l) * b.W + w]; }
enum { LU_C = 0, LU_ALPHA, LU_BETA, LU_GAMMA, LU_IMU };
TS_HD double& lu_at(const Bufs& b, const Geo& g, int naz, int col, int az, int which, int row, int w) {
    return b.lu[(((size_t(col) * naz + az) * 5 + which) * (2 * g.nlyr) + row) * b.W + w];
}
TS_HD double& lam_at(const Bufs& b, const Geo& g, int naz, int view, int col, int az, int row, int w) {
    return b.lam[(((size_t(view) * g.ncol + col) * naz + az) * (2 * g.nlyr) + row) * b.W + w];
}
TS_HD double& g5_at(const Bufs& b, const Geo& g, int view, int col, int l, int dir, int w) {
    return b.g5[(((size_t(view) * g.ncol + col) * g.nlyr + l) * 5 + dir) * b.W + w];
}
TS_HD double& gsum_at(const Bufs& b, const Geo& g, int view, int which, int l, int w) {
    return b.gsum[((size_t(view) * 5 + which) * g.nlyr + l) * b.W + w];
}
TS_HD double& path_at(const Bufs& b, const Geo& g, int view, int which, int seg, int w) {
    return b.path[((size_t(view) * 3 + which) * g.max_seg + seg) * b.W + w];
}
TS_HD double& vstate_at(const Bufs& b, const Geo& g, int view, int which, int w) { return b.vstate[(size_t(view) * 2 + which) * b.W + w]; }
TS_HD double& glev_at(const Bufs& b, const Geo& g, int view, int level, int w) { return b.glev[(size_t(view) * g.ngrid + level) * b.W + w]; }

// level values of wavelength gw, levels counted from the top (load_extinction ... load_emission, :863-917)
struct LevelIn { double ext, ssa, b1, em; };
TS_HD LevelIn load_level(const Geo& g, const Bufs& b, size_t gw, int t) {
    const size_t o = gw * g.ngrid + (g.ngrid - 1 - t);
    LevelIn v;
    v.ext = b.ext[o];
    v.ssa = b.ssa[o];
    const double f = b.delta_f ? b.delta_f[o] : 0.0;
    v.b1 = (b.nleg > 1 ? b.leg[o * b.nleg + 1] : 0.0) - 3.0 * f / (1.0 - f);
    v.em = b.emission ? b.emission[o] : 0.0;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// solve: layer optics, source amplitudes and the LU sweep of the boundary-value problem of every column
// `lds`: n values per thread (layer optical depths), element i of this thread at lds[i * ls]
// ---------------------------------------------------------------------------------------------------------------------------
// FUSED (plane-parallel geometry with ONE line of sight): the back substitution is left to the upward pass (view_adjoint_thread /
// fused_radiance_thread with FUSED) and, with weighting functions, the rows of dF/dc of that line of sight are formed here, top-down, and
// taken through the forward half of the transposed solve as the LU sweep produces its rows -- the solution c, alpha and beta are then
// never read by a later kernel.
template <bool SOLAR, bool FUSED = false> TS_HD void solve_thread(const Geo& g, const Bufs& b, int w, double* lds, int ls) {
    const int n = g.nlyr, naz = naz_of(SOLAR), N = 2 * n;
    const size_t gw = size_t(b.w0) + w;
    const double mu = g.mu;
    const double albedo = b.albedo[gw];
    const double irradiance = SOLAR ? (b.irradiance ? b.irradiance[gw] : 1.0) : 0.0;
    const double surface_emission = (!SOLAR && b.surface_emission) ? b.surface_emission[gw] : 0.0;

    // layer optics (prepare_explicit_column, :1926-1955)
    LevelIn top = load_level(g, b, gw, 0);
    auto keep_level = [&](int t, const LevelIn& v) {
        if (!b.derivs) return;
        lev_at(b, g, 0, t, w) = v.ext;
        lev_at(b, g, 1, t, w) = v.ssa;
        lev_at(b, g, 2, t, w) = v.b1;
        if (!SOLAR) lev_at(b, g, 3, t, w) = v.em;
    };
    keep_level(0, top);
    for (int l = 0; l < n; ++l) {
        const LevelIn bot = load_level(g, b, gw, l + 1);
        keep_level(l + 1, bot);
        const double scat_top = top.ext * top.ssa, scat_bot = bot.ext * bot.ssa;
        const double mean_ext = 0.5 * (top.ext + bot.ext), mean_scat = 0.5 * (scat_top + scat_bot);
        const double od = mean_ext * g.thickness[l];
        const double ssa = fmin(mean_ext > 0.0 ? mean_scat / mean_ext : 0.0, kSsaCap);
        const double b1 = mean_scat > 0.0 ? 0.5 * (scat_top * top.b1 + scat_bot * bot.b1) / mean_scat : 0.0;
        lds[l * ls] = od;
        lay_at(b, g, 0, l, w) = od;
        lay_at(b, g, 1, l, w) = ssa;
        lay_at(b, g, 2, l, w) = b1;
        if (!SOLAR) {
            double slope = 0.0;
            if (od > kThermalMinOd && top.em > kThermalMinEmission && bot.em > kThermalMinEmission) slope = log1p((top.em - bot.em) / bot.em) / od;
            amp_at(b, g, 0, l, w) = top.em;
            rate_at(b, g, 0, l, w) = slope;
        }
        top = bot;
    }
    const int ncol = SOLAR ? g.ncol : 1;
    for (int col = 0; col < ncol; ++col) {
        const double csz = g.csz[col];
        if (SOLAR) {
            // slant transmission to every boundary and the mean secant of every layer (:1956-1973)
            amp_at(b, g, col, 0, w) = irradiance;
            double att_prev = 0.0;
            const double* chap = g.chapman + size_t(col) * n * n;
            const double uniform = g.chap_uniform[col];
            double column_od = 0.0;
            for (int bd = 0; bd < n; ++bd) {
                double slant = 0.0;
                if (uniform != 0.0) {
                    // plane parallel: every layer above the boundary with the same factor (a running sum instead of the table row)
                    column_od += lds[bd * ls];
                    slant = column_od * uniform;
                } else {
                    // (zero factors contribute nothing: the row is used as it is, four entries per scalar load)
                    int l = 0;
                    for (; l + 4 <= n; l += 4) {
                        const double f0 = chap[size_t(bd) * n + l], f1 = chap[size_t(bd) * n + l + 1], f2 = chap[size_t(bd) * n + l + 2],
                                     f3 = chap[size_t(bd) * n + l + 3];
                        slant += lds[l * ls] * f0 + lds[(l + 1) * ls] * f1 + lds[(l + 2) * ls] * f2 + lds[(l + 3) * ls] * f3;
                    }
                    for (; l < n; ++l) slant += lds[l * ls] * chap[size_t(bd) * n + l];
                }
                amp_at(b, g, col, bd + 1, w) = exp(-slant) * irradiance;
                const double od = lds[bd * ls];
                rate_at(b, g, col, bd, w) = od > 0.0 ? (att_prev + slant) / od : 0.0;
                att_prev = -slant;
            }
        }
        const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
        const bool rows = FUSED && b.derivs;
        const double vc = FUSED ? g.view_cos[0] : 1.0, vinv = 1.0 / vc;
        for (int az = 0; az < naz; ++az) {
            // rows of the five-diagonal system layer by layer, eliminated as they appear (:1976-2028, :1861-1888)
            const double delta = az == 0 ? 1.0 : 0.0;
            const double azw = FUSED ? (az == 0 ? 1.0 : cos(g.view_azimuth[0])) : 0.0;
            double attenuation = 1.0, path_od = 0.0, ym1 = 0.0, ym2 = 0.0, pend = 0.0;   // FUSED: y of rows i - 1, i - 2; dF/dc of row 2l - 1
            double a1 = 0, a2 = 0, b1r = 0, b2r = 0, z1 = 0, z2 = 0;   // alpha, beta, z of rows i - 1 and i - 2
            double pxp = 0, pxm = 0, pom = 0, pgpb = 0, pgmb = 0;      // previous layer
            int row = 0;
            auto eliminate = [&](double e, double c, double d, double a, double bb, double rhs) {
                const double gamma = c - a2 * e;
                const double imu = 1.0 / (d - b2r * e - a1 * gamma);
                const double alpha = (a - b1r * gamma) * imu;
                const double beta = bb * imu;
                const double z = (rhs - z2 * e - z1 * gamma) * imu;
                lu_at(b, g, naz, col, az, LU_C, row, w) = z;
                lu_at(b, g, naz, col, az, LU_ALPHA, row, w) = alpha;
                lu_at(b, g, naz, col, az, LU_BETA, row, w) = beta;
                if (b.derivs) {   // (the second sub-diagonal e is xp omega of a layer: the adjoint recomputes it)
                    lu_at(b, g, naz, col, az, LU_GAMMA, row, w) = gamma;
                    lu_at(b, g, naz, col, az, LU_IMU, row, w) = imu;
                }
                a2 = a1; a1 = alpha; b2r = b1r; b1r = beta; z2 = z1; z1 = z;
                ++row;
            };
            // (the layer parameters are fetched one layer ahead: the stores of the elimination would otherwise hold the next loads back)
            auto fetch = [&](int l) { return LayerP<double>{lds[l * ls], lay_at(b, g, 1, l, w), lay_at(b, g, 2, l, w), amp_at(b, g, col, l, w), rate_at(b, g, col, l, w)}; };
            LayerP<double> next = fetch(0);
            for (int l = 0; l < n; ++l) {
                const LayerP<double> p = next;
                if (l + 1 < n) next = fetch(l + 1);
                const LayerSol<double> h = layer_solution<double, SOLAR>(az, p, mu, csz, angular);
                const Boundary<double> q = boundary_particular(p, h);
                double beta_two_back = 0.0;
                if (rows && l > 0) {
                    // U^T y = dF/dc: row 2l - 1 needs alpha of row 2l - 2 and beta of row 2l - 3, the last rows eliminated so far
                    const double y = pend - a1 * ym1 - b2r * ym2;
                    lam_at(b, g, naz, 0, 0, az, 2 * l - 1, w) = y;
                    ym2 = ym1;
                    ym1 = y;
                    beta_two_back = b1r;   // beta of row 2l - 2
                }
                if (l == 0) {
                    eliminate(0.0, 0.0, h.xp, h.xm * h.omega, 0.0, -q.gpt);
                } else {
                    eliminate(0.0, pxm * pom, pxp, -h.xm, -h.xp * h.omega, q.gmt - pgmb);
                    eliminate(pxp * pom, pxm, -h.xp, -h.xm * h.omega, 0.0, q.gpt - pgpb);
                }
                pxp = h.xp; pxm = h.xm; pom = h.omega; pgpb = q.gpb; pgmb = q.gmb;
                if (FUSED) {
                    const double beam = exp(-p.od * vinv);
                    if (rows) {
                        // dF/dc of the two amplitudes of this layer (explicit_plane_view, :2598-2640) and the surface term under the last layer
                        const bool fast = plane_nonresonant(p.rate, h.k, p.od, vinv);
                        const double source_integral = plane_source_integral(p, h.expo, beam, vc, fast);
                        double U, V, P;
                        plane_source_terms<double, SOLAR>(az, p, h, vc, mu, fast, beam, source_integral, U, V, P);
                        double r0 = attenuation * azw * U, r1 = attenuation * azw * V;
                        if (l == n - 1 && az == 0) {
                            const double wsurf = attenuation * beam * (2.0 * mu) * albedo;
                            r0 += wsurf * h.xp * h.omega;
                            r1 += wsurf * h.xm;
                        }
                        // row 2l: alpha of row 2l - 1 (the row before the last one eliminated) and beta of row 2l - 2
                        const double y = l == 0 ? r0 : r0 - a2 * ym1 - beta_two_back * ym2;
                        lam_at(b, g, naz, 0, 0, az, 2 * l, w) = y;
                        ym2 = ym1;
                        ym1 = y;
                        pend = r1;
                    }
                    path_od += p.od * vinv;
                    attenuation *= beam;
                }
            }
            if (rows) lam_at(b, g, naz, 0, 0, az, N - 1, w) = pend - a1 * ym1 - b2r * ym2;   // row 2n - 1: alpha of 2n - 2, beta of 2n - 3
            if (FUSED && az == 0) {
                vstate_at(b, g, 0, 0, w) = attenuation;   // weight of the surface term in the radiance
                vstate_at(b, g, 0, 1, w) = path_od;
            }
            const double direct = SOLAR ? delta * csz * albedo / kPi * amp_at(b, g, col, n, w) : surface_emission;
            eliminate(0.0, (pxm - 2.0 * mu * albedo * delta * pxp) * pom, pxp - 2.0 * mu * albedo * delta * pxm, 0.0, 0.0,
                      direct - (pgmb - 2.0 * delta * mu * albedo * pgpb));
            if (FUSED) continue;   // the upward pass substitutes back as it goes
            // back substitution (:1889-1898)
            // (four rows per round trip: the loads of a block are issued together, ahead of the stores of the block before)
            double x1 = 0, x2 = 0;
            for (int top = N - 1; top >= 0; top -= 4) {
                double z[4], al[4], be[4];
                for (int j = 0; j < 4; ++j) {
                    const int i = top - j;
                    z[j] = i >= 0 ? lu_at(b, g, naz, col, az, LU_C, i, w) : 0.0;
                    al[j] = i >= 0 ? lu_at(b, g, naz, col, az, LU_ALPHA, i, w) : 0.0;
                    be[j] = i >= 0 ? lu_at(b, g, naz, col, az, LU_BETA, i, w) : 0.0;
                }
                for (int j = 0; j < 4; ++j) {
                    const int i = top - j;
                    if (i < 0) break;
                    const double x = z[j] - al[j] * x1 - be[j] * x2;
                    lu_at(b, g, naz, col, az, LU_C, i, w) = x;
                    x2 = x1;
                    x1 = x;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// view: radiance of one line of sight (view_forward_thread) and its weighting functions (view_adjoint_thread)
// ---------------------------------------------------------------------------------------------------------------------------
template <bool SOLAR> TS_HD LayerP<double> load_p(const Geo& g, const Bufs& b, int col, int l, int w) {
    return {lay_at(b, g, 0, l, w), lay_at(b, g, 1, l, w), lay_at(b, g, 2, l, w), amp_at(b, g, SOLAR ? col : 0, l, w), rate_at(b, g, SOLAR ? col : 0, l, w)};
}
template <class S> TS_HD LayerP<S> lift(const LayerP<double>& p) { return make_p(p.od, p.ssa, p.b1, p.amp, p.rate, S()); }

// upwelling radiance leaving the Lambertian surface divided by (2 mu albedo): the downwelling stream at the bottom of the last layer
// (explicit_surface_source, :2403-2419)
template <class S> TS_HD S bottom_plus(const LayerSol<S>& h, const Boundary<S>& q, double c0, double c1) {
    return q.gpb + c0 * (h.xp * h.omega) + c1 * h.xm;
}

template <bool SOLAR>
TS_HD void segment_adjoint(int az, const LayerP<double>& p, const LayerSol<double>& h, double mu, double csz, double angular, double frac_top,
                           double cosine, double c0, double c1, double Ws, double g[5], double& U, double& V);

// DUAL (tests only): the segment sources of the spherical reverse sweep differentiated in dual numbers instead of segment_adjoint
template <bool SOLAR, bool DERIVS, bool SPHERICAL, bool DUAL = false> TS_HD void view_forward_thread(const Geo& g, const Bufs& b, int w, int view) {
    constexpr int NAZ = SOLAR ? 2 : 1;
    const int n = g.nlyr, N = 2 * n;
    const size_t gw = size_t(b.w0) + w;
    const double mu = g.mu;
    const double albedo = b.albedo[gw];
    const double surface_emission = (!SOLAR && b.surface_emission) ? b.surface_emission[gw] : 0.0;
    double radiance = 0.0, path_od = 0.0, surf_weight = 0.0;

    if (!SPHERICAL) {
        // ---- plane parallel (:2550-2668): layer-integrated sources attenuated to the top of the atmosphere.  With DERIVS the rows of
        // dF/dc appear top-down, which is the order of the forward half of the transposed solve (U^T y = dF/dc): it is done on the fly
        const double vc = g.view_cos[view], inv = 1.0 / vc, csz = g.csz[0];
        const double azw[2] = {1.0, cos(g.view_azimuth[view])};
        const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
        double attenuation = 1.0;
        double y1[NAZ], y2[NAZ], al1[NAZ], be1[NAZ], be2[NAZ];   // y, alpha, beta of the rows i - 1 and i - 2
        for (int az = 0; az < NAZ; ++az) y1[az] = y2[az] = al1[az] = be1[az] = be2[az] = 0.0;
        // everything a layer reads from scratch, fetched one layer ahead of its use
        struct Fetch {
            LayerP<double> p;
            double c0[NAZ], c1[NAZ], al_a[NAZ], be_a[NAZ], al_b[NAZ], be_b[NAZ];
        };
        auto fetch = [&](int l) {
            Fetch f;
            f.p = load_p<SOLAR>(g, b, 0, l, w);
            for (int az = 0; az < NAZ; ++az) {
                f.c0[az] = lu_at(b, g, NAZ, 0, az, LU_C, 2 * l, w);
                f.c1[az] = lu_at(b, g, NAZ, 0, az, LU_C, 2 * l + 1, w);
                if (DERIVS) {
                    f.al_a[az] = lu_at(b, g, NAZ, 0, az, LU_ALPHA, 2 * l, w);
                    f.be_a[az] = lu_at(b, g, NAZ, 0, az, LU_BETA, 2 * l, w);
                    f.al_b[az] = lu_at(b, g, NAZ, 0, az, LU_ALPHA, 2 * l + 1, w);
                    f.be_b[az] = lu_at(b, g, NAZ, 0, az, LU_BETA, 2 * l + 1, w);
                }
            }
            return f;
        };
        Fetch next = fetch(0);
        for (int l = 0; l < n; ++l) {
            const Fetch cur = next;
            if (l + 1 < n) next = fetch(l + 1);
            const LayerP<double>& p = cur.p;
            path_od += p.od * inv;
            const double beam = exp(-p.od * inv);
            double source = 0.0;
            for (int az = 0; az < NAZ; ++az) {
                const LayerSol<double> h = layer_solution<double, SOLAR>(az, p, mu, csz, angular);
                const bool fast = plane_nonresonant(p.rate, h.k, p.od, inv);
                const double source_integral = plane_source_integral(p, h.expo, beam, vc, fast);
                double U, V, P;
                plane_source_terms<double, SOLAR>(az, p, h, vc, mu, fast, beam, source_integral, U, V, P);
                const double c0 = cur.c0[az], c1 = cur.c1[az];
                source += azw[az] * (c0 * U + c1 * V + P);
                if (!SOLAR && az == 0) source += p.amp * source_integral * (1.0 - p.ssa);
                if (DERIVS) {
                    double r0 = attenuation * azw[az] * U, r1 = attenuation * azw[az] * V;
                    if (l == n - 1 && az == 0) {   // the surface term 2 mu albedo (downwelling stream at the ground), attenuated by the whole column
                        const double wsurf = attenuation * beam * (2.0 * mu) * albedo;
                        r0 += wsurf * h.xp * h.omega;
                        r1 += wsurf * h.xm;
                    }
                    const double ya = r0 - al1[az] * y1[az] - be2[az] * y2[az];
                    const double yb = r1 - cur.al_a[az] * ya - be1[az] * y1[az];
                    lam_at(b, g, NAZ, view, 0, az, 2 * l, w) = ya;
                    lam_at(b, g, NAZ, view, 0, az, 2 * l + 1, w) = yb;
                    y2[az] = ya;
                    y1[az] = yb;
                    be2[az] = cur.be_a[az];
                    al1[az] = cur.al_b[az];
                    be1[az] = cur.be_b[az];
                }
                if (l == n - 1 && az == 0) {
                    const Boundary<double> q = boundary_particular(p, h);
                    surf_weight = attenuation * beam;
                    radiance += surf_weight * (bottom_plus(h, q, c0, c1) * (2.0 * mu) * albedo + surface_emission);
                }
            }
            radiance += source * attenuation;
            attenuation *= beam;
        }
    } else {
        // ---- spherical (:1196-1233): R <- R t + S (1 - t) over the traced segments, far end first
        const int ncol = SOLAR ? g.ncol : 1;
        const int s0 = g.ray_start[view], s1 = g.ray_start[view + 1];
        const int ground_lo = SOLAR ? g.ground_col[2 * view] : 0, ground_up = SOLAR ? g.ground_col[2 * view + 1] : 0;
        const double ground_uw = SOLAR ? g.ground_colw[view] : 0.0;
        auto surface_of = [&](int col) {
            const LayerP<double> p = load_p<SOLAR>(g, b, col, n - 1, w);
            const LayerSol<double> h = layer_solution<double, SOLAR>(0, p, mu, g.csz[col], 0.0);
            const Boundary<double> q = boundary_particular(p, h);
            const double base = bottom_plus(h, q, lu_at(b, g, NAZ, col, 0, LU_C, N - 2, w), lu_at(b, g, NAZ, col, 0, LU_C, N - 1, w));
            return base * (2.0 * mu) * albedo + (SOLAR ? 0.0 : surface_emission);
        };
        if (g.ground_hit[view]) {
            radiance = surface_of(ground_lo);
            if (ground_lo != ground_up) radiance = radiance * (1.0 - ground_uw) + surface_of(ground_up) * ground_uw;
        }
        auto segment_source = [&](int s, int col) {
            const int l = g.seg_layer[s];
            const LayerP<double> p = load_p<SOLAR>(g, b, col, l, w);
            const double csz = g.csz[col];
            const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
            const double azw[2] = {1.0, cos(g.seg_azimuth[s])};
            double src = 0.0;
            for (int az = 0; az < NAZ; ++az) {
                const LayerSol<double> h = layer_solution<double, SOLAR>(az, p, mu, csz, angular);
                double U, V, P;
                local_source_terms<double, SOLAR>(az, p, h, g.seg_frac[s], g.seg_cos[s], mu, U, V, P);
                src += azw[az] * (lu_at(b, g, NAZ, col, az, LU_C, 2 * l, w) * U + lu_at(b, g, NAZ, col, az, LU_C, 2 * l + 1, w) * V + P);
            }
            if (!SOLAR) src += local_emission(p, g.seg_frac[s]);
            return src;
        };
        for (int s = s0; s < s1; ++s) {
            double od = 0.0;
            for (int j = g.od_start[s]; j < g.od_start[s + 1]; ++j) od += b.ext[gw * g.ngrid + (g.ngrid - 1 - g.od_level[j])] * g.od_weight[j];
            const double t = exp(-od);
            path_od += od;
            const int lo = SOLAR ? g.seg_col[2 * s] : 0, up = SOLAR ? g.seg_col[2 * s + 1] : 0;
            double src = segment_source(s, lo);
            if (lo != up) {
                const double uw = g.seg_colw[s];
                src = src * (1.0 - uw) + segment_source(s, up) * uw;
            }
            if (DERIVS) {
                path_at(b, g, view, 0, s - s0, w) = t;
                path_at(b, g, view, 1, s - s0, w) = src;
                path_at(b, g, view, 2, s - s0, w) = radiance;
            }
            radiance = radiance * t + src * (1.0 - t);
        }
        const int col_first = SOLAR ? g.view_cols[2 * view] : 0, col_last = SOLAR ? g.view_cols[2 * view + 1] : 0;   // columns this line of sight refers to
        if (DERIVS) {
            for (int col = col_first; col <= col_last; ++col) {
                for (int az = 0; az < NAZ; ++az)
                    for (int i = 0; i < N; ++i) lam_at(b, g, NAZ, view, col, az, i, w) = 0.0;
                for (int l = 0; l < n; ++l)
                    for (int d = 0; d < 5; ++d) g5_at(b, g, view, col, l, d, w) = 0.0;
            }
            for (int t = 0; t < g.ngrid; ++t) glev_at(b, g, view, t, w) = 0.0;
            // reverse sweep over the segments: weights of the sources, level-extinction gradient of the transmissions, and the
            // source gradients (dual numbers) gathered per (column, layer)
            double d_rad = 1.0;
            for (int s = s1 - 1; s >= s0; --s) {
                const double t = path_at(b, g, view, 0, s - s0, w), src = path_at(b, g, view, 1, s - s0, w), incoming = path_at(b, g, view, 2, s - s0, w);
                const double d_od = -d_rad * (incoming - src) * t;
                for (int j = g.od_start[s]; j < g.od_start[s + 1]; ++j) glev_at(b, g, view, g.od_level[j], w) += d_od * g.od_weight[j];
                const double d_src = d_rad * (1.0 - t);
                const int lo = SOLAR ? g.seg_col[2 * s] : 0, up = SOLAR ? g.seg_col[2 * s + 1] : 0;
                const double uw = (lo != up) ? g.seg_colw[s] : 0.0;
                const int l = g.seg_layer[s];
                for (int side = 0; side < (lo != up ? 2 : 1); ++side) {
                    const int col = side == 0 ? lo : up;
                    const double wgt = d_src * (side == 0 ? 1.0 - uw : uw);
                    const LayerP<double> pv = load_p<SOLAR>(g, b, col, l, w);
                    const double csz = g.csz[col];
                    const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
                    const double azw[2] = {1.0, cos(g.seg_azimuth[s])};
                    double G[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                    if (!DUAL) {
                        for (int az = 0; az < NAZ; ++az) {
                            const LayerSol<double> h = layer_solution<double, SOLAR>(az, pv, mu, csz, angular);
                            const double c0 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l, w), c1 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l + 1, w);
                            double U, V;
                            segment_adjoint<SOLAR>(az, pv, h, mu, csz, angular, g.seg_frac[s], g.seg_cos[s], c0, c1, wgt * azw[az], G, U, V);
                            lam_at(b, g, NAZ, view, col, az, 2 * l, w) += wgt * azw[az] * U;
                            lam_at(b, g, NAZ, view, col, az, 2 * l + 1, w) += wgt * azw[az] * V;
                        }
                        if (!SOLAR) {
                            // the emission itself seen along the path: amp exp(-rate od frac) (1 - ssa)
                            const double ft = g.seg_frac[s], e = exp(-pv.rate * pv.od * ft), v = wgt * e;
                            G[3] += v * (1.0 - pv.ssa);
                            G[1] -= v * pv.amp;
                            G[4] -= v * pv.amp * (1.0 - pv.ssa) * pv.od * ft;
                            G[0] -= v * pv.amp * (1.0 - pv.ssa) * pv.rate * ft;
                        }
                    } else {
                        const LayerP<D5> p = lift<D5>(pv);
                        D5 acc = mkd(0.0);
                        for (int az = 0; az < NAZ; ++az) {
                            const LayerSol<D5> h = layer_solution<D5, SOLAR>(az, p, mu, csz, angular);
                            D5 U, V, P;
                            local_source_terms<D5, SOLAR>(az, p, h, g.seg_frac[s], g.seg_cos[s], mu, U, V, P);
                            const double c0 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l, w), c1 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l + 1, w);
                            acc += azw[az] * (c0 * U + c1 * V + P);
                            lam_at(b, g, NAZ, view, col, az, 2 * l, w) += wgt * azw[az] * U.v;
                            lam_at(b, g, NAZ, view, col, az, 2 * l + 1, w) += wgt * azw[az] * V.v;
                        }
                        if (!SOLAR) acc += local_emission(p, g.seg_frac[s]);
                        for (int d = 0; d < 5; ++d) G[d] = wgt * acc.d[d];
                    }
                    for (int d = 0; d < 5; ++d) g5_at(b, g, view, col, l, d, w) += G[d];
                }
                d_rad *= t;
            }
            surf_weight = g.ground_hit[view] ? d_rad : 0.0;
            // dF/dc of the surface term, then the forward half of the transposed solve, U^T y = dF/dc, in place
            for (int col = col_first; col <= col_last; ++col) {
                double ws = surf_weight;
                if (SOLAR) {
                    ws = 0.0;
                    if (col == ground_lo) ws += surf_weight * (ground_lo != ground_up ? 1.0 - ground_uw : 1.0);
                    if (col == ground_up && ground_lo != ground_up) ws += surf_weight * ground_uw;
                }
                if (ws != 0.0) {
                    const double wsurf = ws * (2.0 * mu) * albedo;
                    const LayerP<double> p = load_p<SOLAR>(g, b, col, n - 1, w);
                    const LayerSol<double> h = layer_solution<double, SOLAR>(0, p, mu, g.csz[col], 0.0);
                    lam_at(b, g, NAZ, view, col, 0, N - 2, w) += wsurf * h.xp * h.omega;
                    lam_at(b, g, NAZ, view, col, 0, N - 1, w) += wsurf * h.xm;
                }
                for (int az = 0; az < NAZ; ++az) {
                    double r1 = 0, r2 = 0;
                    for (int i = 0; i < N; ++i) {
                        double r = lam_at(b, g, NAZ, view, col, az, i, w);
                        if (i >= 1) r -= lu_at(b, g, NAZ, col, az, LU_ALPHA, i - 1, w) * r1;
                        if (i >= 2) r -= lu_at(b, g, NAZ, col, az, LU_BETA, i - 2, w) * r2;
                        lam_at(b, g, NAZ, view, col, az, i, w) = r;
                        r2 = r1;
                        r1 = r;
                    }
                }
            }
        }
    }
    {
        double* dst = b.radiance + gw * g.nlos + view;
        *dst = b.accumulate ? *dst + radiance : radiance;
        if (b.los_od) b.los_od[gw * g.nlos + view] = path_od;
    }
    if (DERIVS) {
        vstate_at(b, g, view, 0, w) = surf_weight;
        vstate_at(b, g, view, 1, w) = path_od;
    }
}

// radiance of the single plane-parallel line of sight after solve_thread<.., FUSED> when no weighting functions are asked for: back
// substitution and layer sources on the way up
template <bool SOLAR> TS_HD void fused_radiance_thread(const Geo& g, const Bufs& b, int w) {
    constexpr int NAZ = SOLAR ? 2 : 1;
    const int n = g.nlyr;
    const size_t gw = size_t(b.w0) + w;
    const double mu = g.mu, csz = g.csz[0], vc = g.view_cos[0], inv = 1.0 / vc;
    const double albedo = b.albedo[gw];
    const double surface_emission = (!SOLAR && b.surface_emission) ? b.surface_emission[gw] : 0.0;
    const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
    double radiance = 0.0;
    for (int az = 0; az < NAZ; ++az) {
        const double azw = az == 0 ? 1.0 : cos(g.view_azimuth[0]);
        double tau_over_vc = vstate_at(b, g, 0, 1, w), cn0 = 0.0, cn1 = 0.0;
        for (int l = n - 1; l >= 0; --l) {
            const LayerP<double> p = load_p<SOLAR>(g, b, 0, l, w);
            const double c1 = lu_at(b, g, NAZ, 0, az, LU_C, 2 * l + 1, w) - lu_at(b, g, NAZ, 0, az, LU_ALPHA, 2 * l + 1, w) * cn0 -
                              lu_at(b, g, NAZ, 0, az, LU_BETA, 2 * l + 1, w) * cn1;
            const double c0 = lu_at(b, g, NAZ, 0, az, LU_C, 2 * l, w) - lu_at(b, g, NAZ, 0, az, LU_ALPHA, 2 * l, w) * c1 -
                              lu_at(b, g, NAZ, 0, az, LU_BETA, 2 * l, w) * cn0;
            cn1 = c1;
            cn0 = c0;
            tau_over_vc -= p.od * inv;
            const double attenuation = exp(-fmax(tau_over_vc, 0.0)), beam = exp(-p.od * inv);
            const LayerSol<double> h = layer_solution<double, SOLAR>(az, p, mu, csz, angular);
            const bool fast = plane_nonresonant(p.rate, h.k, p.od, inv);
            const double source_integral = plane_source_integral(p, h.expo, beam, vc, fast);
            double U, V, P;
            plane_source_terms<double, SOLAR>(az, p, h, vc, mu, fast, beam, source_integral, U, V, P);
            double source = azw * (c0 * U + c1 * V + P);
            if (!SOLAR && az == 0) source += p.amp * source_integral * (1.0 - p.ssa);
            radiance += attenuation * source;
            if (l == n - 1 && az == 0) {
                const Boundary<double> q = boundary_particular(p, h);
                radiance += vstate_at(b, g, 0, 0, w) * (bottom_plus(h, q, c0, c1) * (2.0 * mu) * albedo + surface_emission);
            }
        }
    }
    double* dst = b.radiance + gw * g.nlos;
    *dst = b.accumulate ? *dst + radiance : radiance;
    if (b.los_od) b.los_od[gw * g.nlos] = vstate_at(b, g, 0, 1, w);
}

// ---------------------------------------------------------------------------------------------------------------------------
// hand-written reverse sweep of one (layer, azimuth order) of the Lagrangian.  The value differentiated is
//   Tp top_plus + Tm top_minus + Bp bot_plus + Bm bot_minus + Ws (c0 U + c1 V + P [+ thermal emission term])
// with top_plus = gpt + c0 xp + c1 xm omega, top_minus = gmt + c0 xm + c1 xp omega, bot_plus = gpb + c0 xp omega + c1 xm,
// bot_minus = gmb + c0 xm omega + c1 xp (the two streams at the layer boundaries) and (U, V, P) of plane_source_terms; PLANE = false leaves
// the source out (spherical geometry, where the segment sources are differentiated in the forward kernel).  g += d/d(od, ssa, b1, amp, rate).
// Every quotient of exponentials enters as a function of (k, rate, od) with its three partial derivatives: the direct form from the
// exponentials at hand, or -- per quotient, where its own exponent difference is near the removable singularity -- the series of
// exp_difference / integrated_exp_difference (:64-128).  view_adjoint_thread<.., DUAL = true> differentiates the same value in dual
// numbers: the cross-check of this function in the tests.
// ---------------------------------------------------------------------------------------------------------------------------
// the common tail of the reverse sweeps: from the adjoints of the particular-solution amplitudes (upper, lower), the eigenvector components
// (xp, xm), omega = exp(-k od) and the eigenvalue to the layer parameters
template <bool SOLAR>
TS_HD void solution_chain(int az, const LayerP<double>& p, const LayerSol<double>& h, double mu, double csz, double angular, double a_upper,
                          double a_lower, double a_xp, double a_xm, double a_om, double a_k, double a_od, double a_ssa, double a_b1, double a_amp,
                          double a_rate, double g[5]) {
    const double xp = h.xp, xm = h.xm, om = h.omega, k = h.k, upper = h.upper, lower = h.lower;
    const double od = p.od, ssa = p.ssa, b1 = p.b1;
    // ---- particular-solution amplitudes
    const double norm = mu * (xp * xp - xm * xm), i_norm = 1.0 / norm;
    double a_norm;
    if (SOLAR) {
        double qp, qm;
        if (az == 0) {
            qp = ssa * (1.0 + b1 * (csz * mu)) / kFourPi;
            qm = ssa * (1.0 - b1 * (csz * mu)) / kFourPi;
        } else {
            qp = qm = ssa * b1 * (angular / kFourPi);
        }
        const double a_qp = (a_upper * xp + a_lower * xm) * i_norm, a_qm = (a_upper * xm + a_lower * xp) * i_norm;
        a_xp += (a_upper * qp + a_lower * qm) * i_norm;
        a_xm += (a_upper * qm + a_lower * qp) * i_norm;
        a_norm = -(a_upper * upper + a_lower * lower) * i_norm;
        if (az == 0) {
            a_ssa += (a_qp * (1.0 + b1 * (csz * mu)) + a_qm * (1.0 - b1 * (csz * mu))) / kFourPi;
            a_b1 += ssa * (csz * mu) * (a_qp - a_qm) / kFourPi;
        } else {
            a_ssa += (a_qp + a_qm) * b1 * (angular / kFourPi);
            a_b1 += (a_qp + a_qm) * ssa * (angular / kFourPi);
        }
    } else {
        const double a_at = a_upper + a_lower;
        a_ssa -= a_at * (xp + xm) * i_norm;
        a_xp += a_at * (1.0 - ssa) * i_norm;
        a_xm += a_at * (1.0 - ssa) * i_norm;
        a_norm = -a_at * upper * i_norm;
    }
    a_xp += a_norm * 2.0 * mu * xp;
    a_xm -= a_norm * 2.0 * mu * xm;
    // ---- omega = exp(-k od) (the exponentials inside the quotients are accounted for by their own partial derivatives),
    // xp / xm = (1 -/+ s / k) / 2, k = sqrt(s d)
    a_k -= a_om * om * od;
    a_od -= a_om * om * k;
    double s, d;
    if (az == 0) {
        d = ssa * b1 * mu - 1.0 / mu;
        s = (ssa - 1.0) / mu;
    } else {
        d = -1.0 / mu;
        s = (ssa * b1 * (1.0 - mu * mu) - 2.0) / (2.0 * mu);
    }
    const double i_k = 1.0 / k, a_sk = 0.5 * (a_xm - a_xp);
    a_k -= a_sk * s * i_k * i_k;
    const double a_s = a_sk * i_k + a_k * 0.5 * d * i_k, a_d = a_k * 0.5 * s * i_k;
    if (az == 0) {
        a_ssa += a_d * b1 * mu + a_s / mu;
        a_b1 += a_d * ssa * mu;
    } else {
        a_ssa += a_s * b1 * (1.0 - mu * mu) / (2.0 * mu);
        a_b1 += a_s * ssa * (1.0 - mu * mu) / (2.0 * mu);
    }
    g[0] += a_od;
    g[1] += a_ssa;
    g[2] += a_b1;
    g[3] += a_amp;
    g[4] += a_rate;
}

// (exp_a - exp_b) / (b - a) with exp_a = exp(-a t), exp_b = exp(-b t) and its partial derivatives with respect to a, b, t
TS_HD Stable edr_partials(double exp_a, double exp_b, double a, double b, double t) {
    const double delta = b - a;
    if (fabs(delta * t) <= 1.0e-5) return exp_difference(a, b, t);
    const double num = exp_a - exp_b, inv = 1.0 / delta, inv2 = inv * inv;
    return {num * inv, (-t * exp_a * delta + num) * inv2, (t * exp_b * delta - num) * inv2, (-a * exp_a + b * exp_b) * inv};
}
// integral over [0, t] of edr(a, b, x) given ga = edr(0, a, t), gb = edr(0, b, t) (values and d/db) and eab = edr(a, b, t)
TS_HD Stable iedr_partials(const Stable& ga, const Stable& gb, double eab, double a, double b, double t) {
    const double delta = b - a;
    if (fabs(delta * t) <= 1.0e-4) return integrated_exp_difference(a, b, t);
    const double num = ga.value - gb.value, inv = 1.0 / delta, inv2 = inv * inv;
    return {num * inv, (ga.db * delta + num) * inv2, (-gb.db * delta - num) * inv2, eab};
}

template <bool SOLAR, bool PLANE>
TS_HD void layer_adjoint(int az, const LayerP<double>& p, const LayerSol<double>& h, double mu, double csz, double angular, double c0, double c1,
                         double Tp, double Tm, double Bp, double Bm, double Ws, double vc, double g[5], double& bot_plus, double& source) {
    const double xp = h.xp, xm = h.xm, om = h.omega, ex = h.expo, k = h.k, upper = h.upper, lower = h.lower;
    const double od = p.od, ssa = p.ssa, b1 = p.b1, amp = p.amp, rate = p.rate;
    const Stable cpr = edr_partials(om, ex, k, rate, od);              // d/dk = da, d/drate = db
    const Stable cmr = edr_partials(1.0, om * ex, 0.0, rate + k, od);  // d/dk = d/drate = db
    const double cp = amp * cpr.value, cm = amp * cmr.value;
    bot_plus = upper * cp * xp + c0 * xp * om + c1 * xm;
    // ---- the two streams at the boundaries
    const double zt = Tp * xm + Tm * xp, zb = Bp * xp + Bm * xm;
    double a_lower = cm * zt, a_upper = cp * zb;
    const double a_cm = lower * zt, a_cp = upper * zb;
    const double k_xp = Tp * c0 + Bm * c1, k_xm = Tm * c0 + Bp * c1, k_xpo = Tm * c1 + Bp * c0, k_xmo = Tp * c1 + Bm * c0;
    double a_xp = lower * cm * Tm + upper * cp * Bp + k_xp + om * k_xpo;
    double a_xm = lower * cm * Tp + upper * cp * Bm + k_xm + om * k_xmo;
    const double a_om = k_xpo * xp + k_xmo * xm;
    // cp = amp cpr, cm = amp cmr
    const double a_cpr = a_cp * amp, a_cmr = a_cm * amp;
    double a_amp = a_cp * cpr.value + a_cm * cmr.value;
    double a_k = a_cpr * cpr.da + a_cmr * cmr.db, a_rate = a_cpr * cpr.db + a_cmr * cmr.db, a_od = a_cpr * cpr.dt + a_cmr * cmr.dt;
    double a_ssa = 0.0, a_b1 = 0.0;
    source = 0.0;
    if (PLANE) {
        // ---- layer-integrated source toward the viewing cosine (explicit_plane_view, :2598-2640)
        const double inv = 1.0 / vc, phase_mu = vc * mu, phase_sine = 0.25 * sqrt(fmax(0.0, (1.0 - vc * vc) * (1.0 - mu * mu)));
        double lp, lm;
        if (az == 0) {
            lp = 0.5 * ssa * (1.0 - b1 * phase_mu);
            lm = 0.5 * ssa * (1.0 + b1 * phase_mu);
        } else {
            lp = lm = SOLAR ? ssa * b1 * phase_sine : 0.0;
        }
        const double yp = lp * xp + lm * xm, ym = lp * xm + lm * xp;
        const double beam = exp(-od * inv);
        const Stable e_hm = edr_partials(om, beam, k, inv, od);             // hm = inv e_hm:  d/dk = da
        const Stable e_hp = edr_partials(1.0, om * beam, 0.0, k + inv, od); // hp = inv e_hp:  d/dk = db
        const Stable e_si = edr_partials(1.0, ex * beam, 0.0, rate + inv, od);   // source_integral = inv e_si:  d/drate = db
        // dp_ratio = inv iedr(rate + k, rate + inv), dm_ratio = inv iedr(k + inv, rate + inv)
        const Stable e_dp = iedr_partials(cmr, e_si, ex * e_hm.value, rate + k, rate + inv, od);    // d/dk = da, d/drate = da + db
        const Stable e_dm = iedr_partials(e_hp, e_si, beam * cpr.value, k + inv, rate + inv, od);   // d/dk = da, d/drate = db
        const double hm = inv * e_hm.value, hp = inv * e_hp.value, si = inv * e_si.value, dpr = inv * e_dp.value, dmr = inv * e_dm.value;
        source = c0 * yp * hp + c1 * ym * hm + amp * (upper * yp * dmr + lower * ym * dpr);
        double a_si = 0.0;
        if (!SOLAR && az == 0) {
            source += amp * si * (1.0 - ssa);
            a_amp += Ws * si * (1.0 - ssa);
            a_si = Ws * amp * (1.0 - ssa);
            a_ssa -= Ws * amp * si;
        }
        const double a_yp = Ws * (c0 * hp + amp * upper * dmr), a_ym = Ws * (c1 * hm + amp * lower * dpr);
        const double a_hp = Ws * c0 * yp * inv, a_hm = Ws * c1 * ym * inv;                            // adjoints of e_hp, e_hm
        const double a_dm = Ws * amp * upper * yp * inv, a_dp = Ws * amp * lower * ym * inv;          // adjoints of e_dm, e_dp
        a_si *= inv;                                                                                   // adjoint of e_si
        a_upper += Ws * amp * yp * dmr;
        a_lower += Ws * amp * ym * dpr;
        a_amp += Ws * (upper * yp * dmr + lower * ym * dpr);
        a_k += a_hm * e_hm.da + a_hp * e_hp.db + a_dp * e_dp.da + a_dm * e_dm.da;
        a_rate += a_si * e_si.db + a_dp * (e_dp.da + e_dp.db) + a_dm * e_dm.db;
        a_od += a_hm * e_hm.dt + a_hp * e_hp.dt + a_si * e_si.dt + a_dp * e_dp.dt + a_dm * e_dm.dt;
        // yp = lp xp + lm xm, ym = lp xm + lm xp
        const double a_lp = a_yp * xp + a_ym * xm, a_lm = a_yp * xm + a_ym * xp;
        a_xp += a_yp * lp + a_ym * lm;
        a_xm += a_yp * lm + a_ym * lp;
        if (az == 0) {
            a_ssa += 0.5 * (a_lp * (1.0 - b1 * phase_mu) + a_lm * (1.0 + b1 * phase_mu));
            a_b1 += 0.5 * ssa * phase_mu * (a_lm - a_lp);
        } else if (SOLAR) {
            a_ssa += (a_lp + a_lm) * b1 * phase_sine;
            a_b1 += (a_lp + a_lm) * ssa * phase_sine;
        }
    }
    solution_chain<SOLAR>(az, p, h, mu, csz, angular, a_upper, a_lower, a_xp, a_xm, a_om, a_k, a_od, a_ssa, a_b1, a_amp, a_rate, g);
}

// reverse sweep of the source function of one traced segment (local_source_terms), one azimuth order: g += Ws d(c0 U + c1 V + P) / d(od, ssa,
// b1, amp, rate); U and V (the dF/dc rows of the segment) are returned
template <bool SOLAR>
TS_HD void segment_adjoint(int az, const LayerP<double>& p, const LayerSol<double>& h, double mu, double csz, double angular, double frac_top,
                           double cosine, double c0, double c1, double Ws, double g[5], double& U, double& V) {
    const double xp = h.xp, xm = h.xm, k = h.k, upper = h.upper, lower = h.lower, ex = h.expo;
    const double od = p.od, ssa = p.ssa, b1 = p.b1, amp = p.amp, rate = p.rate;
    const double frac_bottom = 1.0 - frac_top;
    const double phase_mu = cosine * mu, phase_sine = 0.25 * sqrt(fmax(0.0, (1.0 - cosine * cosine) * (1.0 - mu * mu)));
    double lp, lm;
    if (az == 0) {
        lp = 0.5 * ssa * (1.0 - b1 * phase_mu);
        lm = 0.5 * ssa * (1.0 + b1 * phase_mu);
    } else {
        lp = lm = SOLAR ? ssa * b1 * phase_sine : 0.0;
    }
    const double yp = lp * xp + lm * xm, ym = lp * xm + lm * xp;
    const double top_e = exp(-k * od * frac_top), bot_e = exp(-k * od * frac_bottom), beam = exp(-rate * od * frac_top);
    const Stable e_plus = edr_partials(top_e, beam, k, rate, od * frac_top);                                            // d/dk = da, d/drate = db, d/dod = dt frac_top
    const Stable e_minus = edr_partials(beam, ex * bot_e, rate * frac_top, rate + k * frac_bottom, od);                  // a = rate frac_top, b = rate + k frac_bottom
    const double plus = amp * e_plus.value, minus = amp * frac_bottom * e_minus.value;
    U = yp * top_e;
    V = ym * bot_e;
    const double a_yp = Ws * (c0 * top_e + upper * plus), a_ym = Ws * (c1 * bot_e + lower * minus);
    const double a_top = Ws * c0 * yp, a_bot = Ws * c1 * ym;
    const double a_upper = Ws * yp * plus, a_lower = Ws * ym * minus;
    const double a_e1 = Ws * upper * yp * amp, a_e2 = Ws * lower * ym * amp * frac_bottom;
    const double a_amp = Ws * (upper * yp * e_plus.value + lower * ym * frac_bottom * e_minus.value);
    double a_k = a_e1 * e_plus.da + a_e2 * e_minus.db * frac_bottom - (a_top * top_e * frac_top + a_bot * bot_e * frac_bottom) * od;
    double a_rate = a_e1 * e_plus.db + a_e2 * (e_minus.da * frac_top + e_minus.db);
    double a_od = a_e1 * e_plus.dt * frac_top + a_e2 * e_minus.dt - (a_top * top_e * frac_top + a_bot * bot_e * frac_bottom) * k;
    const double a_lp = a_yp * xp + a_ym * xm, a_lm = a_yp * xm + a_ym * xp;
    const double a_xp = a_yp * lp + a_ym * lm, a_xm = a_yp * lm + a_ym * lp;
    double a_ssa = 0.0, a_b1 = 0.0;
    if (az == 0) {
        a_ssa = 0.5 * (a_lp * (1.0 - b1 * phase_mu) + a_lm * (1.0 + b1 * phase_mu));
        a_b1 = 0.5 * ssa * phase_mu * (a_lm - a_lp);
    } else if (SOLAR) {
        a_ssa = (a_lp + a_lm) * b1 * phase_sine;
        a_b1 = (a_lp + a_lm) * ssa * phase_sine;
    }
    solution_chain<SOLAR>(az, p, h, mu, csz, angular, a_upper, a_lower, a_xp, a_xm, 0.0, a_k, a_od, a_ssa, a_b1, a_amp, a_rate, g);
}

// Weighting functions of one line of sight: the backward half of the transposed solve (L^T lambda = y) fused with the Lagrangian of every
// layer, bottom-up; then the chain from the layer parameters to the level profiles.
// DUAL = false (the product): the hand-written reverse sweep of the layers (layer_adjoint); DUAL = true (tests only): the same Lagrangian
// differentiated in dual numbers.
// `lds`: n + 1 values per thread, element i of this thread at lds[i * ls]
// FUSED (plane-parallel geometry with one line of sight, after solve_thread<.., FUSED>): the solution c is substituted back on the way up
// instead of being read, and the radiance -- the same layer sources this sweep evaluates anyway -- is written here.
template <bool SOLAR, bool SPHERICAL, bool DUAL = false, bool FUSED = false>
TS_HD void view_adjoint_thread(const Geo& g, const Bufs& b, int w, int view, double* lds, int ls) {
    constexpr int NAZ = SOLAR ? 2 : 1;
    const int n = g.nlyr, N = 2 * n, ncol = SOLAR ? g.ncol : 1;
    const size_t gw = size_t(b.w0) + w;
    const double mu = g.mu;
    const double albedo = b.albedo[gw];
    const double surface_emission = (!SOLAR && b.surface_emission) ? b.surface_emission[gw] : 0.0;
    const double surf_weight = vstate_at(b, g, view, 0, w);
    constexpr bool plane = !SPHERICAL;
    const double vc = plane ? g.view_cos[view] : 1.0, inv = 1.0 / vc;
    const int ground_lo = (!plane && SOLAR) ? g.ground_col[2 * view] : 0, ground_up = (!plane && SOLAR) ? g.ground_col[2 * view + 1] : 0;
    const double ground_uw = (!plane && SOLAR) ? g.ground_colw[view] : 0.0;
    double g_albedo = 0.0, g_surface_emission = 0.0, fused_radiance = 0.0;
    // columns no source of this line of sight refers to have dF/dc = 0 and contribute nothing
    const int col_first = (!plane && SOLAR) ? g.view_cols[2 * view] : 0, col_last = (!plane && SOLAR) ? g.view_cols[2 * view + 1] : ncol - 1;

    for (int col = col_first; col <= col_last; ++col) {
        const double csz = g.csz[col];
        const double angular = sqrt(fmax(0.0, (1.0 - mu * mu) * (1.0 - csz * csz)));
        double ws = surf_weight;   // weight of this column's surface source in the radiance
        if (!plane && SOLAR) {
            ws = 0.0;
            if (col == ground_lo) ws += surf_weight * (ground_lo != ground_up ? 1.0 - ground_uw : 1.0);
            if (col == ground_up && ground_lo != ground_up) ws += surf_weight * ground_uw;
        }
        if (!SOLAR) g_surface_emission += ws;
        const double wsurf = ws * (2.0 * mu) * albedo;   // multiplies the downwelling stream at the ground
        for (int i = 0; i <= n; ++i) lds[i * ls] = 0.0;  // gradient with respect to the solar attenuation exponent at every boundary
        // one azimuth order after the other (their Lagrangians add up): half the live state of a fused loop
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int az = 0; az < NAZ; ++az) {
            const double delta = az == 0 ? 1.0 : 0.0;
            const double azw = plane ? (az == 0 ? 1.0 : cos(g.view_azimuth[view])) : 0.0;
            double xa = 0.0, xb = 0.0;   // lambda of rows 2l + 1 and 2l + 2
            // plane parallel: optical depth from the top to the layer (over the view cosine) and the radiance gathered below the layer
            double tau_over_vc = plane ? vstate_at(b, g, view, 1, w) : 0.0, below = 0.0;
            // everything a layer reads from scratch, fetched one layer ahead of its use (the stores of a layer would otherwise hold the
            // next layer's loads back)
            struct Fetch {
                LayerP<double> p;
                double c0, c1, y_plus, y_minus, gamma_plus, gamma_minus, imu_plus, imu_minus;
                double al0, be0, al1, be1;   // FUSED: c0 / c1 hold z of the rows 2l, 2l + 1, these their alpha and beta
            };
            auto fetch = [&](int l) {
                Fetch f;
                f.p = load_p<SOLAR>(g, b, col, l, w);
                f.c0 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l, w);
                f.c1 = lu_at(b, g, NAZ, col, az, LU_C, 2 * l + 1, w);
                if (FUSED) {
                    f.al0 = lu_at(b, g, NAZ, col, az, LU_ALPHA, 2 * l, w);
                    f.be0 = lu_at(b, g, NAZ, col, az, LU_BETA, 2 * l, w);
                    f.al1 = lu_at(b, g, NAZ, col, az, LU_ALPHA, 2 * l + 1, w);
                    f.be1 = lu_at(b, g, NAZ, col, az, LU_BETA, 2 * l + 1, w);
                }
                f.y_plus = lam_at(b, g, NAZ, view, col, az, 2 * l, w);
                f.gamma_plus = lu_at(b, g, NAZ, col, az, LU_GAMMA, 2 * l + 1, w);
                f.imu_plus = lu_at(b, g, NAZ, col, az, LU_IMU, 2 * l, w);
                f.y_minus = l > 0 ? lam_at(b, g, NAZ, view, col, az, 2 * l - 1, w) : 0.0;
                f.gamma_minus = l > 0 ? lu_at(b, g, NAZ, col, az, LU_GAMMA, 2 * l, w) : 0.0;
                f.imu_minus = l > 0 ? lu_at(b, g, NAZ, col, az, LU_IMU, 2 * l - 1, w) : 0.0;
                return f;
            };
            Fetch next = fetch(n - 1);
            xa = lam_at(b, g, NAZ, view, col, az, N - 1, w) * lu_at(b, g, NAZ, col, az, LU_IMU, N - 1, w);
            double cn0 = 0.0, cn1 = 0.0;   // FUSED: c of rows 2l + 2, 2l + 3
            for (int l = n - 1; l >= 0; --l) {
                const Fetch cur = next;
                if (l > 0) next = fetch(l - 1);
                const LayerP<double>& pv = cur.p;
                double attenuation = 0.0;
                if (plane) {
                    tau_over_vc -= pv.od * inv;
                    attenuation = exp(-fmax(tau_over_vc, 0.0));
                }
                double G[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                const LayerSol<double> hv = layer_solution<double, SOLAR>(az, pv, mu, csz, angular);
                double c0 = cur.c0, c1 = cur.c1;
                if (FUSED) {   // back substitution (:1889-1898)
                    c1 = cur.c1 - cur.al1 * cn0 - cur.be1 * cn1;
                    c0 = cur.c0 - cur.al0 * c1 - cur.be0 * cn0;
                    cn1 = c1;
                    cn0 = c0;
                }
                // lambda of the rows above this layer: L^T lambda = y backwards (pentadiagonal_transpose_solve, :1911-1920); the
                // second sub-diagonal of row 2l + 2 is xp omega of this layer (:2008)
                const double e_below = l < n - 1 ? hv.xp * hv.omega : 0.0;
                const double lam_top_plus = (cur.y_plus - cur.gamma_plus * xa - e_below * xb) * cur.imu_plus;
                const double lam_top_minus = l > 0 ? (cur.y_minus - cur.gamma_minus * lam_top_plus) * cur.imu_minus : 0.0;
                // weights of the four boundary streams of this layer in the Lagrangian (rows 2l - 1, 2l above; 2l + 1, 2l + 2 or the surface row below)
                const double lam_last = xa;
                const double Tp = l == 0 ? -lam_top_plus : lam_top_plus, Tm = l == 0 ? 0.0 : lam_top_minus;
                double Bp, Bm;
                if (l == n - 1) {
                    Bp = lam_last * (2.0 * mu * albedo * delta) + (az == 0 ? wsurf : 0.0);
                    Bm = -lam_last;
                } else {
                    Bp = -xb;
                    Bm = -xa;
                }
                const double Ws = attenuation * azw;
                double bot_plus_v = 0.0, source_v = 0.0;
                if (!DUAL) {
                    layer_adjoint<SOLAR, plane>(az, pv, hv, mu, csz, angular, c0, c1, Tp, Tm, Bp, Bm, Ws, vc, G, bot_plus_v, source_v);
                } else {
                    const LayerP<D5> p = lift<D5>(pv);
                    const LayerSol<D5> h = layer_solution<D5, SOLAR>(az, p, mu, csz, angular);
                    const Boundary<D5> q = boundary_particular(p, h);
                    const D5 xpo = h.xp * h.omega, xmo = h.xm * h.omega;
                    const D5 top_plus = q.gpt + c0 * h.xp + c1 * xmo, top_minus = q.gmt + c0 * h.xm + c1 * xpo;
                    const D5 bot_plus = q.gpb + c0 * xpo + c1 * h.xm, bot_minus = q.gmb + c0 * xmo + c1 * h.xp;
                    D5 L = Tp * top_plus + Tm * top_minus + Bp * bot_plus + Bm * bot_minus;
                    bot_plus_v = bot_plus.v;
                    if (plane) {
                        const D5 beam = t_exp(-(p.od * inv));
                        const bool fast = plane_nonresonant(pv.rate, h.k.v, pv.od, inv);
                        const D5 source_integral = plane_source_integral(p, h.expo, beam, vc, fast);
                        D5 U, V, P;
                        plane_source_terms<D5, SOLAR>(az, p, h, vc, mu, fast, beam, source_integral, U, V, P);
                        D5 src = c0 * U + c1 * V + P;
                        if (!SOLAR && az == 0) src += p.amp * source_integral * (1.0 - p.ssa);
                        L += Ws * src;
                        source_v = src.v;
                    }
                    for (int d = 0; d < 5; ++d) G[d] += L.d[d];
                }
                if (l == n - 1) {
                    // the surface row also holds the albedo, the transmission to the ground and the surface emission
                    const double amp_ground = SOLAR ? amp_at(b, g, col, n, w) : 0.0;
                    g_albedo += lam_last * (delta * csz / kPi * amp_ground + 2.0 * mu * delta * bot_plus_v);
                    if (az == 0) {
                        g_albedo += ws * (2.0 * mu) * bot_plus_v;
                        if (plane) below = ws * (bot_plus_v * (2.0 * mu) * albedo + surface_emission);
                    }
                    if (SOLAR) lds[n * ls] += lam_last * delta * csz * albedo / kPi * amp_ground;
                    else g_surface_emission += lam_last;
                }
                xb = lam_top_plus;
                xa = lam_top_minus;

                double g_od = G[0], g_ssa = G[1], g_b1 = G[2], g_amp = G[3], g_rate = G[4];
                if (plane) {
                    // every source below this layer is attenuated by exp(-od / vc) of this one
                    g_od -= below * inv;
                    below += Ws * source_v;
                } else if (az == 0) {
                    g_od += g5_at(b, g, view, col, l, 0, w);
                    g_ssa += g5_at(b, g, view, col, l, 1, w);
                    g_b1 += g5_at(b, g, view, col, l, 2, w);
                    g_amp += g5_at(b, g, view, col, l, 3, w);
                    g_rate += g5_at(b, g, view, col, l, 4, w);
                }
                if (SOLAR) {
                    // amplitude = irradiance exp(att_l); rate = (att_l - att_(l+1)) / od  (map_explicit_to_levels, :2843-2865)
                    if (l > 0) lds[l * ls] += g_amp * pv.amp;
                    if (pv.od > 0.0) {
                        const double wgt = g_rate / pv.od;
                        lds[l * ls] += wgt;
                        lds[(l + 1) * ls] -= wgt;
                        g_od -= g_rate * pv.rate / pv.od;
                    }
                } else {
                    // (thermal: amplitude = emission at the top level, rate = log(top / bottom) / od, :2868-2887 -- mapped with the levels below)
                    gsum_at(b, g, view, 3, l, w) = g_amp;
                    gsum_at(b, g, view, 4, l, w) = g_rate;
                }
                if (col == col_first && az == 0) {
                    gsum_at(b, g, view, 0, l, w) = g_od;
                    gsum_at(b, g, view, 1, l, w) = g_ssa;
                    gsum_at(b, g, view, 2, l, w) = g_b1;
                } else {
                    gsum_at(b, g, view, 0, l, w) += g_od;
                    gsum_at(b, g, view, 1, l, w) += g_ssa;
                    gsum_at(b, g, view, 2, l, w) += g_b1;
                }
            }
            if (FUSED) fused_radiance += below;   // surface term and every layer source of this azimuth order, attenuated to the top
        }
        if (SOLAR) {
            // attenuation exponents are minus the Chapman-weighted optical depths above the boundary
            const double* chap_t = g.chapman_t + size_t(col) * n * n;   // [layer][boundary below]
            const double uniform = g.chap_uniform[col];
            double below_sum = 0.0;
            for (int l = n - 1; l >= 0; --l) {
                double acc = 0.0;
                if (uniform != 0.0) {
                    // plane parallel: a layer is above every boundary below it, all with the same factor
                    below_sum += lds[(l + 1) * ls];
                    acc = below_sum * uniform;
                } else {
                    int bd = 0;
                    for (; bd + 4 <= n; bd += 4) {
                        const double f0 = chap_t[size_t(l) * n + bd], f1 = chap_t[size_t(l) * n + bd + 1], f2 = chap_t[size_t(l) * n + bd + 2],
                                     f3 = chap_t[size_t(l) * n + bd + 3];
                        acc += lds[(bd + 1) * ls] * f0 + lds[(bd + 2) * ls] * f1 + lds[(bd + 3) * ls] * f2 + lds[(bd + 4) * ls] * f3;
                    }
                    for (; bd < n; ++bd) acc += lds[(bd + 1) * ls] * chap_t[size_t(l) * n + bd];
                }
                gsum_at(b, g, view, 0, l, w) -= acc;
            }
        }
    }

    if (FUSED) {
        double* dst = b.radiance + gw * g.nlos + view;
        *dst = b.accumulate ? *dst + fused_radiance : fused_radiance;
        if (b.los_od) b.los_od[gw * g.nlos + view] = vstate_at(b, g, view, 1, w);
    }
    // ---- from the layers to the level profiles (map_explicit_to_levels, :2889-2919) and out
    double carry_ext = 0.0, carry_ssa = 0.0, carry_b1 = 0.0, carry_em = 0.0;
    auto level = [&](int t) { return LevelIn{lev_at(b, g, 0, t, w), lev_at(b, g, 1, t, w), lev_at(b, g, 2, t, w), SOLAR ? 0.0 : lev_at(b, g, 3, t, w)}; };
    LevelIn top = level(0);
    for (int t = 0; t <= n; ++t) {
        double o_ext = carry_ext, o_ssa = carry_ssa, o_b1 = carry_b1, o_em = carry_em;
        carry_ext = carry_ssa = carry_b1 = carry_em = 0.0;
        if (t < n) {
            const int l = t;
            const LevelIn bot = level(l + 1);
            const double od = lay_at(b, g, 0, l, w), ssa = lay_at(b, g, 1, l, w), b1 = lay_at(b, g, 2, l, w);
            double g_od = gsum_at(b, g, view, 0, l, w);
            const double g_ssa = ssa >= kSsaCap ? 0.0 : gsum_at(b, g, view, 1, l, w), g_b1 = gsum_at(b, g, view, 2, l, w);
            if (!SOLAR) {
                const double g_amp = gsum_at(b, g, view, 3, l, w), g_rate = gsum_at(b, g, view, 4, l, w);
                o_em += g_amp;
                if (od > kThermalMinOd && top.em > kThermalMinEmission && bot.em > kThermalMinEmission) {
                    o_em += g_rate / (top.em * od);
                    carry_em -= g_rate / (bot.em * od);
                    g_od -= g_rate * rate_at(b, g, 0, l, w) / od;
                }
            }
            const double dz = g.thickness[l];
            const double mean_ext = od / dz, mean_scat = ssa * mean_ext;
            const double c_od = 0.5 * g_od * dz;
            const double c_ssa = mean_ext > 0.0 ? 0.5 * g_ssa / mean_ext : 0.0;
            const double c_b1 = mean_scat > 0.0 ? 0.5 * g_b1 / mean_scat : 0.0;
            const double db_top = top.b1 - b1, db_bot = bot.b1 - b1;
            o_ext += c_od + c_ssa * (top.ssa - ssa) + c_b1 * top.ssa * db_top;
            o_ssa += c_ssa * top.ext + c_b1 * top.ext * db_top;
            o_b1 += c_b1 * top.ssa * top.ext;
            carry_ext = c_od + c_ssa * (bot.ssa - ssa) + c_b1 * bot.ssa * db_bot;
            carry_ssa = c_ssa * bot.ext + c_b1 * bot.ext * db_bot;
            carry_b1 = c_b1 * bot.ssa * bot.ext;
            top = bot;
        }
        if (!plane) o_ext += glev_at(b, g, view, t, w);
        const size_t o = (size_t(g.ngrid - 1 - t) * b.nw_total + gw) * g.nlos + view;
        if (b.accumulate) {
            b.wf_ext[o] += o_ext;
            b.wf_ssa[o] += o_ssa;
            b.wf_b1[o] += o_b1;
            if (!SOLAR && b.wf_em) b.wf_em[o] += o_em;
        } else {
            b.wf_ext[o] = o_ext;
            b.wf_ssa[o] = o_ssa;
            b.wf_b1[o] = o_b1;
            if (b.wf_em) b.wf_em[o] = SOLAR ? 0.0 : o_em;
        }
    }
    const size_t o = gw * g.nlos + view;
    if (b.accumulate) {
        b.wf_alb[o] += g_albedo;
        if (b.wf_sem) b.wf_sem[o] += g_surface_emission;
    } else {
        b.wf_alb[o] = g_albedo;
        if (b.wf_sem) b.wf_sem[o] = g_surface_emission;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// scratch sizes in doubles for W lanes (shared by the HIP driver and the CPU emulation of the tests)
// ---------------------------------------------------------------------------------------------------------------------------
struct ScratchSizes {
    size_t lay, lev, colv, lu, lam, g5, gsum, vstate, path, glev;
    size_t total() const { return lay + lev + colv + lu + lam + g5 + gsum + vstate + path + glev; }
};
inline ScratchSizes scratch_sizes(int nlyr, int ngrid, int nlos, int ncol, int max_seg, bool spherical, bool derivs, size_t W) {
    ScratchSizes z{};
    const size_t n = size_t(nlyr);
    z.lay = 3 * n * W;
    z.colv = size_t(ncol) * (2 * n + 1) * W;
    z.lu = size_t(ncol) * 2 * 5 * (2 * n) * W;
    z.vstate = size_t(nlos) * 2 * W;
    if (derivs) {
        z.lev = 4 * size_t(ngrid) * W;
        z.lam = size_t(nlos) * ncol * 2 * (2 * n) * W;
        z.gsum = size_t(nlos) * 5 * n * W;
        if (spherical) {
            z.g5 = size_t(nlos) * ncol * n * 5 * W;
            z.path = size_t(nlos) * 3 * size_t(max_seg) * W;
            z.glev = size_t(nlos) * size_t(ngrid) * W;
        }
    }
    return z;
}

}  // namespace skt
