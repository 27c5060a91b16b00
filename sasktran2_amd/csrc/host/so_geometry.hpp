// Wavelength-independent geometry of the exact single-scatter source and of the successive-orders ("HR", legacy) diffuse field.
// Host side only, computed once per engine and flattened into index / weight tables; everything per wavelength happens in the
// consumers of these tables (the HIP kernels of the product, the CPU oracle of the tests).
//
// Reference (behaviour restated, nothing copied):
//   cpp/lib/solar/singlescattersource.cpp        SingleScatterSource (exact and table solar transmission)
//   cpp/lib/solar/solartransmissionexact.cpp:35-93, solartransmissiontable.cpp:117-246
//   cpp/lib/phasefunction/phasehandler.cpp:42-362 (scatter angle and Stokes rotation per ray)
//   cpp/include/sasktran2/math/scattering.h:76-260 (stokes_scattering_factors)
//   cpp/lib/hr/diffuse_table.cpp, cpp/lib/hr/diffuse_point.cpp   (diffuse points, incoming rays, interpolation weights, scattering matrices)
//   cpp/lib/unitsphere/lebedev.cpp, unitsphere_ground.cpp, cpp/lib/grids/altitude_sza_source_location_interpolator.cpp
//   cpp/lib/sourceintegrator/sourceintegrator.cpp:666-760 (accumulation triplets)
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <utility>
#include <vector>

#include "geometry.hpp"
#include "lebedev.hpp"

namespace skso {
using skgeo::Vec3;

constexpr double kMinimumShellSize = 0.0001;   // MINIMUM_SHELL_SIZE_M, cpp/include/sasktran2/internal_common.h:23
constexpr double kScatteringEpsilon = 1e-8;    // SCATTERING_EPSILSON, math/scattering.h:7

// ---------------------------------------------------------------------------------------------------------------------------
// Unit spheres (LebedevSphere, UnitSphereGround)
// ---------------------------------------------------------------------------------------------------------------------------
struct UnitSphere {
    std::vector<Vec3> dir;        // quadrature directions of THIS sphere (ground: the up-looking subset)
    std::vector<double> weight;   // quadrature weights (full sphere: sum 1; ground: renormalised to 1/2)
    // ground variant
    bool ground = false;
    std::vector<Vec3> full_dir;
    std::vector<int> reverse;     // full index -> index in this sphere, -1 when looking down
    int size() const { return (int)dir.size(); }

    static UnitSphere lebedev(int npoints) {
        UnitSphere s;
        const std::vector<double> xyzw = skleb::sphere(npoints);
        s.dir.resize(npoints);
        s.weight.resize(npoints);
        for (int i = 0; i < npoints; ++i) {
            s.dir[i] = Vec3(xyzw[4 * i], xyzw[4 * i + 1], xyzw[4 * i + 2]);
            s.weight[i] = xyzw[4 * i + 3];
        }
        s.full_dir = s.dir;
        return s;
    }
    // UnitSphereGround: the points of the full sphere that look up at `location`
    static UnitSphere ground_of(const UnitSphere& full, const Vec3& location) {
        UnitSphere s;
        s.ground = true;
        s.full_dir = full.dir;
        s.reverse.assign(full.size(), -1);
        double norm = 0;
        std::vector<int> contributing;
        for (int i = 0; i < full.size(); ++i)
            if (full.dir[i].dot(location) > 0) {
                s.reverse[i] = (int)contributing.size();
                contributing.push_back(i);
                norm += full.weight[i];
            }
        for (int i : contributing) {
            s.dir.push_back(full.dir[i]);
            s.weight.push_back(full.weight[i] / norm * 0.5);
        }
        return s;
    }

    // three nearest quadrature points, inverse-distance weights (LebedevSphere::interpolate); ties keep the lower index
    void interpolate(const Vec3& d, std::array<int, 3>& idx, std::array<double, 3>& w) const {
        std::array<std::pair<int, double>, 3> iw;
        for (auto& e : iw) e = {0, 9999.0};
        for (int i = 0; i < (int)full_dir.size(); ++i) {
            const Vec3 diff = full_dir[i] - d;
            double sq = diff.dot(diff);
            for (int j = 0; j < 3; ++j) {
                if (sq < iw[j].second) {
                    auto temp = iw[j];
                    for (int k = j + 1; k < 3; ++k) std::swap(temp, iw[k]);
                    iw[j] = {i, sq};
                    sq = 99999;
                }
            }
        }
        bool exact = false;
        double total = 0;
        for (int i = 0; i < 3 && !exact; ++i) {
            if (iw[i].second < 1e-8) {
                for (int j = 0; j < 3; ++j) iw[j].second = 0;
                iw[i].second = 1;
                exact = true;
            } else {
                total += 1 / std::sqrt(iw[i].second);
            }
        }
        if (!exact)
            for (int i = 0; i < 3; ++i) iw[i].second = (1 / std::sqrt(iw[i].second)) / total;
        if (ground) {
            double tw = 0;
            for (auto& e : iw) {
                if (reverse[e.first] < 0) e = {0, 0.0};
                else e.first = reverse[e.first];
                tw += e.second;
            }
            if (tw > 0)
                for (auto& e : iw) e.second /= tw;
        }
        for (int i = 0; i < 3; ++i) { idx[i] = iw[i].first; w[i] = iw[i].second; }
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// Stokes rotation helpers
// ---------------------------------------------------------------------------------------------------------------------------
inline double clamp1(double v) { return v > 1 ? 1 : (v < -1 ? -1 : v); }

// math::stokes_scattering_factors: incoming and outgoing PROPAGATION directions; rotation angles relative to the global z axis
inline void stokes_scattering_factors(const Vec3& incoming, const Vec3& outgoing, double& theta, double& C1, double& C2, double& S1, double& S2) {
    const double cos_scatter = clamp1(incoming.dot(outgoing));
    theta = std::acos(cos_scatter);
    const double sin_scatter = std::sin(theta);
    C1 = C2 = 1;
    S1 = S2 = 0;
    if (std::abs(sin_scatter) < kScatteringEpsilon) return;
    const double costh_inc = clamp1(incoming.z), costh_scat = clamp1(outgoing.z);
    const double sinth_inc = std::sin(std::acos(costh_inc)), sinth_scat = std::sin(std::acos(costh_scat));
    if (std::abs(sinth_inc) < kScatteringEpsilon) return;
    double costh1 = (costh_scat - costh_inc * cos_scatter) / (sinth_inc * sin_scatter);
    const Vec3 hi = Vec3(incoming.x, incoming.y, 0).normalized();
    const double phi_inc = std::atan2(hi.y, hi.x);
    if (std::abs(sinth_scat) < kScatteringEpsilon) return;
    double costh2 = (costh_inc - costh_scat * cos_scatter) / (sinth_scat * sin_scatter);
    const Vec3 hs = Vec3(outgoing.x, outgoing.y, 0).normalized();
    const double phi_scat = std::atan2(hs.y, hs.x);
    double phi_diff = phi_inc - phi_scat;
    if (phi_diff < 0) phi_diff += 2 * skgeo::kPi;
    costh1 = clamp1(costh1);
    costh2 = clamp1(costh2);
    C1 = 2 * costh1 * costh1 - 1;
    C2 = 2 * costh2 * costh2 - 1;
    S1 = 2 * std::sqrt(1 - costh1 * costh1) * costh1;
    S2 = 2 * std::sqrt(1 - costh2 * costh2) * costh2;
    if (phi_diff < skgeo::kPi) { S1 *= -1; S2 *= -1; }
}

// Coordinates::stokes_standard_to_observer_z (cpp/lib/geometry/geometry.cpp:309-395; the early-out for an observer over the pole has
// no return statement there, so the general expression is always evaluated)
inline std::pair<double, double> stokes_standard_to_observer_z(const skgeo::Coordinates& c, const Vec3& look, const Vec3& position) {
    const Vec3 pn = position.normalized();
    const double cos_obs_lat = pn.dot(c.zu);
    const Vec3 axis = c.zu.cross(pn).normalized();
    const double angle = -std::acos(cos_obs_lat);
    const Vec3 rl = skgeo::rotate(look, axis, angle), rs = skgeo::rotate(c.sun, axis, angle);
    const Vec3 pz0 = (c.zu - look * c.zu.dot(look)).normalized(), ps0 = (c.sun - look * c.sun.dot(look)).normalized();
    const Vec3 pz1 = (c.zu - rl * c.zu.dot(rl)).normalized(), ps1 = (rs - rl * rs.dot(rl)).normalized();
    const double a0 = std::acos(clamp1(pz0.dot(ps0))), a1 = std::acos(clamp1(pz1.dot(ps1)));
    return {std::cos(2 * (a0 - a1)), -std::sin(2 * (a0 - a1))};
}

// ---------------------------------------------------------------------------------------------------------------------------
// Flattened rays
// ---------------------------------------------------------------------------------------------------------------------------
struct Csr {
    std::vector<int> start{0}, idx;
    std::vector<double> w;
    void begin_row() {}
    void push(int i, double v) { idx.push_back(i); w.push_back(v); }
    void end_row() { start.push_back((int)idx.size()); }
    int rows() const { return (int)start.size() - 1; }
};

// One set of traced rays (the lines of sight, or the incoming rays of all diffuse points) with everything the single-scatter source
// and the source integration need per layer.  Layers are stored far end first, as the reference traces them.
struct RaySet {
    int nrays = 0;
    std::vector<int> layer_start;               // [nrays + 1] into the layer arrays
    std::vector<int> endpoint_start;            // [nrays + 1] into the endpoint arrays (layers + 1 per ray, 0 for an empty ray)
    std::vector<char> ground_is_hit;            // [nrays]
    std::vector<double> ground_mu_in;           // [nrays] cosine of the solar zenith angle at the ground intersection (<= 0: no direct beam)
    // outgoing zenith cosine and azimuth difference there (SingleScatterSource::ground_scattering_geometry, cpp/lib/solar/singlescattersource.cpp:264-279):
    // what a non-Lambertian BRDF is evaluated at for the direct bounce
    std::vector<double> ground_mu_out, ground_phi;
    // per layer
    Csr od;                                     // shell optical depth = od . extinction
    std::vector<int> stencil_idx;               // [nlayers][4] grid indices (sorted union of the two endpoints), -1 = unused
    std::vector<double> w_entrance, w_exit;     // [nlayers][4] interpolation weights of the layer's entrance (observer side) / exit
    std::vector<double> quad_start, quad_end;   // od_quad_start_fraction * layer_distance, od_quad_end_fraction * layer_distance
    std::vector<double> frac_start, frac_end;   // od_quad_start_fraction, od_quad_end_fraction (emission sources)
    std::vector<double> distance;               // layer_distance
    std::vector<char> tiny;                     // layer_distance < MINIMUM_SHELL_SIZE_M: no single-scatter source
    std::vector<int> mid_idx;                   // [nlayers][2] atmosphere stencil of the layer mid point (single-scatter albedo of the diffuse source)
    std::vector<double> mid_w;
    // per endpoint (index_map of the reference: endpoint e0 + j is the exit of layer j, e0 + j + 1 its entrance)
    Csr solar;                                  // exact: solar optical depth = solar . extinction; table: interpolation over the table points
    std::vector<char> solar_blocked;            // exact: the solar ray hits the ground
    // per ray: scattering angle of the single scatter, Stokes rotation to the LOS reference frame, Wigner functions
    std::vector<double> theta, C2, S2;          // [nrays]
    std::vector<double> d00, d02;               // [nrays][nmoments]
    int nlayers() const { return layer_start.empty() ? 0 : layer_start.back(); }
};

struct SolarTable {                             // SolarTransmissionTable: optical depth toward the sun on an (altitude, cos sza) grid
    int nalt = 0, ncsz = 0;
    Csr od;                                     // [nalt * ncsz] rows over the atmosphere grid (empty row where the sun is below the horizon)
};

struct Settings {
    int nstokes = 1;
    int num_ss_moments = 16;       // Config::num_singlescatter_moments
    int num_ms_moments = 16;       // Config::num_do_streams: the Legendre orders of the diffuse scattering matrices (diffuse_table.cpp:83)
    int num_incoming = 110, num_outgoing = 110;
    int num_sza = 1;               // Config::num_do_sza
    bool single_scatter = true;    // SingleScatterSource::Exact on the lines of sight
    bool solar_table = false;      // SingleScatterSource::SolarTable: solar optical depths of the lines of sight interpolated in an (altitude, cos sza) table
    bool multiple_scatter = true;  // MultipleScatterSource::SuccessiveOrdersLegacy
    // line-of-sight source terms without scattering (cpp/lib/emission/emission_source.cpp, cpp/lib/solar/occultation.cpp)
    int emission = 0;              // 0 none, 1 EmissionSource::standard ((1 - ssa) x emission_source, e.g. Planck), 2 volume_emission_rate
    bool occultation = false;      // OccultationSource::standard: unit source behind rays that leave the atmosphere (radiance = transmission)
};

struct SoGeometry {
    Settings cfg;
    int ngrid = 0, nlos = 0;
    RaySet los;                                 // exact solar transmission
    std::vector<double> stokes_C, stokes_S;     // [nlos] rotation to the configured Stokes basis (standard: 1, 0)
    // diffuse field
    int npoints = 0, ninterior = 0;
    std::vector<int> in_start, out_start;       // [npoints + 1] offsets of the incoming rays / outgoing directions of each point
    std::vector<int> point_pair;                // [npoints] sphere pair: 0 = interior, 1 + g = ground point g
    std::vector<int> point_leg_idx;             // [npoints][2] atmosphere stencil of the point (Legendre coefficients of its scattering matrix)
    std::vector<double> point_leg_w;
    RaySet incoming;                            // table solar transmission (`solar` rows index the table points)
    SolarTable table;
    Csr accumulation;                           // [incoming rays] over outgoing directions: interior weight (before omega (1 - e^-od) e^-od_before)
    std::vector<int> accumulation_layer;        // per accumulation entry: layer index in `incoming` (-1: ground term)
    Csr los_source, los_ground;                 // [los layers] / [los rays] over outgoing directions
    // scattering matrices: pair 0 per Legendre order and Greek constant (1 for NSTOKES = 1; a1, a2, a3, b1 for NSTOKES = 3),
    // (nout S) x (nin S) row major; ground pairs: 4 pi mu_in w_in per incoming direction
    std::vector<double> scat_basis;             // [num_ms_moments][ngreek][nout S][nin S]
    int nin = 0, nout = 0;
    std::vector<double> in_dir, out_dir, in_weight;   // interior sphere pair: unit vectors [n][3], incoming quadrature weights
    std::vector<std::vector<double>> ground_in_factor;   // [ground point][nin_g]
    std::vector<int> ground_nin, ground_nout;
};

namespace detail {
inline void append_rays(const skgeo::Geometry1D& geo, const std::vector<skgeo::TracedRay>& rays, int nmoments, int nstokes, RaySet& rs) {
    rs.nrays = (int)rays.size();
    rs.layer_start.assign(1, 0);
    rs.endpoint_start.assign(1, 0);
    for (const auto& ray : rays) {
        const int n = (int)ray.layers.size();
        rs.ground_is_hit.push_back(ray.ground_is_hit ? 1 : 0);
        double mu_in = 0, mu_out = 0, saa = 0;
        if (ray.ground_is_hit && n > 0) {
            skgeo::calculate_csz_saz(geo.coords.sun, ray.layers[0].exit, ray.layers[0].average_look_away, mu_in, saa, geo.coords.geotype);
            mu_out = -ray.layers[0].exit.cos_zenith_angle(ray.layers[0].average_look_away);
        }
        rs.ground_mu_in.push_back(mu_in);
        rs.ground_mu_out.push_back(mu_out);
        rs.ground_phi.push_back(saa);
        for (const auto& L : ray.layers) {
            for (int i = 0; i < L.nw; ++i)
                if (L.w_od[i] != 0.0) rs.od.push(L.widx[i], L.w_od[i]);
            rs.od.end_row();
            for (int i = 0; i < 4; ++i) {
                rs.stencil_idx.push_back(i < L.nw ? L.widx[i] : -1);
                rs.w_entrance.push_back(i < L.nw ? L.w_entrance[i] : 0.0);
                rs.w_exit.push_back(i < L.nw ? L.w_exit[i] : 0.0);
            }
            // od_quad_*_fraction (raytracing.h:486-560): 1/2 each for tiny / lower-interpolated layers, else the share of the quadrature weight
            double fs = 0.5, fe = 0.5;
            const double dr = L.exit.radius() - L.entrance.radius();
            if (geo.alt.interp != skgeo::lower && !(std::abs(dr) < 0.001 || geo.coords.geotype == skgeo::planeparallel || geo.alt.interp == skgeo::shell)) {
                fs = L.od_quad_start / (L.od_quad_start + L.od_quad_end);
                fe = L.od_quad_end / (L.od_quad_start + L.od_quad_end);
            }
            rs.quad_start.push_back(fs * L.layer_distance);
            rs.quad_end.push_back(fe * L.layer_distance);
            rs.frac_start.push_back(fs);
            rs.frac_end.push_back(fe);
            rs.distance.push_back(L.layer_distance);
            rs.tiny.push_back(L.layer_distance < kMinimumShellSize ? 1 : 0);
            skgeo::Location mid;
            mid.position = (L.entrance.position + L.exit.position) / 2.0;
            std::vector<std::pair<int, double>> mw;
            geo.interpolation_weights(mid, mw);
            for (int i = 0; i < 2; ++i) {
                rs.mid_idx.push_back(i < (int)mw.size() ? mw[i].first : 0);
                rs.mid_w.push_back(i < (int)mw.size() ? mw[i].second : 0.0);
            }
        }
        rs.layer_start.push_back(rs.layer_start.back() + n);
        rs.endpoint_start.push_back(rs.endpoint_start.back() + (n > 0 ? n + 1 : 0));
        // one scattering angle per (straight) ray, from the far-end layer (phasehandler.cpp:247-262)
        double th = 0, C1 = 1, C2 = 1, S1 = 0, S2 = 0;
        if (n > 0) {
            const Vec3 look = ray.layers[0].average_look_away;
            stokes_scattering_factors(geo.coords.sun * -1.0, look * -1.0, th, C1, C2, S1, S2);
            if (nstokes == 3) {
                const auto r = stokes_standard_to_observer_z(geo.coords, look, ray.observer_and_look.observer.position);
                const double c2 = C2 * r.first - S2 * r.second, s2 = C2 * r.second + S2 * r.first;
                C2 = c2;
                S2 = s2;
            }
        }
        rs.theta.push_back(th);
        rs.C2.push_back(C2);
        rs.S2.push_back(S2);
        std::vector<double> d(nmoments);
        skgeo::wigner_d_m0(0, std::cos(th), nmoments, d.data());
        rs.d00.insert(rs.d00.end(), d.begin(), d.end());
        skgeo::wigner_d_mn(0, 2, std::cos(th), nmoments, d.data());
        rs.d02.insert(rs.d02.end(), d.begin(), d.end());
    }
}

// optical-depth row of the ray from `from` toward the sun, or blocked
inline bool solar_row(const skgeo::Geometry1D& geo, const skgeo::SphericalShellRayTracer& tracer, const skgeo::Location& from, std::vector<double>& dense) {
    skgeo::ViewingRay to_sun;
    to_sun.observer = from;
    to_sun.look_away = geo.coords.sun;
    skgeo::TracedRay tr;
    tracer.trace(to_sun, tr);
    std::fill(dense.begin(), dense.end(), 0.0);
    if (tr.ground_is_hit) return false;
    for (const auto& L : tr.layers)
        for (int i = 0; i < L.nw; ++i) dense[L.widx[i]] += L.w_od[i];
    return true;
}
inline void push_dense(Csr& c, const std::vector<double>& dense) {
    for (int i = 0; i < (int)dense.size(); ++i)
        if (dense[i] != 0.0) c.push(i, dense[i]);
    c.end_row();
}

// exact solar transmission at every layer boundary of every ray (SolarTransmissionExact::generate_geometry_matrix)
inline void exact_solar(const skgeo::Geometry1D& geo, const skgeo::SphericalShellRayTracer& tracer, const std::vector<skgeo::TracedRay>& rays, RaySet& rs) {
    std::vector<double> dense(geo.ngrid());
    for (const auto& ray : rays)
        for (int j = 0; j < (int)ray.layers.size(); ++j) {
            if (j == 0) {
                const bool ok = solar_row(geo, tracer, ray.layers[j].exit, dense);
                rs.solar_blocked.push_back(ok ? 0 : 1);
                push_dense(rs.solar, dense);
            }
            const bool ok = solar_row(geo, tracer, ray.layers[j].entrance, dense);
            rs.solar_blocked.push_back(ok ? 0 : 1);
            push_dense(rs.solar, dense);
        }
}

struct LocationInterpolator {   // AltitudeSZASourceLocationInterpolator
    skgeo::Grid alt, csz;
    int num_ground() const { return csz.size(); }
    int num_interior() const { return csz.size() * alt.size(); }
    Vec3 grid_location(const skgeo::Coordinates& c, int i) const { return c.solar_coordinate_vector(csz.v[i / alt.size()], 0.0, alt.v[i % alt.size()]); }
    Vec3 ground_location(const skgeo::Coordinates& c, int g) const { return c.solar_coordinate_vector(csz.v[g], 0.0, 0.0); }
    void interior_weights(const skgeo::Coordinates& c, const Vec3& position, std::vector<std::pair<int, double>>& out) const {
        std::array<int, 2> ai, si;
        std::array<double, 2> aw, sw;
        int na, ns;
        const double altitude = (c.geotype == skgeo::spherical) ? position.norm() - c.earth_radius : position.z - c.earth_radius;
        const double cos_sza = position.normalized().dot(c.sun);
        csz.weights(cos_sza, si, sw, ns);
        alt.weights(altitude, ai, aw, na);
        out.resize(size_t(na) * ns);
        for (int i = 0; i < na; ++i)
            for (int j = 0; j < ns; ++j) out[i + j * na] = {ai[i] + si[j] * alt.size(), sw[j] * aw[i]};
    }
    void ground_weights(const skgeo::Coordinates& c, const Vec3& position, std::vector<std::pair<int, double>>& out) const {
        std::array<int, 2> si;
        std::array<double, 2> sw;
        int ns;
        csz.weights(position.normalized().dot(c.sun), si, sw, ns);
        out.resize(ns);
        for (int j = 0; j < ns; ++j) out[j] = {num_interior() + si[j], sw[j]};
    }
};

inline std::pair<double, double> min_max_cos_sza(const std::vector<skgeo::TracedRay>& rays) {
    double lo = 1, hi = -1;
    for (const auto& r : rays)
        for (const auto& L : r.layers) {
            lo = std::min(std::min(L.cos_sza_entrance, L.cos_sza_exit), lo);
            hi = std::max(std::max(L.cos_sza_entrance, L.cos_sza_exit), hi);
        }
    return {lo, hi};
}
}  // namespace detail

// Solar transmission of SingleScatterSource<SolarTransmissionTable> at every layer boundary of every ray: the optical depth toward the sun
// interpolated (bilinear in altitude and cos sza) in a table over the altitude grid x 100 cosines between the smallest and largest cos(SZA) the
// rays meet (SolarTransmissionTable::initialize_geometry / generate_interpolation_matrix, solartransmissiontable.cpp:113-238).  Optical depth
// is linear in the extinction profile, so the interpolation is folded into one stencil over the atmosphere grid per boundary -- the consumers
// see the same rows as for the exact source.  Table points whose solar ray hits the ground keep an empty row (optical depth 0) and nothing
// is flagged as blocked, as in the reference.
inline void table_solar(const skgeo::Geometry1D& geo, const skgeo::SphericalShellRayTracer& tracer, const std::vector<skgeo::TracedRay>& rays, RaySet& rs) {
    using namespace detail;
    const auto mm = min_max_cos_sza(rays);
    LocationInterpolator ti;
    ti.alt = skgeo::Grid(geo.alt.v, skgeo::constant_spacing, skgeo::linear);
    ti.csz = skgeo::Grid(skgeo::linspaced(100, mm.first, mm.second), skgeo::constant_spacing, skgeo::linear);
    const int ng = geo.ngrid();
    std::vector<double> table(size_t(ti.num_interior()) * ng, 0.0), dense(ng);
    for (int i = 0; i < ti.num_interior(); ++i) {
        skgeo::Location from;
        from.position = ti.grid_location(geo.coords, i);
        if (solar_row(geo, tracer, from, dense)) std::copy(dense.begin(), dense.end(), table.begin() + size_t(i) * ng);
    }
    std::vector<std::pair<int, double>> w;
    auto endpoint = [&](const skgeo::Location& loc) {
        ti.interior_weights(geo.coords, loc.position, w);
        std::fill(dense.begin(), dense.end(), 0.0);
        for (auto& e : w)
            for (int q = 0; q < ng; ++q) dense[q] += e.second * table[size_t(e.first) * ng + q];
        push_dense(rs.solar, dense);
        rs.solar_blocked.push_back(0);
    };
    for (const auto& ray : rays)
        for (int j = 0; j < (int)ray.layers.size(); ++j) {
            if (j == 0) endpoint(ray.layers[j].exit);
            endpoint(ray.layers[j].entrance);
        }
}

// Builds everything.  `specs` are the lines of sight.
inline void build_so_geometry(const skgeo::Geometry1D& geo, const std::vector<skgeo::RaySpec>& specs, const Settings& cfg, SoGeometry& g, int stokes_basis = 0) {
    using namespace detail;
    if (geo.coords.geotype != skgeo::spherical) throw std::invalid_argument("the exact single-scatter / successive-orders sources need spherical geometry");
    if (geo.alt.interp == skgeo::lower) throw std::invalid_argument("lower interpolation is not supported by the successive-orders path");
    if (cfg.nstokes != 1 && cfg.nstokes != 3) throw std::invalid_argument("NSTOKES must be 1 or 3");
    g = SoGeometry();
    g.cfg = cfg;
    g.ngrid = geo.ngrid();
    g.nlos = (int)specs.size();
    const int S = cfg.nstokes;
    skgeo::SphericalShellRayTracer tracer(geo);
    std::vector<skgeo::TracedRay> los(specs.size());
    for (size_t i = 0; i < specs.size(); ++i) {
        tracer.trace(skgeo::construct_ray(specs[i], geo.coords), los[i]);
        double C = 1, Sn = 0;
        if (S == 3 && stokes_basis == 1) geo.coords.stokes_standard_to_solar(los[i].observer_and_look.look_away, C, Sn);
        if (S == 3 && stokes_basis == 2) geo.coords.stokes_standard_to_observer(los[i].observer_and_look.look_away, los[i].observer_and_look.observer.position, C, Sn);
        g.stokes_C.push_back(C);
        g.stokes_S.push_back(Sn);
    }
    append_rays(geo, los, cfg.num_ss_moments, S, g.los);
    if (cfg.solar_table) table_solar(geo, tracer, los, g.los);
    else exact_solar(geo, tracer, los, g.los);
    if (!cfg.multiple_scatter) return;

    // ---- diffuse points (DiffuseTable::initialize_geometry, construct_diffuse_points)
    const auto mm = min_max_cos_sza(los);
    LocationInterpolator li;
    {
        std::vector<double> mid(geo.ngrid() - 1);
        for (int i = 0; i + 1 < geo.ngrid(); ++i) mid[i] = (geo.alt.v[i] + geo.alt.v[i + 1]) / 2.0;
        li.alt = skgeo::Grid(mid, skgeo::constant_spacing, skgeo::linear);
        std::vector<double> cs = cfg.num_sza > 1 ? skgeo::linspaced(cfg.num_sza, mm.first, mm.second) : std::vector<double>{geo.coords.cos_sza_at_reference()};
        li.csz = skgeo::Grid(cs, skgeo::constant_spacing, skgeo::linear);
    }
    const UnitSphere full_in = UnitSphere::lebedev(cfg.num_incoming), full_out = UnitSphere::lebedev(cfg.num_outgoing);
    std::vector<UnitSphere> in_spheres{full_in}, out_spheres{full_out};
    for (int gidx = 0; gidx < li.num_ground(); ++gidx) {
        const Vec3 loc = li.ground_location(geo.coords, gidx);
        in_spheres.push_back(UnitSphere::ground_of(full_in, loc));
        out_spheres.push_back(UnitSphere::ground_of(full_out, loc));
    }
    g.ninterior = li.num_interior();
    g.npoints = g.ninterior + li.num_ground();
    g.nin = full_in.size();
    g.nout = full_out.size();
    for (int i = 0; i < full_in.size(); ++i) {
        g.in_dir.insert(g.in_dir.end(), {full_in.dir[i].x, full_in.dir[i].y, full_in.dir[i].z});
        g.in_weight.push_back(full_in.weight[i]);
    }
    for (int o = 0; o < full_out.size(); ++o) g.out_dir.insert(g.out_dir.end(), {full_out.dir[o].x, full_out.dir[o].y, full_out.dir[o].z});
    std::vector<Vec3> point_position(g.npoints);
    g.in_start.assign(1, 0);
    g.out_start.assign(1, 0);
    for (int p = 0; p < g.npoints; ++p) {
        Vec3 pos;
        if (p < g.ninterior) {
            pos = li.grid_location(geo.coords, p);
            g.point_pair.push_back(0);
        } else {
            pos = li.ground_location(geo.coords, p - g.ninterior);
            pos = pos + pos.normalized() * 0.01;   // 1 cm above the surface (diffuse_table.cpp:173-174)
            g.point_pair.push_back(1 + p - g.ninterior);
        }
        point_position[p] = pos;
        skgeo::Location loc;
        loc.position = pos;
        std::vector<std::pair<int, double>> lw;
        geo.interpolation_weights(loc, lw);
        for (int i = 0; i < 2; ++i) {
            g.point_leg_idx.push_back(i < (int)lw.size() ? lw[i].first : 0);
            g.point_leg_w.push_back(i < (int)lw.size() ? lw[i].second : 0.0);
        }
        g.in_start.push_back(g.in_start.back() + in_spheres[g.point_pair[p]].size());
        g.out_start.push_back(g.out_start.back() + out_spheres[g.point_pair[p]].size());
    }
    // ---- incoming rays
    std::vector<skgeo::TracedRay> inc(g.in_start.back());
    for (int p = 0; p < g.npoints; ++p) {
        const UnitSphere& sp = in_spheres[g.point_pair[p]];
        for (int j = 0; j < sp.size(); ++j) {
            skgeo::ViewingRay vr;
            vr.observer.position = point_position[p];
            vr.look_away = sp.dir[j];
            tracer.trace(vr, inc[g.in_start[p] + j]);
        }
    }
    append_rays(geo, inc, cfg.num_ss_moments, S, g.incoming);
    // ---- solar transmission table of the first-order source on the incoming rays (SolarTransmissionTable::initialize_geometry)
    {
        const auto mmi = min_max_cos_sza(inc);
        LocationInterpolator ti;
        ti.alt = skgeo::Grid(geo.alt.v, skgeo::constant_spacing, skgeo::linear);
        ti.csz = skgeo::Grid(skgeo::linspaced(100, mmi.first, mmi.second), skgeo::constant_spacing, skgeo::linear);
        g.table.nalt = ti.alt.size();
        g.table.ncsz = ti.csz.size();
        std::vector<double> dense(geo.ngrid());
        for (int i = 0; i < ti.num_interior(); ++i) {
            skgeo::Location from;
            from.position = ti.grid_location(geo.coords, i);
            solar_row(geo, tracer, from, dense);   // blocked rows stay empty: optical depth 0 (the reference never zeroes them for this source)
            push_dense(g.table.od, dense);
        }
        std::vector<std::pair<int, double>> w;
        auto endpoint = [&](const skgeo::Location& loc) {
            ti.interior_weights(geo.coords, loc.position, w);
            for (auto& e : w) g.incoming.solar.push(e.first, e.second);
            g.incoming.solar.end_row();
            g.incoming.solar_blocked.push_back(0);
        };
        for (const auto& ray : inc)
            for (int j = 0; j < (int)ray.layers.size(); ++j) {
                if (j == 0) endpoint(ray.layers[j].exit);
                endpoint(ray.layers[j].entrance);
            }
    }
    // ---- interpolation of the outgoing sources along rays (generate_source_interpolation_weights)
    auto rotate_unit_vector = [&](const Vec3& v, const Vec3& from, const Vec3& to) {
        skgeo::Location t;
        t.position = from;
        double csz, saa;
        skgeo::calculate_csz_saz(geo.coords.sun, t, v, csz, saa);
        return geo.coords.look_vector_from_azimuth(to, -saa, t.cos_zenith_angle(v));
    };
    auto source_weights = [&](const std::vector<skgeo::TracedRay>& rays, Csr& interior, Csr& ground, std::vector<int>* layer_of, int layer0) {
        std::vector<std::pair<int, double>> lw;
        std::array<int, 3> di;
        std::array<double, 3> dw;
        int layer = layer0;
        for (const auto& ray : rays) {
            for (const auto& L : ray.layers) {
                const Vec3 mid = (L.entrance.position + L.exit.position) / 2.0;
                li.interior_weights(geo.coords, mid, lw);
                for (const auto& pw : lw) {
                    const int p = pw.first;
                    const Vec3 rot = rotate_unit_vector(L.average_look_away, mid, point_position[p]);
                    out_spheres[g.point_pair[p]].interpolate(rot, di, dw);
                    for (int k = 0; k < 3; ++k) interior.push(g.out_start[p] + di[k], pw.second * dw[k]);
                }
                interior.end_row();
                (void)layer_of;
                ++layer;
            }
            if (ray.ground_is_hit && !ray.layers.empty()) {
                const auto& L = ray.layers[0];
                li.ground_weights(geo.coords, L.exit.position, lw);
                for (const auto& pw : lw) {
                    const int p = pw.first;
                    const Vec3 rot = rotate_unit_vector(L.average_look_away * -1.0, L.exit.position, point_position[p]);
                    out_spheres[g.point_pair[p]].interpolate(rot, di, dw);
                    for (int k = 0; k < 3; ++k) ground.push(g.out_start[p] + di[k], pw.second * dw[k]);
                }
            }
            ground.end_row();
        }
    };
    {
        // accumulation matrix rows: all layer entries of a ray (near layer last) followed by its ground entries; `accumulation_layer`
        // tells the consumer which optical-depth factor an entry takes
        Csr interior, ground;
        source_weights(inc, interior, ground, nullptr, 0);
        for (int r = 0; r < (int)inc.size(); ++r) {
            for (int l = g.incoming.layer_start[r]; l < g.incoming.layer_start[r + 1]; ++l)
                for (int e = interior.start[l]; e < interior.start[l + 1]; ++e) {
                    g.accumulation.push(interior.idx[e], interior.w[e]);
                    g.accumulation_layer.push_back(l);
                }
            for (int e = ground.start[r]; e < ground.start[r + 1]; ++e) {
                g.accumulation.push(ground.idx[e], ground.w[e]);
                g.accumulation_layer.push_back(-1);
            }
            g.accumulation.end_row();
        }
    }
    source_weights(los, g.los_source, g.los_ground, nullptr, 0);
    // ---- scattering matrices of the interior sphere pair per Legendre order (IncomingOutgoingSpherePair::configure_geometry)
    {
        const int ngreek = (S == 1) ? 1 : 4, nin = g.nin, nout = g.nout, rows = nout * S, cols = nin * S;
        g.scat_basis.assign(size_t(cfg.num_ms_moments) * ngreek * rows * cols, 0.0);
        std::vector<double> d00(cfg.num_ms_moments), d22(cfg.num_ms_moments), d02(cfg.num_ms_moments), d2m2(cfg.num_ms_moments);
        for (int i = 0; i < nin; ++i)
            for (int o = 0; o < nout; ++o) {
                auto at = [&](int l, int greek, int r, int c) -> double& {
                    return g.scat_basis[((size_t(l) * ngreek + greek) * rows + (size_t(o) * S + r)) * cols + size_t(i) * S + c];
                };
                const double wq = full_in.weight[i];
                if (S == 1) {
                    const double ct = clamp1(full_in.dir[i].dot(full_out.dir[o]));
                    skgeo::wigner_d_m0(0, std::cos(std::acos(ct)), cfg.num_ms_moments, d00.data());
                    for (int l = 0; l < cfg.num_ms_moments; ++l) at(l, 0, 0, 0) = wq * d00[l];
                    continue;
                }
                double th, C1, C2, S1, S2;
                stokes_scattering_factors(full_in.dir[i] * -1.0, full_out.dir[o] * -1.0, th, C1, C2, S1, S2);
                const double x = std::cos(th);
                skgeo::wigner_d_m0(0, x, cfg.num_ms_moments, d00.data());
                skgeo::wigner_d_mn(2, 2, x, cfg.num_ms_moments, d22.data());
                skgeo::wigner_d_mn(0, 2, x, cfg.num_ms_moments, d02.data());
                skgeo::wigner_d_mn(2, -2, x, cfg.num_ms_moments, d2m2.data());
                for (int l = 0; l < cfg.num_ms_moments; ++l) {
                    const double wa = d22[l] + d2m2[l], wm = d22[l] - d2m2[l];
                    at(l, 0, 0, 0) = d00[l];
                    at(l, 3, 1, 0) = -C2 * d02[l];
                    at(l, 3, 2, 0) = -S2 * d02[l];
                    at(l, 3, 0, 1) = -C1 * d02[l];
                    at(l, 1, 1, 1) += 0.5 * C1 * C2 * wa - 0.5 * S1 * S2 * wm;
                    at(l, 2, 1, 1) += 0.5 * C1 * C2 * wm - 0.5 * S1 * S2 * wa;
                    at(l, 1, 2, 1) += 0.5 * C1 * S2 * wa + 0.5 * S1 * C2 * wm;
                    at(l, 2, 2, 1) += 0.5 * C1 * S2 * wm + 0.5 * S1 * C2 * wa;
                    at(l, 3, 0, 2) = S1 * d02[l];
                    at(l, 1, 1, 2) += -0.5 * S1 * C2 * wa - 0.5 * C1 * S2 * wm;
                    at(l, 2, 1, 2) += -0.5 * S1 * C2 * wm - 0.5 * C1 * S2 * wa;
                    at(l, 1, 2, 2) += -0.5 * S1 * S2 * wa + 0.5 * C1 * C2 * wm;
                    at(l, 2, 2, 2) += -0.5 * S1 * S2 * wm + 0.5 * C1 * C2 * wa;
                    for (int greek = 0; greek < 4; ++greek)
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) at(l, greek, r, c) *= wq;
                }
            }
    }
    // ---- ground pairs: outgoing = 4 pi brdf mu_in w_in (calculate_ground_scattering_matrix), same for every outgoing direction
    for (int gidx = 0; gidx < li.num_ground(); ++gidx) {
        const UnitSphere& sp = in_spheres[1 + gidx];
        skgeo::Location loc;
        loc.position = point_position[g.ninterior + gidx];
        std::vector<double> f(sp.size());
        for (int i = 0; i < sp.size(); ++i) f[i] = 4 * skgeo::kPi * loc.cos_zenith_angle(sp.dir[i]) * sp.weight[i];
        g.ground_in_factor.push_back(f);
        g.ground_nin.push_back(sp.size());
        g.ground_nout.push_back(out_spheres[1 + gidx].size());
    }
}
}  // namespace skso
