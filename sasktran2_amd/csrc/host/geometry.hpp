// Wavelength-independent geometry precompute for the DISCO hot path (host only).
//
// Everything in this header runs ONCE per engine on the CPU and produces flat
// tables that the per-wavelength solver (HIP kernels, or the CPU oracle) reads:
// stream quadrature, traced lines of sight, sparse optical-depth rows, Chapman
// factors, the diffuse-source table layout and the LOS->table interpolators.
//
// Behaviour follows the reference (paths relative to /root/reference):
//   grids:            cpp/lib/grids/grid.cpp:5-299
//   coordinates:      cpp/lib/geometry/geometry.cpp:7-228
//   viewing rays:     cpp/lib/viewinggeometry/{tangentaltitudesolar,groundviewing}.cpp
//   shell ray tracer: cpp/lib/raytracing/spherical_shell.cpp:6-552,
//                     cpp/include/sasktran2/raytracing.h:319-638,826-866
//   1D interpolation: cpp/lib/geometry/geometry1d.cpp:20-78
//   DO layer geometry / Chapman: cpp/lib/sktran_disco/sktran_do_geometrylayerarray.cpp:6-185
//   SZA grid:         cpp/lib/sktran_disco/source_term/do_source.cpp:62-124
//   diffuse table:    cpp/lib/sktran_disco/source_term/do_source_diffuse_storage.cpp:7-267,416-433
//   plane-parallel LOS: cpp/lib/sktran_disco/source_term/do_source_planeparallel.cpp:598-640
//
// It is shared by the product host code and by the CPU oracle; it is pinned by
// the reference's LOS optical-depth and radiance regression values
// (tests/engine/test_1d_solver_regression.py:120-139,210-230).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace skgeo {

constexpr double kPi = 3.14159265358979323846;

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vec3 operator/(double s) const { return {x / s, y / s, z / s}; }
    double dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3& o) const {
        return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x};
    }
    double norm() const { return std::sqrt(dot(*this)); }
    Vec3 normalized() const {
        double n = norm();
        return n > 0 ? (*this) / n : *this;
    }
};
inline Vec3 operator*(double s, const Vec3& v) { return v * s; }

// Rodrigues rotation of v about unit axis k by angle (what Eigen::AngleAxis does)
inline Vec3 rotate(const Vec3& v, const Vec3& k, double angle) {
    double c = std::cos(angle), s = std::sin(angle);
    return v * c + k.cross(v) * s + k * (k.dot(v) * (1 - c));
}

enum GeometryType { planeparallel = 0, pseudospherical = 1, spherical = 2, ellipsoidal = 3 };
enum Interpolation { shell = 0, linear = 1, lower = 2 };
enum Spacing { constant_spacing, variable_spacing, automatic_spacing };

// ---------------------------------------------------------------------------
// 1-D grid with the reference's interpolation rules (out of bounds = extend)
// ---------------------------------------------------------------------------
struct Grid {
    std::vector<double> v;
    Spacing spacing = variable_spacing;
    Interpolation interp = linear;
    double x0 = 0, dx = 0;

    Grid() = default;
    Grid(std::vector<double> values, Spacing sp, Interpolation in) : v(std::move(values)), spacing(sp), interp(in) {
        if (spacing == automatic_spacing) {
            spacing = constant_spacing;
            if (v.size() > 1) {
                double d0 = v[1] - v[0];
                for (size_t i = 1; i < v.size(); ++i) {
                    double d = v[i] - v[i - 1];
                    if (!(std::abs(d - d0) <= std::min(std::abs(d), std::abs(d0)) * 1e-12)) {
                        spacing = variable_spacing;
                        break;
                    }
                }
            }
        }
        if (spacing == constant_spacing && v.size() > 1) {
            x0 = v[0];
            dx = v[1] - v[0];
        }
    }
    int size() const { return (int)v.size(); }

    void weights(double x, std::array<int, 2>& idx, std::array<double, 2>& w, int& n) const {
        const int N = size();
        idx = {0, 0};
        w = {1.0, 0.0};
        n = 1;
        if (N == 1) return;
        if (interp == lower) {
            for (int i = 0; i < N - 1; ++i) {
                if (x + 0.1 >= v[i] && x < v[i + 1]) {
                    idx[0] = i;
                    return;
                }
            }
            if (x < v[0]) { idx[0] = 0; return; }
            idx[0] = N - 2;
            return;
        }
        if (spacing == constant_spacing) {
            if (x < x0) { idx[0] = 0; return; }
            int i = (int)std::floor((x - x0) / dx);
            if (i >= N - 1) { idx[0] = N - 1; return; }
            idx = {i, i + 1};
            if (interp == linear) {
                w[1] = (x - v[i]) / dx;
                w[0] = 1 - w[1];
            } else {
                w = {0.5, 0.5};
            }
            n = 2;
            return;
        }
        if (x < v[0]) { idx[0] = 0; return; }
        if (x > v[N - 1]) { idx[0] = N - 1; return; }
        int i = (int)(std::lower_bound(v.begin(), v.end(), x) - v.begin());
        if (i == 0) i = 1;
        idx = {i - 1, i};
        if (interp == shell) {
            w = {0.5, 0.5};
        } else {
            w[1] = (x - v[i - 1]) / (v[i] - v[i - 1]);
            w[0] = 1 - w[1];
        }
        n = 2;
    }
};

inline std::vector<double> linspaced(int n, double lo, double hi) {
    std::vector<double> r(n);
    if (n == 1) { r[0] = hi; return r; }
    double step = (hi - lo) / (n - 1);
    bool flip = std::abs(hi) < std::abs(lo);
    for (int i = 0; i < n; ++i) {
        if (!flip) r[i] = (i == n - 1) ? hi : lo + i * step;
        else r[i] = (i == 0) ? lo : hi - (n - 1 - i) * step;
    }
    return r;
}

// ---------------------------------------------------------------------------
// Coordinates (reference: Coordinates::Coordinates(cos_sza, saa, ...))
// ---------------------------------------------------------------------------
struct Coordinates {
    Vec3 xu{1, 0, 0}, yu{0, 1, 0}, zu{0, 0, 1}, sun;
    double earth_radius = 0;
    GeometryType geotype = spherical;

    Coordinates() = default;
    Coordinates(double cos_sza, double saa, double R, GeometryType g) : earth_radius(R), geotype(g) {
        Vec3 sun_horiz = xu * std::cos(saa) + yu * std::sin(saa);
        sun = zu * cos_sza + sun_horiz * std::sqrt(1 - cos_sza * cos_sza);
    }
    double cos_sza_at_reference() const { return zu.normalized().dot(sun); }

    Vec3 solar_coordinate_vector(double cos_sza, double saa, double altitude) const {
        if (geotype == planeparallel || geotype == pseudospherical) return zu * (altitude + earth_radius);
        Vec3 normal = sun.cross(zu);
        if (normal.norm() == 0) normal = yu; else normal = normal.normalized();
        Vec3 v = rotate(sun, normal, std::acos(cos_sza));
        v = rotate(v, sun, saa);
        return v * (altitude + earth_radius);
    }

    // Rotation (cos 2a, sin 2a) of (Q, U) from the standard basis to the solar / observer bases
    //   cpp/lib/geometry/geometry.cpp:232-306 (stokes_standard_to_solar, stokes_standard_to_observer)
    void stokes_standard_to_solar(const Vec3& look, double& C, double& S) const {
        C = 1.0; S = 0.0;
        if (std::fabs(sun.dot(look)) >= 1 || std::fabs(zu.dot(look)) >= 1) return;
        Vec3 perp_z = (zu - look * zu.dot(look)).normalized();
        Vec3 perp_sun = (sun - look * sun.dot(look)).normalized();
        double ca = std::min(1.0, std::max(-1.0, perp_z.dot(perp_sun)));
        double a = std::acos(ca);
        C = std::cos(2 * a);
        S = -std::sin(2 * a);
    }
    void stokes_standard_to_observer(const Vec3& look, const Vec3& position, double& C, double& S) const {
        C = 1.0; S = 0.0;
        Vec3 pn = position.normalized();
        if (std::fabs(pn.dot(look)) >= 1 || std::fabs(zu.dot(look)) >= 1) return;
        Vec3 perp_z = (zu - look * zu.dot(look)).normalized();
        double ca = std::min(1.0, std::max(-1.0, perp_z.dot(pn)));  // as written in the reference (:291)
        double a = std::acos(ca);
        C = std::cos(2 * a);
        S = std::sin(2 * a);
    }

    Vec3 look_vector_from_azimuth(const Vec3& location, double saa, double cos_viewing) const {
        Vec3 up = (geotype == spherical) ? location.normalized() : zu;
        Vec3 sun_horiz = sun - up * up.dot(sun);
        if (sun_horiz.norm() == 0) sun_horiz = yu;
        sun_horiz = sun_horiz.normalized();
        Vec3 horiz_look = rotate(sun_horiz, up, -saa);
        double viewing_angle = kPi / 2 - std::acos(-cos_viewing);
        return rotate(horiz_look, up.cross(horiz_look), viewing_angle);
    }
};

struct Location {
    Vec3 position;
    bool on_exact_altitude = false;
    int lower_alt_index = -1;
    double radius() const { return position.norm(); }
    double cos_zenith_angle(const Vec3& other) const { return position.dot(other) / (position.norm() * other.norm()); }
};

struct ViewingRay {
    Location observer;
    Vec3 look_away;
    double relative_azimuth = 0;
    double cos_viewing() const { return observer.cos_zenith_angle(look_away); }
};

// user-level ray specifications (cpp/include/c_api/viewing_geometry.h:15-24)
struct RaySpec {
    // viewing_up_solar: observer at `observer_altitude`, look direction given by the zenith cosine and the azimuth relative to
    // the sun at the observer (ViewingUpSolar, cpp/lib/viewinggeometry/groundviewing.cpp:66-87)
    // tangent_altitude: limb ray given by the angle of its tangent point from the reference point in the solar plane and the
    // viewing azimuth in the local horizontal plane there (TangentAltitude, cpp/lib/viewinggeometry/tangentaltitude.cpp:5-71)
    enum Kind { tangent_altitude_solar = 0, ground_viewing_solar = 1, viewing_up_solar = 2, tangent_altitude_angles = 3 } kind;
    double cos_sza, relative_azimuth, observer_altitude;
    double horizontal_angle = 0;  // tangent_altitude (relative_azimuth holds the viewing azimuth)
    double tangent_altitude;    // tangent_altitude_solar
    double cos_viewing_zenith;  // ground_viewing_solar, viewing_up_solar
};

inline ViewingRay construct_ray(const RaySpec& s, const Coordinates& c) {
    ViewingRay ray;
    if (s.kind == RaySpec::tangent_altitude_solar) {
        if (c.geotype != spherical) throw std::invalid_argument("TangentAltitudeSolar rays need spherical geometry");
        Vec3 tp = c.solar_coordinate_vector(s.cos_sza, 0.0, s.tangent_altitude);
        ray.look_away = c.look_vector_from_azimuth(tp, s.relative_azimuth, 0);
        double a = c.earth_radius + s.observer_altitude, b = c.earth_radius + s.tangent_altitude;
        double dist = std::sqrt(a * a - b * b);
        ray.observer.position = tp - ray.look_away * dist;
        ray.relative_azimuth = s.relative_azimuth;
    } else if (s.kind == RaySpec::tangent_altitude_angles) {
        if (c.geotype != spherical) throw std::invalid_argument("TangentAltitude rays need spherical geometry");
        if (!(s.observer_altitude >= s.tangent_altitude)) throw std::invalid_argument("observer below the tangent altitude");
        // rotation of the reference frame about y by the horizontal angle (Coordinates::unit_vector_from_angles / local_x_y_from_angles
        // with zero out-of-plane angle, cpp/lib/geometry/geometry.cpp:102-129)
        Vec3 uv = rotate(c.zu, c.yu, s.horizontal_angle), lx = rotate(c.xu, c.yu, s.horizontal_angle), ly = c.yu;
        Vec3 tp = uv * (c.earth_radius + s.tangent_altitude);
        ray.look_away = lx * std::cos(s.relative_azimuth) + ly * std::sin(s.relative_azimuth);
        double a = c.earth_radius + s.observer_altitude, b = c.earth_radius + s.tangent_altitude;
        ray.observer.position = tp - ray.look_away * std::sqrt(a * a - b * b);
        ray.relative_azimuth = s.relative_azimuth;
    } else if (s.kind == RaySpec::viewing_up_solar) {
        Vec3 at = c.solar_coordinate_vector(s.cos_sza, 0.0, s.observer_altitude);
        ray.look_away = c.look_vector_from_azimuth(at, s.relative_azimuth, s.cos_viewing_zenith);
        ray.observer.position = at;
        ray.relative_azimuth = s.relative_azimuth;
    } else {
        Vec3 ground = c.solar_coordinate_vector(s.cos_sza, 0.0, 0.0);
        ray.look_away = c.look_vector_from_azimuth(ground, -(kPi - s.relative_azimuth), s.cos_viewing_zenith) * -1.0;
        double dist = 0;
        if (c.geotype == planeparallel || c.geotype == pseudospherical) {
            dist = s.observer_altitude / s.cos_viewing_zenith;
        } else {
            double b = 2.0 * c.earth_radius * s.cos_viewing_zenith;
            double cc = -(2.0 * c.earth_radius * s.observer_altitude + s.observer_altitude * s.observer_altitude);
            dist = (-b + std::sqrt(b * b - 4 * cc)) / 2;
        }
        ray.observer.position = ground - ray.look_away * dist;
        ray.relative_azimuth = s.relative_azimuth;
    }
    return ray;
}

// ---------------------------------------------------------------------------
// Traced rays
// ---------------------------------------------------------------------------
struct TracedLayer {
    Location entrance, exit;  // entrance is the end closer to the observer
    double r_entrance = 0, r_exit = 0;
    Vec3 average_look_away;
    double layer_distance = 0, curvature_factor = 1;
    double od_quad_start = 0, od_quad_end = 0;
    double saz_entrance = 0, saz_exit = 0, cos_sza_entrance = 0, cos_sza_exit = 0;
    // sorted union stencil over the altitude grid
    int nw = 0;
    std::array<int, 4> widx{};
    std::array<double, 4> w_entrance{}, w_exit{}, w_od{};
};

struct TracedRay {
    ViewingRay observer_and_look;
    bool ground_is_hit = false;
    double tangent_radius = std::numeric_limits<double>::quiet_NaN();
    std::vector<TracedLayer> layers;  // from the far end of the ray toward the observer
};

inline void calculate_csz_saz(const Vec3& sun, const Location& loc, const Vec3& look, double& csz, double& saa,
                              GeometryType geo = spherical) {
    Vec3 up = (geo == spherical) ? loc.position.normalized() : Vec3(0, 0, 1);
    csz = up.dot(sun);
    Vec3 los_p = (look - up * look.dot(up)).normalized();
    Vec3 sun_p = (sun - up * sun.dot(up)).normalized();
    Vec3 y_axis = up.cross(sun_p);
    saa = std::atan2(y_axis.dot(los_p), sun_p.dot(los_p));
}

struct Geometry1D {
    Coordinates coords;
    Grid alt;  // ascending altitude grid, spacing automatic, extend
    Geometry1D() = default;
    Geometry1D(double cos_sza, double saa, double R, std::vector<double> grid, Interpolation in, GeometryType g)
        : coords(cos_sza, saa, R, g), alt(std::move(grid), automatic_spacing, in) {
        if (alt.size() < 2) throw std::invalid_argument("altitude grid needs at least 2 points");
        for (int i = 1; i < alt.size(); ++i)
            if (!(alt.v[i] > alt.v[i - 1])) throw std::invalid_argument("altitude grid must be strictly increasing");
    }
    int ngrid() const { return alt.size(); }

    // Geometry1D::assign_interpolation_weights
    void interpolation_weights(const Location& loc, std::vector<std::pair<int, double>>& out) const {
        double a = (coords.geotype == planeparallel) ? loc.position.z - coords.earth_radius
                                                     : loc.radius() - coords.earth_radius;
        bool valid = loc.lower_alt_index >= 0 && loc.lower_alt_index < alt.size();
        bool matches = valid && (coords.geotype == pseudospherical || std::abs(a - alt.v[loc.lower_alt_index]) <= 1.0);
        if (loc.on_exact_altitude && matches) {
            out.assign(1, {loc.lower_alt_index, 1.0});
            return;
        }
        std::array<int, 2> idx;
        std::array<double, 2> w;
        int n;
        alt.weights(a, idx, w, n);
        out.resize(n);
        for (int i = 0; i < n; ++i) out[i] = {idx[i], w[i]};
    }
};

class SphericalShellRayTracer {
    const Geometry1D& g_;
    const std::vector<double>& alt_;
    double R_;
    enum Dir { up = -1, down = 1 };

    void complete_layer(TracedLayer& L, size_t exit_index, Dir d) const {
        L.r_entrance = alt_[exit_index + d] + R_;
        L.r_exit = alt_[exit_index] + R_;
        L.entrance.on_exact_altitude = true;
        L.entrance.lower_alt_index = int(exit_index + d);
        L.exit.on_exact_altitude = true;
        L.exit.lower_alt_index = int(exit_index);
    }
    void partial_layer(TracedLayer& L, const ViewingRay& ray, size_t start_index, Dir d) const {
        L.r_entrance = ray.observer.radius();
        L.r_exit = alt_[start_index] + R_;
        L.exit.on_exact_altitude = true;
        L.exit.lower_alt_index = int(start_index);
        L.entrance.on_exact_altitude = false;
        L.entrance.lower_alt_index = d < 0 ? int(start_index + d) : int(start_index);
    }
    void tangent_exact(size_t upper_index, double tangent_altitude, int& tidx, bool& exact) const {
        tidx = int(upper_index) - 1;
        exact = std::abs(tangent_altitude - alt_[tidx]) <= 1e-4;
        if (!exact && std::abs(tangent_altitude - alt_[upper_index]) <= 1e-4) {
            tidx = int(upper_index);
            exact = true;
        }
    }
    void tangent_layer(TracedLayer& L, size_t upper_index, double tangent_altitude, Dir d) const {
        int tidx;
        bool exact;
        tangent_exact(upper_index, tangent_altitude, tidx, exact);
        if (d == up) {
            L.r_entrance = tangent_altitude + R_;
            L.r_exit = alt_[upper_index] + R_;
            L.exit.on_exact_altitude = true;
            L.exit.lower_alt_index = int(upper_index);
            L.entrance.on_exact_altitude = exact;
            L.entrance.lower_alt_index = tidx;
        } else {
            L.r_exit = tangent_altitude + R_;
            L.r_entrance = alt_[upper_index] + R_;
            L.entrance.on_exact_altitude = true;
            L.entrance.lower_alt_index = int(upper_index);
            L.exit.on_exact_altitude = exact;
            L.exit.lower_alt_index = tidx;
        }
    }
    void partial_tangent_layer(TracedLayer& L, const ViewingRay& ray, size_t start_index, double tangent_altitude) const {
        int tidx;
        bool exact;
        tangent_exact(start_index, tangent_altitude, tidx, exact);
        L.entrance.on_exact_altitude = false;
        L.entrance.lower_alt_index = int(start_index) - 1;
        L.exit.on_exact_altitude = exact;
        L.exit.lower_alt_index = tidx;
        L.r_exit = tangent_altitude + R_;
        L.r_entrance = ray.observer.radius();
    }
    size_t upper_index(double altitude) const {
        return size_t(std::upper_bound(alt_.begin(), alt_.end(), altitude) - alt_.begin());
    }
    // distance from the observer to the TOA shell, looking down, near side
    double distance_to_altitude_down_near(const ViewingRay& ray, double altitude) const {
        double cz = std::abs(ray.observer.cos_zenith_angle(ray.look_away));
        double ro = ray.observer.radius();
        double re = R_ + altitude;
        double rtsq = ro * ro * (1 - cz * cz);
        double tangent_distance = ro * cz;
        double from_tangent;
        if (rtsq > re * re) {
            if (std::abs(rtsq - re * re) < 100) from_tangent = 0.0;
            else throw std::runtime_error("requested distance to a shell that does not exist");
        } else {
            from_tangent = std::sqrt(std::abs(re * re - rtsq));
        }
        return tangent_distance - from_tangent;
    }

    void add_od_quadrature(TracedLayer& L) const {
        double r0 = L.entrance.radius(), r1 = L.exit.radius(), dr = r1 - r0;
        L.average_look_away = (L.exit.position - L.entrance.position).normalized();
        double path = L.layer_distance * L.curvature_factor;
        if (g_.alt.interp == lower) {
            if (r0 < r1) { L.od_quad_start = path; L.od_quad_end = 0; }
            else { L.od_quad_start = 0; L.od_quad_end = path; }
            return;
        }
        if (std::abs(dr) < 0.001 || g_.alt.interp == shell) {
            L.od_quad_start = path / 2;
            L.od_quad_end = path / 2;
            return;
        }
        double c0 = L.entrance.cos_zenith_angle(L.average_look_away);
        double c1 = L.exit.cos_zenith_angle(L.average_look_away);
        double t0 = r0 * c0, t1 = r1 * c1;
        double rt = r0 * std::sqrt(1.0 - c0 * c0);
        double dt1, dt2;
        if (t1 >= t0) {
            dt1 = t1 - t0;
            if (std::abs(rt) < 10) dt2 = 0.5 * (r1 * t1 - r0 * t0);
            else dt2 = 0.5 * ((r1 * t1 - r0 * t0) + rt * rt * std::log((r1 + t1) / (r0 + t0)));
        } else {
            dt1 = t0 - t1;
            if (std::abs(rt) < 10) dt2 = 0.5 * (r0 + t0 - r1 * t1);
            else dt2 = 0.5 * ((r0 * t0 - r1 * t1) + rt * rt * std::log((r0 + t0) / (r1 + t1)));
        }
        L.od_quad_start = (r1 * dt1 - dt2) / dr * L.curvature_factor;
        L.od_quad_end = -1 * (r0 * dt1 - dt2) / dr * L.curvature_factor;
    }

    void add_interpolation_weights(TracedLayer& L) const {
        std::vector<std::pair<int, double>> e, x;
        g_.interpolation_weights(L.entrance, e);
        g_.interpolation_weights(L.exit, x);
        int cnt = 0;
        auto add = [&](const std::vector<std::pair<int, double>>& s) {
            for (auto& iw : s) {
                if (iw.second == 0.0) continue;
                if (std::find(L.widx.begin(), L.widx.begin() + cnt, iw.first) == L.widx.begin() + cnt) {
                    if (cnt == 4) throw std::runtime_error("traced layer touches more than four grid points");
                    L.widx[cnt++] = iw.first;
                }
            }
        };
        add(e);
        add(x);
        if (cnt == 0) throw std::runtime_error("unable to resolve traced layer interpolation weights");
        std::sort(L.widx.begin(), L.widx.begin() + cnt);
        L.nw = cnt;
        L.w_entrance = {0, 0, 0, 0};
        L.w_exit = {0, 0, 0, 0};
        auto acc = [&](const std::vector<std::pair<int, double>>& s, std::array<double, 4>& w) {
            for (auto& iw : s) {
                if (iw.second == 0.0) continue;
                auto pos = std::find(L.widx.begin(), L.widx.begin() + cnt, iw.first);
                w[pos - L.widx.begin()] += iw.second;
            }
        };
        acc(e, L.w_entrance);
        acc(x, L.w_exit);
        for (int i = 0; i < cnt; ++i) L.w_od[i] = L.w_entrance[i] * L.od_quad_start + L.w_exit[i] * L.od_quad_end;
    }

    void finalize(TracedRay& r) const {
        const double toa = alt_.back();
        const double rt = r.tangent_radius;
        const int n = (int)r.layers.size();
        for (int i = 0; i < n; ++i) {
            TracedLayer& L = r.layers[n - i - 1];
            if (i == 0) {
                if (r.observer_and_look.observer.radius() - R_ < toa) {
                    L.entrance.position = r.observer_and_look.observer.position;
                } else {
                    L.entrance.position = r.observer_and_look.observer.position +
                                          r.observer_and_look.look_away * distance_to_altitude_down_near(r.observer_and_look, toa);
                }
            } else {
                const TracedLayer& prev = r.layers[n - i];
                // keep this layer's exact-altitude flags but take the position of the previous exit
                L.entrance = prev.exit;
            }
            L.curvature_factor = 1;
            L.layer_distance = std::abs(std::sqrt(std::fmax(L.r_entrance * L.r_entrance - rt * rt, 0)) -
                                        std::sqrt(std::fmax(L.r_exit * L.r_exit - rt * rt, 0.0)));
            L.average_look_away = r.observer_and_look.look_away;
            L.exit.position = L.entrance.position + L.average_look_away * L.layer_distance;
            add_od_quadrature(L);
            add_interpolation_weights(L);
            calculate_csz_saz(g_.coords.sun, L.entrance, L.average_look_away, L.cos_sza_entrance, L.saz_entrance);
            calculate_csz_saz(g_.coords.sun, L.exit, L.average_look_away, L.cos_sza_exit, L.saz_exit);
        }
    }

  public:
    explicit SphericalShellRayTracer(const Geometry1D& g) : g_(g), alt_(g.alt.v), R_(g.coords.earth_radius) {}

    void trace(const ViewingRay& ray, TracedRay& r) const {
        r = TracedRay();
        r.observer_and_look = ray;
        const int ng = (int)alt_.size();
        double cv = ray.cos_viewing();
        double rt = ray.observer.radius() * std::sqrt(std::max(0.0, 1 - cv * cv));
        r.tangent_radius = rt;
        double talt = rt - R_;
        double oalt = ray.observer.radius() - R_;
        if (oalt >= alt_.back()) {
            if (cv > 0) return;  // looking up from outside: empty
            if (talt >= alt_.back()) return;  // misses the atmosphere
            if (talt > alt_[0]) {
                size_t above = upper_index(talt);
                size_t nl = 2 * (ng - above);
                r.layers.resize(nl);
                if (nl == 0) return;
                size_t c = 0;
                for (size_t i = ng - 1; i != above; --i) complete_layer(r.layers[c++], i, up);
                tangent_layer(r.layers[c++], above, talt, up);
                tangent_layer(r.layers[c++], above, talt, down);
                for (int i = (int)above; i < ng - 1; ++i) complete_layer(r.layers[c++], i, down);
            } else {
                r.ground_is_hit = true;
                r.layers.resize(ng - 1);
                for (int i = 0; i < ng - 1; ++i) complete_layer(r.layers[i], i, down);
            }
        } else {
            if (cv > 0) {
                size_t start = upper_index(oalt);
                r.layers.resize(ng - start);
                size_t c = 0;
                for (size_t i = ng - 1; i != start; --i) complete_layer(r.layers[c++], i, up);
                partial_layer(r.layers[c], ray, start, up);
            } else if (talt > alt_[0]) {
                size_t obs = upper_index(oalt);
                size_t above = upper_index(talt);
                int nl = int(ng - 1) - int(above) + 2;
                if (obs > above) nl += int(obs - above);
                r.layers.resize(nl);
                size_t c = 0;
                for (size_t i = ng - 1; i != above; --i) complete_layer(r.layers[c++], i, up);
                tangent_layer(r.layers[c++], above, talt, up);
                if (above == obs) partial_tangent_layer(r.layers[c++], ray, above, talt);
                else tangent_layer(r.layers[c++], above, talt, down);
                for (size_t i = above; i + 1 < obs; ++i) complete_layer(r.layers[c++], i, down);
                if (obs > above) partial_layer(r.layers[c++], ray, obs - 1, down);
            } else {
                r.ground_is_hit = true;
                size_t start = upper_index(oalt);
                if (start == 0) return;
                r.layers.resize(start);
                for (size_t i = 0; i + 1 < start; ++i) complete_layer(r.layers[i], i, down);
                partial_layer(r.layers[start - 1], ray, start - 1, down);
            }
        }
        finalize(r);
    }
};

// ---------------------------------------------------------------------------
// Stream quadrature and generalized spherical functions (scalar: d^l_{m0})
//   cpp/lib/sktran_disco/sktran_do_quadrature.cpp:5-76, cpp/include/sasktran2/math/wigner.h:40-176
// ---------------------------------------------------------------------------
inline void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0);
    w.assign(n, 0);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        long double z = std::cos(3.14159265358979323846264338327950288L * (i + 0.75L) / (n + 0.5L));
        long double pp = 0;
        for (int it = 0; it < 100; ++it) {
            long double p1 = 1, p2 = 0;
            for (int j = 0; j < n; ++j) {
                long double p3 = p2;
                p2 = p1;
                p1 = ((2.0L * j + 1.0L) * z * p2 - j * p3) / (j + 1.0L);
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0L);
            long double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs((double)(z - z1)) < 1e-18) break;
        }
        // ascending order: -z first
        x[i] = (double)(-z);
        x[n - 1 - i] = (double)z;
        double ww = (double)(2.0L / ((1.0L - z * z) * pp * pp));
        w[i] = ww;
        w[n - 1 - i] = ww;
    }
}

// double-Gauss streams: first N entries mu>0 (ascending), last N mirrored
inline void streams_and_weights(int nstr, std::vector<double>& mu, std::vector<double>& wt) {
    mu.assign(nstr, 0);
    wt.assign(nstr, 0);
    if (nstr == 2) {
        mu = {0.5, -0.5};
        wt = {1, 1};
        return;
    }
    int N = nstr / 2;
    std::vector<double> x, w;
    gauss_legendre(N, x, w);
    for (int i = 0; i < N; ++i) {
        mu[i] = 0.5 * x[i] + 0.5;
        mu[i + N] = -0.5 * x[i] - 0.5;
        wt[i] = 0.5 * w[i];
        wt[i + N] = 0.5 * w[i];
    }
}

// d^l_{m0}(theta) with x = cos(theta); returns values for l = 0..lmax-1
inline void wigner_d_m0(int m, double x, int lmax, double* out) {
    for (int l = 0; l < lmax; ++l) out[l] = 0.0;
    if (m >= lmax) return;
    // recurrence start d^m_{m0}: zeta * 2^-m * sqrt((2m)!/(m! m!)) * (1-x)^(m/2) (1+x)^(m/2)
    double factorial = 1;
    for (int i = 2 * m; i > 1; --i) {
        factorial *= double(i);
        if (i <= m) factorial /= double(i);
        if (i <= m) factorial /= double(i);
    }
    int zeta = (m % 2 == 0) ? 1 : -1;  // n=0 < m => (-1)^(m-n); m=0 => 1
    if (m == 0) zeta = 1;
    double start = zeta * std::pow(2.0, -double(m)) * std::sqrt(factorial) * std::pow(1 - x, m / 2.0) * std::pow(1 + x, m / 2.0);
    double val = start, prev = 0;
    out[m] = val;
    for (int l = m + 1; l < lmax; ++l) {
        double mult = 1.0 / (std::sqrt(double(l * l - m * m)) * l);
        double cur = (2 * l - 1) * (l * x);
        double pri = l * std::sqrt(double((l - 1) * (l - 1) - m * m));
        double t = val;
        val = mult * (cur * val - pri * prev);
        prev = t;
        out[l] = val;
    }
}

// d^l_{mn}(theta) for n != 0 (Mishchenko's recurrence F.4.1 with start F.4.3), x = cos(theta); l = 0..lmax-1
//   cpp/include/sasktran2/math/wigner.h:52-160
inline void wigner_d_mn(int m, int n, double x, int lmax, double* out) {
    for (int l = 0; l < lmax; ++l) out[l] = 0.0;
    const int lmin = std::max(std::abs(m), std::abs(n));
    if (lmin >= lmax) return;
    int zeta = 1;
    if (n < m && ((m - n) % 2 != 0)) zeta = -1;
    const int den1 = std::abs(m - n), den2 = std::abs(m + n);
    double factorial = 1;
    for (int i = 2 * lmin; i > 1; --i) {
        factorial *= double(i);
        if (i <= den1) factorial /= double(i);
        if (i <= den2) factorial /= double(i);
    }
    double val = zeta * std::pow(2.0, -double(lmin)) * std::sqrt(factorial) * std::pow(1 - x, den1 / 2.0) * std::pow(1 + x, den2 / 2.0);
    double prev = 0;
    out[lmin] = val;
    for (int l = lmin + 1; l < lmax; ++l) {
        double mult = 1.0 / ((l - 1) * std::sqrt(double(l * l - m * m)) * std::sqrt(double(l * l - n * n)));
        double cur = (2 * l - 1) * (l * (l - 1) * x - n * m);
        double pri = l * std::sqrt(double((l - 1) * (l - 1) - m * m)) * std::sqrt(double((l - 1) * (l - 1) - n * n));
        double t = val;
        val = mult * (cur * val - pri * prev);
        prev = t;
        out[l] = val;
    }
}

// polarised companions of d^l_{m0}:  R = -(d^l_{m2} + d^l_{m,-2})/2,  T = -(d^l_{m2} - d^l_{m,-2})/2
//   cpp/include/sktran_disco/sktran_do_polarization_types.h:92-107
inline void gsf_RT(int m, double x, int lmax, double* R, double* T) {
    std::vector<double> dp(lmax), dn(lmax);
    wigner_d_mn(m, 2, x, lmax, dp.data());
    wigner_d_mn(m, -2, x, lmax, dn.data());
    for (int l = 0; l < lmax; ++l) {
        R[l] = -0.5 * (dp[l] + dn[l]);
        T[l] = -0.5 * (dp[l] - dn[l]);
    }
}

// ---------------------------------------------------------------------------
// Complete precomputed geometry for one DISCO engine
// ---------------------------------------------------------------------------
struct SparseRow { std::vector<int> idx; std::vector<double> w; };

struct TableEntry { int angle, alt, sza; double w; };      // interior table point (azimuth handled with cos(k*azi))
struct GroundEntry { int angle, sza; double w; };

struct RayLayerGeom {
    double layer_distance;
    SparseRow od;          // shell OD = od . total_extinction
    SparseRow ssa_interp;  // ssa at the source point (layer midpoint)
    double azi;            // relative azimuth used for cos(k*azi)
    std::vector<TableEntry> table;  // <= 8 entries
};

struct RayGeom {
    bool ground_is_hit = false;
    std::vector<RayLayerGeom> layers;  // far end -> observer
    double ground_saa = 0;             // cos(k*(pi - saa))
    std::vector<GroundEntry> ground;   // <= 4 entries
};

struct PlaneParallelLOS { double coszenith, azimuth, observer_altitude; };

struct DiscoGeometry {
    int nstr = 0, nlyr = 0, ngrid = 0, nsza = 1, nlos = 0;
    int nstokes = 1;                    // 1 or 3; the *_R / *_T tables are filled for 3 only
    std::vector<double> stokes_C, stokes_S;  // [nlos] rotation of (Q, U) into the configured Stokes basis (1, 0 for the standard basis)
    bool spherical_pp = false;          // true: spherical interpolated post-processing, false: plane-parallel
    std::vector<double> mu, wt;         // nstr
    std::vector<double> lp_mu;          // [m][stream i < nstr][l]
    std::vector<double> lp_mu_R, lp_mu_T, lp_angle_R, lp_angle_T, lp_los_R, lp_los_T;  // same shapes as lp_mu / lp_angle / lp_los
    std::vector<double> altitudes;      // ngrid
    std::vector<double> ceil_h, floor_h;  // nlyr (p=0 at TOA)
    std::vector<double> interp;         // [nlyr][ngrid] optical interpolator
    std::vector<double> csz;            // nsza
    std::vector<double> lp_csz;         // [sza][m][l]
    std::vector<double> chapman;        // [sza][nlyr][nlyr]
    // spherical table
    int nangle = 100;
    std::vector<double> cos_angles;     // nangle
    std::vector<double> lp_angle;       // [m][angle][l]
    std::vector<uint8_t> need_point;    // [sza][alt][angle]
    std::vector<uint8_t> need_ground;   // [sza][angle]
    std::vector<RayGeom> rays;
    // plane parallel
    std::vector<PlaneParallelLOS> pp_los;
    std::vector<double> lp_los;         // [los][m][l]

    double lpmu(int m, int i, int l) const { return lp_mu[(size_t(m) * nstr + i) * nstr + l]; }
    double lpcsz(int s, int m, int l) const { return lp_csz[(size_t(s) * nstr + m) * nstr + l]; }
    double lpang(int m, int a, int l) const { return lp_angle[(size_t(m) * nangle + a) * nstr + l]; }
    double lplos(int j, int m, int l) const { return lp_los[(size_t(j) * nstr + m) * nstr + l]; }
    double chap(int s, int p, int r) const { return chapman[(size_t(s) * nlyr + p) * nlyr + r]; }
    double W(int p, int q) const { return interp[size_t(p) * ngrid + q]; }
};

inline void build_disco_geometry(const Geometry1D& g, const std::vector<RaySpec>& specs, int nstr, int num_sza,
                                 DiscoGeometry& out, int nstokes = 1, int stokes_basis = 0) {
    out = DiscoGeometry();
    out.nstokes = nstokes;
    const Coordinates& c = g.coords;
    const int ng = g.ngrid(), nlyr = ng - 1;
    out.nstr = nstr;
    out.nlyr = nlyr;
    out.ngrid = ng;
    out.nlos = (int)specs.size();
    out.altitudes = g.alt.v;
    out.spherical_pp = (c.geotype == spherical);
    streams_and_weights(nstr, out.mu, out.wt);
    out.lp_mu.assign(size_t(nstr) * nstr * nstr, 0);
    for (int m = 0; m < nstr; ++m)
        for (int i = 0; i < nstr; ++i)
            wigner_d_m0(m, std::cos(std::acos(out.mu[i])), nstr, &out.lp_mu[(size_t(m) * nstr + i) * nstr]);
    if (nstokes == 3) {
        out.lp_mu_R.assign(out.lp_mu.size(), 0);
        out.lp_mu_T.assign(out.lp_mu.size(), 0);
        for (int m = 0; m < nstr; ++m)
            for (int i = 0; i < nstr; ++i)
                gsf_RT(m, std::cos(std::acos(out.mu[i])), nstr, &out.lp_mu_R[(size_t(m) * nstr + i) * nstr], &out.lp_mu_T[(size_t(m) * nstr + i) * nstr]);
    }

    // layers
    out.ceil_h.resize(nlyr);
    out.floor_h.resize(nlyr);
    for (int p = 0; p < nlyr; ++p) {
        out.ceil_h[p] = g.alt.v[ng - 1 - p];
        out.floor_h[p] = (p == nlyr - 1) ? g.alt.v[0] : g.alt.v[ng - 2 - p];
    }
    out.interp.assign(size_t(nlyr) * ng, 0);
    for (int p = 0; p < nlyr; ++p) {
        std::array<int, 2> idx;
        std::array<double, 2> w;
        int n;
        g.alt.weights((out.ceil_h[p] + out.floor_h[p]) / 2.0, idx, w, n);
        for (int q = 0; q < n; ++q) out.interp[size_t(p) * ng + idx[q]] = w[q];
    }

    // trace LOS rays
    std::vector<TracedRay> traced(specs.size());
    std::vector<ViewingRay> vrays(specs.size());
    for (size_t i = 0; i < specs.size(); ++i) vrays[i] = construct_ray(specs[i], c);
    // Stokes basis of the output: 0 standard, 1 solar, 2 observer (Output::initialize, cpp/lib/output/output.cpp:27-55)
    out.stokes_C.assign(specs.size(), 1.0);
    out.stokes_S.assign(specs.size(), 0.0);
    for (size_t i = 0; i < specs.size(); ++i) {
        if (stokes_basis == 1) c.stokes_standard_to_solar(vrays[i].look_away, out.stokes_C[i], out.stokes_S[i]);
        if (stokes_basis == 2) c.stokes_standard_to_observer(vrays[i].look_away, vrays[i].observer.position, out.stokes_C[i], out.stokes_S[i]);
    }

    if (c.geotype == spherical) {
        SphericalShellRayTracer tracer(g);
        for (size_t i = 0; i < specs.size(); ++i) tracer.trace(vrays[i], traced[i]);

        // SZA grid
        double mn = 1, mx = -1;
        for (auto& r : traced)
            for (auto& L : r.layers) {
                mn = std::min({mn, L.cos_sza_entrance, L.cos_sza_exit});
                mx = std::max({mx, L.cos_sza_entrance, L.cos_sza_exit});
            }
        if (num_sza == 1) out.csz = {c.cos_sza_at_reference()};
        else out.csz = linspaced(num_sza, mn, mx);
        out.nsza = (int)out.csz.size();
        Grid sza_grid(out.csz, constant_spacing, linear);

        // Chapman factors from traced solar rays
        out.chapman.assign(size_t(out.nsza) * nlyr * nlyr, 0);
        for (int s = 0; s < out.nsza; ++s) {
            for (int p = 0; p < nlyr; ++p) {
                ViewingRay sr;
                sr.look_away = c.sun;
                sr.observer.position = c.solar_coordinate_vector(out.csz[s], 0.0, out.floor_h[p]);
                TracedRay tr;
                tracer.trace(sr, tr);
                double* row = &out.chapman[(size_t(s) * nlyr + p) * nlyr];
                if (tr.ground_is_hit) {
                    for (int q = 0; q < nlyr; ++q) row[q] = -1.0;
                    continue;
                }
                for (auto& L : tr.layers) {
                    double dist = L.layer_distance * L.curvature_factor;
                    double avg_alt = (L.entrance.radius() + L.exit.radius()) / 2.0 - c.earth_radius;
                    int q = 0;
                    for (; q < nlyr; ++q)
                        if (avg_alt >= out.floor_h[q] && avg_alt <= out.ceil_h[q]) break;
                    if (q == nlyr) q = nlyr - 1;
                    row[q] += dist / (out.ceil_h[q] - out.floor_h[q]);
                }
            }
        }

        // diffuse table grids
        std::vector<double> mid(nlyr);
        for (int p = 0; p < nlyr; ++p) mid[nlyr - 1 - p] = (out.ceil_h[p] + out.floor_h[p]) / 2.0;
        Grid alt_mid(mid, variable_spacing, linear);
        out.cos_angles = linspaced(out.nangle, -1, 1);
        Grid ang_grid(out.cos_angles, variable_spacing, linear);
        out.lp_angle.assign(size_t(nstr) * out.nangle * nstr, 0);
        for (int m = 0; m < nstr; ++m)
            for (int a = 0; a < out.nangle; ++a)
                wigner_d_m0(m, std::cos(std::acos(out.cos_angles[a])), nstr, &out.lp_angle[(size_t(m) * out.nangle + a) * nstr]);
        if (nstokes == 3) {
            out.lp_angle_R.assign(out.lp_angle.size(), 0);
            out.lp_angle_T.assign(out.lp_angle.size(), 0);
            for (int m = 0; m < nstr; ++m)
                for (int a = 0; a < out.nangle; ++a)
                    gsf_RT(m, std::cos(std::acos(out.cos_angles[a])), nstr, &out.lp_angle_R[(size_t(m) * out.nangle + a) * nstr],
                           &out.lp_angle_T[(size_t(m) * out.nangle + a) * nstr]);
        }
        out.need_point.assign(size_t(out.nsza) * nlyr * out.nangle, 0);
        out.need_ground.assign(size_t(out.nsza) * out.nangle, 0);

        out.rays.resize(traced.size());
        std::vector<std::pair<int, double>> iw;
        for (size_t i = 0; i < traced.size(); ++i) {
            const TracedRay& tr = traced[i];
            RayGeom& rg = out.rays[i];
            rg.ground_is_hit = tr.ground_is_hit;
            rg.layers.resize(tr.layers.size());
            for (size_t j = 0; j < tr.layers.size(); ++j) {
                const TracedLayer& L = tr.layers[j];
                RayLayerGeom& o = rg.layers[j];
                o.layer_distance = L.layer_distance;
                for (int k = 0; k < L.nw; ++k)
                    if (L.w_od[k] != 0.0) { o.od.idx.push_back(L.widx[k]); o.od.w.push_back(L.w_od[k]); }
                Location midp;
                midp.position = (L.entrance.position + L.exit.position) / 2.0;
                g.interpolation_weights(midp, iw);
                for (auto& e : iw) { o.ssa_interp.idx.push_back(e.first); o.ssa_interp.w.push_back(e.second); }
                double altitude = (L.entrance.radius() + L.exit.radius()) / 2.0 - c.earth_radius;
                double cos_angle = -(L.entrance.cos_zenith_angle(L.average_look_away) + L.exit.cos_zenith_angle(L.average_look_away)) / 2.0;
                o.azi = (L.saz_entrance + L.saz_exit) / 2.0;
                double cos_sza = (L.cos_sza_entrance + L.cos_sza_exit) / 2.0;
                std::array<int, 2> ai, gi, si;
                std::array<double, 2> aw, gw, sw;
                int na, ngc, ns;
                alt_mid.weights(altitude, ai, aw, na);
                ang_grid.weights(cos_angle, gi, gw, ngc);
                sza_grid.weights(cos_sza, si, sw, ns);
                for (int s = 0; s < ns; ++s)
                    for (int a = 0; a < na; ++a)
                        for (int q = 0; q < ngc; ++q) {
                            o.table.push_back({gi[q], ai[a], si[s], aw[a] * gw[q] * sw[s]});
                            out.need_point[(size_t(si[s]) * nlyr + ai[a]) * out.nangle + gi[q]] = 1;
                        }
            }
            if (tr.ground_is_hit && !tr.layers.empty()) {
                const TracedLayer& L0 = tr.layers[0];
                Location tmp;
                tmp.position = L0.entrance.position;
                double csz_g, saa_g;
                calculate_csz_saz(c.sun, tmp, L0.average_look_away, csz_g, saa_g);
                rg.ground_saa = saa_g;
                std::array<int, 2> si, gi;
                std::array<double, 2> sw, gw;
                int ns, ngc;
                sza_grid.weights(csz_g, si, sw, ns);
                double mu = L0.entrance.position.normalized().dot(L0.average_look_away * -1.0);
                ang_grid.weights(mu, gi, gw, ngc);
                if (out.cos_angles[gi[0]] < 0) gi[0] = gi[1];
                for (int s = 0; s < ns; ++s)
                    for (int q = 0; q < ngc; ++q) {
                        // the reference assigns (not accumulates) sparse coefficients: a repeated index keeps the last weight
                        bool replaced = false;
                        for (auto& e : rg.ground)
                            if (e.angle == gi[q] && e.sza == si[s]) { e.w = sw[s] * gw[q]; replaced = true; }
                        if (!replaced) rg.ground.push_back({gi[q], si[s], sw[s] * gw[q]});
                        out.need_ground[size_t(si[s]) * out.nangle + gi[q]] = 1;
                    }
            }
        }
    } else {
        // plane-parallel: one SZA, analytic Chapman factors, LOS described by (cos zenith, azimuth, observer altitude)
        out.csz = {c.cos_sza_at_reference()};
        out.nsza = 1;
        out.chapman.assign(size_t(nlyr) * nlyr, 0);
        if (c.geotype == pseudospherical) {
            // pseudo-spherical: plane-parallel diffuse field, solar beam attenuated along its true path through spherical shells
            // (GeometryLayerArray::calculate_chapman_factors_raytracer, sktran_do_geometrylayerarray.cpp:114-183)
            SphericalShellRayTracer tracer(g);
            for (int p = 0; p < nlyr; ++p) {
                ViewingRay sr;
                sr.look_away = c.sun;
                sr.observer.position = c.solar_coordinate_vector(out.csz[0], 0.0, out.floor_h[p]);
                TracedRay tr;
                tracer.trace(sr, tr);
                double* row = &out.chapman[size_t(p) * nlyr];
                if (tr.ground_is_hit) {
                    for (int q = 0; q < nlyr; ++q) row[q] = -1.0;
                    continue;
                }
                for (auto& L : tr.layers) {
                    double dist = L.layer_distance * L.curvature_factor;
                    double avg_alt = (L.entrance.radius() + L.exit.radius()) / 2.0 - c.earth_radius;
                    int q = 0;
                    for (; q < nlyr; ++q)
                        if (avg_alt >= out.floor_h[q] && avg_alt <= out.ceil_h[q]) break;
                    if (q == nlyr) q = nlyr - 1;
                    row[q] += dist / (out.ceil_h[q] - out.floor_h[q]);
                }
            }
        } else {
            for (int p = 0; p < nlyr; ++p)
                for (int q = 0; q <= p; ++q) out.chapman[size_t(p) * nlyr + q] = 1 / out.csz[0];
        }
        out.pp_los.resize(specs.size());
        out.lp_los.assign(specs.size() * size_t(nstr) * nstr, 0);
        if (nstokes == 3) {
            out.lp_los_R.assign(out.lp_los.size(), 0);
            out.lp_los_T.assign(out.lp_los.size(), 0);
        }
        for (size_t i = 0; i < specs.size(); ++i) {
            PlaneParallelLOS& l = out.pp_los[i];
            l.coszenith = -vrays[i].look_away.z;
            if (l.coszenith < 0) throw std::invalid_argument("only upwelling radiances are supported in plane parallel mode");
            l.azimuth = -vrays[i].relative_azimuth;
            l.observer_altitude = vrays[i].observer.position.z - c.earth_radius;
            for (int m = 0; m < nstr; ++m)
                wigner_d_m0(m, std::cos(std::acos(l.coszenith)), nstr, &out.lp_los[(i * nstr + m) * nstr]);
            if (nstokes == 3)
                for (int m = 0; m < nstr; ++m)
                    gsf_RT(m, std::cos(std::acos(l.coszenith)), nstr, &out.lp_los_R[(i * nstr + m) * nstr], &out.lp_los_T[(i * nstr + m) * nstr]);
        }
    }
    out.lp_csz.assign(size_t(out.nsza) * nstr * nstr, 0);
    for (int s = 0; s < out.nsza; ++s)
        for (int m = 0; m < nstr; ++m)
            wigner_d_m0(m, std::cos(std::acos(out.csz[s])), nstr, &out.lp_csz[(size_t(s) * nstr + m) * nstr]);
}

}  // namespace skgeo
