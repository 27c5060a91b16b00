// TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle (loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).
//
// Threading mirrors the reference engine: one `#pragma omp parallel for` over wavelengths with
// thread-private solver workspaces (cpp/lib/engine/engine.cpp:610-622).
#include <malloc.h>
#include <algorithm>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "disco_oracle.hpp"
#include "so_oracle.hpp"
#include "ts_oracle.hpp"

namespace {
struct Handle {
    skgeo::Geometry1D geo;
    skgeo::DiscoGeometry dg;
    int ss_in_do = 1;
    const double* delta_f = nullptr;
    const double *emission = nullptr, *surface_emission = nullptr;
    int forced_azimuth = 0;              // plane-parallel post-processing: > 0 solves exactly this many azimuth orders
    std::vector<double> flux_altitudes;  // flux observers (m)
    double* flux_out = nullptr;          // [nwavel][observer][3] (upwelling, downwelling, actinic) of the next orc_run calls
    int brdf_kind = 0, brdf_nargs = 1;   // non-Lambertian surface of the next orc_run calls (orc_set_brdf)
    const double* brdf_args = nullptr;   // [nwavel][nargs]
    double* d_brdf_out = nullptr;        // [nwavel][nlos][S][nargs]
    std::string error;
};
}  // namespace

namespace {
// Worker threads allocate and free their per-wavelength work arrays (hundreds of KB each) all the time: keep them inside the thread's malloc
// arena instead of mmap / munmap per array (the address-space lock would serialise the OpenMP threads of the CPU baseline).
struct MallocTuning {
    MallocTuning() {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
    }
} malloc_tuning;
}  // namespace

extern "C" {

// ray_params: per ray [cos_sza, relative_azimuth, observer_altitude, tangent_altitude | cos_viewing_zenith]
void* orc_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method,
                 int geotype, int nstr, int num_sza, int nrays, const int* ray_kinds, const double* ray_params,
                 int ss_in_do, int nstokes, int stokes_basis) {
    try {
        auto h = std::make_unique<Handle>();
        h->geo = skgeo::Geometry1D(cos_sza, saa, earth_radius, std::vector<double>(altitudes, altitudes + ngrid),
                                   (skgeo::Interpolation)interp_method, (skgeo::GeometryType)geotype);
        std::vector<skgeo::RaySpec> specs(nrays);
        for (int i = 0; i < nrays; ++i) {
            specs[i].kind = (skgeo::RaySpec::Kind)ray_kinds[i];
            specs[i].cos_sza = ray_params[4 * i + 0];
            specs[i].horizontal_angle = ray_params[4 * i + 0];  // kind 3 (tangent_altitude)
            specs[i].relative_azimuth = ray_params[4 * i + 1];
            specs[i].observer_altitude = ray_params[4 * i + 2];
            specs[i].tangent_altitude = ray_params[4 * i + 3];
            specs[i].cos_viewing_zenith = ray_params[4 * i + 3];
        }
        skgeo::build_disco_geometry(h->geo, specs, nstr, num_sza, h->dg, nstokes, stokes_basis);
        h->ss_in_do = ss_in_do;
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "orc_create: %s\n", e.what());
        return nullptr;
    }
}

void orc_destroy(void* h) { delete static_cast<Handle*>(h); }

int orc_num_sza(void* hv) { return static_cast<Handle*>(hv)->dg.nsza; }
int orc_chapman(void* hv, double* out) {
    auto* h = static_cast<Handle*>(hv);
    std::copy(h->dg.chapman.begin(), h->dg.chapman.end(), out);
    return 0;
}
// geometry tables of one line of sight, for the independent geometry check (tests/test_independent_geometry.py): per ray layer (far end ->
// observer) its length, the azimuth used for cos(m azi) and the diffuse-table stencil (sza index, altitude index, angle index, weight);
// flattened as  [nlayers, ground_is_hit, ground_saa, nground, (angle, sza, w) x nground, then per layer: distance, azi, n, (sza, alt, angle, w) x n].
// Returns the number of doubles needed (the caller passes a buffer of `cap` doubles; nothing is written past it).
int orc_ray_tables(void* hv, int ray, double* out, int cap) {
    auto* h = static_cast<Handle*>(hv);
    if (ray < 0 || ray >= int(h->dg.rays.size())) return -1;
    const skgeo::RayGeom& r = h->dg.rays[ray];
    std::vector<double> v{double(r.layers.size()), r.ground_is_hit ? 1.0 : 0.0, r.ground_saa, double(r.ground.size())};
    for (const auto& g : r.ground) { v.push_back(g.angle); v.push_back(g.sza); v.push_back(g.w); }
    for (const auto& L : r.layers) {
        v.push_back(L.layer_distance); v.push_back(L.azi); v.push_back(double(L.table.size()));
        for (const auto& e : L.table) { v.push_back(e.sza); v.push_back(e.alt); v.push_back(e.angle); v.push_back(e.w); }
    }
    for (int i = 0; i < int(v.size()) && i < cap; ++i) out[i] = v[i];
    return int(v.size());
}
int orc_streams(void* hv, double* mu, double* wt) {
    auto* h = static_cast<Handle*>(hv);
    std::copy(h->dg.mu.begin(), h->dg.mu.end(), mu);
    std::copy(h->dg.wt.begin(), h->dg.wt.end(), wt);
    return 0;
}

// Outputs (C order, S = nstokes): radiance[nwavel][nlos][S], d_k[nwavel][nlos][S][ngrid], d_ssa[nwavel][nlos][S][ngrid],
// d_albedo[nwavel][nlos][S], los_od[nwavel][nlos], n_azimuth[nwavel].  Derivative outputs may be null.
// leg: [nleg (*4 for S = 3: a1, a2, a3, b1 per order), ngrid, nwavel] as in the reference storage.
// delta-M truncation fraction f[ngrid, nwavel] (location fastest) of an already scaled storage for the next orc_run calls
int orc_set_delta_f(void* hv, const double* f) {
    auto* h = static_cast<Handle*>(hv);
    if (!h) return -1;
    h->delta_f = f;
    return 0;
}
// thermal source of the discrete-ordinates equations for the next orc_run calls: emission_source [ngrid, nwavel], surface emission [nwavel]
int orc_set_emission(void* hv, const double* emission, const double* surface_emission) {
    auto* h = static_cast<Handle*>(hv);
    if (!h) return -1;
    h->emission = emission;
    h->surface_emission = surface_emission;
    return 0;
}
// plane-parallel post-processing: number of azimuth orders to solve (<= 0: all of them), flux observers and where the next orc_run calls
// write their fluxes [nwavel][observer][3] (upwelling, downwelling, actinic)
int orc_set_forced_azimuth(void* hv, int n) {
    auto* h = static_cast<Handle*>(hv);
    if (!h) return -1;
    h->forced_azimuth = n;
    return 0;
}
int orc_set_flux_observers(void* hv, const double* altitudes, int n, double* flux_out) {
    auto* h = static_cast<Handle*>(hv);
    if (!h || n < 0 || (n > 0 && (!altitudes || !flux_out))) return -1;
    if (n > 0 && h->dg.spherical_pp) return -2;
    h->flux_altitudes.assign(altitudes, altitudes + n);
    h->flux_out = n > 0 ? flux_out : nullptr;
    return 0;
}
// surface of the next orc_run calls: kind 0 Lambertian (the albedo argument of orc_run), 1 SnowKokhanovsky (1 argument), 2 MODIS (3 arguments);
// args [nwavel][nargs]; d_brdf_out [nwavel][nlos][S][nargs] receives the derivatives wrt the arguments (or null)
int orc_set_brdf(void* hv, int kind, int nargs, const double* args, double* d_brdf_out) {
    auto* h = static_cast<Handle*>(hv);
    if (!h || kind < 0 || kind > 3 || (kind != 0 && (!args || nargs != (kind == 2 ? 3 : 1)))) return -1;
    h->brdf_kind = kind;
    h->brdf_nargs = kind == 0 ? 1 : nargs;
    h->brdf_args = kind == 0 ? nullptr : args;
    h->d_brdf_out = kind == 0 ? nullptr : d_brdf_out;
    return 0;
}
int orc_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg,
            const double* solar_irradiance, const double* albedo, int derivatives, int nthreads, double* radiance,
            double* d_k, double* d_ssa, double* d_albedo, double* los_od, int* n_azimuth) {
    auto* h = static_cast<Handle*>(hv);
    if (!h) return -1;
    const int nlos = h->dg.nlos, ngrid = h->dg.ngrid, S = h->dg.nstokes;
    oracle::AtmosphereView atm;
    atm.ngrid = ngrid;
    atm.nwavel = nwavel;
    atm.nleg = nleg;
    atm.nstokes = S;
    atm.extinction = extinction;
    atm.ssa = ssa;
    atm.leg = leg;
    atm.solar_irradiance = solar_irradiance;
    atm.albedo = albedo;
    atm.brdf_kind = h->brdf_kind;
    atm.brdf_nargs = h->brdf_nargs;
    atm.brdf_args = h->brdf_args;
    atm.delta_f = h->delta_f;
    atm.emission = h->emission;
    atm.surface_emission = h->surface_emission;
    atm.derivatives = derivatives != 0;
    int status = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        std::unique_ptr<oracle::Solver> solver;
        oracle::Result res;
        try {
            solver = std::make_unique<oracle::Solver>(h->dg, atm.derivatives, h->ss_in_do != 0);
            solver->forced_azimuth = h->forced_azimuth;
            solver->flux_altitudes = h->flux_altitudes;
        } catch (const std::exception& e) {
            std::fprintf(stderr, "orc_run: %s\n", e.what());
            status = -3;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int w = 0; w < nwavel; ++w) {
            if (status != 0 || !solver) continue;
            try {
                solver->solve(atm, w, res);
                for (int r = 0; r < nlos; ++r) {
                    if (los_od) los_od[size_t(w) * nlos + r] = res.los_od[r];
                    for (int sc = 0; sc < S; ++sc) {
                        const size_t o = (size_t(w) * nlos + r) * S + sc;
                        // rotation of (Q, U) into the configured Stokes basis (OutputC::assign_lane, cpp/lib/output/outputc.cpp:108-228)
                        const double C = h->dg.stokes_C[r], Sn = h->dg.stokes_S[r];
                        const int nn = solver->num_native();
                        auto mix = [&](const double* base, size_t stride, size_t off) {
                            if (sc == 0 || S == 1) return base[(size_t(r) * S + sc) * stride + off];
                            const double q = base[(size_t(r) * S + 1) * stride + off], u = base[(size_t(r) * S + 2) * stride + off];
                            return sc == 1 ? C * q - Sn * u : Sn * q + C * u;
                        };
                        radiance[o] = mix(res.radiance.data(), 1, 0);
                        if (atm.derivatives) {
                            for (int q = 0; q < ngrid; ++q) {
                                if (d_k) d_k[o * ngrid + q] = mix(res.deriv.data(), nn, q);
                                if (d_ssa) d_ssa[o * ngrid + q] = mix(res.deriv.data(), nn, ngrid + q);
                            }
                            if (d_albedo) d_albedo[o] = mix(res.deriv.data(), nn, 2 * ngrid);
                        }
                    }
                }
                if (n_azimuth) n_azimuth[w] = res.num_azimuth_used;
                if (h->flux_out) std::copy(res.flux.begin(), res.flux.end(), h->flux_out + size_t(w) * res.flux.size());
                // derivatives wrt the BRDF arguments: one pass per argument (the native surface entry [2 ngrid] belongs to the selected one)
                if (atm.derivatives && h->brdf_kind != 0 && h->d_brdf_out) {
                    const int nn = solver->num_native(), na = h->brdf_nargs;
                    for (int a = 0; a < na; ++a) {
                        if (a > 0) { solver->surface_deriv_arg = a; solver->solve(atm, w, res); }
                        for (int r = 0; r < nlos; ++r)
                            for (int sc = 0; sc < S; ++sc) {
                                const double C = h->dg.stokes_C[r], Sn = h->dg.stokes_S[r];
                                const double* base = res.deriv.data();
                                double v = base[(size_t(r) * S + sc) * nn + 2 * ngrid];
                                if (S == 3 && sc > 0) {
                                    const double q = base[(size_t(r) * S + 1) * nn + 2 * ngrid], u = base[(size_t(r) * S + 2) * nn + 2 * ngrid];
                                    v = sc == 1 ? C * q - Sn * u : Sn * q + C * u;
                                }
                                h->d_brdf_out[((size_t(w) * nlos + r) * S + sc) * na + a] = v;
                            }
                    }
                    solver->surface_deriv_arg = 0;
                }
            } catch (const std::exception& e) {
                std::fprintf(stderr, "orc_run (wavelength %d): %s\n", w, e.what());
                status = -3;
            }
        }
    }
    return status;
}

// ---- exact single scatter + legacy successive orders (so_oracle.hpp)
struct SoHandle {
    skgeo::Geometry1D geo;
    skso::SoGeometry sg;
    int num_iterations = 50;
    const double *emission = nullptr, *surface_emission = nullptr;
    oracle_so::Solver::FixedPoint fixed_point;
    const double* los_albedo = nullptr;   // [nwavel][nlos] pi x brdf at the lines of sight's ground geometry (orso_set_brdf), or null
};

void* orso_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method, int geotype,
                  int nrays, const int* ray_kinds, const double* ray_params, int nstokes, int stokes_basis, int num_ss_moments,
                  int num_ms_moments, int num_incoming, int num_outgoing, int num_sza, int num_iterations, int single_scatter,
                  int multiple_scatter, int emission, int occultation) {
    try {
        auto h = std::make_unique<SoHandle>();
        h->geo = skgeo::Geometry1D(cos_sza, saa, earth_radius, std::vector<double>(altitudes, altitudes + ngrid),
                                   (skgeo::Interpolation)interp_method, (skgeo::GeometryType)geotype);
        std::vector<skgeo::RaySpec> specs(nrays);
        for (int i = 0; i < nrays; ++i) {
            specs[i].kind = (skgeo::RaySpec::Kind)ray_kinds[i];
            specs[i].cos_sza = ray_params[4 * i + 0];
            specs[i].horizontal_angle = ray_params[4 * i + 0];
            specs[i].relative_azimuth = ray_params[4 * i + 1];
            specs[i].observer_altitude = ray_params[4 * i + 2];
            specs[i].tangent_altitude = ray_params[4 * i + 3];
            specs[i].cos_viewing_zenith = ray_params[4 * i + 3];
        }
        skso::Settings cfg;
        cfg.nstokes = nstokes;
        cfg.num_ss_moments = num_ss_moments;
        cfg.num_ms_moments = num_ms_moments;
        cfg.num_incoming = num_incoming;
        cfg.num_outgoing = num_outgoing;
        cfg.num_sza = num_sza;
        cfg.single_scatter = single_scatter != 0;   // 1: exact, 2: solar table
        cfg.solar_table = single_scatter == 2;
        cfg.multiple_scatter = multiple_scatter != 0;
        cfg.emission = emission;
        cfg.occultation = occultation != 0;
        skso::build_so_geometry(h->geo, specs, cfg, h->sg, stokes_basis);
        h->num_iterations = num_iterations;
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "orso_create: %s\n", e.what());
        return nullptr;
    }
}
void orso_destroy(void* h) { delete static_cast<SoHandle*>(h); }
// emission_source [ngrid, nwavel] and surface emission [nwavel] for the following orso_run calls
int orso_set_emission(void* hv, const double* emission, const double* surface_emission) {
    auto* h = static_cast<SoHandle*>(hv);
    if (!h) return -1;
    h->emission = emission;
    h->surface_emission = surface_emission;
    return 0;
}
// MultipleScatterSource::SuccessiveOrders: fixed point with Anderson mixing instead of a fixed number of legacy passes (so_oracle.hpp)
int orso_set_fixed_point(void* hv, int enabled, double relative_tolerance, double absolute_tolerance, double damping, int anderson_depth) {
    auto* h = static_cast<SoHandle*>(hv);
    if (!h || anderson_depth < 0 || anderson_depth > 3 || !(damping > 0 && damping <= 1)) return -1;
    h->fixed_point.enabled = enabled != 0;
    h->fixed_point.relative_tolerance = relative_tolerance;
    h->fixed_point.absolute_tolerance = absolute_tolerance;
    h->fixed_point.damping = damping;
    h->fixed_point.anderson_depth = anderson_depth;
    return 0;
}
// Non-Lambertian surface under the exact single scatter of the next orso_run calls (kind as orc_set_brdf; 0 switches it off): fills
// los_albedo [nwavel][nlos] = pi x brdf(mu_in, mu_out, phi) at every line of sight's ground geometry (it must stay alive) and, if given,
// d_los_albedo [nwavel][nlos][nargs] = pi x d brdf / d argument -- the single scatter is linear in that value, so the weighting function of
// argument a is (d I / d albedo of the run) x d_los_albedo[.., a]   (cpp/lib/solar/singlescattersource.cpp:264-279,574-636)
int orso_set_brdf(void* hv, int kind, int nargs, const double* args, int nwavel, double* los_albedo, double* d_los_albedo) {
    auto* h = static_cast<SoHandle*>(hv);
    if (!h || kind < 0 || kind > 3) return -1;
    h->los_albedo = nullptr;
    if (kind == 0) return 0;
    if (!args || !los_albedo || nargs != (kind == 2 ? 3 : 1)) return -1;
    const skso::RaySet& lo = h->sg.los;
    for (int w = 0; w < nwavel; ++w)
        for (int r = 0; r < lo.nrays; ++r) {
            const bool lit = lo.ground_is_hit[r] && lo.ground_mu_in[r] > 0;
            const double* a = args + size_t(w) * nargs;
            los_albedo[size_t(w) * lo.nrays + r] = lit ? skgeo::kPi * oracle::brdf_value(kind, lo.ground_mu_in[r], lo.ground_mu_out[r], lo.ground_phi[r], a) : 0.0;
            for (int k = 0; k < nargs && d_los_albedo; ++k)
                d_los_albedo[(size_t(w) * lo.nrays + r) * nargs + k] =
                    lit ? skgeo::kPi * oracle::brdf_deriv(kind, k, lo.ground_mu_in[r], lo.ground_mu_out[r], lo.ground_phi[r], a) : 0.0;
        }
    h->los_albedo = los_albedo;
    return 0;
}
int orso_lebedev(int npoints, double* xyzw) {
    try {
        const std::vector<double> s = skleb::sphere(npoints);
        std::copy(s.begin(), s.end(), xyzw);
        return 0;
    } catch (const std::exception&) { return -1; }
}
// three-nearest interpolation on a Lebedev sphere (or, with `ground_location` non-null, on its up-looking half at that location) as the shared
// geometry precompute performs it (skso::UnitSphere::interpolate), for the independent check in tests/test_independent_geometry.py
int orso_sphere_interpolate(int npoints, const double* direction, const double* ground_location, int* idx, double* w) {
    try {
        skso::UnitSphere s = skso::UnitSphere::lebedev(npoints);
        if (ground_location) s = skso::UnitSphere::ground_of(s, skgeo::Vec3(ground_location[0], ground_location[1], ground_location[2]));
        std::array<int, 3> i3;
        std::array<double, 3> w3;
        s.interpolate(skgeo::Vec3(direction[0], direction[1], direction[2]), i3, w3);
        for (int k = 0; k < 3; ++k) { idx[k] = i3[k]; w[k] = w3[k]; }
        return s.size();
    } catch (const std::exception&) { return -1; }
}
// sizes of the diffuse discretisation: points, incoming rays, outgoing directions, accumulation-matrix entries
int orso_sizes(void* hv, int* out) {
    auto* h = static_cast<SoHandle*>(hv);
    if (!h) return -1;
    out[0] = h->sg.npoints;
    out[1] = h->sg.in_start.empty() ? 0 : h->sg.in_start.back();
    out[2] = h->sg.out_start.empty() ? 0 : h->sg.out_start.back();
    out[3] = (int)h->sg.accumulation.w.size();
    return 0;
}
// radiance[nwavel][nlos][S], los_od[nwavel][nlos]
// d_k, d_ssa [nwavel][nlos][S][ngrid], d_albedo [nwavel][nlos][S] (or null): native weighting functions of the exact single scatter
int orso_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg, const double* solar_irradiance,
             const double* albedo, int nthreads, double* radiance, double* los_od, double* d_k, double* d_ssa, double* d_albedo,
             double* d_emission, double* d_surface_emission) {   // the last two [nwavel][nlos][S][ngrid] / [nwavel][nlos][S], or null
    auto* h = static_cast<SoHandle*>(hv);
    if (!h) return -1;
    const int nlos = h->sg.nlos, S = h->sg.cfg.nstokes;
    oracle_so::AtmosphereView atm;
    atm.ngrid = h->sg.ngrid;
    atm.nwavel = nwavel;
    atm.nleg = nleg;
    atm.nstokes = S;
    atm.extinction = extinction;
    atm.ssa = ssa;
    atm.leg = leg;
    atm.solar_irradiance = solar_irradiance;
    atm.albedo = albedo;
    atm.los_albedo = h->los_albedo;
    atm.emission = h->emission;
    atm.surface_emission = h->surface_emission;
    if (h->sg.cfg.emission != 0 && !atm.emission) return -2;
    atm.derivatives = d_k != nullptr;
    const int ngrid = h->sg.ngrid, nn = 3 * ngrid + 2;
    int status = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        oracle_so::Solver solver(h->sg);
        solver.num_iterations = h->num_iterations;
        solver.fixed_point = h->fixed_point;
        oracle_so::Result res;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int w = 0; w < nwavel; ++w) {
            if (status != 0) continue;
            try {
                solver.solve(atm, w, res);
                for (int i = 0; i < nlos * S; ++i) radiance[size_t(w) * nlos * S + i] = res.radiance[i];
                if (los_od)
                    for (int r = 0; r < nlos; ++r) los_od[size_t(w) * nlos + r] = res.los_od[r];
                if (atm.derivatives)
                    for (int i = 0; i < nlos * S; ++i) {
                        const double* src = res.deriv.data() + size_t(i) * nn;
                        for (int q = 0; q < ngrid; ++q) {
                            d_k[(size_t(w) * nlos * S + i) * ngrid + q] = src[q];
                            d_ssa[(size_t(w) * nlos * S + i) * ngrid + q] = src[ngrid + q];
                        }
                        d_albedo[size_t(w) * nlos * S + i] = src[2 * ngrid];
                        if (d_emission)
                            for (int q = 0; q < ngrid; ++q) d_emission[(size_t(w) * nlos * S + i) * ngrid + q] = src[2 * ngrid + 1 + q];
                        if (d_surface_emission) d_surface_emission[size_t(w) * nlos * S + i] = src[3 * ngrid + 1];
                    }
            } catch (const std::exception& e) {
                std::fprintf(stderr, "orso_run (wavelength %d): %s\n", w, e.what());
                status = -3;
            }
        }
    }
    return status;
}

// ---- two-stream source (ts_oracle.hpp)
struct TsHandle {
    skgeo::Geometry1D geo;
    skts::TsGeometry tg;
};

void* orts_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method, int geotype,
                  int nrays, const int* ray_kinds, const double* ray_params, int num_sza, int solar, int thermal) {
    try {
        auto h = std::make_unique<TsHandle>();
        h->geo = skgeo::Geometry1D(cos_sza, saa, earth_radius, std::vector<double>(altitudes, altitudes + ngrid),
                                   (skgeo::Interpolation)interp_method, (skgeo::GeometryType)geotype);
        std::vector<skgeo::RaySpec> specs(nrays);
        for (int i = 0; i < nrays; ++i) {
            specs[i].kind = (skgeo::RaySpec::Kind)ray_kinds[i];
            specs[i].cos_sza = ray_params[4 * i + 0];
            specs[i].horizontal_angle = ray_params[4 * i + 0];
            specs[i].relative_azimuth = ray_params[4 * i + 1];
            specs[i].observer_altitude = ray_params[4 * i + 2];
            specs[i].tangent_altitude = ray_params[4 * i + 3];
            specs[i].cos_viewing_zenith = ray_params[4 * i + 3];
        }
        skts::Settings cfg;
        cfg.num_sza = num_sza;
        cfg.solar = solar != 0;
        cfg.thermal = thermal != 0;
        skts::build_ts_geometry(h->geo, specs, cfg, h->tg);
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "orts_create: %s\n", e.what());
        return nullptr;
    }
}
void orts_destroy(void* h) { delete static_cast<TsHandle*>(h); }
// sizes: columns, segments, optical-depth stencil entries
int orts_sizes(void* hv, int* out) {
    auto* h = static_cast<TsHandle*>(hv);
    if (!h) return -1;
    out[0] = h->tg.ncol;
    out[1] = h->tg.nseg();
    out[2] = (int)h->tg.od_level.size();
    return 0;
}
// Inputs in the engine's storage layout (location fastest): extinction, ssa, emission [nwavel][ngrid], leg [nwavel][ngrid][nleg],
// delta_f [nwavel][ngrid] or null (truncation fraction already applied to the optical properties), albedo / irradiance / surface emission
// [nwavel] (irradiance and the emission arrays may be null).  Outputs: radiance [nwavel][nlos] (solar + thermal two-stream sources as
// configured), d_k / d_ssa / d_b1 / d_emission [nwavel][nlos][ngrid], d_albedo / d_surface_emission [nwavel][nlos] (all or none).
int orts_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg, const double* delta_f,
             const double* emission, const double* surface_emission, const double* solar_irradiance, const double* albedo, int nthreads,
             double* radiance, double* d_k, double* d_ssa, double* d_b1, double* d_emission, double* d_albedo, double* d_surface_emission) {
    auto* h = static_cast<TsHandle*>(hv);
    if (!h) return -1;
    const skts::TsGeometry& g = h->tg;
    const int nlos = g.nlos, ngrid = g.ngrid;
    if (g.cfg.thermal && !emission) return -2;
    const bool derivs = d_k != nullptr;
    int status = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        std::vector<double> b1(ngrid);
        oracle_ts::Result res;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int w = 0; w < nwavel; ++w) {
            if (status != 0) continue;
            try {
                const size_t o = size_t(w) * ngrid;
                for (int q = 0; q < ngrid; ++q) {
                    const double f = delta_f ? delta_f[o + q] : 0.0;
                    b1[q] = (nleg > 1 ? leg[(o + q) * nleg + 1] : 0.0) - 3.0 * f / (1.0 - f);
                }
                for (int i = 0; i < nlos; ++i) radiance[size_t(w) * nlos + i] = 0.0;
                if (derivs) {
                    for (size_t i = 0; i < size_t(nlos) * ngrid; ++i) {
                        const size_t j = size_t(w) * nlos * ngrid + i;
                        d_k[j] = d_ssa[j] = d_b1[j] = d_emission[j] = 0.0;
                    }
                    for (int i = 0; i < nlos; ++i) d_albedo[size_t(w) * nlos + i] = d_surface_emission[size_t(w) * nlos + i] = 0.0;
                }
                for (int pass = 0; pass < 2; ++pass) {
                    const bool solar = pass == 0;
                    if (solar ? !g.cfg.solar : !g.cfg.thermal) continue;
                    oracle_ts::run_wavelength(g, solar, extinction + o, ssa + o, b1.data(), emission ? emission + o : nullptr, albedo[w],
                                              solar_irradiance ? solar_irradiance[w] : 1.0, surface_emission ? surface_emission[w] : 0.0,
                                              derivs, res);
                    for (int i = 0; i < nlos; ++i) radiance[size_t(w) * nlos + i] += res.radiance[i];
                    if (!derivs) continue;
                    for (size_t i = 0; i < size_t(nlos) * ngrid; ++i) {
                        const size_t j = size_t(w) * nlos * ngrid + i;
                        d_k[j] += res.d_extinction[i];
                        d_ssa[j] += res.d_ssa[i];
                        d_b1[j] += res.d_b1[i];
                        d_emission[j] += res.d_emission[i];
                    }
                    for (int i = 0; i < nlos; ++i) {
                        d_albedo[size_t(w) * nlos + i] += res.d_albedo[i];
                        d_surface_emission[size_t(w) * nlos + i] += res.d_surface_emission[i];
                    }
                }
            } catch (const std::exception& e) {
                std::fprintf(stderr, "orts_run (wavelength %d): %s\n", w, e.what());
                status = -3;
            }
        }
    }
    return status;
}

}  // extern "C"
