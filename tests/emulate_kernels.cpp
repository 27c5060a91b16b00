// TEST INFRASTRUCTURE: runs the HIP kernel bodies (sasktran2_amd/csrc/hip/disco_core.hpp, compiled here with g++)
// on the CPU, one loop iteration per GPU thread, in the same launch order as the HIP driver
// (sasktran2_amd/csrc/hip/disco_engine.hip).  Lets the kernel math be checked against the oracle in the GPU-less
// authoring container; it is not part of the product library and is never a fallback.
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "../sasktran2_amd/csrc/host/geo_tables.hpp"

namespace {
// thermal source inside the discrete-ordinates equations (emu_set_thermal): inputs and the emission weighting-function outputs
const double* g_emission = nullptr;
const double* g_surface_emission = nullptr;
double* g_wf_emission = nullptr;
double* g_wf_surface_emission = nullptr;
struct Handle {
    skgeo::Geometry1D geo;
    skgeo::DiscoGeometry dg;
    skhost::GeoTables tab;
};

template <int N>
void run_n(const skd::Geo& g, skd::Bufs& b, int derivs) {
    const int nw = b.nw;
    for (int w = 0; w < nw; ++w) skd::k0_layer_optics(g, b, w);
    if (g.pp_mode) {
        // plane-parallel launch order of the HIP driver: every azimuth order, K3 only for the record's E / PPM entries
        for (size_t i = 0; i < size_t(nw) * g.nlos; ++i) b.radiance[i] = 0;
        for (int w = 0; w < nw; ++w) {
            skd::k0_transmission(g, b, 0, w);
            b.norders[w] = g.nstr;
        }
        for (int m = 0; m < g.nstr; ++m) {
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < g.nlyr; ++p) skd::k1_layer_solution<N>(g, b, 0, m, p, w);
            for (int w = 0; w < nw; ++w) skd::k2a_bvp<N>(g, b, 0, m, w);
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < g.nlyr; ++p) skd::k3_table_values<N>(g, b, 0, m, p, w);
            if (derivs)
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < g.nlyr; ++p) skd::k2c_layer_ops<N>(g, b, 0, m, p, w);
            for (int w = 0; w < nw; ++w)
                for (int l = 0; l < g.nlos; ++l) skd::kpp_los<N>(g, b, 0, m, l, w);
        }
        if (derivs)
            for (int w = 0; w < nw; ++w)
                for (int l = 0; l < g.nlos; ++l) skd::k4a_chain(g, b, 0, l, w);
        if (g_emission) {
            // thermal pass in the launch order of the HIP driver (disco_engine.hip, run_chunk): order 0 once more with the thermal "beam"
            skd::Bufs bt = b;
            bt.emission = g_emission;
            bt.surf_em = g_surface_emission;
            bt.thermal_pass = 1;
            const int s = g.nsza;
            for (int w = 0; w < nw; ++w) skd::k0_thermal(g, bt, w);
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < g.nlyr; ++p) skd::k1_layer_solution<N>(g, bt, s, 0, p, w);
            for (int w = 0; w < nw; ++w) skd::k2a_bvp<N>(g, bt, s, 0, w);
            if (derivs) {
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < g.nlyr; ++p) skd::k3_table_values<N>(g, bt, s, 0, p, w);
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < g.nlyr; ++p) skd::k2c_layer_ops<N>(g, bt, s, 0, p, w);
            }
            for (int w = 0; w < nw; ++w)
                for (int l = 0; l < g.nlos; ++l) skd::kpp_los<N>(g, bt, s, 0, l, w);
            if (derivs)
                for (int w = 0; w < nw; ++w)
                    for (int l = 0; l < g.nlos; ++l) skd::k4a_thermal(g, bt, l, w, 0, nw, g_wf_emission, g_wf_surface_emission);
        }
        return;
    }
    for (int s = 0; s < g.nsza; ++s) {
        for (int w = 0; w < nw; ++w) {
            skd::k0_transmission(g, b, s, w);
            b.norders[w] = g.nstr;
            b.notconv[w] = 0;
        }
        for (int m = 0; m < g.nstr; ++m) {
            int active = 0;
            for (int w = 0; w < nw; ++w) active += (m < b.norders[w]);
            if (!active) break;
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < g.nlyr; ++p)
                    if (m < b.norders[w]) skd::k1_layer_solution<N>(g, b, s, m, p, w);
            for (int w = 0; w < nw; ++w)
                if (m < b.norders[w]) skd::k2a_bvp<N>(g, b, s, m, w);
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < g.nlyr; ++p)
                    if (m < b.norders[w]) skd::k3_table_values<N>(g, b, s, m, p, w);
            if (derivs) {
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < g.nlyr; ++p)
                        if (m < b.norders[w]) skd::k2c_layer_ops<N>(g, b, s, m, p, w);
                for (int w = 0; w < nw; ++w)
                    for (int l = 0; l < g.nlos; ++l)
                        if (m < b.norders[w]) skd::k2b_adjoint<N>(g, b, s, m, l, w);
            }
            for (int w = 0; w < nw; ++w) skd::k_stop_rule(g, b, s, m, w);
        }
        if (derivs)
            for (int w = 0; w < nw; ++w)
                for (int l = 0; l < g.nlos; ++l) skd::k4a_chain(g, b, s, l, w);
    }
    for (int w = 0; w < nw; ++w)
        for (int l = 0; l < g.nlos; ++l) skd::k4b_los(g, b, l, w);
}
}  // namespace

extern "C" {
void* emu_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method,
                 int geotype, int nstr, int num_sza, int nrays, const int* ray_kinds, const double* ray_params, int ss_in_do) {
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
        skgeo::build_disco_geometry(h->geo, specs, nstr, num_sza, h->dg);
        h->tab.include_ss = ss_in_do;
        h->tab.build(h->dg);
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emu_create: %s\n", e.what());
        return nullptr;
    }
}
void emu_destroy(void* h) { delete static_cast<Handle*>(h); }

const double* g_delta_f = nullptr;
void emu_set_delta_f(const double* f) { g_delta_f = f; }
// emission [w][q], surface emission [w] (or null), outputs wf_emission [q][w][los], wf_surface_emission [w][los] (or null); all null = off
void emu_set_thermal(const double* emission, const double* surface_emission, double* wf_emission, double* wf_surface_emission) {
    g_emission = emission;
    g_surface_emission = surface_emission;
    g_wf_emission = wf_emission;
    g_wf_surface_emission = wf_surface_emission;
}
int emu_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg,
            const double* solar_irradiance, const double* albedo, int derivatives, double* radiance, double* d_k,
            double* d_ssa, double* d_albedo, double* los_od, int* status) {
    auto* h = static_cast<Handle*>(hv);
    const skhost::GeoTables& t = h->tab;
    skd::Geo g = t.host_view();
    const int N = t.N, nlyr = t.nlyr, nstr = t.nstr, nw = nwavel, nlos = t.nlos, ngrid = t.ngrid;
    const size_t nt = size_t(nw) * nlos;
    int recsize = 4 * N * N + 41 * N;
    std::vector<double> tau(size_t(nlyr) * nw), om(tau.size()), kbar(tau.size()), a(size_t(nlyr) * nstr * nw),
        T(size_t(nlyr + 1) * nw), sec(tau.size()), od(size_t(t.nq_total) * nw), omloc(od.size()), srcw(od.size()),
        att_total(size_t(nlos) * nw), rec(size_t(nlyr) * recsize * nw), dinv(size_t(nlyr) * 4 * N * N * nw),
        xsol(size_t(nlyr) * 2 * N * nw), J(size_t(nstr) * t.nslot * nw, 0.0), z(size_t(nlyr) * 2 * N * nt),
        grad(size_t(4 * nlyr + 2) * nt, 0.0), ops(size_t(nlyr) * (8 * N * N + 26 * N + 5) * nw), qlbuf(size_t(nlyr) * nstr * nt), dnat(size_t(2 * ngrid + 1) * nt, 0.0), rad(nt), lod(nt);
    std::vector<uint8_t> conv(size_t(t.nslot) * nw);
    for (int sl = 0; sl < t.nslot; ++sl)
        for (int w = 0; w < nw; ++w) conv[size_t(w) * t.nslot + sl] = t.slot_is_ground[sl] ? 1 : 255;
    std::vector<int> notconv(nw, 0), norders(nw, nstr), stat(nw, 0);
    skd::Bufs b{};
    b.nw = nw; b.ext = extinction; b.ssa = ssa; b.leg = leg; b.albedo = albedo; b.irradiance = solar_irradiance; b.delta_f = g_delta_f; b.nleg = nleg;
    b.tau = tau.data(); b.om = om.data(); b.kbar = kbar.data(); b.a = a.data(); b.T = T.data(); b.sec = sec.data();
    b.od = od.data(); b.omloc = omloc.data(); b.srcw = srcw.data(); b.att_total = att_total.data(); b.rec = rec.data();
    b.dinv = dinv.data(); b.xsol = xsol.data(); b.J = J.data(); b.conv_order = conv.data(); b.notconv = notconv.data();
    b.norders = norders.data(); b.status = stat.data(); b.ops = ops.data(); b.qlbuf = qlbuf.data(); b.z = z.data(); b.grad = grad.data(); b.dnat = dnat.data();
    b.radiance = rad.data(); b.los_od = lod.data(); b.derivs = derivatives;
    switch (N) {
        case 1: run_n<1>(g, b, derivatives); break;
        case 2: run_n<2>(g, b, derivatives); break;
        case 4: run_n<4>(g, b, derivatives); break;
        case 8: run_n<8>(g, b, derivatives); break;
        case 16: run_n<16>(g, b, derivatives); break;
        default: return -2;
    }
    for (int w = 0; w < nw; ++w) {
        if (status) status[w] = stat[w];
        for (int l = 0; l < nlos; ++l) {
            size_t tid = size_t(w) * nlos + l;
            radiance[tid] = rad[tid];
            if (los_od) los_od[tid] = lod[tid];
            if (derivatives) {
                for (int q = 0; q < ngrid; ++q) {
                    if (d_k) d_k[tid * ngrid + q] = dnat[size_t(q) * nt + tid];
                    if (d_ssa) d_ssa[tid * ngrid + q] = dnat[size_t(ngrid + q) * nt + tid];
                }
                if (d_albedo) d_albedo[tid] = dnat[size_t(2 * ngrid) * nt + tid];
            }
        }
    }
    return 0;
}
}
