// TEST INFRASTRUCTURE: runs the task bodies of the polarised path (sasktran2_amd/csrc/hip/disco_vec_core.hpp, compiled here
// with g++) on the CPU, one loop iteration per GPU thread, in the launch order of the HIP driver
// (sasktran2_amd/csrc/hip/disco_vec_engine.inc).  Not part of the product library and never a fallback.
#include <cstdio>
#include <cstring>
#include <memory>
#include <cstdlib>
#include <vector>

#include "../sasktran2_amd/csrc/host/geo_tables.hpp"

namespace {
struct Handle {
    skgeo::Geometry1D geo;
    skgeo::DiscoGeometry dg;
    skhost::GeoTables tab;
    skhost::VecTables vtab;
    // non-Lambertian surface of the next emuv_run calls (emuv_set_brdf)
    int brdf_kind = 0, brdf_nargs = 1;
    const double* brdf_args = nullptr;   // [nwavel][nargs]
    double* d_brdf_out = nullptr;        // [nwavel][nlos][S][nargs]
};
}  // namespace

extern "C" {
void* emuv_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method,
                  int geotype, int nstr, int num_sza, int nrays, const int* ray_kinds, const double* ray_params, int ss_in_do,
                  int nstokes, int stokes_basis) {
    try {
        auto h = std::make_unique<Handle>();
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
        skgeo::build_disco_geometry(h->geo, specs, nstr, num_sza, h->dg, nstokes, stokes_basis);
        h->tab.include_ss = ss_in_do;
        h->tab.build(h->dg);
        h->vtab.build(h->dg, h->tab);
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emuv_create: %s\n", e.what());
        return nullptr;
    }
}
void emuv_destroy(void* h) { delete static_cast<Handle*>(h); }
int emuv_set_brdf(void* hv, int kind, int nargs, const double* args, double* d_brdf_out) {
    auto* h = static_cast<Handle*>(hv);
    if (!h || kind < 0 || kind > 3) return -1;
    h->brdf_kind = kind;
    h->brdf_nargs = kind == 0 ? 1 : nargs;
    h->brdf_args = kind == 0 ? nullptr : args;
    h->d_brdf_out = kind == 0 ? nullptr : d_brdf_out;
    return 0;
}

// radiance: [nwavel][nlos][nstokes];  leg: [nwavel][ngrid][nleg][4] (reference storage order, nstokes = 3) or [..][nleg]
int emuv_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg,
             const double* solar_irradiance, const double* albedo, const double* delta_f, double* radiance, double* los_od, int* status,
             int derivatives, double* d_k, double* d_ssa, double* d_albedo, int ngroups, const double* d_legendre, double* d_scat) {
    auto* h = static_cast<Handle*>(hv);
    const skhost::GeoTables& t = h->tab;
    skd::Geo g = t.host_view();
    skv::VGeo vg = h->vtab.host_view();
    const int S = vg.S, N = t.N, n = S * N, nlyr = t.nlyr, nstr = t.nstr, nw = nwavel, nlos = t.nlos;
    const size_t nt = size_t(nw) * nlos;
    std::vector<double> tau(size_t(nlyr) * nw), om(tau.size()), kbar(tau.size()), a(size_t(nlyr) * nstr * nw), T(size_t(nlyr + 1) * nw),
        sec(tau.size()), od(size_t(t.nq_total) * nw), omloc(od.size()), srcw(od.size()), att_total(size_t(nlos) * nw), lod(nt);
    std::vector<uint8_t> conv(size_t(t.nslot) * nw);
    for (int sl = 0; sl < t.nslot; ++sl)
        for (int w = 0; w < nw; ++w) conv[size_t(w) * t.nslot + sl] = (t.slot_is_ground[sl] && h->brdf_kind == 0) ? 1 : 255;
    std::vector<int> notconv(nw, 0), norders(nw, nstr), stat(nw, 0);
    skd::Bufs b{};
    b.nw = nw; b.ext = extinction; b.ssa = ssa; b.leg = nullptr; b.nleg = 0; b.albedo = albedo; b.irradiance = solar_irradiance; b.delta_f = delta_f;
    b.tau = tau.data(); b.om = om.data(); b.kbar = kbar.data(); b.a = a.data(); b.T = T.data(); b.sec = sec.data();
    b.od = od.data(); b.omloc = omloc.data(); b.srcw = srcw.data(); b.att_total = att_total.data();
    b.conv_order = conv.data(); b.notconv = notconv.data(); b.norders = norders.data(); b.status = stat.data(); b.los_od = lod.data();
    const skv::VRec R(n);
    std::vector<double> a4(size_t(nlyr) * nstr * 4 * nw), rec(size_t(nlyr) * skv::VRec(n, derivatives ? ngroups : 0).size() * nw), dinv(size_t(nlyr) * 4 * n * n * nw),
        xsol(size_t(nlyr) * 2 * n * nw), J(size_t(nstr) * std::max(t.nslot, 1) * nw * S, 0.0), rad(nt * S, 0.0);
    const int NL = skv::legendre_space(g, S), ngrid = t.ngrid;
    const size_t nt3 = nt * S;
    std::vector<double> scratch(std::max({skv::kv1_scratch(n), skv::kv2_scratch(n), skv::kv3_scratch(n, nstr), skv::kvpp_scratch(n, ngroups),
                                          skv::kv2c_scratch(n, NL, ngroups), skv::kv2b_scratch(n, NL), skv::kv2bt_scratch(n, NL, nlos * S)}));
    const int G = derivatives ? ngroups : 0;
    std::vector<double> ops(derivatives ? size_t(nlyr) * skv::VOps(n, NL, G).size() * nw : 1), zb(derivatives ? size_t(nlyr) * 2 * n * nt3 : 1),
        qlb(derivatives ? size_t(nlyr) * NL * nt3 : 1), grad(derivatives ? size_t(4 * nlyr + 2 + G * nlyr + 2) * nt3 : 1, 0.0),
        dnat(derivatives ? size_t(2 * ngrid + 1 + G * ngrid + 2) * nt3 : 1, 0.0), da4(size_t(std::max(G, 1)) * nlyr * nstr * 4 * nw, 0.0);
    skv::VBufs v{};
    v.S = S; v.n = n; v.leg = leg; v.nleg = nleg; v.a4 = a4.data(); v.rec = rec.data(); v.dinv = dinv.data(); v.xsol = xsol.data();
    v.J = J.data(); v.radiance = rad.data(); v.scratch = scratch.data(); v.scratch_stride = 1;
    v.ops = ops.data(); v.z = zb.data(); v.qlbuf = qlb.data(); v.grad = grad.data(); v.dnat = dnat.data(); v.derivs = derivatives;
    v.ngroups = G; v.dleg = d_legendre; v.dleg_nw = nw; v.dleg_w0 = 0; v.dfg = nullptr; v.da4 = da4.data();
    // non-Lambertian surface: the geometry tables and the reflection factors of every (pair, wavelength) for one solar zenith index, as the
    // device driver forms them (kern_surface_expand once per solar zenith index)
    skhost::SurfaceTables stab;
    std::vector<double> surf, surf_ops;
    v.brdf_kind = h->brdf_kind;
    v.brdf_nargs = h->brdf_nargs;
    v.surf_nlos = g.pp_mode ? nlos : 0;
    v.surf_slot0 = 0;
    auto surface_for = [&](int sz) {
        if (h->brdf_kind == 0) return;
        stab = skhost::SurfaceTables();
        stab.build(t, [](double mi, double mo, double phi, double& kv, double& kg) { skv::modis_kernels(mi, mo, phi, kv, kg); },
                   [](double mi, double mo, double phi, double& r0, double& kk) { skv::kokhanovsky_parts(mi, mo, phi, r0, kk); }, sz);
        skv::SurfGeo sg{stab.NP, stab.NQ, stab.nstr, stab.cw.data(), stab.factor0.data(), stab.factor.data(), stab.modis_ke.data(), stab.kok_r0.data(), stab.kok_kk.data()};
        surf.assign(size_t(nw) * nstr * (1 + h->brdf_nargs) * stab.NP, 0.0);
        surf_ops.assign(size_t(nw) * N * (2 * n + 1), 0.0);
        for (int w = 0; w < nw; ++w)
            for (int pr = 0; pr < stab.NP; ++pr)
                skv::surface_expand_pair(sg, h->brdf_kind, h->brdf_nargs, h->brdf_args + size_t(w) * h->brdf_nargs, pr,
                                         surf.data() + size_t(w) * nstr * (1 + h->brdf_nargs) * stab.NP);
        v.surf = surf.data();
        v.surf_ops = surf_ops.data();
        v.surf_nlos = g.pp_mode ? nlos : t.gpt_start[sz + 1] - t.gpt_start[sz];
        v.surf_slot0 = (g.pp_mode || v.surf_nlos == 0) ? 0 : t.gpt_slot[t.gpt_start[sz]];
    };
    v.k1_compact = skv::kv1_compact_layout(n, G);   // the work-space layout the device launcher chooses (its aliasing is the same on the CPU)
    const skv::Scr scr{scratch.data(), 1};
    const skv::Team team{0, 1};
    double* ws = scratch.data();
    // SKV_EMU_PARK=1: the eigen stage in its parked layout (E+- outside the work space, four matrix slots; the launcher's choice at 32 polarised streams)
    const char* park_env = std::getenv("SKV_EMU_PARK");
    std::vector<double> parkbuf((park_env && atoi(park_env) && G == 0) ? 2 * size_t(n) * n : 0);
    double* park = parkbuf.empty() ? nullptr : parkbuf.data();
    const char* k3_env = std::getenv("SKV_EMU_K3_TEAM");
    const bool k3_team = k3_env && atoi(k3_env) != 0;
    for (int w = 0; w < nw; ++w) {
        skd::k0_layer_optics(g, b, w);  // tau / omega / LOS shells; its scalar a_l is not used (nleg = 0)
        skv::kv0_coefficients(g, b, v, w);
    }
    if (g.pp_mode) {
        surface_for(0);
        for (int w = 0; w < nw; ++w) skd::k0_transmission(g, b, 0, w);
        for (int m = 0; m < nstr; ++m) {
            for (int w = 0; w < nw; ++w)
                for (int p = 0; p < nlyr; ++p) skv::kv1_layer_solution(team, g, vg, b, v, 0, m, p, w, ws, derivatives != 0, park);
            for (int w = 0; w < nw; ++w) skv::kv2_bvp(team, g, b, v, 0, m, w, ws, ws + skv::kv2_scratch_main(n));
            if (derivatives)
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < nlyr; ++p) skv::kv2c_layer_ops(team, g, vg, b, v, 0, m, p, w, ws);
            for (int w = 0; w < nw; ++w)
                for (int l = 0; l < nlos; ++l) skv::kvpp_los(g, vg, b, v, 0, m, l, w, scr);
        }
        if (derivatives) {
            skd::Geo g3 = g;
            g3.nlos = nlos * S;
            skd::Bufs b3 = b;
            b3.grad = grad.data();
            b3.dnat = dnat.data();
            for (int w = 0; w < nw; ++w)
                for (int l3 = 0; l3 < nlos * S; ++l3) {
                    skd::k4a_chain(g3, b3, 0, l3, w);
                    skv::kv4g_chain(g, b, v, w, size_t(w) * nlos * S + l3, nt3);
                    if (v.brdf_kind != 0) skv::kv4s_chain(g, v, size_t(w) * nlos * S + l3, nt3);
                }
        }
    } else {
        for (int s = 0; s < g.nsza; ++s) {
            surface_for(s);
            for (int w = 0; w < nw; ++w) {
                skd::k0_transmission(g, b, s, w);
                norders[w] = nstr;
                notconv[w] = 0;
            }
            for (int m = 0; m < nstr; ++m) {
                int active = 0;
                for (int w = 0; w < nw; ++w) active += (m < norders[w]);
                if (!active) break;
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < nlyr; ++p)
                        if (m < norders[w]) skv::kv1_layer_solution(team, g, vg, b, v, s, m, p, w, ws, derivatives != 0, park);
                for (int w = 0; w < nw; ++w)
                    if (m < norders[w]) skv::kv2_bvp(team, g, b, v, s, m, w, ws, ws + skv::kv2_scratch_main(n));
                for (int w = 0; w < nw; ++w)
                    for (int p = 0; p < nlyr; ++p)
                        if (m < norders[w]) {
                            if (k3_team) skv::kv3_table_values_team(team, g, vg, b, v, s, m, p, w, ws);   // SKV_EMU_K3_TEAM=1: the device's wavefront form
                            else skv::kv3_table_values(g, vg, b, v, s, m, p, w, scr);
                        }
                if (derivatives) {
                    for (int w = 0; w < nw; ++w)
                        for (int p = 0; p < nlyr; ++p)
                            if (m < norders[w]) skv::kv2c_layer_ops(team, g, vg, b, v, s, m, p, w, ws);
                    for (int w = 0; w < nw; ++w)
                        if (m < norders[w]) skv::kv2b_team(team, g, vg, b, v, s, m, w, ws);
                }
                for (int w = 0; w < nw; ++w) skd::k_stop_rule(g, b, s, m, w);
            }
            if (derivatives) {
                // reverse chain beam / optics -> native derivatives: the scalar task body with (LOS, Stokes) as the line-of-sight index
                skd::Geo g3 = g;
                g3.nlos = nlos * S;
                skd::Bufs b3 = b;
                b3.grad = grad.data();
                b3.dnat = dnat.data();
                for (int w = 0; w < nw; ++w)
                    for (int l3 = 0; l3 < nlos * S; ++l3) {
                        skd::k4a_chain(g3, b3, s, l3, w);
                        skv::kv4g_chain(g, b, v, w, size_t(w) * nlos * S + l3, nt3);
                        if (v.brdf_kind != 0) skv::kv4s_chain(g, v, size_t(w) * nlos * S + l3, nt3);
                    }
            }
        }
        for (int w = 0; w < nw; ++w)
            for (int l = 0; l < nlos; ++l) skv::kv4_los(g, vg, b, v, l, w);
    }
    for (size_t i = 0; i < nt; ++i) skv::stokes_rotate(vg, int(i % nlos), &rad[i * S], &radiance[i * S], S);
    for (size_t i = 0; i < nt; ++i)
        if (los_od) los_od[i] = lod[i];
    if (derivatives) {
        // [w][los][S][ngrid], Stokes rotation applied to the derivatives as to the radiance
        for (size_t i = 0; i < nt; ++i) {
            const int los = int(i % nlos);
            for (int q = 0; q <= 2 * ngrid; ++q) {
                double in[3] = {0, 0, 0}, out[3];
                for (int c = 0; c < S; ++c) in[c] = dnat[size_t(q) * nt3 + i * S + c];
                skv::stokes_rotate(vg, los, in, out, S);
                for (int c = 0; c < S; ++c) {
                    if (q < ngrid) { if (d_k) d_k[(i * S + c) * ngrid + q] = out[c]; }
                    else if (q < 2 * ngrid) { if (d_ssa) d_ssa[(i * S + c) * ngrid + (q - ngrid)] = out[c]; }
                    else if (d_albedo) d_albedo[i * S + c] = out[c];
                }
            }
            // BRDF arguments: d_brdf [w][los][S][nargs]
            for (int a = 0; a < v.brdf_nargs && h->d_brdf_out; ++a) {
                double in[3] = {0, 0, 0}, out[3];
                for (int c = 0; c < S; ++c) in[c] = dnat[skv::surf_dnat_row(g, v, a) * nt3 + i * S + c];
                skv::stokes_rotate(vg, los, in, out, S);
                for (int c = 0; c < S; ++c) h->d_brdf_out[(i * S + c) * v.brdf_nargs + a] = out[c];
            }
            // scattering groups: d_scat [w][los][S][G][ngrid]
            for (int gi = 0; gi < G; ++gi)
                for (int q = 0; q < ngrid; ++q) {
                    double in[3] = {0, 0, 0}, out[3];
                    for (int c = 0; c < S; ++c) in[c] = dnat[size_t(2 * ngrid + 1 + gi * ngrid + q) * nt3 + i * S + c];
                    skv::stokes_rotate(vg, los, in, out, S);
                    for (int c = 0; c < S; ++c)
                        if (d_scat) d_scat[((i * S + c) * G + gi) * ngrid + q] = out[c];
                }
        }
    }
    for (int w = 0; w < nw; ++w)
        if (status) status[w] = stat[w];
    return 0;
}
}
