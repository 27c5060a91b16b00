// TEST INFRASTRUCTURE: runs the two-stream kernel bodies (sasktran2_amd/csrc/hip/ts_core.hpp, compiled here with g++) on the CPU, one
// loop iteration per GPU thread, in the launch order of the HIP driver (sasktran2_amd/csrc/hip/ts_engine.hip).  Lets the kernel math
// be checked against the oracle in the GPU-less authoring container; it is not part of the product library and is never a fallback.
#include <cstdio>
#include <memory>
#include <vector>

#include "../sasktran2_amd/csrc/host/ts_geometry.hpp"
#include "../sasktran2_amd/csrc/hip/ts_core.hpp"

namespace {
struct Handle {
    skgeo::Geometry1D geo;
    skts::TsGeometry tg;
};
}  // namespace

extern "C" {

void* emts_create(double cos_sza, double saa, double earth_radius, const double* altitudes, int ngrid, int interp_method, int geotype,
                  int nrays, const int* ray_kinds, const double* ray_params, int num_sza, int solar, int thermal) {
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
        skts::Settings cfg;
        cfg.num_sza = num_sza;
        cfg.solar = solar != 0;
        cfg.thermal = thermal != 0;
        skts::build_ts_geometry(h->geo, specs, cfg, h->tg);
        return h.release();
    } catch (const std::exception& e) {
        std::fprintf(stderr, "emts_create: %s\n", e.what());
        return nullptr;
    }
}
void emts_destroy(void* h) { delete static_cast<Handle*>(h); }

// same arguments as orts_run (oracle/oracle_capi.cpp) except the output layout, which is the product's:
// radiance [nwavel][nlos]; wf_ext / wf_ssa / wf_b1 / wf_em [ngrid][nwavel][nlos]; wf_alb / wf_sem [nwavel][nlos] (all or none)
int emts_run(void* hv, int nwavel, int nleg, const double* extinction, const double* ssa, const double* leg, const double* delta_f,
             const double* emission, const double* surface_emission, const double* solar_irradiance, const double* albedo,
             double* radiance, double* wf_ext, double* wf_ssa, double* wf_b1, double* wf_em, double* wf_alb, double* wf_sem, int dual_adjoint, int allow_fused) {
    // dual_adjoint 0: as the HIP driver (hand-written reverse sweep of the layers); 1: the same Lagrangian differentiated in dual numbers
    auto* h = static_cast<Handle*>(hv);
    if (!h) return -1;
    const skts::TsGeometry& tg = h->tg;
    skt::Geo g{};
    g.nlyr = tg.nlyr; g.ngrid = tg.ngrid; g.nlos = tg.nlos; g.ncol = tg.ncol; g.spherical = tg.spherical ? 1 : 0; g.mu = tg.mu;
    g.max_seg = 0;
    for (int i = 0; i + 1 < (int)tg.ray_start.size(); ++i) g.max_seg = std::max(g.max_seg, tg.ray_start[i + 1] - tg.ray_start[i]);
    g.thickness = tg.thickness.data(); g.csz = tg.csz.data(); g.chapman = tg.chapman.data();
    g.chapman_t = tg.chapman_t.data(); g.chap_uniform = tg.chap_uniform.data();
    g.view_cos = tg.view_cos.data(); g.view_azimuth = tg.view_azimuth.data();
    g.view_cols = tg.view_cols.data();
    g.ray_start = tg.ray_start.data(); g.ground_hit = tg.ground_hit.data(); g.ground_col = tg.ground_col.data(); g.ground_colw = tg.ground_colw.data();
    g.seg_layer = tg.seg_layer.data(); g.seg_frac = tg.seg_frac.data(); g.seg_cos = tg.seg_cos.data(); g.seg_azimuth = tg.seg_azimuth.data();
    g.seg_col = tg.seg_col.data(); g.seg_colw = tg.seg_colw.data();
    g.od_start = tg.od_start.data(); g.od_level = tg.od_level.data(); g.od_weight = tg.od_weight.data();

    const bool derivs = wf_ext != nullptr;
    skt::Bufs b{};
    b.nw = nwavel; b.W = nwavel; b.w0 = 0; b.nw_total = nwavel; b.nleg = nleg; b.derivs = derivs ? 1 : 0;
    b.ext = extinction; b.ssa = ssa; b.leg = leg; b.delta_f = delta_f; b.albedo = albedo; b.irradiance = solar_irradiance;
    b.emission = emission; b.surface_emission = surface_emission;
    b.radiance = radiance; b.wf_ext = wf_ext; b.wf_ssa = wf_ssa; b.wf_b1 = wf_b1; b.wf_alb = wf_alb; b.wf_em = wf_em; b.wf_sem = wf_sem;
    const skt::ScratchSizes z = skt::scratch_sizes(g.nlyr, g.ngrid, g.nlos, g.ncol, g.max_seg, g.spherical != 0, derivs, size_t(b.W));
    std::vector<double> lay(z.lay), lev(z.lev), colv(z.colv), lu(z.lu), lam(z.lam), g5(z.g5), gsum(z.gsum), vstate(z.vstate), path(z.path), glev(z.glev);
    b.lay = lay.data(); b.lev = lev.data(); b.colv = colv.data(); b.lu = lu.data(); b.lam = lam.data(); b.g5 = g5.data(); b.gsum = gsum.data();
    b.path = path.data(); b.glev = glev.data(); b.vstate = vstate.data();
    std::vector<double> lds(size_t(g.nlyr) + 1);
    int pass = 0;
    for (int solar = 1; solar >= 0; --solar) {
        if (solar ? !tg.cfg.solar : !tg.cfg.thermal) continue;
        if (!solar && !emission) return -2;
        b.accumulate = pass > 0 ? 1 : 0;
        // plane-parallel geometry with one line of sight: the two fused passes of the HIP driver
        const bool fused = allow_fused && !g.spherical && g.nlos == 1 && !dual_adjoint;
        if (fused) {
            for (int w = 0; w < nwavel; ++w) {
                if (solar) skt::solve_thread<true, true>(g, b, w, lds.data(), 1);
                else skt::solve_thread<false, true>(g, b, w, lds.data(), 1);
            }
            for (int w = 0; w < nwavel; ++w) {
                if (derivs) {
                    if (solar) skt::view_adjoint_thread<true, false, false, true>(g, b, w, 0, lds.data(), 1);
                    else skt::view_adjoint_thread<false, false, false, true>(g, b, w, 0, lds.data(), 1);
                } else {
                    if (solar) skt::fused_radiance_thread<true>(g, b, w);
                    else skt::fused_radiance_thread<false>(g, b, w);
                }
            }
            ++pass;
            continue;
        }
        for (int w = 0; w < nwavel; ++w) {
            if (solar) skt::solve_thread<true>(g, b, w, lds.data(), 1);
            else skt::solve_thread<false>(g, b, w, lds.data(), 1);
        }
        auto forward = [&](int w, int v) {
            const int key = (solar ? 4 : 0) + (derivs ? 2 : 0) + (g.spherical ? 1 : 0);
            switch (key) {
                case 0: skt::view_forward_thread<false, false, false>(g, b, w, v); break;
                case 1: skt::view_forward_thread<false, false, true>(g, b, w, v); break;
                case 2: skt::view_forward_thread<false, true, false>(g, b, w, v); break;
                case 3: dual_adjoint ? skt::view_forward_thread<false, true, true, true>(g, b, w, v) : skt::view_forward_thread<false, true, true>(g, b, w, v); break;
                case 4: skt::view_forward_thread<true, false, false>(g, b, w, v); break;
                case 5: skt::view_forward_thread<true, false, true>(g, b, w, v); break;
                case 6: skt::view_forward_thread<true, true, false>(g, b, w, v); break;
                default: dual_adjoint ? skt::view_forward_thread<true, true, true, true>(g, b, w, v) : skt::view_forward_thread<true, true, true>(g, b, w, v); break;
            }
        };
        auto adjoint = [&](int w, int v) {
            const int key = (solar ? 4 : 0) + (g.spherical ? 2 : 0) + (dual_adjoint ? 1 : 0);
            switch (key) {
                case 0: skt::view_adjoint_thread<false, false, false>(g, b, w, v, lds.data(), 1); break;
                case 1: skt::view_adjoint_thread<false, false, true>(g, b, w, v, lds.data(), 1); break;
                case 2: skt::view_adjoint_thread<false, true, false>(g, b, w, v, lds.data(), 1); break;
                case 3: skt::view_adjoint_thread<false, true, true>(g, b, w, v, lds.data(), 1); break;
                case 4: skt::view_adjoint_thread<true, false, false>(g, b, w, v, lds.data(), 1); break;
                case 5: skt::view_adjoint_thread<true, false, true>(g, b, w, v, lds.data(), 1); break;
                case 6: skt::view_adjoint_thread<true, true, false>(g, b, w, v, lds.data(), 1); break;
                default: skt::view_adjoint_thread<true, true, true>(g, b, w, v, lds.data(), 1); break;
            }
        };
        for (int v = 0; v < g.nlos; ++v)
            for (int w = 0; w < nwavel; ++w) forward(w, v);
        if (derivs)
            for (int v = 0; v < g.nlos; ++v)
                for (int w = 0; w < nwavel; ++w) adjoint(w, v);
        ++pass;
    }
    return 0;
}

}  // extern "C"
