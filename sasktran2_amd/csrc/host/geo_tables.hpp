// Flattens skgeo::DiscoGeometry into the contiguous tables the device code (skd::Geo) reads.
// Host only; the arrays are uploaded once per engine.
#pragma once
#include <vector>

#include "../hip/disco_core.hpp"
#include "../hip/disco_vec_core.hpp"
#include "geometry.hpp"

namespace skhost {

struct GeoTables {
    int nstr = 0, N = 0, nlyr = 0, ngrid = 0, nlos = 0, nsza = 0, nangle = 0, npoint = 0, nground = 0, nslot = 0, nq_total = 0;
    std::vector<double> mu, wt, lpmu, lpcsz, lpang, dh, lfrac, wq_w, chapman, csz;
    std::vector<int> wq_idx, pt_start, pt_angle, pt_slot, gpt_start, gpt_slot;
    std::vector<int> ray_qstart, ray_ground, rl_od_start, od_idx, rl_ssa_start, ssa_idx, rl_active, rl_tab_start, tab_slot;
    std::vector<double> ray_gsaa, od_w, ssa_w, rl_azi, rl_cosm, rl_cosm_t, tab_w;
    std::vector<int> lp_start, lp_q, lp_angle, lp_slot, g_start, g_slot, g_sza;
    std::vector<double> lp_w, g_w, lp_pc;
    std::vector<int> slot_is_ground;  // [nslot]
    std::vector<double> gpt_mu;       // cos-angle grid value of every ground point of gpt_slot (outgoing zenith cosine of its ground source)
    // plane-parallel post-processing
    int pp_mode = 0;
    int include_ss = 1;  // set by the engine from single_scatter_source before the tables are uploaded
    std::vector<double> pp_mu, pp_azi, pp_lf, lplos;
    std::vector<int> pp_first;

    void build(const skgeo::DiscoGeometry& G) {
        nstr = G.nstr; N = nstr / 2; nlyr = G.nlyr; ngrid = G.ngrid; nlos = G.nlos; nsza = G.nsza; nangle = G.nangle;
        mu.assign(G.mu.begin(), G.mu.begin() + N);
        wt.assign(G.wt.begin(), G.wt.begin() + N);
        lpmu.resize(size_t(nstr) * N * nstr);
        for (int m = 0; m < nstr; ++m)
            for (int i = 0; i < N; ++i)
                for (int l = 0; l < nstr; ++l) lpmu[(size_t(m) * N + i) * nstr + l] = G.lpmu(m, i, l);
        lpcsz = G.lp_csz;
        lpang = G.lp_angle;
        csz = G.csz;
        // one more "solar zenith index" (s = nsza) for the thermal pass of the plane-parallel engine: an isotropic source is the solar
        // machinery with Legendre functions of the sun (1, 0, 0, ...) at order 0 and no direct beam on the ground (csz = 0)
        lpcsz.resize(size_t(nsza + 1) * nstr * nstr, 0.0);
        lpcsz[size_t(nsza) * nstr * nstr] = 1.0;
        csz.push_back(0.0);
        chapman = G.chapman;
        dh.resize(nlyr); lfrac.resize(nlyr); wq_idx.assign(2 * nlyr, -1); wq_w.assign(2 * nlyr, 0.0);
        for (int p = 0; p < nlyr; ++p) {
            dh[p] = G.ceil_h[p] - G.floor_h[p];
            double altitude = (G.ceil_h[p] + G.floor_h[p]) / 2.0;
            lfrac[p] = (G.ceil_h[p] - altitude) / (G.ceil_h[p] - G.floor_h[p]);
            int c = 0;
            for (int q = 0; q < ngrid; ++q)
                if (G.W(p, q) > 0) {
                    if (c >= 2) throw std::runtime_error("optical interpolator row has more than two entries");
                    wq_idx[2 * p + c] = q;
                    wq_w[2 * p + c] = G.W(p, q);
                    ++c;
                }
        }
        pp_mode = G.spherical_pp ? 0 : 1;
        if (pp_mode) { build_plane_parallel(G); return; }
        // slots
        std::vector<int> point_slot(G.need_point.size(), -1), ground_slot(G.need_ground.size(), -1);
        npoint = 0;
        for (size_t i = 0; i < G.need_point.size(); ++i)
            if (G.need_point[i]) point_slot[i] = npoint++;
        nground = 0;
        for (size_t i = 0; i < G.need_ground.size(); ++i)
            if (G.need_ground[i]) ground_slot[i] = npoint + nground++;
        nslot = npoint + nground;
        slot_is_ground.assign(nslot, 0);
        for (int i = npoint; i < nslot; ++i) slot_is_ground[i] = 1;
        pt_start.assign(size_t(nsza) * nlyr + 1, 0);
        for (int s = 0; s < nsza; ++s)
            for (int p = 0; p < nlyr; ++p) {
                const int lidx = nlyr - 1 - p;
                for (int a = 0; a < nangle; ++a) {
                    int slot = point_slot[(size_t(s) * nlyr + lidx) * nangle + a];
                    if (slot >= 0) { pt_angle.push_back(a); pt_slot.push_back(slot); }
                }
                pt_start[size_t(s) * nlyr + p + 1] = (int)pt_slot.size();
            }
        gpt_start.assign(nsza + 1, 0);
        for (int s = 0; s < nsza; ++s) {
            for (int a = 0; a < nangle; ++a) {
                int slot = ground_slot[size_t(s) * nangle + a];
                if (slot >= 0) { gpt_slot.push_back(slot); gpt_mu.push_back(G.cos_angles[a]); }
            }
            gpt_start[s + 1] = (int)gpt_slot.size();
        }
        // rays
        ray_qstart.assign(nlos + 1, 0); ray_ground.assign(nlos, 0); ray_gsaa.assign(nlos, 0.0);
        rl_od_start.assign(1, 0); rl_ssa_start.assign(1, 0); rl_tab_start.assign(1, 0);
        g_start.assign(nlos + 1, 0);
        struct LpE { int q, angle, slot; double w; };
        std::vector<std::vector<LpE>> lp(size_t(nlos) * nsza * nlyr);
        int q = 0;
        for (int l = 0; l < nlos; ++l) {
            const auto& ray = G.rays[l];
            ray_ground[l] = ray.ground_is_hit ? 1 : 0;
            ray_gsaa[l] = ray.ground_saa;
            for (const auto& L : ray.layers) {
                for (size_t t = 0; t < L.od.idx.size(); ++t) { od_idx.push_back(L.od.idx[t]); od_w.push_back(L.od.w[t]); }
                rl_od_start.push_back((int)od_idx.size());
                for (size_t t = 0; t < L.ssa_interp.idx.size(); ++t) { ssa_idx.push_back(L.ssa_interp.idx[t]); ssa_w.push_back(L.ssa_interp.w[t]); }
                rl_ssa_start.push_back((int)ssa_idx.size());
                rl_azi.push_back(L.azi);
                rl_active.push_back(L.layer_distance < 1e-4 ? 0 : 1);
                for (const auto& e : L.table) {
                    int slot = point_slot[(size_t(e.sza) * nlyr + e.alt) * nangle + e.angle];
                    // (zero-weight stencil entries stay table points that are solved and tested for convergence, as in the reference, but the
                    // line-of-sight sums do not read them)
                    if (e.w != 0.0) {
                        tab_slot.push_back(slot);
                        tab_w.push_back(e.w);
                    }
                    const int p = nlyr - 1 - e.alt;
                    // (an entry of weight zero adds nothing to any sum over the group: a segment that spans a whole layer sits exactly on a
                    // table altitude, and half of its 2 x 2 stencil then carries 0 -- at BASELINE configs[1] half of all entries)
                    if (e.w != 0.0) lp[(size_t(l) * nsza + e.sza) * nlyr + p].push_back({q, e.angle, slot, e.w});
                }
                rl_tab_start.push_back((int)tab_slot.size());
                ++q;
            }
            ray_qstart[l + 1] = q;
            for (const auto& e : ray.ground) {
                g_slot.push_back(ground_slot[size_t(e.sza) * nangle + e.angle]);
                g_sza.push_back(e.sza);
                g_w.push_back(e.w);
            }
            g_start[l + 1] = (int)g_slot.size();
        }
        nq_total = q;
        rl_cosm.resize(size_t(nstr) * nq_total);
        for (int m = 0; m < nstr; ++m)
            for (int i = 0; i < nq_total; ++i) rl_cosm[size_t(m) * nq_total + i] = std::cos(m * rl_azi[i]);
        rl_cosm_t.resize(rl_cosm.size());
        for (int m = 0; m < nstr; ++m)
            for (int i = 0; i < nq_total; ++i) rl_cosm_t[size_t(i) * nstr + m] = rl_cosm[size_t(m) * nq_total + i];
        // 16 streams (matrix-core sweeps): every non-empty (LOS, sza, layer) group is padded to a multiple of 4 entries with zero-weight
        // copies of its first entry, so that kern_k2b_mfma8 reads four weights with two 16-byte loads and their live flags as half a byte
        if (nstr == 16)
            for (auto& grp : lp)
                while (grp.size() % 4 != 0) { LpE e = grp.front(); e.w = 0.0; grp.push_back(e); }
        lp_start.assign(lp.size() + 1, 0);
        for (size_t i = 0; i < lp.size(); ++i) {
            for (const auto& e : lp[i]) { lp_q.push_back(e.q); lp_angle.push_back(e.angle); lp_slot.push_back(e.slot); lp_w.push_back(e.w); }
            lp_start[i + 1] = (int)lp_q.size();
        }
        if (nstr == 16) {
            // per azimuth order and entry: cos(m azimuth of the entry's ray layer) P_l^m(table angle), l >= m, in the operand order of the
            // sweeps -- lp_pc[((m n_lp + e) 4 + q) 4 + kk] = value for l = 4 kk + q: one contiguous 32 bytes per lane, nothing indirect
            const size_t ne = lp_q.size();
            lp_pc.assign(size_t(nstr) * ne * 16, 0.0);
            for (int m = 0; m < nstr; ++m)
                for (size_t e = 0; e < ne; ++e) {
                    const double cm = rl_cosm[size_t(m) * nq_total + lp_q[e]];
                    for (int qq = 0; qq < 4; ++qq)
                        for (int kk = 0; kk < 4; ++kk) {
                            const int l = 4 * kk + qq;
                            lp_pc[((size_t(m) * ne + e) * 4 + qq) * 4 + kk] = (l >= m) ? cm * lpang[(size_t(m) * nangle + lp_angle[e]) * nstr + l] : 0.0;
                        }
                }
        }
    }

    // plane-parallel mode: no diffuse table and no traced rays; every line of sight is (cos zenith, azimuth, observer altitude).
    // The observer layer / layer fraction follow OpticalLayerArray::opticalDepthAt and the loop of
    // DOSourcePlaneParallelPostProcessing::calculate (do_source_planeparallel.cpp:19-218).
    void build_plane_parallel(const skgeo::DiscoGeometry& G) {
        npoint = nground = nslot = nq_total = 0;
        pt_start.assign(size_t(nsza) * nlyr + 1, 0);
        gpt_start.assign(nsza + 1, 0);
        ray_qstart.assign(nlos + 1, 0); ray_ground.assign(nlos, 0); ray_gsaa.assign(nlos, 0.0);
        rl_od_start.assign(1, 0); rl_ssa_start.assign(1, 0); rl_tab_start.assign(1, 0);
        g_start.assign(nlos + 1, 0);
        lp_start.assign(size_t(nlos) * nsza * nlyr + 1, 0);
        lplos = G.lp_los;
        pp_mu.resize(nlos); pp_azi.resize(nlos); pp_lf.assign(nlos, 1.0); pp_first.assign(nlos, 0);
        for (int l = 0; l < nlos; ++l) {
            const auto& los = G.pp_los[l];
            pp_mu[l] = los.coszenith;
            pp_azi[l] = los.azimuth;
            const double alt = los.observer_altitude;
            if (alt < 0 || alt > G.ceil_h[0]) continue;  // observer outside the atmosphere: the whole column is seen
            int pl = nlyr - 1;
            for (int p = 0; p < nlyr; ++p)
                if (alt <= G.ceil_h[p] && alt >= G.floor_h[p]) { pl = p; break; }
            pp_first[l] = pl;
            pp_lf[l] = 1.0 - (G.ceil_h[pl] - alt) / (G.ceil_h[pl] - G.floor_h[pl]);
        }
    }

    // view over host memory (emulation) -- the HIP driver builds the same struct over device copies
    skd::Geo host_view() const {
        skd::Geo g{};
        fill_scalars(g);
        g.mu = mu.data(); g.wt = wt.data(); g.lpmu = lpmu.data(); g.lpcsz = lpcsz.data(); g.lpang = lpang.data();
        g.dh = dh.data(); g.lfrac = lfrac.data(); g.wq_idx = wq_idx.data(); g.wq_w = wq_w.data();
        g.chapman = chapman.data(); g.csz = csz.data(); g.pt_start = pt_start.data(); g.pt_angle = pt_angle.data();
        g.pt_slot = pt_slot.data(); g.gpt_start = gpt_start.data(); g.gpt_slot = gpt_slot.data();
        g.ray_qstart = ray_qstart.data(); g.ray_ground = ray_ground.data(); g.ray_gsaa = ray_gsaa.data();
        g.rl_od_start = rl_od_start.data(); g.od_idx = od_idx.data(); g.od_w = od_w.data();
        g.rl_ssa_start = rl_ssa_start.data(); g.ssa_idx = ssa_idx.data(); g.ssa_w = ssa_w.data();
        g.rl_azi = rl_azi.data(); g.rl_cosm = rl_cosm.data(); g.rl_cosm_t = rl_cosm_t.data(); g.rl_active = rl_active.data(); g.rl_tab_start = rl_tab_start.data();
        g.tab_slot = tab_slot.data(); g.tab_w = tab_w.data(); g.lp_start = lp_start.data(); g.lp_q = lp_q.data();
        g.lp_angle = lp_angle.data(); g.lp_slot = lp_slot.data(); g.lp_w = lp_w.data(); g.lp_pc = lp_pc.data(); g.g_start = g_start.data();
        g.g_slot = g_slot.data(); g.g_sza = g_sza.data(); g.g_w = g_w.data();
        g.pp_mu = pp_mu.data(); g.pp_azi = pp_azi.data(); g.pp_first = pp_first.data(); g.pp_lf = pp_lf.data(); g.lplos = lplos.data();
        return g;
    }
    void fill_scalars(skd::Geo& g) const {
        g.nstr = nstr; g.N = N; g.nlyr = nlyr; g.ngrid = ngrid; g.nlos = nlos; g.nsza = nsza; g.nangle = nangle;
        g.npoint = npoint; g.nground = nground; g.nslot = nslot; g.nq_total = nq_total;
        g.pp_mode = pp_mode;
        g.n_lp = (int)lp_q.size();
        g.include_ss = include_ss;
    }
};

// Tables of the polarised (NSTOKES = 3) path that go with a GeoTables: the R / T companions of the Legendre tables (same
// shapes; zero-filled for NSTOKES = 1) and the sine azimuth factors of the ray layers (U component).
// Geometry tables of a non-Lambertian surface for plane-parallel / pseudo-spherical post-processing (skv::SurfGeo): built once per engine.
// Pairs in the order of skv::VSurf: (stream i, stream t), (stream i, sun), (line of sight, stream t), (line of sight, sun).
struct SurfaceTables {
    int NP = 0, NQ = 0, nstr = 0;
    std::vector<double> cw, factor0, factor, modis_ke, kok_r0, kok_kk;
    // sza: solar zenith index the sun pairs (and, in spherical geometry, the ground points) belong to
    template <class KernelsFn, class PartsFn> void build(const GeoTables& t, KernelsFn modis_kernels, PartsFn kokhanovsky_parts, int sza = 0) {
        const double pi = 3.14159265358979323846;
        const bool pp = !t.pp_mu.empty();
        // "line of sight" rows: the lines of sight with plane-parallel post-processing, the ground points of this sza in spherical geometry
        const int N = t.N, g0 = pp ? 0 : t.gpt_start[sza], nlos = pp ? t.nlos : t.gpt_start[sza + 1] - g0;
        nstr = t.nstr;
        NP = N * N + N + nlos * N + nlos;
        // the two 512-point Gauss-Legendre panels over the azimuth of SurfaceStorage::compute_expansion (sktran_do_surface.h:62-93)
        std::vector<double> x, w, aq, wq;
        skgeo::gauss_legendre(512, x, w);
        for (size_t i = 0; i < x.size(); ++i)
            if (x[i] > 0)
                for (double a : {0.5 * x[i] + 0.5, -0.5 * x[i] + 0.5, 0.5 * x[i] - 0.5, -0.5 * x[i] - 0.5}) { aq.push_back(a); wq.push_back(0.5 * w[i]); }
        NQ = int(aq.size());
        cw.resize(size_t(nstr) * NQ);
        for (int m = 0; m < nstr; ++m)
            for (int q = 0; q < NQ; ++q) cw[size_t(m) * NQ + q] = wq[q] * std::cos(m * pi * aq[q]) * 0.5 * pi * (m == 0 ? 1.0 : 2.0);
        std::vector<double> mo(NP), mi(NP);
        factor0.resize(NP); factor.resize(NP);
        const double csz0 = t.csz[sza];
        auto set = [&](int pr, double mu_out, double mu_in, int in_stream) {
            mo[pr] = mu_out; mi[pr] = mu_in;
            const double f = in_stream >= 0 ? t.wt[in_stream] * t.mu[in_stream] : csz0 / pi;   // (1 + delta_m0) w mu for a stream, cos(sza) / pi for the sun
            factor0[pr] = in_stream >= 0 ? 2.0 * f : f;
            factor[pr] = f;
        };
        for (int i = 0; i < N; ++i) {
            for (int tt = 0; tt < N; ++tt) set(i * N + tt, t.mu[i], t.mu[tt], tt);
            set(N * N + i, t.mu[i], csz0, -1);
        }
        for (int l = 0; l < nlos; ++l) {
            const double mu_l = pp ? t.pp_mu[l] : t.gpt_mu[g0 + l];
            for (int tt = 0; tt < N; ++tt) set(N * N + N + l * N + tt, mu_l, t.mu[tt], tt);
            set(N * N + N + nlos * N + l, mu_l, csz0, -1);   // (spherical geometry: unused -- the direct bounce belongs to the single-scatter source)
        }
        modis_ke.assign(size_t(3) * nstr * NP, 0.0);
        kok_r0.resize(size_t(NP) * NQ); kok_kk.resize(NP);
        for (int pr = 0; pr < NP; ++pr) {
            for (int q = 0; q < NQ; ++q) {
                double kv, kg, r0, kk;
                modis_kernels(mi[pr], mo[pr], pi * aq[q], kv, kg);
                kokhanovsky_parts(mi[pr], mo[pr], pi * aq[q], r0, kk);
                kok_r0[size_t(pr) * NQ + q] = r0;
                kok_kk[pr] = kk;
                for (int m = 0; m < nstr; ++m) {
                    const double c = cw[size_t(m) * NQ + q] / pi;   // the kernels enter the BRDF divided by pi
                    modis_ke[(size_t(0) * nstr + m) * NP + pr] += c * kv;
                    modis_ke[(size_t(1) * nstr + m) * NP + pr] += c * kg;
                    // kind 3: abs(cos(caz_diff)) * abs(1 - abs(abs(out) - abs(in))) in the legacy convention rho = pi * brdf (test_scalar.cpp:826-831)
                    modis_ke[(size_t(2) * nstr + m) * NP + pr] += c * std::abs(std::cos(pi * aq[q])) * std::abs(1 - std::abs(std::abs(mo[pr]) - std::abs(mi[pr])));
                }
            }
        }
    }
};

struct VecTables {
    int S = 1;
    std::vector<double> lpmuR, lpmuT, lpangR, lpangT, lplosR, lplosT, rl_sinm_t, stokes_c, stokes_s;
    void build(const skgeo::DiscoGeometry& G, const GeoTables& t) {
        S = G.nstokes;
        const int nstr = t.nstr, N = t.N;
        lpmuR.assign(t.lpmu.size(), 0.0);
        lpmuT.assign(t.lpmu.size(), 0.0);
        lpangR.assign(t.lpang.size(), 0.0);
        lpangT.assign(t.lpang.size(), 0.0);
        lplosR.assign(t.lplos.size(), 0.0);
        lplosT.assign(t.lplos.size(), 0.0);
        if (S == 3) {
            for (int m = 0; m < nstr; ++m)
                for (int i = 0; i < N; ++i)
                    for (int l = 0; l < nstr; ++l) {
                        lpmuR[(size_t(m) * N + i) * nstr + l] = G.lp_mu_R[(size_t(m) * nstr + i) * nstr + l];
                        lpmuT[(size_t(m) * N + i) * nstr + l] = G.lp_mu_T[(size_t(m) * nstr + i) * nstr + l];
                    }
            if (!G.lp_angle_R.empty() && lpangR.size() == G.lp_angle_R.size()) { lpangR = G.lp_angle_R; lpangT = G.lp_angle_T; }
            if (!G.lp_los_R.empty() && lplosR.size() == G.lp_los_R.size()) { lplosR = G.lp_los_R; lplosT = G.lp_los_T; }
        }
        stokes_c = G.stokes_C;
        stokes_s = G.stokes_S;
        rl_sinm_t.assign(t.rl_azi.size() * size_t(nstr), 0.0);
        for (size_t q = 0; q < t.rl_azi.size(); ++q)
            for (int m = 0; m < nstr; ++m) rl_sinm_t[q * nstr + m] = std::sin(m * t.rl_azi[q]);
    }
    skv::VGeo host_view() const {
        skv::VGeo v{};
        v.S = S;
        v.lpmuR = lpmuR.data(); v.lpmuT = lpmuT.data(); v.lpangR = lpangR.data(); v.lpangT = lpangT.data();
        v.lplosR = lplosR.data(); v.lplosT = lplosT.data(); v.rl_sinm_t = rl_sinm_t.data();
        v.stokes_c = stokes_c.data(); v.stokes_s = stokes_s.data();
        return v;
    }
};

}  // namespace skhost
