// Wavelength-independent geometry of the two-stream source (multiple_scatter_source = twostream, emission_source = twostream).
// Host side only, computed once per engine; the per-wavelength work is in the consumers (HIP kernels of the product, CPU oracle of the
// tests).
//
// Reference (behaviour restated, nothing copied):
//   cpp/lib/sktran_disco/cpp_twostream_source.cpp:3348-3511   CppTwoStreamSourceAdapter::initialize_geometry
//     :3361-3386  solar-zenith columns: `num_do_sza` cosines evenly spaced between the smallest and largest cos(SZA) met by the
//                 traced lines of sight (spherical geometry only), otherwise the reference cos(SZA)
//     :3394-3412  one column per cosine: layer thickness and Chapman factors of the discrete-ordinates layer array
//     :3416-3428  plane parallel: one (cosine, relative azimuth) pair per line of sight, upwelling only
//     :3429-3505  spherical: one segment per traced layer, far end first: layer / fraction-from-top of the source point, the local
//                 propagation cosine, relative azimuth, cos(SZA) and the optical-depth stencil over the levels (top-down numbering)
//   :796-806      normalized_source_weights (split of a segment between its two ends)
//   :1128-1146    interpolation of a cos(SZA) in the column grid
#pragma once
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <vector>

#include "geometry.hpp"

namespace skts {

struct Settings {
    int num_sza = 1;
    bool solar = true;      // multiple_scatter_source = twostream
    bool thermal = false;   // emission_source = twostream
};

struct TsGeometry {
    Settings cfg;
    int nlyr = 0, ngrid = 0, nlos = 0, ncol = 1;
    bool spherical = false;
    double mu = 0.5;                     // the quadrature cosine of the two-stream model (ColumnGeometry::quadrature_cosine)
    std::vector<double> thickness;       // [nlyr], layers numbered from the top
    std::vector<double> csz;             // [ncol]
    std::vector<double> chapman;         // [ncol][boundary below layer b: nlyr][layer: nlyr]
    std::vector<double> chapman_t;       // [ncol][layer][boundary]: the same, transposed (reverse sweeps)
    std::vector<double> chap_uniform;    // [ncol] plane-parallel column: the one factor of every layer above a boundary; otherwise 0
    // plane parallel
    std::vector<double> view_cos, view_azimuth;   // [nlos]
    // spherical
    std::vector<int> ray_start;          // [nlos + 1] into the segment arrays
    std::vector<int> ground_hit;         // [nlos]
    std::vector<int> ground_col;         // [nlos][2] columns bracketing the cos(SZA) at the ground
    std::vector<double> ground_colw;     // [nlos] weight of the upper column
    std::vector<int> view_cols;          // [nlos][2] first and last column any source of the line of sight refers to
    std::vector<int> seg_layer;          // layer (from the top) holding the source point
    std::vector<double> seg_frac;        // fraction of the layer between its top and the source point
    std::vector<double> seg_cos;         // cosine of the propagation direction against the local vertical
    std::vector<double> seg_azimuth;     // relative azimuth
    std::vector<int> seg_col;            // [nseg][2]
    std::vector<double> seg_colw;        // [nseg] weight of the upper column
    std::vector<int> od_start;           // [nseg + 1] optical-depth stencil, level index counted from the top
    std::vector<int> od_level;
    std::vector<double> od_weight;

    int nseg() const { return (int)seg_layer.size(); }
    double chap(int col, int boundary, int layer) const { return chapman[(size_t(col) * nlyr + boundary) * nlyr + layer]; }
    // columns bracketing a cos(SZA) and the weight of the upper one (cpp_twostream_source.cpp:1128-1146)
    void column_weights(double cosine, int& lower, int& upper, double& upper_weight) const {
        lower = upper = 0;
        upper_weight = 0.0;
        if (ncol == 1 || cosine <= csz.front()) return;
        if (cosine >= csz.back()) { lower = upper = ncol - 1; return; }
        upper = int(std::lower_bound(csz.begin(), csz.end(), cosine) - csz.begin());
        lower = upper - 1;
        upper_weight = (cosine - csz[lower]) / (csz[upper] - csz[lower]);
    }
};

inline void build_ts_geometry(const skgeo::Geometry1D& geo, const std::vector<skgeo::RaySpec>& specs, const Settings& cfg, TsGeometry& g) {
    g = TsGeometry();
    g.cfg = cfg;
    // layers, Chapman factors and the cos(SZA) grid are those of the two-stream discrete-ordinates layer array
    skgeo::DiscoGeometry dg;
    const bool sph = geo.coords.geotype == skgeo::spherical;
    skgeo::build_disco_geometry(geo, specs, 2, (cfg.solar && sph) ? std::max(cfg.num_sza, 1) : 1, dg);
    g.nlyr = dg.nlyr;
    g.ngrid = dg.ngrid;
    g.nlos = (int)specs.size();
    g.spherical = sph;
    g.thickness.resize(g.nlyr);
    for (int p = 0; p < g.nlyr; ++p) g.thickness[p] = dg.ceil_h[p] - dg.floor_h[p];
    g.csz = dg.csz;
    if (g.csz.size() > 1 && !(g.csz.back() - g.csz.front() > 1e-14)) g.csz = {geo.coords.cos_sza_at_reference()};
    g.ncol = (int)g.csz.size();
    if (g.ncol == (int)dg.csz.size()) {
        g.chapman = dg.chapman;
    } else {
        // degenerate cos(SZA) range: a single column at the reference sun (cpp_twostream_source.cpp:3371-3386)
        skgeo::DiscoGeometry one;
        skgeo::build_disco_geometry(geo, specs, 2, 1, one);
        g.chapman = one.chapman;
    }

    // derived forms of the Chapman table
    g.chapman_t.assign(g.chapman.size(), 0.0);
    g.chap_uniform.assign(g.ncol, 0.0);
    for (int c = 0; c < g.ncol; ++c) {
        const double f = g.chap(c, 0, 0);
        bool uniform = f > 0.0;
        for (int bd = 0; bd < g.nlyr; ++bd)
            for (int l = 0; l < g.nlyr; ++l) {
                g.chapman_t[(size_t(c) * g.nlyr + l) * g.nlyr + bd] = g.chap(c, bd, l);
                uniform = uniform && g.chap(c, bd, l) == (l <= bd ? f : 0.0);
            }
        if (uniform) g.chap_uniform[c] = f;
    }
    if (!sph) {
        g.view_cos.resize(g.nlos);
        g.view_azimuth.resize(g.nlos);
        for (int i = 0; i < g.nlos; ++i) {
            if (!(dg.pp_los[i].coszenith > 0.0)) throw std::invalid_argument("two-stream source: only upwelling plane-parallel radiances are supported");
            g.view_cos[i] = dg.pp_los[i].coszenith;
            g.view_azimuth[i] = dg.pp_los[i].azimuth;
        }
        return;
    }

    const std::vector<double>& alt = geo.alt.v;
    const int nlevel = g.ngrid;
    const double R = geo.coords.earth_radius;
    skgeo::SphericalShellRayTracer tracer(geo);
    g.ray_start.assign(1, 0);
    g.od_start.assign(1, 0);
    for (int i = 0; i < g.nlos; ++i) {
        skgeo::TracedRay tr;
        tracer.trace(skgeo::construct_ray(specs[i], geo.coords), tr);
        g.ground_hit.push_back(tr.ground_is_hit ? 1 : 0);
        int lo, up;
        double uw;
        g.column_weights(tr.layers.empty() ? geo.coords.cos_sza_at_reference() : tr.layers.front().cos_sza_exit, lo, up, uw);
        g.ground_col.push_back(lo);
        g.ground_col.push_back(up);
        g.ground_colw.push_back(uw);
        for (const skgeo::TracedLayer& L : tr.layers) {
            double ws = 0.5, we = 0.5;
            const double sum = L.od_quad_start + L.od_quad_end;
            if (std::isfinite(L.od_quad_start) && std::isfinite(L.od_quad_end) && sum > 0.0) {
                ws = L.od_quad_start / sum;
                we = L.od_quad_end / sum;
            }
            const skgeo::Vec3 position = L.entrance.position * ws + L.exit.position * we;
            const double altitude = position.norm() - R;
            int bottom = int(std::upper_bound(alt.begin(), alt.end(), altitude) - alt.begin()) - 1;
            bottom = std::min(std::max(bottom, 0), nlevel - 2);
            const double frac = std::clamp((alt[bottom + 1] - altitude) / (alt[bottom + 1] - alt[bottom]), 0.0, 1.0);
            const skgeo::Vec3 up_dir = position.normalized();
            const double cosine = std::clamp(-L.average_look_away.dot(up_dir), -1.0, 1.0);
            const double s_az = ws * std::sin(L.saz_entrance) + we * std::sin(L.saz_exit);
            const double c_az = ws * std::cos(L.saz_entrance) + we * std::cos(L.saz_exit);
            g.seg_layer.push_back(g.nlyr - 1 - bottom);
            g.seg_frac.push_back(frac);
            g.seg_cos.push_back(cosine);
            g.seg_azimuth.push_back(std::atan2(s_az, c_az));
            g.column_weights(std::clamp(ws * L.cos_sza_entrance + we * L.cos_sza_exit, -1.0, 1.0), lo, up, uw);
            g.seg_col.push_back(lo);
            g.seg_col.push_back(up);
            g.seg_colw.push_back(uw);
            for (int k = 0; k < L.nw; ++k)
                if (L.w_od[k] != 0.0) {
                    if (L.widx[k] < 0 || L.widx[k] >= nlevel) throw std::runtime_error("two-stream source: optical-depth stencil outside the altitude grid");
                    g.od_level.push_back(nlevel - 1 - L.widx[k]);
                    g.od_weight.push_back(L.w_od[k]);
                }
            g.od_start.push_back((int)g.od_level.size());
        }
        g.ray_start.push_back(g.nseg());
        int clo = g.ground_col[2 * i], chi = g.ground_col[2 * i + 1];
        if (!tr.ground_is_hit) { clo = g.ncol; chi = -1; }
        for (int sgm = g.ray_start[i]; sgm < g.ray_start[i + 1]; ++sgm) {
            clo = std::min(clo, g.seg_col[2 * sgm]);
            chi = std::max(chi, g.seg_col[2 * sgm + 1]);
        }
        if (chi < clo) clo = chi = 0;
        g.view_cols.push_back(clo);
        g.view_cols.push_back(chi);
    }
}

}  // namespace skts
