// Optical-property front end on the device: the atmosphere a set of constituents describes -- total extinction,
// single scatter albedo, Legendre coefficients, emission [wavelength][location] in the reference's storage layout -- and the derivative
// mappings of the constituents' parameters are formed in HBM from what is small: profiles on the altitude grid [ngrid] and spectra on
// the wavelength grid [nwavel] (cross sections, scattering cross sections, phase moments).  Nothing of size ngrid x nwavel crosses PCIe.
//
// Formulation (not the reference's): the reference lets every constituent add its own [ngrid x nwavel] arrays to the storage and then
// normalises in place (atmosphere.py:745-757, atmosphere_storage.rs:113-153), and every derivative mapping is filled constituent by
// constituent the same way.  Here every contribution is a rank-one TERM profile[q] x spectrum[w] (a temperature-dependent cross-section
// table is one term per table temperature, its profile carrying the interpolation weight), and one kernel sums all terms in registers,
// normalises and writes each output element exactly once; a mapping is the same sum with other per-location weights, combined with
// the totals in closed form:
//     e = sum_r c_r[q] ext_r[w],   s = sum_r c_r[q] scat_r[w],   P_l = sum_r c_r[q] scat_r[w] leg_r[w][l]
//     d_extinction = e            d_ssa = e (s / e - ssa) / k        scat_factor = s / (ssa k)        d_legendre_l = P_l / s - leg_l
// which reproduces rayleigh.rs:209-243 (e = s = sigma), vmr_alt_absorber.rs:14-62 (s = 0) and number_density_scatterer.rs:343-424.
// All kernels are pure HBM streams (a few FMAs per element written); the spectra and profiles they re-read live in L2.
#pragma once
#include <hip/hip_runtime.h>

namespace skfe {

struct Term {            // device pointers
    const double* prof;  // [ngrid]  assembly: amount profile; mapping: per-location weight or null (= 1)
    const double* ext;   // [nwavel] extinction cross section
    const double* scat;  // [nwavel] scattering cross section, or null (absorber)
    const double* leg;   // [nwavel][nleg] phase moments per scattering event (first dimension of leg_coeff), or null
};

// extinction, single scatter albedo (normalize_by_extinctions: clamped to 1, 0 where nothing absorbs or scatters)
__global__ void __launch_bounds__(256) k_fe_assemble(const Term* __restrict__ terms, int nterms, int ngrid, int nw, double* __restrict__ ext,
                                                     double* __restrict__ ssa) {
    const size_t n = size_t(ngrid) * nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / ngrid), q = int(i - size_t(w) * ngrid);
        double k = 0.0, s = 0.0;
        for (int r = 0; r < nterms; ++r) {
            const Term t = terms[r];
            const double p = t.prof[q];
            if (p == 0.0) continue;   // a table node this level does not touch, a level a one-level term does not cover
            k += t.ext[w] * p;
            if (t.scat) s += t.scat[w] * p;
        }
        ext[i] = k;
        ssa[i] = k > 0.0 ? fmin(s / k, 1.0) : 0.0;
    }
}

// Legendre coefficients [wavelength][location][l]: scattering-extinction weighted mean of the terms' moments
__global__ void __launch_bounds__(256) k_fe_legendre(const Term* __restrict__ terms, int nterms, int ngrid, int nw, int nleg,
                                                     double* __restrict__ leg) {
    const size_t n = size_t(ngrid) * nw * nleg;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t cell = i / nleg;
        const int l = int(i - cell * nleg), w = int(cell / ngrid), q = int(cell - size_t(w) * ngrid);
        double s = 0.0, P = 0.0;
        for (int r = 0; r < nterms; ++r) {
            const Term t = terms[r];
            if (!t.scat) continue;
            const double p = t.prof[q];
            if (p == 0.0) continue;
            const double v = t.scat[w] * p;
            s += v;
            if (t.leg) P += v * t.leg[size_t(w) * nleg + l];
        }
        leg[i] = P / (s == 0.0 ? 1.0 : s);
    }
}

// emission.rs:9-30: Planck radiance W / (m^2 sr nm) and its temperature derivative
__device__ inline double planck_nm(double temperature, double wavelength_nm, bool derivative) {
    const double h = 6.62607015e-34, c = 299792458.0, kb = 1.380649e-23;
    const double wl = wavelength_nm * 1.0e-9;
    const double ex = exp(h * c / (wl * kb * temperature));
    const double wl2 = wl * wl, wl5 = wl2 * wl2 * wl;
    if (!derivative) return (2.0 * h * c * c / wl5) / (ex - 1.0) * 1.0e-9;
    return (2.0 * h * h * c * c * c * ex) / (wl5 * wl * kb * temperature * temperature * (ex - 1.0) * (ex - 1.0)) * 1.0e-9;
}
__global__ void __launch_bounds__(256) k_fe_planck(const double* __restrict__ temperature, const double* __restrict__ wavelength_nm, int ngrid,
                                                   int nw, int derivative, int accumulate, double* __restrict__ out) {
    const size_t n = size_t(ngrid) * nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / ngrid), q = int(i - size_t(w) * ngrid);
        const double v = planck_nm(temperature[q], wavelength_nm[w], derivative != 0);
        out[i] = accumulate ? out[i] + v : v;
    }
}

// a derivative mapping: d_extinction, d_ssa and (scatterers) scat_factor from the mapping's terms and the assembled totals
__global__ void __launch_bounds__(256) k_fe_map(const Term* __restrict__ terms, int nterms, int ngrid, int nw, const double* __restrict__ ext,
                                                const double* __restrict__ ssa, double* __restrict__ d_ext, double* __restrict__ d_ssa,
                                                double* __restrict__ scat_factor) {
    const size_t n = size_t(ngrid) * nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / ngrid), q = int(i - size_t(w) * ngrid);
        double e = 0.0, s = 0.0;
        for (int r = 0; r < nterms; ++r) {
            const Term t = terms[r];
            const double c = t.prof ? t.prof[q] : 1.0;
            e += t.ext[w] * c;
            if (t.scat) s += t.scat[w] * c;
        }
        const double k = ext[i], om = ssa[i];
        const double species_ssa = e != 0.0 ? s / e : 1.0;   // number_density_scatterer.rs:316-324
        d_ext[i] = e;
        d_ssa[i] = k == 0.0 ? 0.0 : e * (species_ssa - om) / k;
        if (scat_factor) scat_factor[i] = om * k != 0.0 ? s / (om * k) : 0.0;
    }
}
__global__ void __launch_bounds__(256) k_fe_map_legendre(const Term* __restrict__ terms, int nterms, int ngrid, int nw, int nleg,
                                                         const double* __restrict__ leg, double* __restrict__ d_leg) {
    const size_t n = size_t(ngrid) * nw * nleg;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t cell = i / nleg;
        const int l = int(i - cell * nleg), w = int(cell / ngrid), q = int(cell - size_t(w) * ngrid);
        double s = 0.0, P = 0.0;
        for (int r = 0; r < nterms; ++r) {
            const Term t = terms[r];
            if (!t.scat) continue;
            const double v = t.scat[w] * (t.prof ? t.prof[q] : 1.0);
            s += v;
            if (t.leg) P += v * t.leg[size_t(w) * nleg + l];
        }
        d_leg[i] = (s != 0.0 ? P / s : 0.0) - leg[i];
    }
}

// All pending mappings in one launch each: the totals (extinction, ssa; leg_coeff) are read ONCE per element and every mapping's arrays are
// written from that read -- with M scatterer mappings the d_legendre stage moves (1 + M) instead of 2 M arrays.
struct MapDesc {             // device
    int first, count;        // the mapping's terms in the table of the launch
    double *d_ext, *d_ssa;   // [nwavel][ngrid]
    double *scat_factor;     // [nwavel][ngrid] or null (plain mapping)
    double *d_leg;           // [nwavel][ngrid][nleg] or null
};
__global__ void __launch_bounds__(256) k_fe_map_all(const Term* __restrict__ terms, const MapDesc* __restrict__ maps, int nmaps, int ngrid, int nw,
                                                    const double* __restrict__ ext, const double* __restrict__ ssa) {
    const size_t n = size_t(ngrid) * nw;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const int w = int(i / ngrid), q = int(i - size_t(w) * ngrid);
        const double k = ext[i], om = ssa[i];
        for (int m = 0; m < nmaps; ++m) {
            const MapDesc d = maps[m];
            double e = 0.0, s = 0.0;
            for (int r = d.first; r < d.first + d.count; ++r) {
                const Term t = terms[r];
                const double c = t.prof ? t.prof[q] : 1.0;
                e += t.ext[w] * c;
                if (t.scat) s += t.scat[w] * c;
            }
            const double species_ssa = e != 0.0 ? s / e : 1.0;
            // (written once, read by the engine's mapping kernel much later: streamed past the caches)
            __builtin_nontemporal_store(e, &d.d_ext[i]);
            __builtin_nontemporal_store(k == 0.0 ? 0.0 : e * (species_ssa - om) / k, &d.d_ssa[i]);
            if (d.scat_factor) __builtin_nontemporal_store(om * k != 0.0 ? s / (om * k) : 0.0, &d.scat_factor[i]);
        }
    }
}
__global__ void __launch_bounds__(256) k_fe_map_legendre_all(const Term* __restrict__ terms, const MapDesc* __restrict__ maps, int nmaps, int ngrid,
                                                             int nw, int nleg, const double* __restrict__ leg) {
    const size_t n = size_t(ngrid) * nw * nleg;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const size_t cell = i / nleg;
        const int l = int(i - cell * nleg), w = int(cell / ngrid), q = int(cell - size_t(w) * ngrid);
        const double beta = leg[i];
        for (int m = 0; m < nmaps; ++m) {
            const MapDesc d = maps[m];
            if (!d.d_leg) continue;
            double s = 0.0, P = 0.0;
            for (int r = d.first; r < d.first + d.count; ++r) {
                const Term t = terms[r];
                if (!t.scat) continue;
                const double v = t.scat[w] * (t.prof ? t.prof[q] : 1.0);
                s += v;
                if (t.leg) P += v * t.leg[size_t(w) * nleg + l];
            }
            __builtin_nontemporal_store((s != 0.0 ? P / s : 0.0) - beta, &d.d_leg[i]);   // written once, read by another kernel much later
        }
    }
}

// validate_input_atmosphere for storage that lives in HBM: first offending element (index, kind 1 non-finite / 2 below / 3 above) or -1
__global__ void __launch_bounds__(256) k_fe_validate(const double* __restrict__ a, size_t n, double lo, double hi, unsigned long long* first) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const double v = a[i];
        if (isfinite(v) && v >= lo && v <= hi) continue;
        const unsigned long long kind = !isfinite(v) ? 1ull : (v < lo ? 2ull : 3ull);
        atomicMin(first, (static_cast<unsigned long long>(i) << 2) | kind);
    }
}


// delta-M scaling of a device-resident atmosphere (Atmosphere<NSTOKES>::apply_delta_m_scaling, cpp/lib/atmosphere/atmosphere.cpp:73-200; the
// host form is sk_atmosphere_apply_delta_m_scaling).  The mappings are scaled FIRST, from the still unscaled storage:
//   f = beta_order / (2 order + 1),  k* = (1 - w f) k,  w* = (1 - f) w / (1 - w f),  beta_l* = beta_l / (1 - f)
//   d_k <- d_k (1 - w f) - k f d_w,   d_w <- d_w (1 - f (1 - w*)) / (1 - w f)
//   scattering groups: d_f = d_beta_order / (2 order + 1), d_beta_l <- (d_beta_l + beta_l* d_f) / (1 - f), and the d_f terms of d_k / d_w
__global__ void __launch_bounds__(256) k_fe_delta_m_mapping(const double* __restrict__ ext, const double* __restrict__ ssa, const double* __restrict__ leg,
                                                            int nleg, int slot, double norm, size_t n, double* __restrict__ d_ext,
                                                            double* __restrict__ d_ssa, double* __restrict__ scat_factor, double* __restrict__ d_leg,
                                                            double* __restrict__ d_f_out) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const double f = leg[i * nleg + slot] / norm, w_u = ssa[i], k_u = ext[i];
        const double w_s = (1 - f) / (1 - w_u * f) * w_u;
        double de = d_ext[i] * (1 - w_u * f) - k_u * f * d_ssa[i];
        double ds = d_ssa[i] * ((1 - f * (1 - w_s)) / (1 - w_u * f));
        if (d_leg) {
            const double d_f = d_leg[i * nleg + slot] / norm;
            d_f_out[i] = d_f;
            for (int l = 0; l < nleg; ++l) d_leg[i * nleg + l] = (d_leg[i * nleg + l] + leg[i * nleg + l] / (1 - f) * d_f) / (1 - f);
            const double dfs = d_f * scat_factor[i];
            de -= w_u * k_u * dfs;
            ds += dfs * w_u / (1 - w_u * f) * (w_s - 1);
        }
        d_ext[i] = de;
        d_ssa[i] = ds;
    }
}
__global__ void __launch_bounds__(256) k_fe_delta_m_storage(double* __restrict__ ext, double* __restrict__ ssa, double* __restrict__ leg, int nleg,
                                                            int slot, double norm, size_t n, double* __restrict__ delta_f) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const double f = leg[i * nleg + slot] / norm, w_u = ssa[i], k_u = ext[i];
        delta_f[i] = f;
        ext[i] = k_u * (1 - w_u * f);
        ssa[i] = (1 - f) / (1 - w_u * f) * w_u;
        for (int l = 0; l < nleg; ++l) leg[i * nleg + l] /= (1 - f);
    }
}

inline unsigned stream_grid(size_t n) { return unsigned(std::min<size_t>((n + 255) / 256, 256 * 32)); }

}  // namespace skfe
