// DISCO hot path for MI355X -- per-task device math (scalar NSTOKES=1).
//
// Every function here is `SKD_HD` (host + device): the HIP kernels in disco_kernels.hip call them
// with one GPU thread per task, and tests/emulate_kernels.cpp compiles the very same source with
// g++ to run the kernel bodies on the CPU-only authoring container (test infrastructure only; the
// product library is always the HIP build and fails loudly without a GPU).
//
// The formulation is NOT the reference's (and not the oracle's):
//   * homogeneous solution: the streams x streams problem S-S+ x = k^2 x is symmetrized
//     (B+- = T S+- T^-1, T = diag sqrt(w mu)), reduced with a Cholesky factor of B+ and solved by
//     cyclic Jacobi -- no Hessenberg/QR, no pivoting, fixed data flow;
//     reference: RTESolver::solveHomogeneous cpp/lib/sktran_disco/sktran_do_rte.cpp:382-553
//   * eigen-pair derivatives by bi-orthogonal perturbation theory (left vectors Y = W M S+ X come for
//     free) instead of N bordered LU solves; reference: linearizeHomogeneous sktran_do_rte.cpp:197-298
//   * boundary value problem as a block-tridiagonal system with 2N x 2N blocks (one block row per
//     layer: downwelling continuity at the layer top, upwelling continuity at its bottom) instead
//     of LAPACK band storage; reference: solveBVP sktran_do_rte.cpp:1620-1790
//   * weighting functions by an ADJOINT sweep: one transposed block solve per line of sight and a
//     reverse-mode chain through beam transmission / average secant, instead of the reference's
//     dense forward sweep with one right-hand side per atmospheric parameter
//     (sktran_do_rte.cpp:1730-1789, do_source_diffuse_storage.cpp:957-1094).
// Results are mathematically identical; parity is asserted on radiances and weighting functions.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SKD_HD __host__ __device__ inline
#else
#define SKD_HD inline
#endif

namespace skd {

constexpr double kPi = 3.14159265358979323846;
constexpr double kGreensEps = 1e-4;  // sktran_do_types.h:11
constexpr double kSSADither = 1e-9;  // sktran_do_specs.h:104

// error bits written to the per-wavelength status word
enum : int { ERR_CHOLESKY = 1, ERR_EIGEN = 2, ERR_SINGULAR_BVP = 4, ERR_NAN = 8 };

// ------------------------------------------------------------------------------------------------
// Geometry tables (wavelength independent, uploaded once per engine)
// ------------------------------------------------------------------------------------------------
struct Geo {
    int nstr, N, nlyr, ngrid, nlos, nsza, nangle, npoint, nground, nslot, nq_total;
    const double* mu;       // [N]
    const double* wt;       // [N]
    const double* lpmu;     // [m][i<N][l<nstr]
    const double* lpcsz;    // [s][m][l]
    const double* lpang;    // [m][a][l]
    const double* dh;       // [p]
    const double* lfrac;    // [p]  relative position of the table altitude inside the layer
    const int* wq_idx;      // [p][2] optical interpolator (index -1 = unused)
    const double* wq_w;     // [p][2]
    const double* chapman;  // [s][p][r]
    const double* csz;      // [s]
    const int* pt_start;    // [s*nlyr+p .. +1] needed table angles of (sza, layer)
    const int* pt_angle;
    const int* pt_slot;
    const int* gpt_start;   // [s .. +1] needed ground slots
    const int* gpt_slot;
    const int* ray_qstart;  // [los .. +1] ray layers (far end -> observer)
    const int* ray_ground;  // [los]
    const double* ray_gsaa; // [los]
    const int* rl_od_start; // [q .. +1]
    const int* od_idx;
    const double* od_w;
    const int* rl_ssa_start;
    const int* ssa_idx;
    const double* ssa_w;
    const double* rl_azi;   // [q]
    const double* rl_cosm;  // [m][q]  cos(m * azi_q)
    const double* rl_cosm_t;  // [q][m]  the same, order index fastest (LOS integration)
    const int* rl_active;   // [q] 0 when the shell is shorter than MINIMUM_SHELL_SIZE_M
    const int* rl_tab_start;  // [q .. +1] table entries of a ray layer
    const int* tab_slot;
    const double* tab_w;
    const int* lp_start;    // [(los*nsza+s)*nlyr+p .. +1] table entries of a LOS grouped by (sza, layer)
    const int* lp_q;
    const int* lp_angle;
    const int* lp_slot;
    const double* lp_w;
    const double* lp_pc;    // [m][entry][q<4][kk<4] = cos(m azimuth of the entry's ray layer) P_{4kk+q}^m(entry's table angle), 0 for l < m  (nstr = 16 only)
    int n_lp;               // number of (LOS, sza, layer) table entries
    const int* g_start;     // [los .. +1] ground entries
    const int* g_slot;
    const int* g_sza;
    const double* g_w;
    // plane-parallel post-processing (DOSourcePlaneParallelPostProcessing, do_source_planeparallel.cpp:19-218)
    int pp_mode;            // 1: analytic layer integration per line of sight instead of the spherical diffuse table
    int include_ss;         // the solar single-scatter term is part of the DO source (single_scatter_source = DO)
    const double* pp_mu;    // [los] cosine of the viewing zenith angle (upwelling, > 0)
    const double* pp_azi;   // [los] relative azimuth
    const int* pp_first;    // [los] topmost layer the line of sight crosses
    const double* pp_lf;    // [los] fraction of that layer below the observer (1 when the observer is above it)
    const double* lplos;    // [los][m][l] Legendre functions at the viewing zenith
};

// ------------------------------------------------------------------------------------------------
// Per-run device buffers.  Unless noted every array is [element][thread] with the thread index
// (wavelength, or wavelength*nlos+los) fastest, so that consecutive lanes touch consecutive doubles.
// ------------------------------------------------------------------------------------------------
struct Bufs {
    int nw;                 // wavelengths in this launch
    // inputs in the reference's storage layout (location fastest), cpp/include/c_api/atmosphere.h:77-83
    const double* ext;      // [w][q]
    const double* ssa;      // [w][q]
    const double* leg;      // [w][q][l<nleg]
    const double* albedo;   // [w]
    const double* irradiance;  // [w] or null
    const double* delta_f;  // [w][q] delta-M truncation fraction of the (already scaled) storage, or null
    // thermal pass (EmissionSource::discrete_ordinates, plane-parallel post-processing): emission_source [w][q], surface emission [w]
    const double* emission;
    const double* surf_em;  // surface emission [w] or null, set only while the thermal pass runs
    int thermal_pass;       // 1 while the thermal pass runs
    int nleg;
    // layer optics (K0)
    double* tau;            // [p][w]
    double* om;             // [p][w]
    double* kbar;           // [p][w]
    double* a;              // [p][l][w]
    double* T;              // [lev<=nlyr][w]    (current sza)
    double* sec;            // [p][w]
    // LOS optics (K0)
    double* od;             // [q][w]
    double* omloc;          // [q][w]
    double* srcw;           // [q][w]  omega_loc (1-exp(-od)) * attenuation(q -> observer)
    double* att_total;      // [los][w]
    double* cwbuf;          // [w][entry]  srcw of the entry's ray layer x its interpolation weight (16-stream sweeps)
    uint8_t* qlmask;        // [w][entry / 8]  bit j: entry 8 i + j still contributes at the azimuth order of the adjoint stage (kern_qlmask)
    // per azimuth order
    double* rec;            // [w][p][REC]   (record-major, see rec_at)
    double* dinv;           // [w][p][4N^2]
    double* xsol;           // [w][p][2N]
    double* eig;            // [p][N + N N][w]  eigen-solutions from kern_kj_lane<N, true> for kern_k1_lane (16 streams, spherical)
    int k1_lane;            // 1: records written by kern_k1_lane -- Rec::DG holds the coefficient vectors, kern_k2c_mfma8 forms DG itself
    double* J;              // [w][slot][m]   (see J_at; the orders of a table point are contiguous)
    uint8_t* conv_order;    // [w][slot]  first azimuth order that is skipped for this table point (see conv_at)
    int* notconv;           // [w]
    int* norders;           // [w] number of azimuth orders that are solved (stop rule)
    int* status;            // [w] error bits
    double* ops;            // [w][p][Ops::SIZE]  LOS-independent adjoint operators (K2c)
    double* qlbuf;          // [p][nstr][w*nlos+los]  Legendre-space LOS weights
    double* z;              // [p][2N][w*nlos+los]
    double* grad;           // [4*nlyr+2][w*nlos+los]   reduced gradient of the current sza
    double* dnat;           // [2*ngrid+1][w*nlos+los]  native derivatives
    double* radiance;       // [w*nlos+los]
    double* los_od;         // [w*nlos+los]
    int derivs;
    // 16 streams: a_l (16), omega and tau of a layer in ONE run per (wavelength, layer), [w][p][18] -- what the per-wavelength kernels of an
    // azimuth order read of the layer optics (from the [p][l][w] arrays every value is its own 64-byte sector); written by kern_k0_tile
    double* at;
    // Azimuth orders batched into one launch (small launches, 16 streams): gridDim.z = orders, order m0 + blockIdx.z works on its own set of
    // per-order buffers, ord_* elements behind the set of order m0 (0 when a launch serves one order: order_view then changes nothing)
    size_t ord_rec, ord_dinv, ord_xsol, ord_eig, ord_ops, ord_z, ord_qlbuf, ord_qlmask, ord_grad;
    // Scattering-group weighting functions on the 16-stream matrix-core kernels (d_legendre / scat_factor mappings,
    // sktran_do_layerarray.cpp:496-513,759-860): per group the direction of the layer's Legendre coefficients (kern_k0g), the derivative of
    // the layer solution along it (kern_k1g_cols8 -> RecG), one more DI / HD / GD row per group in kern_k2c_mfma8 (opsg, hg) and one more
    // gradient row per group and layer in kern_k2b_mfma8 (grad rows 4 nlyr + 2 + group nlyr + p), chained to the grid by kern_k4g
    int ngroups;
    const double* dleg;     // [group][wavelength of the spectrum][q][l < nleg] d_legendre of the group's mapping
    const double* dfg;      // [group][wavelength of the spectrum][q] d_f of the group's mapping (delta-M scaled storage) or null
    int dleg_nw, dleg_w0;   // wavelengths of the spectrum, offset of this block in it
    double* ga;             // [group][p][l][w] direction of a_l
    double* recg;           // [w][p][group][RecG::SIZE]
    double* opsg;           // [w][p][group][OpsG::SIZE]
    double* hg;             // [w][p][tile][64] group rows of the HD part (C/D image rows 8 .. 11 of the forward sweep's second product)
};
#if defined(__HIPCC__)
// the buffers of azimuth order m0 + mo (mo = blockIdx.z of a batched launch)
__device__ inline Bufs order_view(Bufs b, unsigned mo) {
    b.rec += mo * b.ord_rec; b.dinv += mo * b.ord_dinv; b.xsol += mo * b.ord_xsol; b.eig += mo * b.ord_eig; b.ops += mo * b.ord_ops;
    b.z += mo * b.ord_z; b.qlbuf += mo * b.ord_qlbuf; b.qlmask += mo * b.ord_qlmask; b.grad += mo * b.ord_grad;
    return b;
}
#endif

// derivative of a layer's solution along a scattering group's direction (16 streams, lane-path records): dW+[i][j], dW-[i][j] stream-major
// with rows 9 doubles apart, then dk_j, dA+_j, dA-_j, d(A- C-)_j, d(A+ C+)_j as runs of eight, then the direction itself (16).  The eight lanes
// of a problem in kern_k1g_cols8 (lane = solution j) write 64-byte runs; in kern_k2c_mfma8's LDS image (this layout unchanged) the lanes of an
// edge (stream i) are 18 dwords apart and the lanes of a Legendre sum (i, j) consecutive: no bank conflicts either way
template <int N> struct RecG {
    static constexpr int RS = N + 1, DWP = 0, DWM = N * RS, DK = 2 * N * RS, DAP = DK + N, DAM = DAP + N, CGM = DAM + N, CGP = CGM + N;
    static constexpr int DA = CGP + N, SIZE = DA + 2 * N;
    SKD_HD static constexpr int dwp(int i, int j) { return DWP + i * RS + j; }
    SKD_HD static constexpr int dwm(int i, int j) { return DWM + i * RS + j; }
    SKD_HD static constexpr int at(int what, int j) { return what + j; }
};
// group rows of the adjoint operators: DI[edge][i] and GD
template <int N> struct OpsG {
    static constexpr int DI = 0, GD = 4 * N, SIZE = 4 * N + 8;
};
template <int N> SKD_HD double* recg_at(const Bufs& b, int nlyr, int p, int w, int gi) { return b.recg + ((size_t(w) * nlyr + p) * b.ngroups + gi) * RecG<N>::SIZE; }
template <int N> SKD_HD double* opsg_at(const Bufs& b, int nlyr, int p, int w, int gi) { return b.opsg + ((size_t(w) * nlyr + p) * b.ngroups + gi) * OpsG<N>::SIZE; }

template <int N> struct Rec {
    static constexpr int K = 0, DK = N, WP = 2 * N, WM = WP + N * N, DWP = WM + N * N, DWM = DWP + N * N,
                         AP = DWM + N * N, AM = AP + N, DAP = AM + N, DAM = DAP + N, G = DAM + N, DG = G + 4 * N,
                         PPM = DG + 16 * N, E = PPM + 14 * N, SIZE = E + N;
    // PPM[j][14]: post-processing multipliers of solution j (struct PPMult) and E[j] = exp(-k_j tau), written by K3 so
    // that the per-LOS adjoint lanes of K2b never evaluate an exponential
    // G order: 0 Gplus_top, 1 Gminus_top, 2 Gplus_bottom, 3 Gminus_bottom;  DG[theta][which][i], theta: 0 tau 1 omega 2 T 3 sec
    // element accessors (the kernels that serve both record layouts go through these)
    static constexpr bool COLUMN_BLOCKS = false;
    SKD_HD static constexpr int wp(int i, int j) { return WP + i + N * j; }
    SKD_HD static constexpr int wm(int i, int j) { return WM + i + N * j; }
    SKD_HD static constexpr int dwp(int i, int j) { return DWP + i + N * j; }
    SKD_HD static constexpr int dwm(int i, int j) { return DWM + i + N * j; }
    SKD_HD static constexpr int k(int j) { return K + j; }
    SKD_HD static constexpr int dk(int j) { return DK + j; }
    SKD_HD static constexpr int e(int j) { return E + j; }
    SKD_HD static constexpr int ap(int j) { return AP + j; }
    SKD_HD static constexpr int am(int j) { return AM + j; }
    SKD_HD static constexpr int dap(int j) { return DAP + j; }
    SKD_HD static constexpr int dam(int j) { return DAM + j; }
    SKD_HD static constexpr int g(int x) { return G + x; }            // x = which * N + i
    SKD_HD static constexpr int coef(int tk, int j) { return DG + tk * N + j; }   // kern_k1_lane's coefficient vectors, tk = type * 5 + k
    SKD_HD static constexpr int ppm(int j) { return PPM + 14 * j; }
};
// Record of the lane-per-problem K1 (kern_k1_lane, 16 streams): everything K1 knows about SOLUTION j sits in one block of CB doubles
// -- W+(:, j), W-(:, j), dW+(:, j), dW-(:, j), the ten coefficients of the particular solution, k, dk, A+-, dA+- -- because a
// lane produces its record one solution column at a time and writes a column's 48 doubles as three full 128-byte lines (the standard
// layout scatters a column over the record in 64-byte runs: 2.1 x the bytes in HBM write traffic, PMC).  G, exp(-k_j tau) and the post-processing
// multipliers K3 adds follow the blocks; the derivative rows DG are formed where they are needed (kern_k2c_mfma8) and have no place here.
// PAD: unused doubles behind every block (the LDS image of kern_k2c_mfma8 uses 1: a block stride of 49 doubles spreads the eight solutions'
// entries over the banks, 48 puts them on two)
template <int N, int PAD = 0> struct RecC {
    static constexpr int CB = 4 * N + 16 + PAD, WPo = 0, WMo = N, DWPo = 2 * N, DWMo = 3 * N, COEFo = 4 * N, Ko = COEFo + 10, DKo = Ko + 1,
                         APo = Ko + 2, AMo = Ko + 3, DAPo = Ko + 4, DAMo = Ko + 5, G = N * CB, E = G + 4 * N,
                         TAIL0 = N * (4 * N + 16) + 5 * N,                       // end of E in the unpadded record
                         PPM = E + N + ((TAIL0 + 15) / 16 * 16 - TAIL0),         // 128-byte aligned there; the padded image only shifts by N PAD
                         SIZE = PPM + (14 * N + 15) / 16 * 16;
    // (exp(-k_j tau) of the eight solutions in ONE run behind G: the boundary-value kernel reads W+-(:, j) -- line 0 of block j --, E and G only)
    static_assert(Ko + 6 + PAD == CB, "column block");
    static constexpr bool COLUMN_BLOCKS = true;
    SKD_HD static constexpr int wp(int i, int j) { return j * CB + WPo + i; }
    SKD_HD static constexpr int wm(int i, int j) { return j * CB + WMo + i; }
    SKD_HD static constexpr int dwp(int i, int j) { return j * CB + DWPo + i; }
    SKD_HD static constexpr int dwm(int i, int j) { return j * CB + DWMo + i; }
    SKD_HD static constexpr int k(int j) { return j * CB + Ko; }
    SKD_HD static constexpr int dk(int j) { return j * CB + DKo; }
    SKD_HD static constexpr int e(int j) { return E + j; }
    SKD_HD static constexpr int ap(int j) { return j * CB + APo; }
    SKD_HD static constexpr int am(int j) { return j * CB + AMo; }
    SKD_HD static constexpr int dap(int j) { return j * CB + DAPo; }
    SKD_HD static constexpr int dam(int j) { return j * CB + DAMo; }
    SKD_HD static constexpr int g(int x) { return G + x; }
    SKD_HD static constexpr int coef(int tk, int j) { return j * CB + COEFo + tk; }
    SKD_HD static constexpr int ppm(int j) { return PPM + 14 * j; }
};
enum { TH_TAU = 0, TH_OM = 1, TH_T = 2, TH_SEC = 3 };

// Layer records, block inverses and BVP solutions are stored record-major, [wavelength][layer][element], so that a
// workgroup that owns one wavelength streams whole records with unit stride.
struct RecView {
    const double* p;
    SKD_HD double operator()(int e) const { return p[e]; }
};
SKD_HD double& J_at(const Bufs& b, const Geo& g, int m, int slot, int w) { return b.J[(size_t(w) * g.nslot + slot) * g.nstr + m]; }
SKD_HD uint8_t& conv_at(const Bufs& b, const Geo& g, int slot, int w) { return b.conv_order[size_t(w) * g.nslot + slot]; }
template <int N> SKD_HD double* rec_at(const Bufs& b, int nlyr, int p, int w) { return b.rec + (size_t(w) * nlyr + p) * Rec<N>::SIZE; }
template <class R> SKD_HD double* rec_at_r(const Bufs& b, int nlyr, int p, int w) { return b.rec + (size_t(w) * nlyr + p) * R::SIZE; }
template <int N> SKD_HD double* dinv_at(const Bufs& b, int nlyr, int p, int w) { return b.dinv + (size_t(w) * nlyr + p) * (4 * N * N); }
template <int N> SKD_HD double* xsol_at(const Bufs& b, int nlyr, int p, int w) { return b.xsol + (size_t(w) * nlyr + p) * (2 * N); }

// =================================================================================================
// K0: layer optics + beam attenuation + LOS optics        (one task per wavelength)
//   reference: OpticalLayerArray ctor / configureTransmission, sktran_do_layerarray.cpp:332-477,890-979
// =================================================================================================
SKD_HD void k0_layer_optics(const Geo& g, const Bufs& b, int w, bool with_los_shells = true) {
    const int nw = b.nw, nlyr = g.nlyr, nstr = g.nstr, ngrid = g.ngrid;
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    const double* leg = b.leg + size_t(w) * ngrid * b.nleg;
    double ceiling_depth = 0, floor_depth = 0;
    for (int p = 0; p < nlyr; ++p) {
        double od = 0, s = 0;
        for (int l = 0; l < nstr; ++l) b.a[(size_t(p) * nstr + l) * nw + w] = 0;
        for (int t = 0; t < 2; ++t) {
            int q = g.wq_idx[2 * p + t];
            if (q < 0) continue;
            double wt = g.wq_w[2 * p + t];
            double kext = ext[q], kscat = ssa[q] * kext;
            od += kext * wt;
            s += kscat * wt;
            // delta-M: the multiple-scatter expansion drops the truncated forward peak (sktran_do_layerarray.cpp:395-405)
            const double fq = b.delta_f ? b.delta_f[size_t(w) * ngrid + q] : 0.0, fsub = fq / (1 - fq);
            for (int l = 0; l < nstr; ++l)
                b.a[(size_t(p) * nstr + l) * nw + w] += wt * kscat * (((l < b.nleg) ? leg[size_t(q) * b.nleg + l] : 0.0) - (2 * l + 1) * fsub);
        }
        if (s > 0) {
            for (int l = 0; l < nstr; ++l) b.a[(size_t(p) * nstr + l) * nw + w] /= s;
        } else {
            b.a[(size_t(p) * nstr) * nw + w] = 0;
        }
        s /= od;
        od *= g.dh[p];
        floor_depth += od;
        double te = od / g.dh[p];
        double se = fmax(te * s, te * kSSADither);
        double m_ssa = se / te;
        if (1 - m_ssa < kSSADither) m_ssa = 1 - kSSADither;
        b.kbar[size_t(p) * nw + w] = te;
        b.om[size_t(p) * nw + w] = m_ssa;
        b.tau[size_t(p) * nw + w] = floor_depth - ceiling_depth;
        ceiling_depth = floor_depth;
    }
    // LOS shells: optical depth, local ssa, source weights  (the HIP driver runs this part as kern_k0_shells / kern_k0_los)
    if (!with_los_shells) return;
    for (int l = 0; l < g.nlos; ++l) {
        const int q0 = g.ray_qstart[l], q1 = g.ray_qstart[l + 1];
        double total = 0;
        for (int q = q0; q < q1; ++q) {
            double od = 0;
            for (int t = g.rl_od_start[q]; t < g.rl_od_start[q + 1]; ++t) od += g.od_w[t] * ext[g.od_idx[t]];
            double om = 0;
            for (int t = g.rl_ssa_start[q]; t < g.rl_ssa_start[q + 1]; ++t) om += ssa[g.ssa_idx[t]] * g.ssa_w[t];
            b.od[size_t(q) * nw + w] = od;
            b.omloc[size_t(q) * nw + w] = om;
            total += od;
        }
        // attenuation from the near edge of shell q to the observer: product over shells closer to the observer
        double att = 1.0;
        for (int q = q1 - 1; q >= q0; --q) {
            const double e = exp(-b.od[size_t(q) * nw + w]);
            b.srcw[size_t(q) * nw + w] = g.rl_active[q] ? b.omloc[size_t(q) * nw + w] * (1 - e) * att : 0.0;
            att *= e;
        }
        b.att_total[size_t(l) * nw + w] = att;
        b.los_od[size_t(w) * g.nlos + l] = total;
    }
}

// beam transmission at layer ceilings (T[lev]) and average secants for solar zenith index s
SKD_HD void k0_transmission(const Geo& g, const Bufs& b, int s, int w) {
    const int nw = b.nw, nlyr = g.nlyr;
    const double F0 = b.irradiance ? b.irradiance[w] : 1.0;
    double prev = 0;
    b.T[w] = F0;
    for (int p = 0; p < nlyr; ++p) {
        const double* ch = g.chapman + (size_t(s) * nlyr + p) * nlyr;
        if (ch[0] < 0) {
            b.T[size_t(p + 1) * nw + w] = 0.0;
            b.sec[size_t(p) * nw + w] = 1e10;
            prev = -1.0;
            continue;
        }
        double sig = 0;
        for (int r = 0; r < nlyr; ++r) sig += ch[r] * b.tau[size_t(r) * nw + w];
        b.sec[size_t(p) * nw + w] = (sig - prev) / b.tau[size_t(p) * nw + w];
        b.T[size_t(p + 1) * nw + w] = exp(-sig) * F0;
        prev = sig;
    }
}

// Thermal source of the discrete-ordinates equations, S(x) = (1 - omega) b0 exp(-b1 x) per layer with b0 the emission at the layer's
// ceiling and b1 = ln(B_top / B_bottom) / tau (sktran_do_layerarray.cpp:340-475).  Its Green's-function particular solution
// (RTESolver::solveParticularGreenThermal, sktran_do_rte.cpp:1335-1618), post-processing multipliers (sktran_do_opticallayer.cpp:421-431,
// 941-975) and direct term (1 - omega) E_thermal (:513-521) are the SOLAR ones with b0 in place of the beam transmission, b1 in place of
// the average secant and Q+ = Q- = w_i (1 - omega).  The solar kernels compute Q = w_i omega / (4 pi) sum_l a_l P_l(mu_i) P_l(mu_0), so
// the thermal pass reuses them unchanged at order 0 with the "sun" Legendre row (1, 0, 0, ...) of the extra solar index (a_0 = 1) and
//     T[p]   <- b0_p (1 - omega_p) 4 pi / omega_p,      sec[p] <- b1_p
// written over the beam arrays once the solar orders are done.  (task = wavelength)
SKD_HD void k0_thermal(const Geo& g, const Bufs& b, int w) {
    const int nw = b.nw, nlyr = g.nlyr;
    const double* em = b.emission + size_t(w) * g.ngrid;
    for (int p = 0; p < nlyr; ++p) {
        int lo = -1, hi = -1;
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * p + t];
            if (q < 0) continue;
            if (lo < 0 || q < lo) lo = q;
            if (q > hi) hi = q;
        }
        const double top = hi >= 0 ? em[hi] : 0.0, bot = (lo >= 0 && lo != hi) ? em[lo] : top;
        const double th = b.tau[size_t(p) * nw + w], om = b.om[size_t(p) * nw + w];
        double b1 = 0.0;
        if (th > 1e-10 && top > 1e-30 && bot > 1e-30 && fabs(top - bot) > 1e-15 * fmax(top, bot)) b1 = log(top / bot) / th;
        b.T[size_t(p) * nw + w] = top * (1 - om) * (4.0 * kPi) / om;
        b.sec[size_t(p) * nw + w] = b1;
    }
    b.T[size_t(nlyr) * nw + w] = 0.0;
}

// =================================================================================================
// small dense helpers on private arrays (row-major, leading dimension LD)
// =================================================================================================
template <int N> SKD_HD bool cholesky_lower(double* A) {  // in place, lower triangle
    for (int j = 0; j < N; ++j) {
        double d = A[j * N + j];
        for (int k = 0; k < j; ++k) d -= A[j * N + k] * A[j * N + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * N + j] = d;
        for (int i = j + 1; i < N; ++i) {
            double s = A[i * N + j];
            for (int k = 0; k < j; ++k) s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = s / d;
        }
        for (int i = 0; i < j; ++i) A[i * N + j] = 0;
    }
    return true;
}

// cyclic Jacobi for a symmetric N x N matrix (destroyed); eigenvalues -> lam, eigenvectors -> V columns
template <int N> SKD_HD bool jacobi_eigen(double* A, double* lam, double* V) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dia = 0;
        for (int i = 0; i < N; ++i) {
            dia += A[i * N + i] * A[i * N + i];
            for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j];
        }
        if (off <= 1e-30 * dia || off == 0) {  // off-diagonal norm at the rounding level of the rotations
            for (int i = 0; i < N; ++i) lam[i] = A[i * N + i];
            return true;
        }
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                double apq = A[p * N + q];
                if (apq == 0.0) continue;
                double app = A[p * N + p], aqq = A[q * N + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                A[p * N + p] = app - t * apq;
                A[q * N + q] = aqq + t * apq;
                A[p * N + q] = 0;
                A[q * N + p] = 0;
                for (int r = 0; r < N; ++r) {
                    if (r != p && r != q) {
                        double arp = A[r * N + p], arq = A[r * N + q];
                        double np_ = c * arp - s * arq, nq_ = s * arp + c * arq;
                        A[r * N + p] = np_;
                        A[p * N + r] = np_;
                        A[r * N + q] = nq_;
                        A[q * N + r] = nq_;
                    }
                    double vrp = V[r * N + p], vrq = V[r * N + q];
                    V[r * N + p] = c * vrp - s * vrq;
                    V[r * N + q] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < N; ++i) lam[i] = A[i * N + i];
    return false;
}

// Gauss-Jordan inverse with partial pivoting of an M x M matrix (row-major); returns false if singular
template <int M> SKD_HD bool invert(double* A, double* Ainv) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) Ainv[i * M + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < M; ++c) {
        int piv = c;
        double best = fabs(A[c * M + c]);
        for (int r = c + 1; r < M; ++r)
            if (fabs(A[r * M + c]) > best) { best = fabs(A[r * M + c]); piv = r; }
        if (!(best > 0)) return false;
        if (piv != c)
            for (int j = 0; j < M; ++j) {
                double t = A[c * M + j]; A[c * M + j] = A[piv * M + j]; A[piv * M + j] = t;
                t = Ainv[c * M + j]; Ainv[c * M + j] = Ainv[piv * M + j]; Ainv[piv * M + j] = t;
            }
        double inv = 1.0 / A[c * M + c];
        for (int j = 0; j < M; ++j) { A[c * M + j] *= inv; Ainv[c * M + j] *= inv; }
        for (int r = 0; r < M; ++r) {
            if (r == c) continue;
            double f = A[r * M + c];
            if (f == 0.0) continue;
            for (int j = 0; j < M; ++j) { A[r * M + j] -= f * A[c * M + j]; Ainv[r * M + j] -= f * Ainv[c * M + j]; }
        }
    }
    return true;
}

// Green's function multipliers C+-(tau, sec, k; T) with their partial derivatives.
//   reference: solveParticularGreen, sktran_do_rte.cpp:1039-1237 (second-order Taylor forms when |sec -+ k| <= 1e-4)
struct Mult { double v, d_tau, d_sec, d_k; };  // dC/dT = v / T
// exp_te = exp(-th ev), exp_ts = exp(-th sc) are passed in (the second one is the same for every solution of a layer)
SKD_HD void green_multipliers_e(double T, double th, double sc, double ev, double exp_te, double exp_ts, Mult& Cp, Mult& Cm) {
    if (fabs(sc - ev) > kGreensEps) {
        double den = sc - ev;
        Cp.v = T * (exp_te - exp_ts) / den;
        Cp.d_sec = (T * th * exp_ts - Cp.v) / den;
        Cp.d_k = (Cp.v - T * th * exp_te) / den;
        Cp.d_tau = -T / den * (ev * exp_te - sc * exp_ts);
    } else {
        Cp.v = T * exp_te * th * (1 - th / 2 * (sc - ev));
        Cp.d_sec = -th / 2.0 * th * T * exp_te;
        Cp.d_k = -Cp.v * th + th / 2.0 * th * T * exp_te;
        Cp.d_tau = T * exp_te * (1 - th * (sc - ev)) - ev * Cp.v;
    }
    if (fabs(sc + ev) > kGreensEps) {
        double den = sc + ev;
        double exp_sum = exp_te * exp_ts;
        Cm.v = T * (1 - exp_sum) / den;
        Cm.d_sec = (T * th * exp_sum - Cm.v) / den;
        Cm.d_k = Cm.d_sec;
        Cm.d_tau = T * exp_sum;
    } else {
        Cm.v = T * th * (1 - th / 2 * (sc + ev));
        Cm.d_sec = -th / 2 * th * T;
        Cm.d_k = -T * th * th / 2;
        Cm.d_tau = T * (1 - th * (sc + ev));
    }
}
SKD_HD void green_multipliers(double T, double th, double sc, double ev, Mult& Cp, Mult& Cm) {
    green_multipliers_e(T, th, sc, ev, exp(-th * ev), exp(-th * sc), Cp, Cm);
}

// =================================================================================================
// K1: homogeneous + particular solution of one layer for azimuth order m   (task = wavelength x layer)
// =================================================================================================
template <int N> SKD_HD void k1_layer_solution(const Geo& g, const Bufs& b, int s, int m, int p, int w) {
    constexpr int NSTR = 2 * N;
    using R = Rec<N>;
    const int nw = b.nw;
    double* rec = rec_at<N>(b, g.nlyr, p, w);
    auto out = [&](int e) -> double& { return rec[e]; };
    const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
    const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    const double* lpcsz = g.lpcsz + (size_t(s) * NSTR + m) * NSTR;
    double a[NSTR];
    for (int l = 0; l < NSTR; ++l) a[l] = b.a[(size_t(p) * NSTR + l) * nw + w];

    double Ep[N * N], Em[N * N], Bp[N * N], Bm[N * N], X[N * N], U[N * N], V[N * N], Tm[N * N];
    double lam[N], kk[N], sq[N], tt[N];
    for (int i = 0; i < N; ++i) {
        sq[i] = sqrt(g.wt[i] / g.mu[i]);
        tt[i] = sqrt(g.wt[i] * g.mu[i]);
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            double ep = 0, em = 0;
            for (int l = m; l < NSTR; ++l) {
                double t = a[l] * lpmu[i * NSTR + l] * lpmu[j * NSTR + l];
                if ((l - m) & 1) em += t; else ep += t;
            }
            Ep[i * N + j] = Ep[j * N + i] = ep;
            Em[i * N + j] = Em[j * N + i] = em;
            double dlt = (i == j) ? 1.0 / g.mu[i] : 0.0;
            Bp[i * N + j] = Bp[j * N + i] = dlt - sq[i] * sq[j] * om * ep;
            Bm[i * N + j] = Bm[j * N + i] = dlt - sq[i] * sq[j] * om * em;
        }
    int err = 0;
    if (!cholesky_lower<N>(Bp)) err |= ERR_CHOLESKY;  // Bp now holds R (lower), B+ = R R^T
    // As = R^T B- R
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double t = 0;
            for (int r = j; r < N; ++r) t += Bm[i * N + r] * Bp[r * N + j];
            Tm[i * N + j] = t;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = 0;
            for (int r = i; r < N; ++r) t += Bp[r * N + i] * Tm[r * N + j];
            Bm[i * N + j] = Bm[j * N + i] = t;
        }
    if (!jacobi_eigen<N>(Bm, lam, V)) err |= ERR_EIGEN;
    // xhat = R^-T v ; x = T^-1 xhat
    for (int j = 0; j < N; ++j) {
        for (int i = N - 1; i >= 0; --i) {
            double t = V[i * N + j];
            for (int r = i + 1; r < N; ++r) t -= Bp[r * N + i] * X[r * N + j];
            X[i * N + j] = t / Bp[i * N + i];
        }
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) X[i * N + j] /= tt[i];
    // S+ (into Bp), S- (into Bm):  S+- = M^-1 (I - om E+- W)
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double dlt = (i == j) ? 1.0 / g.mu[i] : 0.0;
            Bp[i * N + j] = dlt - om * g.wt[j] * Ep[i * N + j] / g.mu[i];
            Bm[i * N + j] = dlt - om * g.wt[j] * Em[i * N + j] / g.mu[i];
        }
    for (int j = 0; j < N; ++j) {
        if (!(lam[j] > 0)) { err |= ERR_EIGEN; lam[j] = fabs(lam[j]) + 1e-300; }
        kk[j] = sqrt(lam[j]);
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double t = 0;
            for (int r = 0; r < N; ++r) t += Bp[i * N + r] * X[r * N + j];
            U[i * N + j] = t;
        }
    // ---- omega derivatives (bi-orthogonal perturbation theory); Ep is reused for d(Gamma) X, Em for Z/C, Tm for dX, V for dU
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double hm = 0;
            for (int r = 0; r < N; ++r) hm += Bm[i * N + r] * X[r * N + j] / g.mu[r];
            Ep[i * N + j] = (2 * X[i * N + j] * lam[j] - U[i * N + j] / g.mu[i] - hm) / om;
        }
    double dk[N];
    double lmax = 0;
    for (int j = 0; j < N; ++j) lmax = fmax(lmax, lam[j]);
    for (int aidx = 0; aidx < N; ++aidx)
        for (int bidx = 0; bidx < N; ++bidx) {
            double z = 0;
            for (int i = 0; i < N; ++i) z += g.wt[i] * g.mu[i] * U[i * N + aidx] * Ep[i * N + bidx];
            if (aidx == bidx) {
                dk[bidx] = z / (2 * kk[bidx]);
                Em[aidx * N + bidx] = 0;
            } else {
                double den = lam[bidx] - lam[aidx];
                Em[aidx * N + bidx] = (fabs(den) > 1e-13 * lmax) ? z / den : 0.0;
            }
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double t = 0;
            for (int r = 0; r < N; ++r) t += X[i * N + r] * Em[r * N + j];
            Tm[i * N + j] = t;  // dX
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double t = (U[i * N + j] - X[i * N + j] / g.mu[i]) / om;
            for (int r = 0; r < N; ++r) t += Bp[i * N + r] * Tm[r * N + j];
            V[i * N + j] = t;  // dU
        }
    // W+-, dW+- (stored as W(i, solution j) at i + N j), norms
    double nrm[N], dnrm[N];
    for (int j = 0; j < N; ++j) {
        nrm[j] = 0;
        dnrm[j] = 0;
        out(R::K + j) = kk[j];
        out(R::E + j) = exp(-kk[j] * th);
        out(R::DK + j) = dk[j];
        for (int i = 0; i < N; ++i) {
            double uk = U[i * N + j] / kk[j];
            double duk = V[i * N + j] / kk[j] - U[i * N + j] * dk[j] / lam[j];
            double wp = 0.5 * (X[i * N + j] + uk), wm = 0.5 * (X[i * N + j] - uk);
            double dwp = 0.5 * (Tm[i * N + j] + duk), dwm = 0.5 * (Tm[i * N + j] - duk);
            out(R::WP + i + N * j) = wp;
            out(R::WM + i + N * j) = wm;
            out(R::DWP + i + N * j) = dwp;
            out(R::DWM + i + N * j) = dwm;
            // reuse X/U/Tm/V to keep W values in private memory: X<-wp, U<-wm, Tm<-dwp, V<-dwm
            X[i * N + j] = wp;
            U[i * N + j] = wm;
            Tm[i * N + j] = dwp;
            V[i * N + j] = dwm;
            nrm[j] += g.wt[i] * g.mu[i] * (wp * wp - wm * wm);
            dnrm[j] += g.wt[i] * g.mu[i] * 2.0 * (wp * dwp - wm * dwm);
        }
    }
    // ---- particular solution (Green's function)
    double Qp[N], Qm[N];
    bool allzero = true;
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    for (int i = 0; i < N; ++i) {
        double sp = 0, sn = 0;
        for (int l = m; l < NSTR; ++l) {
            double t = a[l] * lpmu[i * NSTR + l] * lpcsz[l];
            sp += t;
            sn += ((l - m) & 1) ? -t : t;
        }
        Qp[i] = sp * fm * g.wt[i] * om;
        Qm[i] = sn * fm * g.wt[i] * om;
        if (Qp[i] != 0 || Qm[i] != 0) allzero = false;
    }
    double G[4][N], dG[4][4][N];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < N; ++i) {
            G[c][i] = 0;
            for (int t = 0; t < 4; ++t) dG[t][c][i] = 0;
        }
    for (int j = 0; j < N; ++j) {
        double Ap = 0, Am = 0, dAp = 0, dAm = 0;
        if (!allzero) {
            for (int i = 0; i < N; ++i) {
                double wp = X[i * N + j], wm = U[i * N + j], dwp = Tm[i * N + j], dwm = V[i * N + j];
                Ap += Qp[i] * wp + Qm[i] * wm;
                Am += Qm[i] * wp + Qp[i] * wm;
                dAp += Qp[i] * dwp + Qp[i] / om * wp + Qm[i] * dwm + Qm[i] / om * wm;
                dAm += Qm[i] * dwp + Qm[i] / om * wp + Qp[i] * dwm + Qp[i] / om * wm;
            }
            Ap /= nrm[j];
            Am /= nrm[j];
            dAp = dAp / nrm[j] - dnrm[j] * Ap / nrm[j];
            dAm = dAm / nrm[j] - dnrm[j] * Am / nrm[j];
            Mult Cp, Cm;
            green_multipliers(T, th, sc, kk[j], Cp, Cm);
            const double dCp[4] = {Cp.d_tau, Cp.d_k * dk[j], T != 0 ? Cp.v / T : 0.0, Cp.d_sec};
            const double dCm[4] = {Cm.d_tau, Cm.d_k * dk[j], T != 0 ? Cm.v / T : 0.0, Cm.d_sec};
            for (int i = 0; i < N; ++i) {
                double wp = X[i * N + j], wm = U[i * N + j], dwp = Tm[i * N + j], dwm = V[i * N + j];
                G[0][i] += Am * Cm.v * wm;
                G[1][i] += Am * Cm.v * wp;
                G[2][i] += Ap * Cp.v * wp;
                G[3][i] += Ap * Cp.v * wm;
                for (int t = 0; t < 4; ++t) {
                    dG[t][0][i] += Am * dCm[t] * wm;
                    dG[t][1][i] += Am * dCm[t] * wp;
                    dG[t][2][i] += Ap * dCp[t] * wp;
                    dG[t][3][i] += Ap * dCp[t] * wm;
                }
                dG[TH_OM][0][i] += dAm * Cm.v * wm + Am * Cm.v * dwm;
                dG[TH_OM][1][i] += dAm * Cm.v * wp + Am * Cm.v * dwp;
                dG[TH_OM][2][i] += dAp * Cp.v * wp + Ap * Cp.v * dwp;
                dG[TH_OM][3][i] += dAp * Cp.v * wm + Ap * Cp.v * dwm;
            }
        }
        out(R::AP + j) = Ap;
        out(R::AM + j) = Am;
        out(R::DAP + j) = dAp;
        out(R::DAM + j) = dAm;
    }
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < N; ++i) {
            out(R::G + c * N + i) = G[c][i];
            for (int t = 0; t < 4; ++t) out(R::DG + (t * 4 + c) * N + i) = dG[t][c][i];
        }
    if (err) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.status[w], err);
#else
        b.status[w] |= err;
#endif
    }
}

// =================================================================================================
// K2a: block-tridiagonal boundary value problem, factor + solve      (task = wavelength)
//   block row p: rows A  -W+_p L_p - W-_p E_p M_p + [W+_{p-1} E_{p-1} L_{p-1} + W-_{p-1} M_{p-1}] = G+top_p - G+bot_{p-1}
//                rows B   W-_p E_p L_p + W+_p M_p - [W-_{p+1} L_{p+1} + W+_{p+1} E_{p+1} M_{p+1}] = G-top_{p+1} - G-bot_p
//                (last layer rows B: ground reflection condition)
//   reference conditions: bvpTOACondition / bvpContinuityCondition / bvpGroundCondition, sktran_do_rte.cpp:1897-2128
// =================================================================================================
template <int N> SKD_HD void k2a_bvp(const Geo& g, const Bufs& b, int s, int m, int w) {
    using R = Rec<N>;
    constexpr int M2 = 2 * N;
    const int nw = b.nw, nlyr = g.nlyr;
    const double rho = (m == 0) ? b.albedo[w] : 0.0;  // Lambertian
    const double kd = 2.0;                            // (1 + delta_m0) with m == 0
    double D[M2 * M2], Dinv[M2 * M2], T1[N * M2], r[M2], rprev[M2], E[N], Eprev[N];
    auto recp = [&](int p) { return RecView{rec_at<N>(b, nlyr, p, w)}; };
    for (int p = 0; p < nlyr; ++p) {
        RecView c = recp(p);
        const double th = b.tau[size_t(p) * nw + w];
        for (int j = 0; j < N; ++j) E[j] = exp(-fabs(c(R::K + j)) * th);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double wp = c(R::WP + i + N * j), wm = c(R::WM + i + N * j);
                D[i * M2 + j] = -wp;
                D[i * M2 + N + j] = -wm * E[j];
                D[(N + i) * M2 + j] = wm * E[j];
                D[(N + i) * M2 + N + j] = wp;
            }
        for (int i = 0; i < N; ++i) {
            r[i] = c(R::G + 0 * N + i);
            r[N + i] = -c(R::G + 3 * N + i);
        }
        if (p == nlyr - 1) {
            if (m == 0) {
                for (int j = 0; j < N; ++j) {
                    double sp = 0, sm = 0;
                    for (int t = 0; t < N; ++t) {
                        sp += g.wt[t] * g.mu[t] * c(R::WP + t + N * j);
                        sm += g.wt[t] * g.mu[t] * c(R::WM + t + N * j);
                    }
                    for (int i = 0; i < N; ++i) {
                        D[(N + i) * M2 + j] -= kd * rho * sp * E[j];
                        D[(N + i) * M2 + N + j] -= kd * rho * sm;
                    }
                }
                double sg = 0;
                for (int t = 0; t < N; ++t) sg += g.wt[t] * g.mu[t] * c(R::G + 2 * N + t);
                const double direct = g.csz[s] * rho / kPi * b.T[size_t(nlyr) * nw + w] + (b.surf_em ? b.surf_em[w] : 0.0);   // + surface emission (thermal pass)
                for (int i = 0; i < N; ++i) r[N + i] += direct + kd * rho * sg;
            }
        } else {
            RecView cn = recp(p + 1);
            for (int i = 0; i < N; ++i) r[N + i] += cn(R::G + 1 * N + i);
        }
        if (p > 0) {
            RecView cp = recp(p - 1);
            // T1 = Sub_p * Dinv_{p-1},  Sub_p rows A = [W+_{p-1} E_{p-1} | W-_{p-1}]
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < M2; ++j) {
                    double t = 0;
                    for (int q = 0; q < N; ++q)
                        t += cp(R::WP + i + N * q) * Eprev[q] * Dinv[q * M2 + j] + cp(R::WM + i + N * q) * Dinv[(N + q) * M2 + j];
                    T1[i * M2 + j] = t;
                }
            // D rows A -= T1[:, N:] * Sup_{p-1} rows B = [-W-_p | -W+_p E_p]
            for (int i = 0; i < N; ++i) {
                for (int j = 0; j < N; ++j) {
                    double t1 = 0, t2 = 0;
                    for (int q = 0; q < N; ++q) {
                        t1 += T1[i * M2 + N + q] * c(R::WM + q + N * j);
                        t2 += T1[i * M2 + N + q] * c(R::WP + q + N * j);
                    }
                    D[i * M2 + j] += t1;
                    D[i * M2 + N + j] += t2 * E[j];
                }
                double t = 0;
                for (int q = 0; q < M2; ++q) t += T1[i * M2 + q] * rprev[q];
                r[i] -= cp(R::G + 2 * N + i) + t;
            }
        }
        if (!invert<M2>(D, Dinv)) {
#if defined(__HIP_DEVICE_COMPILE__)
            atomicOr(&b.status[w], ERR_SINGULAR_BVP);
#else
            b.status[w] |= ERR_SINGULAR_BVP;
#endif
        }
        for (int i = 0; i < M2; ++i) {
            rprev[i] = r[i];
            xsol_at<N>(b, nlyr, p, w)[i] = r[i];
            for (int j = 0; j < M2; ++j) dinv_at<N>(b, nlyr, p, w)[i * M2 + j] = Dinv[i * M2 + j];
        }
        for (int j = 0; j < N; ++j) Eprev[j] = E[j];
    }
    // back substitution:  x_p = Dinv_p (r'_p - Sup_p x_{p+1})
    double xn[M2];
    for (int p = nlyr - 1; p >= 0; --p) {
        for (int i = 0; i < M2; ++i) r[i] = xsol_at<N>(b, nlyr, p, w)[i];
        if (p < nlyr - 1) {
            RecView cn = recp(p + 1);
            const double thn = b.tau[size_t(p + 1) * nw + w];
            for (int i = 0; i < N; ++i) {
                double t = 0;
                for (int j = 0; j < N; ++j)
                    t += cn(R::WM + i + N * j) * xn[j] + cn(R::WP + i + N * j) * exp(-fabs(cn(R::K + j)) * thn) * xn[N + j];
                r[N + i] += t;
            }
        }
        for (int i = 0; i < M2; ++i) {
            double t = 0;
            for (int j = 0; j < M2; ++j) t += dinv_at<N>(b, nlyr, p, w)[i * M2 + j] * r[j];
            D[i] = t;
        }
        for (int i = 0; i < M2; ++i) {
            xn[i] = D[i];
            xsol_at<N>(b, nlyr, p, w)[i] = D[i];
        }
    }
}

// layer-interior multipliers of the post-processing (sktran_do_postprocessing.h:19-188), value + partials
struct PPMult {
    double hp, hm, Dm, Dp;
    double hp_tau, hp_k, hm_tau, hm_k;
    double Dm_tau, Dm_k, Dm_sec, Dp_tau, Dp_k, Dp_sec;  // d/dT = value / T
};
// hp = exp(-k x), x = th lf;  hm = exp(-k (th - x));  exs = exp(-sec x);  ets = exp(-sec th)  are passed in
SKD_HD void pp_multipliers_e(double T, double th, double sc, double ev, double lf, double hp, double hm, double exs, double ets, PPMult& o) {
    const double x = th * lf;
    o.hp = hp;
    o.hm = hm;
    o.hp_tau = -o.hp * ev * lf;
    o.hp_k = -o.hp * th * lf;
    o.hm_tau = -o.hm * ev * (1 - lf);
    o.hm_k = -o.hm * th * (1 - lf);
    {
        const double den = sc - ev, exk = hp, v = exk - exs;
        o.Dm = v * T / den;
        o.Dm_sec = T * (-v / den + exs * x) / den;
        o.Dm_k = T * (-x * exk + v / den) / den;
        o.Dm_tau = lf * T * (exs * sc - exk * ev) / den;
    }
    {
        const double den = ev + sc;
        const double ef2 = ets * hm, v = exs - ef2;
        o.Dp = v * T / den;
        o.Dp_sec = T * (-v / den - x * exs + th * ef2) / den;
        o.Dp_k = T * (-v / den + ef2 * (th - x)) / den;
        o.Dp_tau = T * (-exs * lf * sc + ef2 * sc + ef2 * (ev - lf * ev)) / den;
    }
}
SKD_HD void pp_multipliers(double T, double th, double sc, double ev, double lf, PPMult& o) {
    const double x = th * lf;
    pp_multipliers_e(T, th, sc, ev, lf, exp(-ev * x), exp(-ev * (th - x)), exp(-x * sc), exp(-th * sc), o);
}

SKD_HD void ppm_store(double* dst, const PPMult& o) {
    dst[0] = o.hp; dst[1] = o.hm; dst[2] = o.Dm; dst[3] = o.Dp; dst[4] = o.hp_tau; dst[5] = o.hp_k; dst[6] = o.hm_tau;
    dst[7] = o.hm_k; dst[8] = o.Dm_tau; dst[9] = o.Dm_k; dst[10] = o.Dm_sec; dst[11] = o.Dp_tau; dst[12] = o.Dp_k; dst[13] = o.Dp_sec;
}
SKD_HD void ppm_load(const double* src, PPMult& o) {
    o.hp = src[0]; o.hm = src[1]; o.Dm = src[2]; o.Dp = src[3]; o.hp_tau = src[4]; o.hp_k = src[5]; o.hm_tau = src[6];
    o.hm_k = src[7]; o.Dm_tau = src[8]; o.Dm_k = src[9]; o.Dm_sec = src[10]; o.Dp_tau = src[11]; o.Dp_k = src[12]; o.Dp_sec = src[13];
}

// =================================================================================================
// K3: diffuse source table values J_m at the needed (layer, angle) points + convergence bookkeeping
//   (task = wavelength x layer; the bottom layer's task also fills the ground rows)
//   reference: DOSourceDiffuseStorage::accumulate_sources / accumulate_ground_sources,
//              do_source_diffuse_storage.cpp:435-695,697-1098 (values only)
// =================================================================================================
template <int N> SKD_HD void k3_table_values(const Geo& g, const Bufs& b, int s, int m, int p, int w) {
    constexpr int NSTR = 2 * N;
    using R = Rec<N>;
    const int nw = b.nw, nlyr = g.nlyr;
    RecView c{rec_at<N>(b, nlyr, p, w)};
    const double th = b.tau[size_t(p) * nw + w];
    const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    double a[NSTR], coefL[N], coefM[N], coefP[N], coefQ[N];
    for (int l = 0; l < NSTR; ++l) a[l] = b.a[(size_t(p) * NSTR + l) * nw + w];
    // per solution j: J = (1/om) sum_j [ Y+_j (h+_j L_j + A+_j D-_j) + Y-_j (h-_j M_j + A-_j D+_j) ]
    for (int j = 0; j < N; ++j) {
        PPMult pm;
        pp_multipliers(T, th, sc, c(R::K + j), g.lfrac[p], pm);
        ppm_store(rec_at<N>(b, nlyr, p, w) + R::PPM + 14 * j, pm);
        rec_at<N>(b, nlyr, p, w)[R::E + j] = exp(-fabs(c(R::K + j)) * th);
        double L = xsol_at<N>(b, nlyr, p, w)[j], Mv = xsol_at<N>(b, nlyr, p, w)[N + j];
        coefL[j] = pm.hp * L + c(R::AP + j) * pm.Dm;
        coefM[j] = pm.hm * Mv + c(R::AM + j) * pm.Dp;
    }
    // contract with W first:  J = (1/om) sum_r [ lp_r P_r + lm_r Q_r ],  P_r = sum_j (W+_rj cL_j + W-_rj cM_j),  Q_r = sum_j (W-_rj cL_j + W+_rj cM_j)
    for (int r = 0; r < N; ++r) {
        double pr = 0, qr = 0;
        for (int j = 0; j < N; ++j) {
            double wp = c(R::WP + r + N * j), wm = c(R::WM + r + N * j);
            pr += wp * coefL[j] + wm * coefM[j];
            qr += wm * coefL[j] + wp * coefM[j];
        }
        coefP[r] = pr * 0.5 * g.wt[r];
        coefQ[r] = qr * 0.5 * g.wt[r];
    }
    // fold the stream sums into Legendre space: J(angle) = sum_l a_l P_l(angle) [ (-1)^(l-m) cP_l + cQ_l ]
    double cleg[NSTR];
    for (int l = 0; l < NSTR; ++l) {
        double sp = 0, sq = 0;
        if (l >= m)
            for (int r = 0; r < N; ++r) {
                sp += lpmu[r * NSTR + l] * coefP[r];
                sq += lpmu[r * NSTR + l] * coefQ[r];
            }
        cleg[l] = a[l] * ((((l - m) & 1) ? -sp : sp) + sq);
    }
    int unconverged = 0;
    // (the thermal pass runs with the extra solar index s = nsza, which has no diffuse table)
    const bool tables = s < g.nsza;
    const int t0 = tables ? g.pt_start[s * nlyr + p] : 0, t1 = tables ? g.pt_start[s * nlyr + p + 1] : 0;
    for (int t = t0; t < t1; ++t) {
        const int slot = g.pt_slot[t];
        if (conv_at(b, g, slot, w) <= m) continue;
        const double* lpa = g.lpang + (size_t(m) * g.nangle + g.pt_angle[t]) * NSTR;
        double val = 0;
        for (int l = m; l < NSTR; ++l) val += cleg[l] * lpa[l];
        // (the 1/omega of the table and the omega inside the phase sums cancel)
        J_at(b, g, m, slot, w) = val;
        if (m >= 2) {
            double prev = J_at(b, g, m - 1, slot, w), prev2 = J_at(b, g, m - 2, slot, w);
            if ((fabs(val / prev) < 1e-4 || prev < 1e-10) && (fabs(val / prev2) < 1e-4 || prev2 < 1e-10)) {
                conv_at(b, g, slot, w) = (uint8_t)(m + 1);
            } else {
                unconverged = 1;
            }
        }
    }
    if (p == nlyr - 1 && m == 0) {
        // upwelling diffuse reflection of a Lambertian surface
        double diffuse = 0;
        for (int i = 0; i < N; ++i) {
            double sv = c(R::G + 2 * N + i);
            for (int j = 0; j < N; ++j) {
                double st = exp(-fabs(c(R::K + j)) * th);
                sv += xsol_at<N>(b, nlyr, p, w)[j] * c(R::WP + i + N * j) * st +
                      xsol_at<N>(b, nlyr, p, w)[N + j] * c(R::WM + i + N * j);
            }
            diffuse += 2.0 * g.mu[i] * g.wt[i] * sv * b.albedo[w];
        }
        if (tables)
            for (int t = g.gpt_start[s]; t < g.gpt_start[s + 1]; ++t) J_at(b, g, m, g.gpt_slot[t], w) = diffuse;
    }
    if (unconverged) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&b.notconv[w], 1);
#else
        b.notconv[w] |= 1;
#endif
    }
}

// =================================================================================================
// K2b: adjoint sweep of one line of sight for azimuth order m             (task = wavelength x LOS)
//   radiance_m(los) = sum_p u_p . x_p + particular ;  z = A^-T u ;  reduced gradient wrt
//   (tau_p, omega_p, T_p, sec_p)_p, albedo, T_floor accumulated into b.grad.
// =================================================================================================
// -------------------------------------------------------------------------------------------------
// K2c: line-of-sight independent adjoint operators of one layer               (task = wavelength x layer)
//   Everything a LOS lane needs from layer p is folded into small dense operators acting on the Legendre-space LOS
//   weights ql[l] and on the neighbouring adjoint vectors:
//     y_p = FP ql_p + SP yB_{p-1} (+ gw UG)          forward sweep  (FP = Dinv^T F, SP = -Dinv^T Sup^T)
//     z_p = y_p - BP zA_{p+1}                         backward sweep (BP = Dinv^T Sub^T)
//     grad_theta = sum_edges wI_edge . DI[theta][edge] + HD[theta] . ql_p + (zBsum + gw) GD[theta]
// -------------------------------------------------------------------------------------------------
template <int N> struct Ops {
    static constexpr int NSTR = 2 * N, M2 = 2 * N;
    static constexpr int FP = 0, SP = FP + M2 * NSTR, BP = SP + M2 * N, UG = BP + M2 * N, DI = UG + M2, HD = DI + 16 * N,
                         GD = HD + 4 * NSTR, IPB = GD + 4, SIZE = IPB + 1;
    // DI[(theta*4 + edge)*N + i], edge: 0 I+top, 1 I-top, 2 I+bot, 3 I-bot
};
template <int N> SKD_HD double* ops_at(const Bufs& b, int nlyr, int p, int w) { return b.ops + (size_t(w) * nlyr + p) * Ops<N>::SIZE; }

template <int N> SKD_HD void k2c_layer_ops(const Geo& g, const Bufs& b, int s, int m, int p, int w) {
    constexpr int NSTR = 2 * N, M2 = 2 * N;
    using R = Rec<N>;
    using O = Ops<N>;
    const int nw = b.nw, nlyr = g.nlyr;
    RecView c{rec_at<N>(b, nlyr, p, w)};
    const double* dinv = dinv_at<N>(b, nlyr, p, w);
    const double* xs = xsol_at<N>(b, nlyr, p, w);
    double* o = ops_at<N>(b, nlyr, p, w);
    const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w], T = b.T[size_t(p) * nw + w];
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    const bool last = (p == nlyr - 1);
    const double rho = (m == 0) ? b.albedo[w] : 0.0;
    double F[M2 * NSTR], HDv[4 * NSTR], tmp[M2];
    for (int i = 0; i < 4 * NSTR; ++i) HDv[i] = 0;
    PPMult pm[N];
    for (int j = 0; j < N; ++j) ppm_load(c.p + R::PPM + 14 * j, pm[j]);
    // F = [UL; UM] and the direct-dependence rows HD
    for (int l = 0; l < NSTR; ++l) {
        const double al = (l >= m) ? b.a[(size_t(p) * NSTR + l) * nw + w] : 0.0;
        const double sg = ((l - m) & 1) ? -1.0 : 1.0;
        double val_l = 0;  // coefficient of ql[l] in (omega * layer value)
        for (int j = 0; j < N; ++j) {
            double yp = 0, ym = 0, dyp = 0, dym = 0;
            for (int r = 0; r < N; ++r) {
                const double cm = 0.5 * g.wt[r] * om * al * lpmu[r * NSTR + l];
                const double wp = c(R::WP + r + N * j), wm = c(R::WM + r + N * j);
                const double dwp = c(R::DWP + r + N * j) + wp / om, dwm = c(R::DWM + r + N * j) + wm / om;
                yp += cm * (sg * wp + wm);
                ym += cm * (sg * wm + wp);
                dyp += cm * (sg * dwp + dwm);
                dym += cm * (sg * dwm + dwp);
            }
            F[j * NSTR + l] = yp * pm[j].hp / om;
            F[(N + j) * NSTR + l] = ym * pm[j].hm / om;
            const double L = xs[j], Mv = xs[N + j], dk = c(R::DK + j);
            const double Ap = c(R::AP + j), Am = c(R::AM + j), dAp = c(R::DAP + j), dAm = c(R::DAM + j);
            const double tL = pm[j].hp * L + Ap * pm[j].Dm, tM = pm[j].hm * Mv + Am * pm[j].Dp;
            val_l += yp * tL + ym * tM;
            HDv[TH_TAU * NSTR + l] += yp * (pm[j].hp_tau * L + Ap * pm[j].Dm_tau) + ym * (pm[j].hm_tau * Mv + Am * pm[j].Dp_tau);
            HDv[TH_SEC * NSTR + l] += yp * Ap * pm[j].Dm_sec + ym * Am * pm[j].Dp_sec;
            HDv[TH_T * NSTR + l] += (T != 0) ? (yp * Ap * pm[j].Dm + ym * Am * pm[j].Dp) / T : 0.0;
            HDv[TH_OM * NSTR + l] += dyp * tL + dym * tM + yp * (pm[j].hp_k * dk * L + dAp * pm[j].Dm + Ap * pm[j].Dm_k * dk) +
                                     ym * (pm[j].hm_k * dk * Mv + dAm * pm[j].Dp + Am * pm[j].Dp_k * dk);
        }
        for (int t = 0; t < 4; ++t) HDv[t * NSTR + l] /= om;
        HDv[TH_OM * NSTR + l] -= val_l / (om * om);
    }
    for (int t = 0; t < 4 * NSTR; ++t) o[O::HD + t] = HDv[t];
    // FP = Dinv^T F
    for (int i = 0; i < M2; ++i)
        for (int l = 0; l < NSTR; ++l) {
            double t = 0;
            for (int j = 0; j < M2; ++j) t += dinv[j * M2 + i] * F[j * NSTR + l];
            o[O::FP + i * NSTR + l] = t;
        }
    // SP = Dinv^T SupT with SupT[j][i] = W-(i,j), SupT[N+j][i] = W+(i,j) E_j ;  BP = Dinv^T SubT with SubT[j][i] = W+(i,j) E_j, SubT[N+j][i] = W-(i,j)
    for (int i = 0; i < M2; ++i)
        for (int cidx = 0; cidx < N; ++cidx) {
            double t1 = 0, t2 = 0;
            for (int j = 0; j < N; ++j) {
                const double wp = c(R::WP + cidx + N * j), wm = c(R::WM + cidx + N * j), E = c(R::E + j);
                t1 += dinv[j * M2 + i] * wm + dinv[(N + j) * M2 + i] * wp * E;
                t2 += dinv[j * M2 + i] * wp * E + dinv[(N + j) * M2 + i] * wm;
            }
            o[O::SP + i * N + cidx] = t1;
            o[O::BP + i * N + cidx] = t2;
        }
    // edge-intensity derivatives at fixed (L, M)
    double ipb = 0;
    for (int i = 0; i < N; ++i) {
        double d_tau[4] = {0, 0, 0, 0}, d_om[4] = {0, 0, 0, 0}, ipb_i = c(R::G + 2 * N + i);
        for (int j = 0; j < N; ++j) {
            const double wp = c(R::WP + i + N * j), wm = c(R::WM + i + N * j), dwp = c(R::DWP + i + N * j), dwm = c(R::DWM + i + N * j);
            const double E = c(R::E + j), L = xs[j], Mv = xs[N + j];
            const double dE_tau = -c(R::K + j) * E, dE_om = -c(R::DK + j) * th * E;
            // I+top = W+ L + W- E M ; I-top = W- L + W+ E M ; I+bot = W+ E L + W- M ; I-bot = W- E L + W+ M
            d_tau[0] += wm * dE_tau * Mv;
            d_tau[1] += wp * dE_tau * Mv;
            d_tau[2] += wp * dE_tau * L;
            d_tau[3] += wm * dE_tau * L;
            d_om[0] += dwp * L + dwm * E * Mv + wm * dE_om * Mv;
            d_om[1] += dwm * L + dwp * E * Mv + wp * dE_om * Mv;
            d_om[2] += dwp * E * L + wp * dE_om * L + dwm * Mv;
            d_om[3] += dwm * E * L + wm * dE_om * L + dwp * Mv;
            ipb_i += wp * E * L + wm * Mv;
        }
        ipb += g.wt[i] * g.mu[i] * ipb_i;
        for (int e = 0; e < 4; ++e) {
            o[O::DI + (TH_TAU * 4 + e) * N + i] = c(R::DG + (TH_TAU * 4 + e) * N + i) + d_tau[e];
            o[O::DI + (TH_OM * 4 + e) * N + i] = c(R::DG + (TH_OM * 4 + e) * N + i) + d_om[e];
            o[O::DI + (TH_T * 4 + e) * N + i] = c(R::DG + (TH_T * 4 + e) * N + i);
            o[O::DI + (TH_SEC * 4 + e) * N + i] = c(R::DG + (TH_SEC * 4 + e) * N + i);
        }
    }
    o[O::IPB] = ipb;
    for (int t = 0; t < 4; ++t) {
        double sacc = 0;
        if (last && m == 0)
            for (int i = 0; i < N; ++i) sacc += g.wt[i] * g.mu[i] * o[O::DI + (t * 4 + 2) * N + i];
        o[O::GD + t] = 2.0 * rho * sacc;
    }
    // ground part of u (order 0, last layer): UG = Dinv^T [2 rho sp_j E_j ; 2 rho sm_j]
    for (int i = 0; i < M2; ++i) tmp[i] = 0;
    if (last && m == 0)
        for (int j = 0; j < N; ++j) {
            double sp = 0, sm = 0;
            for (int i = 0; i < N; ++i) {
                sp += g.mu[i] * g.wt[i] * c(R::WP + i + N * j);
                sm += g.mu[i] * g.wt[i] * c(R::WM + i + N * j);
            }
            tmp[j] = 2.0 * rho * sp * c(R::E + j);
            tmp[N + j] = 2.0 * rho * sm;
        }
    for (int i = 0; i < M2; ++i) {
        double t = 0;
        for (int j = 0; j < M2; ++j) t += dinv[j * M2 + i] * tmp[j];
        o[O::UG + i] = t;
    }
}

// =================================================================================================
// K2b: adjoint sweep of one line of sight for azimuth order m             (task = wavelength x LOS)
//   radiance_m(los) = sum_p u_p . x_p + particular ;  z = A^-T u ;  reduced gradient wrt
//   (tau_p, omega_p, T_p, sec_p)_p, albedo, T_floor accumulated into b.grad.  Uses the K2c operators.
// =================================================================================================
template <int N> struct K2bTask {
    static constexpr int NSTR = 2 * N, M2 = 2 * N;
    using O = Ops<N>;
    const Geo& g;
    const Bufs& b;
    const int s, m, los, w, nw, nlyr;
    const size_t tid, nt;
    double gw = 0;
    bool table_weights = true;  // false: the caller supplies u_p directly (plane-parallel), no Legendre-space HD term
    double carry[N];  // yB_{p-1} in the forward sweep, zA_{p+1} in the backward sweep

    SKD_HD K2bTask(const Geo& g_, const Bufs& b_, int s_, int m_, int los_, int w_)
        : g(g_), b(b_), s(s_), m(m_), los(los_), w(w_), nw(b_.nw), nlyr(g_.nlyr), tid(size_t(w_) * g_.nlos + los_),
          nt(size_t(b_.nw) * g_.nlos) {
        // ground weight of this LOS (order 0 only)
        if (!g.pp_mode && m == 0 && g.ray_ground[los]) {
            for (int t = g.g_start[los]; t < g.g_start[los + 1]; ++t)
                if (g.g_sza[t] == s && conv_at(b, g, g.g_slot[t], w) > m) gw += g.g_w[t];
            gw *= b.att_total[size_t(los) * nw + w];
        }
        for (int i = 0; i < N; ++i) carry[i] = 0;
    }

    // Legendre-space LOS weights of a layer: ql[l] = sum_entries srcw * w_e * cos(m azi) * P_l(angle)
    SKD_HD void los_weights(int p, double* ql) const {
#ifdef SKD_EXP_NOQL
        for (int l = 0; l < NSTR; ++l) ql[l] = 1e-3 * (l + p);
        return;
#endif
        for (int l = 0; l < NSTR; ++l) ql[l] = 0;
        const int base = (los * g.nsza + s) * nlyr + p;
        for (int t = g.lp_start[base]; t < g.lp_start[base + 1]; ++t) {
            if (conv_at(b, g, g.lp_slot[t], w) <= m) continue;
            const int q = g.lp_q[t];
            double cw = b.srcw[size_t(q) * nw + w] * g.lp_w[t] * g.rl_cosm[size_t(m) * g.nq_total + q];
            if (cw == 0.0) continue;
            const double* lpa = g.lpang + (size_t(m) * g.nangle + g.lp_angle[t]) * NSTR;
            for (int l = m; l < NSTR; ++l) ql[l] += cw * lpa[l];
        }
    }

    // forward sweep step; `ops` may live in LDS or global memory
    SKD_HD void forward(int p, const double* ops) {
        double ql[NSTR], y[M2];
        los_weights(p, ql);
        for (int l = 0; l < NSTR; ++l) b.qlbuf[(size_t(p) * NSTR + l) * nt + tid] = ql[l];
        for (int i = 0; i < M2; ++i) {
            double t = (p == nlyr - 1) ? gw * ops[O::UG + i] : 0.0;
            for (int l = m; l < NSTR; ++l) t += ops[O::FP + i * NSTR + l] * ql[l];
            for (int cidx = 0; cidx < N; ++cidx) t += ops[O::SP + i * N + cidx] * carry[cidx];
            y[i] = t;
        }
        for (int i = 0; i < M2; ++i) b.z[(size_t(p) * M2 + i) * nt + tid] = y[i];
        for (int i = 0; i < N; ++i) carry[i] = y[N + i];
    }

    // backward sweep step  z_p = y_p - BP zA_{p+1}   (call with carry reset to zero before the first step)
    SKD_HD void backward(int p, const double* ops) {
        double z[M2];
        for (int i = 0; i < M2; ++i) {
            double t = b.z[(size_t(p) * M2 + i) * nt + tid];
            if (p < nlyr - 1)
                for (int cidx = 0; cidx < N; ++cidx) t -= ops[O::BP + i * N + cidx] * carry[cidx];
            z[i] = t;
        }
        for (int i = 0; i < M2; ++i) b.z[(size_t(p) * M2 + i) * nt + tid] = z[i];
        for (int i = 0; i < N; ++i) carry[i] = z[i];
    }

    // reduced-gradient contraction of layer p (z of layers p-1, p, p+1 and ql are read back)
    SKD_HD void gradient(int p, const double* ops) {
        const bool last = (p == nlyr - 1);
        double gth[4] = {0, 0, 0, 0};
        double zBsum = 0;
        for (int i = 0; i < N; ++i) {
            const double zA = b.z[(size_t(p) * M2 + i) * nt + tid], zB = b.z[(size_t(p) * M2 + N + i) * nt + tid];
            const double zAn = last ? 0.0 : b.z[(size_t(p + 1) * M2 + i) * nt + tid];
            const double zBp = (p > 0) ? b.z[(size_t(p - 1) * M2 + N + i) * nt + tid] : 0.0;
            if (last && m == 0) zBsum += zB;
            for (int t = 0; t < 4; ++t)
                gth[t] += zA * ops[O::DI + (t * 4 + 0) * N + i] + zBp * ops[O::DI + (t * 4 + 1) * N + i] -
                          zAn * ops[O::DI + (t * 4 + 2) * N + i] - zB * ops[O::DI + (t * 4 + 3) * N + i];
        }
        if (table_weights)
            for (int l = m; l < NSTR; ++l) {
                const double q = b.qlbuf[(size_t(p) * NSTR + l) * nt + tid];
                for (int t = 0; t < 4; ++t) gth[t] += ops[O::HD + t * NSTR + l] * q;
            }
        if (last && m == 0) {
            for (int t = 0; t < 4; ++t) gth[t] += (zBsum + gw) * ops[O::GD + t];
            const double rho = b.albedo[w], ipb = ops[O::IPB];
            b.grad[size_t(4 * nlyr) * nt + tid] += gw * 2.0 * ipb + zBsum * (2.0 * ipb + g.csz[s] / kPi * b.T[size_t(nlyr) * nw + w]);
            // solar: d / d(beam transmission at the ground); thermal pass: d / d(surface emission), which stands beside the direct bounce
            // in the ground condition without the albedo (k2a ground row)
            b.grad[size_t(4 * nlyr + 1) * nt + tid] += b.thermal_pass ? zBsum : zBsum * g.csz[s] / kPi * rho;
        }
        for (int t = 0; t < 4; ++t) b.grad[size_t(t * nlyr + p) * nt + tid] += gth[t];
    }
};

template <int N> SKD_HD void k2b_adjoint(const Geo& g, const Bufs& b, int s, int m, int los, int w) {
    K2bTask<N> t(g, b, s, m, los, w);
    const int nlyr = g.nlyr;
    for (int p = 0; p < nlyr; ++p) t.forward(p, ops_at<N>(b, nlyr, p, w));
    for (int i = 0; i < N; ++i) t.carry[i] = 0;
    for (int p = nlyr - 1; p >= 0; --p) t.backward(p, ops_at<N>(b, nlyr, p, w));
    for (int p = 0; p < nlyr; ++p) t.gradient(p, ops_at<N>(b, nlyr, p, w));
}

// =================================================================================================
// KPP: plane-parallel post-processing of one line of sight for azimuth order m     (task = wavelength x LOS)
//   I_m(los) = sum_p A_p S_p(mu) + A_ground * (reflected intensity),  A_p = attenuation from the top of layer p to
//   the observer, S_p = analytic integral of the layer source function along the LOS.  The radiance is accumulated
//   over m with cos(m azimuth); the weighting functions use the same adjoint machinery as the spherical mode
//   (u_p = dI/d(L_p, M_p), z = A^-T u, reduced gradient wrt tau, omega, T, sec of every layer, albedo, T_floor).
//   reference: OpticalLayer::integrate_source sktran_do_opticallayer.cpp:93-555, multipliers
//   sktran_do_postprocessing.h:19-188, computeReflectedIntensities sktran_do_layerarray.cpp:5-288,
//   DOSourcePlaneParallelPostProcessing::calculate do_source_planeparallel.cpp:19-218
// =================================================================================================
struct PPLosMult {
    double hp, hp_th, hp_k, hm, hm_th, hm_k;
    double Dm, Dm_th, Dm_k, Dm_T, Dm_sec, Dp, Dp_th, Dp_k, Dp_T, Dp_sec;
};
// source-function multiplier of the solar beam, E(mu, x): value and partials wrt (th, T, sec); x = (1 - lf) th
struct PPBeam { double v, d_th, d_T, d_sec; };
SKD_HD void pp_beam_multiplier(double mu, double lf, double th, double T, double sc, PPBeam& o) {
    const double x = (1 - lf) * th, den = 1.0 + mu * sc;
    const double e1 = exp(-x * sc), e2 = exp(-th * sc) * exp(-(th - x) / mu);
    o.v = T / den * (e1 - e2);
    o.d_T = (e1 - e2) / den;
    o.d_sec = T / den * e2 * th - o.v / den * mu - T / den * e1 * x;
    o.d_th = T / den * (e2 * (sc + lf / mu) - e1 * (1 - lf) * sc);
}
SKD_HD void pp_los_multipliers(double mu, double lf, double th, double T, double sc, double ev, const PPBeam& E, PPLosMult& o) {
    const double x = (1 - lf) * th, r = th - x;
    {
        // h+: 1 + mu k > 1 for an upwelling line of sight, the Taylor branch of the reference is unreachable
        const double den = 1.0 + mu * ev, e1 = exp(-x * ev), e2 = exp(-th * ev) * exp(-r / mu);
        o.hp = (e1 - e2) / den;
        o.hp_th = (-e1 * (1 - lf) * ev + e2 * (ev + lf / mu)) / den;
        o.hp_k = (-e1 * x + e2 * th) / den - o.hp * mu / den;
    }
    {
        const double den = 1.0 - mu * ev;
        if (fabs(den) > 0.0001) {
            const double e1 = exp(-ev * r), e2 = exp(-r / mu);
            o.hm = (e1 - e2) / den;
            o.hm_th = (-e1 * lf * ev + e2 * lf / mu) / den;
            o.hm_k = -e1 * r / den + o.hm * mu / den;
        } else {
            // intensity singularity mu k -> 1: first-order expansion and its derivative exactly as the reference writes them
            const double e1 = exp(-ev * r), c = 1 - r * (ev - 1 / mu);
            o.hm = e1 * r / mu * c;
            o.hm_th = -ev * o.hm + e1 / mu * c - e1 * r / mu * (ev - 1 / mu);
            o.hm_k = -th * o.hm;
        }
    }
    const double ex = exp(-th * sc);
    {
        const double den = sc + ev;
        o.Dp = (-T * ex * o.hm + E.v) / den;
        o.Dp_sec = (T * o.hm * th * ex - o.Dp + E.d_sec) / den;
        o.Dp_T = (-o.hm * ex + E.d_T) / den;
        o.Dp_th = (E.d_th - T * ex * o.hm_th + T * sc * o.hm * ex) / den;
        o.Dp_k = (-o.Dp - T * ex * o.hm_k) / den;
    }
    {
        const double den = sc - ev;
        o.Dm = (T * o.hp - E.v) / den;
        o.Dm_sec = (-o.Dm - E.d_sec) / den;
        o.Dm_T = (o.hp - E.d_T) / den;
        o.Dm_th = (T * o.hp_th - E.d_th) / den;
        o.Dm_k = (T * o.hp_k + o.Dm) / den;
    }
}

template <int N> SKD_HD void kpp_los(const Geo& g, const Bufs& b, int s, int m, int los, int w) {
    constexpr int NSTR = 2 * N, M2 = 2 * N;
    using R = Rec<N>;
    using O = Ops<N>;
    const int nw = b.nw, nlyr = g.nlyr;
    const size_t tid = size_t(w) * g.nlos + los, nt = size_t(nw) * g.nlos;
    const double mu = g.pp_mu[los];
    const double* lpl = g.lplos + (size_t(los) * NSTR + m) * NSTR;
    const double* lpmu = g.lpmu + size_t(m) * N * NSTR;
    const double* lpc = g.lpcsz + (size_t(s) * NSTR + m) * NSTR;
    const int first = g.pp_first[los];
    const double fm = (m == 0 ? 1.0 : 2.0) / (4.0 * kPi);
    K2bTask<N> t(g, b, s, m, los, w);
    t.table_weights = false;
    double Aab = cos(m * g.pp_azi[los]);  // attenuation to the observer x azimuth factor
    double total = 0;
    for (int p = 0; p < nlyr; ++p) {
        double u[M2];
        for (int i = 0; i < M2; ++i) u[i] = 0;
        double contrib = 0;
        if (p >= first) {
            const double lf = (p == first) ? g.pp_lf[los] : 1.0;
            RecView c{rec_at<N>(b, nlyr, p, w)};
            const double* xs = xsol_at<N>(b, nlyr, p, w);
            const double om = b.om[size_t(p) * nw + w], th = b.tau[size_t(p) * nw + w];
            const double T = b.T[size_t(p) * nw + w], sc = b.sec[size_t(p) * nw + w];
            double lp[N], lm[N];
            double q = 0;
            for (int r = 0; r < N; ++r) { lp[r] = 0; lm[r] = 0; }
            for (int l = m; l < NSTR; ++l) {
                const double al = b.a[(size_t(p) * NSTR + l) * nw + w] * lpl[l];
                const double sg = ((l - m) & 1) ? -1.0 : 1.0;
                for (int r = 0; r < N; ++r) {
                    const double tt = al * lpmu[r * NSTR + l];
                    lm[r] += tt;
                    lp[r] += sg * tt;
                }
                q += sg * al * lpc[l];
            }
            for (int r = 0; r < N; ++r) {
                lp[r] *= 0.5 * om * g.wt[r];
                lm[r] *= 0.5 * om * g.wt[r];
            }
            // thermal pass: Q E is the direct emission term (1 - omega) E_thermal, present whatever the single-scatter source
            // (sktran_do_opticallayer.cpp:513-521)
            const double Q = (g.include_ss || b.thermal_pass) ? q * fm * om : 0.0;
            PPBeam E;
            pp_beam_multiplier(mu, lf, th, T, sc, E);
            double S = Q * E.v;
            double dS[4];
            dS[TH_TAU] = Q * E.d_th;
            dS[TH_OM] = Q / om * E.v;
            dS[TH_T] = Q * E.d_T;
            dS[TH_SEC] = Q * E.d_sec;
            for (int j = 0; j < N; ++j) {
                double yp = 0, ym = 0, dyp = 0, dym = 0;
                for (int r = 0; r < N; ++r) {
                    const double wp = c(R::WP + r + N * j), wm = c(R::WM + r + N * j);
                    const double dwp = c(R::DWP + r + N * j), dwm = c(R::DWM + r + N * j);
                    yp += lp[r] * wp + lm[r] * wm;
                    ym += lp[r] * wm + lm[r] * wp;
                    dyp += lp[r] * dwp + lm[r] * dwm;
                    dym += lp[r] * dwm + lm[r] * dwp;
                }
                dyp += yp / om;
                dym += ym / om;
                PPLosMult pm;
                pp_los_multipliers(mu, lf, th, T, sc, c(R::K + j), E, pm);
                const double L = xs[j], Mv = xs[N + j], dk = c(R::DK + j);
                const double Ap = c(R::AP + j), Am = c(R::AM + j), dAp = c(R::DAP + j), dAm = c(R::DAM + j);
                S += yp * pm.hp * L + ym * pm.hm * Mv + Ap * yp * pm.Dm + Am * ym * pm.Dp;
                dS[TH_TAU] += yp * L * pm.hp_th + ym * Mv * pm.hm_th + Ap * yp * pm.Dm_th + Am * ym * pm.Dp_th;
                dS[TH_T] += Ap * yp * pm.Dm_T + Am * ym * pm.Dp_T;
                dS[TH_SEC] += Ap * yp * pm.Dm_sec + Am * ym * pm.Dp_sec;
                dS[TH_OM] += dyp * (pm.hp * L + Ap * pm.Dm) + dym * (pm.hm * Mv + Am * pm.Dp) +
                             yp * (pm.hp_k * dk * L + dAp * pm.Dm + Ap * pm.Dm_k * dk) +
                             ym * (pm.hm_k * dk * Mv + dAm * pm.Dp + Am * pm.Dp_k * dk);
                u[j] = Aab * yp * pm.hp;
                u[N + j] = Aab * ym * pm.hm;
            }
            contrib = Aab * S;
            if (b.derivs)
                for (int th_i = 0; th_i < 4; ++th_i) b.grad[size_t(th_i * nlyr + p) * nt + tid] += Aab * dS[th_i];
            Aab *= exp(-th * lf / mu);
        }
        total += contrib;
        if (b.derivs) {
            b.qlbuf[(size_t(p) * NSTR) * nt + tid] = contrib;
            // forward substitution of A^T z = u:  y_p = Dinv^T u_p + SP yB_{p-1} (+ ground part below)
            const double* dinv = dinv_at<N>(b, nlyr, p, w);
            const double* ops = ops_at<N>(b, nlyr, p, w);
            for (int i = 0; i < M2; ++i) {
                double acc = 0;
                for (int j = 0; j < M2; ++j) acc += dinv[j * M2 + i] * u[j];
                for (int cidx = 0; cidx < N; ++cidx) acc += ops[O::SP + i * N + cidx] * t.carry[cidx];
                b.z[(size_t(p) * M2 + i) * nt + tid] = acc;
            }
            if (p < nlyr - 1)
                for (int i = 0; i < N; ++i) t.carry[i] = b.z[(size_t(p) * M2 + N + i) * nt + tid];
        }
    }
    // Lambertian ground seen through the whole column (order 0 only)
    double ground = 0;
    if (m == 0) {
        RecView c{rec_at<N>(b, nlyr, nlyr - 1, w)};
        const double* xs = xsol_at<N>(b, nlyr, nlyr - 1, w);
        const double th = b.tau[size_t(nlyr - 1) * nw + w], rho = b.albedo[w];
        double ipb = 0;
        for (int i = 0; i < N; ++i) {
            double v = c(R::G + 2 * N + i);
            for (int j = 0; j < N; ++j)
                v += c(R::WP + i + N * j) * exp(-fabs(c(R::K + j)) * th) * xs[j] + c(R::WM + i + N * j) * xs[N + j];
            ipb += g.wt[i] * g.mu[i] * v;
        }
        const double direct = g.include_ss ? g.csz[s] / kPi * b.T[size_t(nlyr) * nw + w] : 0.0;
        // thermal pass: the surface emission joins the reflected intensity where the direct bounce does (sktran_do_layerarray.cpp:255-266)
        ground = Aab * (rho * (2.0 * ipb + direct) + ((g.include_ss && b.surf_em) ? b.surf_em[w] : 0.0));
        total += ground;
        if (b.derivs) {
            t.gw = Aab;
            const double* ops = ops_at<N>(b, nlyr, nlyr - 1, w);
            for (int i = 0; i < M2; ++i) b.z[(size_t(nlyr - 1) * M2 + i) * nt + tid] += t.gw * ops[O::UG + i];
            // direct-beam part of the reflected intensity (the diffuse part is handled by K2bTask::gradient through gw)
            b.grad[size_t(4 * nlyr) * nt + tid] += Aab * direct;
            b.grad[size_t(4 * nlyr + 1) * nt + tid] += b.thermal_pass ? ((g.include_ss && b.surf_em) ? Aab : 0.0)
                                                                          : (g.include_ss ? Aab * rho * g.csz[s] / kPi : 0.0);
        }
    }
    b.radiance[tid] += total;
    if (!b.derivs) return;
    // attenuation of everything below a layer: dI/dtau_q = -(lf_q / mu) * (radiance entering the layer from below)
    {
        double below = ground;
        for (int p = nlyr - 1; p >= first; --p) {
            const double lf = (p == first) ? g.pp_lf[los] : 1.0;
            b.grad[size_t(TH_TAU * nlyr + p) * nt + tid] -= lf / mu * below;
            below += b.qlbuf[(size_t(p) * NSTR) * nt + tid];
        }
    }
    for (int i = 0; i < N; ++i) t.carry[i] = 0;
    for (int p = nlyr - 1; p >= 0; --p) t.backward(p, ops_at<N>(b, nlyr, p, w));
    for (int p = 0; p < nlyr; ++p) t.gradient(p, ops_at<N>(b, nlyr, p, w));
}

// =================================================================================================
// K4a: reverse chain through the beam (T, sec) and the layer optics into native derivatives
//      (task = wavelength x LOS, once per solar zenith index)
//   reference wiring: sktran_do_layerarray.cpp:496-649,729-757 and configureTransmission :890-979
// =================================================================================================
SKD_HD void k4a_chain(const Geo& g, const Bufs& b, int s, int los, int w, bool clear = true) {
    const int nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    const size_t tid = size_t(w) * g.nlos + los, nt = size_t(nw) * g.nlos;
    auto G = [&](int e) -> double& { return b.grad[size_t(e) * nt + tid]; };
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    // d/d tau_r of everything downstream: eight layers r per pass so that the gradients G(sec_p), G(T_{p+1}) are read once per pass
    constexpr int RC = 8;
    for (int r0 = 0; r0 < nlyr; r0 += RC) {
        double gt[RC];
        for (int k = 0; k < RC; ++k) gt[k] = (r0 + k < nlyr) ? G(r0 + k) : 0.0;
        for (int p = 0; p < nlyr; ++p) {
            const double* ch = g.chapman + (size_t(s) * nlyr + p) * nlyr;
            if (ch[0] < 0) continue;
            // sec_p = (sig_p - sig_{p-1}) / tau_p ;  T_{p+1} = F0 exp(-sig_p)
            const double* chm = (p > 0) ? g.chapman + (size_t(s) * nlyr + p - 1) * nlyr : nullptr;
            // when the level above was in darkness the reference differences against the flagged value with zero derivative
            const bool prev_ok = (p > 0) && !(chm[0] < 0);
            const double inv_tau = 1.0 / b.tau[size_t(p) * nw + w];
            const double gsec = G(3 * nlyr + p), sec_p = b.sec[size_t(p) * nw + w];
            const double gT = (p + 1 < nlyr) ? G(2 * nlyr + p + 1) : G(4 * nlyr + 1);
            const double Tn = b.T[size_t(p + 1) * nw + w];
            for (int k = 0; k < RC; ++k) {
                const int r = r0 + k;
                if (r >= nlyr) break;
                const double dsig_p = ch[r];
                double dsec = (dsig_p - (prev_ok ? chm[r] : 0.0)) * inv_tau;
                if (r == p) dsec -= sec_p * inv_tau;
                gt[k] += gsec * dsec - gT * Tn * dsig_p;
            }
        }
        // tau_r -> extinction ; omega_r -> ssa, extinction
        for (int k = 0; k < RC; ++k) {
            const int r = r0 + k;
            if (r >= nlyr) break;
            const double gom = G(nlyr + r);
            const double kb = b.kbar[size_t(r) * nw + w], omr = b.om[size_t(r) * nw + w];
            for (int t = 0; t < 2; ++t) {
                int q = g.wq_idx[2 * r + t];
                if (q < 0) continue;
                double wt = g.wq_w[2 * r + t];
                b.dnat[size_t(q) * nt + tid] += gt[k] * wt * g.dh[r] + gom * wt * (ssa[q] - omr) / kb;
                b.dnat[size_t(ngrid + q) * nt + tid] += gom * wt * ext[q] / kb;
            }
        }
    }
    b.dnat[size_t(2 * ngrid) * nt + tid] += G(4 * nlyr);
    // the gradients accumulate per solar zenith index; after the last one of a solve nothing reads them again (the next solve's
    // initialisation clears them): `clear` = false saves 0.4 GB of stores at the C2 size
    if (clear)
        for (int e = 0; e < 4 * nlyr + 2; ++e) G(e) = 0;
}

// K4a of the thermal pass (k0_thermal): the layer gradients with respect to (tau, omega, T, sec) -- T = b0 (1 - omega) 4 pi / omega and
// sec = b1 = ln(B_top / B_bottom) / tau here -- chained into native derivatives.  Nothing depends on the layers above (no beam):
//   d/d tau_p   = G_tau - G_sec b1 / tau                 d/d omega_p = G_omega - G_T b0 4 pi / omega^2
//   d/d B_top   = G_T (1 - omega) 4 pi / omega + G_sec / (B_top tau)          d/d B_bottom = -G_sec / (B_bottom tau)
// (the reference wires the same partials, sktran_do_layerarray.cpp:565-620, 700-727).  Extinction / ssa parts join the solar ones in
// dnat; the emission profile and surface emission derivatives go straight to the caller's arrays [altitude][w][los], [w][los].
SKD_HD void k4a_thermal(const Geo& g, const Bufs& b, int los, int w, int w0, int nw_total, double* wf_em, double* wf_sem) {
    const int nw = b.nw, nlyr = g.nlyr, ngrid = g.ngrid;
    const size_t tid = size_t(w) * g.nlos + los, nt = size_t(nw) * g.nlos;
    const size_t o = size_t(w0 + w) * g.nlos + los, plane = size_t(nw_total) * g.nlos;
    auto G = [&](int e) -> double& { return b.grad[size_t(e) * nt + tid]; };
    const double* ext = b.ext + size_t(w) * ngrid;
    const double* ssa = b.ssa + size_t(w) * ngrid;
    const double* em = b.emission + size_t(w) * ngrid;
    if (wf_em)
        for (int q = 0; q < ngrid; ++q) wf_em[size_t(q) * plane + o] = 0.0;
    for (int r = 0; r < nlyr; ++r) {
        int lo = -1, hi = -1;
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * r + t];
            if (q < 0) continue;
            if (lo < 0 || q < lo) lo = q;
            if (q > hi) hi = q;
        }
        const double th = b.tau[size_t(r) * nw + w], om = b.om[size_t(r) * nw + w], kb = b.kbar[size_t(r) * nw + w];
        const double top = hi >= 0 ? em[hi] : 0.0, bot = (lo >= 0 && lo != hi) ? em[lo] : top;
        const double b1 = b.sec[size_t(r) * nw + w];   // as k0_thermal left it
        const double gT = G(2 * nlyr + r), gsec = G(3 * nlyr + r);
        const bool slope = th > 1e-10 && top > 1e-30 && bot > 1e-30 && lo != hi && lo >= 0;
        const double gtau = G(r) - (slope ? gsec * b1 / th : 0.0);
        const double gom = G(nlyr + r) - gT * top * (4.0 * kPi) / (om * om);
        if (wf_em && hi >= 0) {
            wf_em[size_t(hi) * plane + o] += gT * (1 - om) * (4.0 * kPi) / om + (slope ? gsec / (top * th) : 0.0);
            if (slope) wf_em[size_t(lo) * plane + o] -= gsec / (bot * th);
        }
        for (int t = 0; t < 2; ++t) {
            const int q = g.wq_idx[2 * r + t];
            if (q < 0) continue;
            const double wt = g.wq_w[2 * r + t];
            b.dnat[size_t(q) * nt + tid] += gtau * wt * g.dh[r] + gom * wt * (ssa[q] - om) / kb;
            b.dnat[size_t(ngrid + q) * nt + tid] += gom * wt * ext[q] / kb;
        }
    }
    b.dnat[size_t(2 * ngrid) * nt + tid] += G(4 * nlyr);
    if (wf_sem) wf_sem[o] = G(4 * nlyr + 1);
    for (int e = 0; e < 4 * nlyr + 2; ++e) G(e) = 0;
}

// =================================================================================================
// K4b: line-of-sight integration of the diffuse table (value + the LOS-level derivative terms)
//      (task = wavelength x LOS)
//   reference: SourceIntegrator::integrate_ray sourceintegrator.cpp:521-580,
//              DOSourceInterpolatedPostProcessing::integrated_source/end_of_ray_source do_source_interpolated_pp.cpp:95-211
// =================================================================================================
SKD_HD void k4b_los(const Geo& g, const Bufs& b, int los, int w) {
    const int nw = b.nw, ngrid = g.ngrid, nstr = g.nstr;
    const size_t tid = size_t(w) * g.nlos + los, nt = size_t(nw) * g.nlos;
    // I = sum_q a_q omega_q (1 - att_q) sv_q + a_total * ground, a_q = attenuation from the near edge of shell q to the observer
    double behind = 0;  // radiance arriving at the far edge of the current shell
    if (g.ray_ground[los]) {
        for (int t = g.g_start[los]; t < g.g_start[los + 1]; ++t) {
            int slot = g.g_slot[t];
            if (conv_at(b, g, slot, w) > 0) behind += g.g_w[t] * J_at(b, g, 0, slot, w);  // order 0 only
        }
    }
    const double att_total = b.att_total[size_t(los) * nw + w];
    double prod_upto = 1.0;  // prod_{q' <= q} att_q'
    for (int q = g.ray_qstart[los]; q < g.ray_qstart[los + 1]; ++q) {
        const double att = exp(-b.od[size_t(q) * nw + w]);
        prod_upto *= att;
        double sv = 0;
        if (g.rl_active[q]) {
            // sum over the table entries of the shell and the azimuth orders each entry's point was solved for
            const double* cs = g.rl_cosm_t + size_t(q) * nstr;
            for (int t = g.rl_tab_start[q]; t < g.rl_tab_start[q + 1]; ++t) {
                const int slot = g.tab_slot[t];
                const int cv = int(conv_at(b, g, slot, w));
                const int mmax = (cv < nstr) ? cv : nstr;
                const double* Jp = &J_at(b, g, 0, slot, w);
                double a = 0;
                for (int m = 0; m < mmax; ++m) a += cs[m] * Jp[m];
                sv += g.tab_w[t] * a;
            }
        }
        const double omega = b.omloc[size_t(q) * nw + w];
        if (b.derivs) {
            const double a_q = (prod_upto > 0) ? att_total / prod_upto : 0.0;
            const double dI_dod = a_q * (-behind * att + omega * att * sv);
            for (int t = g.rl_od_start[q]; t < g.rl_od_start[q + 1]; ++t) b.dnat[size_t(g.od_idx[t]) * nt + tid] += g.od_w[t] * dI_dod;
            if (g.rl_active[q]) {
                const double dI_dom = a_q * (1 - att) * sv;
                for (int t = g.rl_ssa_start[q]; t < g.rl_ssa_start[q + 1]; ++t) b.dnat[size_t(ngrid + g.ssa_idx[t]) * nt + tid] += g.ssa_w[t] * dI_dom;
            }
        }
        behind = behind * att + (g.rl_active[q] ? omega * (1 - att) * sv : 0.0);
    }
    b.radiance[tid] = behind;
}

// stop rule of the azimuth expansion (DOSource::calculate, do_source.cpp:47-57): after order m >= 2 with every needed
// table point converged no further order is solved for this solar zenith index.   (task = wavelength)
// INVARIANT the order pipeline of run_chunk relies on (kernels of other orders read these without an event): norders[w] is only ever
// lowered, and only to m + 1 > m; conv_order entries are only ever lowered, and only to m + 1 (K3 of order m + 1 writes m + 2).
SKD_HD void k_stop_rule(const Geo& g, const Bufs& b, int s, int m, int w) {
    const int nw = b.nw;
    if (m >= b.norders[w]) return;
    if (m >= 2 && b.notconv[w] == 0) {
        b.norders[w] = m + 1;
        for (int p = 0; p < g.nlyr; ++p)
            for (int t = g.pt_start[s * g.nlyr + p]; t < g.pt_start[s * g.nlyr + p + 1]; ++t) {
                uint8_t& c = conv_at(b, g, g.pt_slot[t], w);
                if (c > m + 1) c = (uint8_t)(m + 1);
            }
    }
    b.notconv[w] = 0;
}

}  // namespace skd
