// Micro-benchmark (gfx950): cycles per instruction of dependent / independent FP64 FMAs, v_mov_b64_dpp row_newbcast, rsq / rcp,
// with 1, 2, 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 fp64_latency_probe.hip -o fp64_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int L> __device__ __forceinline__ double bc(double v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false); }
constexpr int ITER = 4096;
__global__ void k_dep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = fma(x, a, b);
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH> __global__ void k_ind(double* out, long long* cyc, double a, double b) {
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = out[threadIdx.x] + c;
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
    }
    long long t1 = clock64();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x], y = x + 1, z = x + 2, w = x + 3;
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = fma(bc<3>(y), a, x);
            y = fma(bc<5>(z), a, y);
            z = fma(bc<7>(w), a, z);
            w = fma(bc<9>(x), a, w);
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the same broadcast as two 32-bit DPP moves (low / high word)
template <int L> __device__ __forceinline__ double bc32(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = int(bits), hi = int(bits >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + L, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + L, 0xf, 0xf, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi2 << 32) | (unsigned)lo2));
}
__global__ void k_dpp32(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x], y = x + 1, z = x + 2, w = x + 3;
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = fma(bc32<3>(y), a, x);
            y = fma(bc32<5>(z), a, y);
            z = fma(bc32<7>(w), a, z);
            w = fma(bc32<9>(x), a, w);
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dppdep(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = bc<3>(x) + a;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rsq(double* out, long long* cyc, double a, double b) {
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = __builtin_amdgcn_rsq(x) + a;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K> void run(const char* name, K kern, int waves_per_simd, int ops) {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8 * 4096); hipMalloc(&cyc, 8 * 4096);
    hipMemset(out, 0, 64 * 8 * 4096);
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD %d: %.2f clock64 ticks per op (wave 0), %.3f ms -> %.2f ns per op per wave\n", name, waves_per_simd, double(c0) / (double(ITER) * ops), ms,
           ms * 1e6 / (double(ITER) * ops));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("dependent fma f64", k_dep, w, 16);
        run("2 independent chains", k_ind<2>, w, 16);
        run("4 independent chains", k_ind<4>, w, 16);
        run("8 independent chains", k_ind<8>, w, 16);
        run("fma with row_newbcast src x4", k_dpp, w, 16);
        run("fma with 2 x b32 row_newbcast x4", k_dpp32, w, 16);
        run("dependent mov_dpp + add", k_dppdep, w, 16);
        run("dependent rsq + add", k_rsq, w, 16);
    }
    return 0;
}
