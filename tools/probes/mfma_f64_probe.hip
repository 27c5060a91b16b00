// Probe: operand/result lane layout and issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (build: hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout(const double* A, const double* B, double* D) {  // A 16x4 row-major, B 4x16 row-major, D 16x16
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
// chained use: C/D registers 2,3 of Y as the B operand of the next product (rows 8..15 of Y)
__global__ void chain(const double* S, const double* Y, double* D) {  // S 16x8 row-major, Y 16x16, D = S * Y[8:16,:]
    const int l = threadIdx.x;
    d4 y;
    for (int r = 0; r < 4; ++r) y[r] = Y[((l >> 4) + 4 * r) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    for (int kk = 0; kk < 2; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[(l & 15) * 8 + 4 * kk + (l >> 4)], y[2 + kk], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
template <int INDEP> __global__ void rate(double* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    d4 acc[INDEP];
    for (int i = 0; i < INDEP; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < INDEP; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < INDEP; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int INDEP> __global__ void rate_fma(double* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    double acc[INDEP];
    for (int i = 0; i < INDEP; ++i) acc[i] = i;
    double a = 1.0 + l * 1e-9, b = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < INDEP; ++i) acc[i] = fma(acc[i], a, b);
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < INDEP; ++i) s += acc[i];
    out[blockIdx.x * 64 + l] = s;
    if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    std::vector<double> A(64), B(64), D(256), S(128), Y(256);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + 0.37 * i + 0.01 * i * i; B[i] = 2 - 0.11 * i + 0.003 * i * i * i; }
    for (int i = 0; i < 128; ++i) S[i] = 0.5 + 0.21 * i - 0.002 * i * i;
    for (int i = 0; i < 256; ++i) Y[i] = 1.5 - 0.13 * i + 0.0007 * i * i;
    double *dA, *dB, *dD, *dS, *dY, *dO; long long* dC;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048); hipMalloc(&dS, 1024); hipMalloc(&dY, 2048);
    hipMalloc(&dO, size_t(8) * 64 * 16384); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dS, S.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), 2048, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 4; ++k) r += A[i * 4 + k] * B[k * 16 + j]; err = fmax(err, fabs(r - D[i * 16 + j]) / fabs(r)); }
    printf("layout max rel err %.3e\n", err);
    chain<<<1, 64>>>(dS, dY, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 8; ++k) r += S[i * 8 + k] * Y[(8 + k) * 16 + j]; err = fmax(err, fabs(r - D[i * 16 + j]) / fabs(r)); }
    printf("chain  max rel err %.3e\n", err);
    long long c;
    const int iters = 2000;
#define RUN(K, NI, NB) { K<NI><<<NB, 64>>>(dO, iters, dC); hipDeviceSynchronize(); hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); printf(#K " indep=%d blocks=%d: %.1f cycles/instr (wave0)\n", NI, NB, double(c) / (double(iters) * NI)); }
    RUN(rate, 1, 1) RUN(rate, 2, 1) RUN(rate, 4, 1) RUN(rate, 8, 1)
    RUN(rate, 4, 1024) RUN(rate, 4, 2048) RUN(rate, 4, 4096)
    RUN(rate_fma, 1, 1) RUN(rate_fma, 8, 1) RUN(rate_fma, 8, 1024) RUN(rate_fma, 8, 4096)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
#define WALL(K, NI, NB, FLOP) { K<NI><<<NB, 64>>>(dO, iters, dC); hipDeviceSynchronize(); hipEventRecord(e0); K<NI><<<NB, 64>>>(dO, iters, dC); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf(#K " indep=%d blocks=%d: %.3f ms, %.1f TFLOP/s\n", NI, NB, ms, double(FLOP) * iters * NI * NB / ms * 1e-9); }
    WALL(rate, 4, 1024, 2048) WALL(rate, 4, 4096, 2048) WALL(rate, 4, 8192, 2048) WALL(rate, 1, 8192, 2048)
    WALL(rate_fma, 8, 1024, 128) WALL(rate_fma, 8, 4096, 128) WALL(rate_fma, 8, 8192, 128) WALL(rate_fma, 1, 8192, 128)
    return 0;
}
