// Probe: gj16_inplace / gj16_scatter (sasktran2_amd/csrc/hip/disco_wave_kernels.hpp) against a host Gauss-Jordan inverse.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../sasktran2_amd/csrc/hip/disco_wave_kernels.hpp"

__global__ void invert(const double* A, double* Ainv, int* sing, double* dbg) {
    const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
    const double* a = A + size_t(blockIdx.x) * 256;
    __shared__ double sP[256];
    skdev::mf4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = a[(4 * r + q) * 16 + col];
    // row max probe
    double key = a[col * 16 + 0];
    dbg[blockIdx.x * 64 + lane] = skdev::row16_max(fabs(key));
    unsigned long long pP, pI;
    long long t0 = clock64();
    int s = skdev::gj16_inplace(acc, col, q, pP, pI, blockIdx.x == 0 ? dbg + 1024 : nullptr);
    long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 1) dbg[1100] = double(t1 - t0);
    __syncthreads();
    skdev::gj16_scatter(acc, col, q, pP, pI, sP);
    __syncthreads();
    for (int r = 0; r < 4; ++r) Ainv[size_t(blockIdx.x) * 256 + (4 * r + q) * 16 + col] = sP[(4 * r + q) * 16 + col];
    if (lane == 0) sing[blockIdx.x] = s;
}
int main() {
    const int nb = 8;
    std::vector<double> A(nb * 256), Ai(nb * 256), dbg(2048);
    unsigned s = 12345;
    for (auto& v : A) { s = s * 1664525u + 1013904223u; v = (double(s >> 8) / double(1 << 24)) - 0.5; }
    for (int b = 0; b < nb; ++b) for (int i = 0; i < 16; ++i) A[b * 256 + i * 16 + ((i * 7 + b) % 16)] += 2.0;  // permuted dominance -> pivoting needed
    double *dA, *dAi, *dD; int* dS;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dAi, A.size() * 8); hipMalloc(&dS, nb * 4); hipMalloc(&dD, dbg.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    invert<<<nb, 64>>>(dA, dAi, dS, dD);
    hipDeviceSynchronize();
    hipMemcpy(Ai.data(), dAi, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(dbg.data(), dD, dbg.size() * 8, hipMemcpyDeviceToHost);
    std::vector<int> sing(nb); hipMemcpy(sing.data(), dS, nb * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
        double err = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double t = 0; for (int k = 0; k < 16; ++k) t += A[b * 256 + i * 16 + k] * Ai[b * 256 + k * 16 + j];
            err = fmax(err, fabs(t - (i == j)));
        }
        double cm = 0; for (int i = 0; i < 16; ++i) cm = fmax(cm, fabs(A[b * 256 + i * 16]));
        printf("matrix %d: |A Ainv - I|max = %.3e sing=%d ; row16_max lanes 0,5,17,63 = %.6f %.6f %.6f %.6f expected %.6f\n", b, err, sing[b], dbg[b * 64], dbg[b * 64 + 5], dbg[b * 64 + 17], dbg[b * 64 + 63], cm);
    }
    printf("gj16_inplace: %.0f clock ticks for 16 pivot steps (one wave alone on its SIMD)\n", dbg[1100]);
    printf("stage ticks (16 steps): gather %.0f keymax %.0f readlanes %.0f rcp %.0f ops+mfma %.0f fixups %.0f\n", dbg[1064], dbg[1065], dbg[1066], dbg[1067], dbg[1068], dbg[1069]);
    // host reference pivot sequence for matrix 0 (in-place GJ, rows not moved)
    {
        std::vector<double> M(A.begin(), A.begin() + 256);
        unsigned used = 0;
        for (int k = 0; k < 16; ++k) {
            int pr = -1; double best = -1;
            for (int i = 0; i < 16; ++i) if (!((used >> i) & 1) && fabs(M[i * 16 + k]) > best) { best = fabs(M[i * 16 + k]); pr = i; }
            double p = M[pr * 16 + k];
            printf("k=%2d host piv row %2d p=% .6f | gpu row %2.0f p=% .6f\n", k, pr, p, dbg[1024 + 2 * k], dbg[1024 + 2 * k + 1]);
            for (int i = 0; i < 16; ++i) if (i != pr) { double f = M[i * 16 + k] / p; for (int j = 0; j < 16; ++j) if (j != k) M[i * 16 + j] -= f * M[pr * 16 + j]; M[i * 16 + k] = -f; }
            for (int j = 0; j < 16; ++j) if (j != k) M[pr * 16 + j] /= p;
            M[pr * 16 + k] = 1 / p;
            used |= 1u << pr;
        }
    }
    return 0;
}
