// Microbenchmark of the building blocks of the polarised kernels on one MI355X: time per call of the register Gauss-Jordan
// inverses and of a team matrix product, one workgroup per CU, operands in LDS.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sasktran2_amd/csrc/hip tools/probes/vec_micro.hip -o tools/probes/_bin/vec_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "disco_vec_core.hpp"

template <int MODE, int M2> __global__ void __launch_bounds__(512) kern(double* out, int reps) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ double lds[];
    double* D = lds;
    double* Dinv = lds + M2 * M2;
    double* buf = Dinv + M2 * M2;
    const skv::Team team{int(threadIdx.x), int(blockDim.x)};
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int e = threadIdx.x; e < M2 * M2; e += blockDim.x) {
            const int i = e / M2, j = e % M2;
            D[e] = (i == j ? 4.0 + 0.01 * i : 0.0) + 1.0 / (1.0 + i + 2 * j + r);
        }
        __syncthreads();
        if constexpr (MODE == 0) {
            if constexpr (M2 == 96) skv::gj_inverse_registers<6>(D, Dinv, buf);
            else skv::gj_inverse_registers<3>(D, Dinv, buf);
        } else if constexpr (MODE == 1) {
            if (threadIdx.x < 64) skv::gj_inverse_wave<M2 / 8>(D, Dinv, reinterpret_cast<int*>(buf));
        } else if constexpr (MODE == 3) {
            // symmetric test matrix (strong diagonal, smooth off-diagonal part) and V = identity
            for (int e = threadIdx.x; e < M2 * M2; e += blockDim.x) {
                const int i = e / M2, j = e % M2;
                D[e] = (i == j ? 1.0 + 0.05 * i : 0.0) + 0.3 / (1.0 + i + j + r);
                Dinv[e] = (i == j) ? 1.0 : 0.0;
            }
            __syncthreads();
            skv::team_jacobi(team, M2, D, Dinv, buf);
        } else {
            skv::team_gemm(team, M2, M2, M2, [&](int i, int k) { return D[i * M2 + k]; }, [&](int k, int j) { return D[k * M2 + j]; },
                           [&](int i, int j, double v) { Dinv[i * M2 + j] = v; });
        }
        __syncthreads();
        acc += Dinv[(threadIdx.x * 7) % (M2 * M2)];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
#endif
}

static int g_blocks = 256;
template <int MODE, int M2> void run(const char* name, int threads) {
    const int reps = 50, blocks = g_blocks;
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 512);
    const size_t bytes = sizeof(double) * (2 * M2 * M2 + 10 * M2 + 128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern<MODE, M2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<MODE, M2>), dim3(blocks), dim3(threads), bytes, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE, M2>), dim3(blocks), dim3(threads), bytes, 0, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> h(16);
    hipMemcpy(h.data(), out, sizeof(double) * 16, hipMemcpyDeviceToHost);
    printf("%-28s M2 %3d threads %3d: %8.2f us per call (checksum %.6e) %s\n", name, M2, threads, 1e3 * ms / reps, h[3], hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);   // workgroups in flight (256 = one per CU)
    printf("workgroups: %d\n", g_blocks);
    run<0, 96>("gj_inverse_registers<6>", 256);
    run<0, 48>("gj_inverse_registers<3>", 256);
    run<1, 48>("gj_inverse_wave<6>", 64);
    run<1, 48>("gj_inverse_wave<6> (256 thr)", 256);
    run<3, 48>("team_jacobi", 256);
    run<3, 48>("team_jacobi", 512);
    run<3, 24>("team_jacobi", 128);
    run<3, 24>("team_jacobi", 64);
    run<2, 96>("team_gemm", 256);
    run<2, 48>("team_gemm", 256);
    run<2, 48>("team_gemm", 64);
    return 0;
}
