// Standalone microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 as a function of the number of
// independent accumulators per wave and of waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NA, int NB>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[NA][NB];
    double a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    for (int j = 0; j < NB; ++j) b[j] = 1.0 - 1e-9 * (threadIdx.x + j);
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    asm volatile("" ::"v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0); }
    if (s == 12345.678) out[2] = s;
}

// the same loop with the accumulators pinned to AccVGPRs (inline asm "a" constraint)
template <int NA, int NB>
__global__ __launch_bounds__(256) void k_agpr(double* out, int iters) {
    d4 acc[NA][NB];
    double a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
    for (int j = 0; j < NB; ++j) b[j] = 1.0 - 1e-9 * (threadIdx.x + j);
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
    double s = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    asm volatile("" ::"v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0); }
    if (s == 12345.678) out[2] = s;
}

template <int NA, int NB, bool AGPR = false>
void run(const char* name, int blocks) {
    double* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000 / (NA * NB);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (AGPR) hipLaunchKernelGGL((k_agpr<NA, NB>), dim3(blocks), dim3(256), 0, 0, d, iters);
        else      hipLaunchKernelGGL((k<NA, NB>), dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    double h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double n_mfma_wave = (double)iters * NA * NB;
    const double tf = (double)blocks * 4 * n_mfma_wave * 2048 / (ms * 1e-3) / 1e12;
    printf("%-10s blocks=%5d waves/SIMD=%.0f  %7.2f TF/s  cycles/MFMA/wave=%7.1f  clock=%.2f GHz\n", name, blocks,
           blocks / 256.0, tf, h[0] / n_mfma_wave, h[0] / h[1] * 0.1);
    hipFree(d);
}

int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<1, 1>("acc1", blocks);
        run<2, 2>("acc4", blocks);
        run<2, 4>("acc8", blocks);
        run<4, 4>("acc16", blocks);
        run<2, 2, true>("acc4/agpr", blocks);
        run<2, 4, true>("acc8/agpr", blocks);
        run<4, 4, true>("acc16/agpr", blocks);
    }
    return 0;
}
