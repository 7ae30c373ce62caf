// Do fp64 MFMA and fp64 VALU work of DIFFERENT waves on one SIMD overlap on gfx950, or do they share the DP units?
// One workgroup of 8 waves per CU (100 KB of LDS keeps a second one out): waves 0-3 and 4-7 land on SIMDs 0-3 pairwise.
//   mode 0: waves 0-3 run a chain of 4 independent v_mfma_f64_16x16x4_f64, waves 4-7 idle
//   mode 1: waves 4-7 run 8 independent v_fma_f64 chains, waves 0-3 idle
//   mode 2: both
//   mode 3: every wave runs both, interleaved in one instruction stream (4 waves only do work: waves 0-3)
//   mode 4 / 5: waves 4-7 run 64 v_fma_f32 per iteration, alone / next to the MFMA waves
//   mode 6 / 7: waves 4-7 run 64 v_mad_u32_u24 per iteration, alone / next to the MFMA waves
// overlap  => t(2) ~ max(t(0), t(1));  shared units => t(2) ~ t(0) + t(1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int mode, int iters, double* out) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    if (mode >= 4) {
        const bool m = (mode == 5 || mode == 7) && wave < 4, v32 = (mode == 4 || mode == 5) && wave >= 4, vi = mode >= 6 && wave >= 4;
        d4 ac[4];
        for (int i = 0; i < 4; ++i) ac[i] = d4{0.0, 0.0, 0.0, 0.0};
        const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        float g[8]; unsigned h[8];
        for (int i = 0; i < 8; ++i) { g[i] = 0.5f + i * 1e-3f; h[i] = threadIdx.x + i; }
        const float gb = 1.0f - threadIdx.x * 1e-7f, ga = 1e-3f;
        if (m)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ac[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, ac[i], 0, 0, 0);
            }
        if (v32)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = __builtin_fmaf(g[i], gb, ga);
            }
        if (vi)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (h[i] ^ (unsigned)it) + (h[i] >> 3);
            }
        double s = 0.0;
        for (int i = 0; i < 4; ++i) s += ac[i][0] + ac[i][1] + ac[i][2] + ac[i][3];
        for (int i = 0; i < 8; ++i) s += g[i] + h[i];
        if (s == 12345.678) out[threadIdx.x] = s + lds[threadIdx.x];
        return;
    }
    const bool do_m = (mode == 0 || mode == 2) ? wave < 4 : (mode == 3 ? wave < 4 : false);
    const bool do_v = (mode == 1 || mode == 2) ? wave >= 4 : (mode == 3 ? wave < 4 : false);
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5 + i * 1e-3;
    if (mode != 3) {
        if (do_m)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
        if (do_v)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)          // 64 fma per iteration = the DP-unit time of 4 MFMAs (4 x 64 cycles)
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = fma(f[i], b, a);
            }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fma(f[j], b, a);
            }
        }
    }
    double s = 0.0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678) out[threadIdx.x] = s + lds[threadIdx.x];
}

int main() {
    double* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 8; ++mode) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, mode, iters, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: 4 MFMA per iteration (mode 0), 64 FMA per iteration (mode 1)
            printf("rep %d mode %d  %.3f ms   cycles/iteration at 2.4 GHz: %.1f\n", rep, mode, ms, ms * 1e-3 * 2.4e9 / iters);
        }
    return 0;
}
