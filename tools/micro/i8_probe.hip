// Standalone microbenchmark: issue rate of v_mfma_i32_16x16x64_i8 on gfx950 (the int8 rate behind an Ozaki-type
// emulation of the fp64 product).  hipcc --offload-arch=gfx950 -O3 i8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NA, int NB>
__global__ __launch_bounds__(256) void k(int* out, int iters) {
    v4i acc[NA][NB], a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = v4i{1 + i, 2, 3, (int)threadIdx.x};
    for (int j = 0; j < NB; ++j) b[j] = v4i{4, 5 + j, 6, 7};
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < NA; ++i) for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123456789) out[0] = s;
}

template <int NA, int NB>
void run(const char* name, int blocks) {
    int* d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400000 / (NA * NB);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NA, NB>), dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_mfma_wave = (double)iters * NA * NB;
    const double tops = (double)blocks * 4 * n_mfma_wave * (16.0 * 16 * 64 * 2) / (ms * 1e-3) / 1e12;
    printf("%-8s blocks=%5d waves/SIMD=%.0f  %8.1f TOP/s (int8 x int8 -> int32)\n", name, blocks, blocks / 256.0, tops);
    (void)hipFree(d);
}

int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<2, 2>("acc4", blocks);
        run<4, 4>("acc16", blocks);
    }
    return 0;
}
