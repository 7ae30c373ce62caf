// Standalone probe: where does the gfx950 dispatcher place consecutive workgroups of a 1-D grid?
// Each workgroup records (XCC, SE, CU) and its start time, then spins ~30 us so that the whole grid is
// co-resident (20 KB LDS + 256 threads per workgroup, like k_predict<64,4>).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result dispatch_probe.hip -o dispatch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long spin) {
    __shared__ double pad[2560];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    pad[threadIdx.x] = (double)hw;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = hw;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (unsigned)t0;
        out[4 * blockIdx.x + 3] = (unsigned)pad[5];
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1280;
    unsigned* d;
    hipMalloc(&d, n * 16);
    unsigned* h = (unsigned*)malloc(n * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 3000ull);      // 100 MHz counter: 30 us
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, n * 16, hipMemcpyDeviceToHost);
    printf("# block xcc se sh cu t0\n");
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[4 * b];
        printf("%d %u %u %u %u %u\n", b, h[4 * b + 1], (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, h[4 * b + 2] - h[2]);
    }
    return 0;
}
