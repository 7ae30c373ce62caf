// Standalone probe: how fast does ONE wave issue the instruction kinds of the Cholesky chain's in-register pass (chol_block.h)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 issue_rate_probe.hip -o issue_rate_probe
// One workgroup of `nw` waves (1 or 2 per SIMD), each runs N copies of one instruction kind between two s_memtime reads; prints
// shader cycles per instruction for wave 0.  The accumulators are independent (8 of them in rotation): issue rate, not latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ __launch_bounds__(512) void k_issue(double* out, unsigned long long* cyc, double seed) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    double m = seed * 0.5 + threadIdx.x;
    double sv = seed;                                  // uniform: lives in scalar registers
    asm volatile("" : "+s"(sv));
    int lo = __double2loint(m), hi = __double2hiint(m), s0 = 0, s1 = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "s"(sv), "v"(m));
            if (KIND == 1) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m));
            if (KIND == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(m));
            if (KIND == 3) { asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(lo)); asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s1) : "v"(hi)); }
            if (KIND == 4) {       // the pass's triple: two readlanes (for a later column) + one fma on an earlier scalar
                asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(lo)); asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s1) : "v"(hi));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "s"(sv), "v"(m));
            }
            if (KIND == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(lo) : "v"(hi), "v"(hi));
            if (KIND == 6) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(m));
            if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(m));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[threadIdx.x] = r + s0 + s1 + lo;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, double* out, unsigned long long* cyc) {
    for (int nw : {1, 4, 8}) {
        unsigned long long best = ~0ull, c;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k_issue<KIND>, dim3(1), dim3(64 * nw), 0, 0, out, cyc, 1.25);
            hipDeviceSynchronize();
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (c < best) best = c;
        }
        printf("%-44s %d wave(s): %6.2f cycles per instruction (%llu cycles / %d)\n", name, nw, (double)best / (512.0 * per_iter), best, 512 * per_iter);
    }
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8);
    run<0>("v_fmac_f64 v, s, v", 1, out, cyc);
    run<1>("v_fmac_f64 v, v, v", 1, out, cyc);
    run<2>("v_fmac_f64_dpp row_newbcast", 1, out, cyc);
    run<6>("v_mov_b64_dpp row_newbcast", 1, out, cyc);
    run<3>("v_readlane_b32 (pairs)", 2, out, cyc);
    run<4>("2 x v_readlane_b32 + v_fmac_f64 v, s, v", 3, out, cyc);
    run<5>("v_fmac_f32 v, v, v", 1, out, cyc);
    run<7>("v_pk_fma_f32", 1, out, cyc);
    return 0;
}
