// Standalone probe: where do the cycles of potf2_inv_64 go (gpbayestools_hic_amd/csrc/chol_block.h)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../gpbayestools_hic_amd/csrc potf2_probe.hip -o potf2_probe
// Factors a random SPD 64x64 block on `nblocks` workgroups (each its own copy), prints the s_memtime differences
// between the phase boundaries (100 MHz ticks -> ns x10) of workgroup 0, the HIP-event time of the launch, and
// the residuals |L L^T - A|, |X L - I| against the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "chol_block.h"
using namespace gpb;

template <bool STAMP>
__global__ __launch_bounds__(CHOL_THREADS, 4) void k_probe(const double* A, double* L, double* X, unsigned long long* stamps) {
    __shared__ CholLds s;
    load_tile(A, 64, s.a, threadIdx.x);
    potf2_inv_64<STAMP>(s, stamps + 64 * blockIdx.x);
    for (int e = threadIdx.x; e < 4096; e += CHOL_THREADS) {
        L[(size_t)blockIdx.x * 4096 + e] = s.a[e >> 6][e & 63];
        X[(size_t)blockIdx.x * 4096 + e] = s.x[e >> 6][e & 63];
    }
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<double> A(4096), B(4096);
    srand(1);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += B[i * 64 + k] * B[j * 64 + k];
            A[i * 64 + j] = s / 64 + (i == j ? 0.5 : 0.0);
        }
    double *dA, *dL, *dX;
    unsigned long long* dS;
    hipMalloc(&dA, 4096 * 8); hipMalloc(&dL, (size_t)nblocks * 4096 * 8); hipMalloc(&dX, (size_t)nblocks * 4096 * 8);
    hipMalloc(&dS, (size_t)nblocks * 64 * 8);
    hipMemcpy(dA, A.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(k_probe<false>, dim3(nblocks), dim3(CHOL_THREADS), 0, 0, dA, dL, dX, dS);
            else hipLaunchKernelGGL(k_probe<true>, dim3(nblocks), dim3(CHOL_THREADS), 0, 0, dA, dL, dX, dS);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: best launch %.2f us (%d workgroups)\n", variant ? "stamped" : "plain  ", best * 1e3, nblocks);
    }
    std::vector<unsigned long long> st(64);
    hipMemcpy(st.data(), dS, 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"zero+pass 0", "update 0", "pass 1", "update 1", "pass 2", "update 2", "pass 3",
                           "inverses 16", "asm16 T", "asm16 X", "asm32 T", "asm32 X"};
    for (int i = 1; i <= 12; ++i) printf("  %-14s %6llu cycles\n", names[i - 1], st[i] - st[i - 1]);
    printf("  total          %6llu cycles (s_memtime counts shader cycles)\n", st[12] - st[0]);
    std::vector<double> L(4096), X(4096);
    hipMemcpy(L.data(), dL, 4096 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dX, 4096 * 8, hipMemcpyDeviceToHost);
    double r1 = 0, r2 = 0;
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < 64; ++k) {       // the routine leaves the strictly upper blocks of s.a untouched: mask them
                s += (k <= i && k <= j) ? L[i * 64 + k] * L[j * 64 + k] : 0.0;
                t += (k <= i && j <= k) ? X[i * 64 + k] * L[k * 64 + j] : 0.0;
            }
            r1 = fmax(r1, fabs(s - A[i * 64 + j]));
            r2 = fmax(r2, fabs(t - (i == j ? 1.0 : 0.0)));
        }
    printf("residuals: |L L^T - A| %.2e   |X L - I| %.2e\n", r1, r2);
    return 0;
}
