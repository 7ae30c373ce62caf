// accuracy of fast_math.h's exp_nonpos (degree-11 polynomial) against long-double expl on the host, next to round 3's Taylor-13
// form and the device library's exp: max and mean error in ulp over random arguments in [-lim, 0].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../gpbayestools_hic_amd/csrc/fast_math.h"
using namespace gpb;

__device__ __forceinline__ double exp_taylor13(double x) {
    x = fmax(x, -746.0);
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    const double c[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                          1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};
    double q = c[0];
#pragma unroll
    for (int k = 1; k < 12; ++k) q = fma(q, r, c[k]);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)n);
}
__global__ void k(const double* x, double* a, double* b, double* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = exp_nonpos(x[i]); b[i] = exp_taylor13(x[i]); c[i] = exp(x[i]); }
}
int main() {
    const int n = 1 << 22;
    for (double lim : {1.0, 40.0, 700.0}) {
        std::vector<double> hx(n), ha(n), hb(n), hc(n);
        srand(7);
        for (int i = 0; i < n; ++i) hx[i] = -lim * ((double)rand() / RAND_MAX) * ((double)rand() / RAND_MAX);
        double *x, *a, *b, *c;
        hipMalloc(&x, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
        hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, a, b, c, n);
        hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost);
        double mx[3] = {0, 0, 0}, sm[3] = {0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const long double t = expl((long double)hx[i]);
            if (t < 1e-300L) continue;
            const double u = std::nextafter((double)t, INFINITY) - (double)t;      // one ulp at the result
            const double e[3] = {(double)fabsl((long double)ha[i] - t) / u, (double)fabsl((long double)hb[i] - t) / u,
                                 (double)fabsl((long double)hc[i] - t) / u};
            for (int q = 0; q < 3; ++q) { mx[q] = fmax(mx[q], e[q]); sm[q] += e[q]; }
        }
        printf("x in [-%g, 0]: ulp error max / mean   degree-11 %.3f / %.3f   Taylor-13 %.3f / %.3f   library exp %.3f / %.3f\n", lim,
               mx[0], sm[0] / n, mx[1], sm[1] / n, mx[2], sm[2] / n);
        hipFree(x); hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
