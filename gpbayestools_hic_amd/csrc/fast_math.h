// fast_math.h — fp64 exp / sqrt / kernel shape functions for the K(X,X) assembly (k_kmat_mfma), where they are most of the
// vector instructions.  (Measured in k_kcross as well, tools/gpu_shard_sim.py A/B in one process: 2.8996 vs 2.8979 ms/step at
// 2048-row batches, 0.4359 vs 0.4349 at 256 — that kernel is bound by its LDS broadcasts, not by exp; it keeps the library forms.)  Each is within 1-2 ulp of the library form that
// sklearn's numpy calls round to (the parity bars on kernel matrices are 1e-13 .. 1e-11).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/gpbayes.h"

namespace gpb {

// exp(x) for x <= 0 without the library's special-case selects, its coefficients in SGPRs (scalar loads from constant
// memory: as immediates the compiler re-materialised them in VGPRs with two v_mov_b32 each — 12 per evaluation, a sixth of
// the kernel's vector instructions).  n = rint(x log2 e), r = x - n ln 2 (two-term), e^r by a polynomial of degree 11
// (|r| <= ln 2 / 2: 1.6e-17; round 3 used the Taylor polynomial of degree 13 — two more multiply-adds per pair of a kernel that
// runs at the package's power limit), 2^n by v_ldexp_f64 (underflows to 0 for n < -1074).  x is clamped at -746 first
// (exp(-746) rounds to 0): without the clamp n = rint(x log2 e) stops being exact beyond |x| ~ 2^53, r stops being small, the
// polynomial overflows and x = -inf gives NaN (fma(-inf, -ln 2, -inf)) where the library exp of k_kcross gives 0 — extreme
// theta then left NaN in K(X,X) and 0 in K*.
// Measured against long-double expl over 4 M random arguments in [-700, 0] (tools/micro/exp_accuracy.hip, profiles/r04_exp_accuracy.txt):
// max 0.94 ulp, mean 0.258 — the device library's exp: 0.86 / 0.258, the Taylor-13 form: 0.88 / 0.257; numpy's exp, which sklearn calls, is
// within 1 ulp: at most 4e-16 relative between the two (the G1 / G2 bars: 1e-13, 1e-11).
static __constant__ double EXP_C[16] = {1.4426950408889634, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
                                 // e^r = 1 + r + r^2 (c2 + c3 r + ... + c11 r^9), |r| <= ln 2 / 2: interpolation of (e^r - 1 - r) / r^2 at
                                 // ten Chebyshev nodes (tools/exp_poly.py; near-minimax: 1.6e-17 relative in exact arithmetic with these
                                 // doubles, the Taylor polynomial needs degree 13 for that) — c11 first
                                 2.51004512318139235428e-08, 2.76202012214868374137e-07, 2.75572683411822321766e-06,
                                 2.48015210588699591568e-05, 1.98412698631052176416e-04, 1.38888889173661989052e-03,
                                 8.33333333333006499866e-03, 4.16666666666238236227e-02, 1.66666666666666685170e-01,
                                 5.00000000000000111022e-01, 0.0, 0.0, 1.0 / 3.0};
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -746.0);
    const double n = __builtin_rint(x * EXP_C[0]);
    double r = fma(n, EXP_C[1], x);
    r = fma(n, EXP_C[2], r);
    double q = EXP_C[3];
#pragma unroll
    for (int k = 4; k <= 12; ++k) q = fma(q, r, EXP_C[k]);      // c10 ... c2
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)n);
}
// The same for N independent arguments, step by step side by side: a wave that evaluates one exp at a time walks a chain of ~17
// dependent fp64 operations at their full latency (k_kcross at three waves per SIMD kept the vector pipes 66 % busy, and a cheaper
// exp changed nothing: latency, not issue); N chains interleaved in the source give the scheduler N operations in flight.
// Element for element the operations of exp_nonpos: same bits.
template <int N>
__device__ __forceinline__ void exp_nonpos_n(const double (&xin)[N], double (&out)[N]) {
    double x[N], n[N], r[N], q[N];
#pragma unroll
    for (int u = 0; u < N; ++u) x[u] = fmax(xin[u], -746.0);
#pragma unroll
    for (int u = 0; u < N; ++u) n[u] = __builtin_rint(x[u] * EXP_C[0]);
#pragma unroll
    for (int u = 0; u < N; ++u) r[u] = fma(n[u], EXP_C[1], x[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) r[u] = fma(n[u], EXP_C[2], r[u]);
#pragma unroll
    for (int u = 0; u < N; ++u) q[u] = EXP_C[3];
#pragma unroll
    for (int k = 4; k <= 12; ++k)
#pragma unroll
        for (int u = 0; u < N; ++u) q[u] = fma(q[u], r[u], EXP_C[k]);
#pragma unroll
    for (int u = 0; u < N; ++u) q[u] = fma(q[u], r[u], 1.0);
#pragma unroll
    for (int u = 0; u < N; ++u) q[u] = fma(q[u], r[u], 1.0);
#pragma unroll
    for (int u = 0; u < N; ++u) out[u] = ldexp(q[u], (int)n[u]);
}

// sqrt(x) for x >= 0 by v_rsq_f64 and two coupled Newton steps (Goldschmidt), 1 ulp, without the library form's scaling
// and special-case selects (17 instructions); x is clamped to 1e-300 first (coincident points: sqrt = 1e-150, K = 1).
__device__ __forceinline__ double sqrt_pos(double x) {
    x = fmax(x, 1e-300);
    const double y = __builtin_amdgcn_rsq(x);
    double sq = x * y, h = 0.5 * y;
    const double e = fma(-h, sq, 0.5);
    sq = fma(sq, e, sq);
    h = fma(h, e, h);
    return fma(fma(-sq, sq, x), h, sq);
}
// shape functions of k_kmat_mfma: as shape_fn, with exp_nonpos, sqrt_pos and t^2 / 3 as a multiplication (each within
// 1-2 ulp of the library form)
template <int KIND>
__device__ __forceinline__ double shape_fn_fast(double r2) {
    if (KIND == GPB_KERNEL_RBF) {
        return exp_nonpos(-0.5 * r2);
    } else if (KIND == GPB_KERNEL_MATERN15) {
        const double t = sqrt_pos(r2) * 1.7320508075688772;
        return (1.0 + t) * exp_nonpos(-t);
    } else {
        const double t = sqrt_pos(r2) * 2.23606797749979;
        return (1.0 + t + (t * t) * EXP_C[15]) * exp_nonpos(-t);
    }
}

// ... for N arguments side by side (see exp_nonpos_n): what k_kcross evaluates for the walkers a lane holds
template <int KIND, int N>
__device__ __forceinline__ void shape_fn_fast_n(const double (&r2)[N], double (&out)[N]) {
    double x[N], e[N];
    if (KIND == GPB_KERNEL_RBF) {
#pragma unroll
        for (int u = 0; u < N; ++u) x[u] = -0.5 * r2[u];
        exp_nonpos_n<N>(x, out);
    } else {
        double t[N];
#pragma unroll
        for (int u = 0; u < N; ++u) t[u] = sqrt_pos(r2[u]) * (KIND == GPB_KERNEL_MATERN15 ? 1.7320508075688772 : 2.23606797749979);
#pragma unroll
        for (int u = 0; u < N; ++u) x[u] = -t[u];
        exp_nonpos_n<N>(x, e);
#pragma unroll
        for (int u = 0; u < N; ++u)
            out[u] = (KIND == GPB_KERNEL_MATERN15 ? (1.0 + t[u]) : (1.0 + t[u] + (t[u] * t[u]) * EXP_C[15])) * e[u];
    }
}

}  // namespace gpb
