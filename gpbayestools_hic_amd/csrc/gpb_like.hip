// gpb_like.hip — PC -> observable transform, batched multivariate-normal log-likelihood,
// prior box, and the emcee-equivalent stretch move.
//   k_obs       Emulator.predict after the per-GP calls            src/emulator.py:555-605
//   k_loglike   Chain._predict block + mvn_loglike, fused          src/mcmc.py:23-65,153-166,288-293
//   k_box       strict prior box + constant                        src/mcmc.py:194-198,275-276,296-297
//   k_propose / k_accept   emcee StretchMove (a=2) as driven by     src/mcmc.py:68-92,372-412
#include "gpb_internal.h"
#include <math.h>

namespace gpb {

// ------------------------------------------------------------------ observable transform (materialised)
// one workgroup per walker; writes mean[w][M] and (optionally) cov[w][M][M]
__global__ __launch_bounds__(256) void k_obs(const double* __restrict__ mean_pc, const double* __restrict__ var_pc,
                                             const double* __restrict__ estd, int64_t Wld, int P, int M, int mode,
                                             const double* __restrict__ A, const double* __restrict__ mu,
                                             const double* __restrict__ scale, const double* __restrict__ C0,
                                             double* __restrict__ mean_out, double* __restrict__ cov_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* zm = sm;            // [P]
    double* zv = sm + P;        // [P]
    double* mo = sm + 2 * P;    // [M]
    const int64_t w = blockIdx.x;
    const int tid = threadIdx.x;
    const bool no_pca = (mode == GPB_MODE_NO_PCA || mode == GPB_MODE_NO_PCA_EXPDIAG);
    const bool expdiag = (mode == GPB_MODE_EXPDIAG || mode == GPB_MODE_NO_PCA_EXPDIAG);
    const double e = estd ? estd[w] : 0.0;
    for (int p = tid; p < P; p += 256) {
        zm[p] = mean_pc[(int64_t)p * Wld + w];
        zv[p] = var_pc ? (var_pc[(int64_t)p * Wld + w] + e * e) : 0.0;     // src/emulator.py:578-579
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) {
        double v;
        if (!no_pca) {
            v = 0.0;
            for (int p = 0; p < P; ++p) v = fma(zm[p], A[p * M + m], v);   // :559-561, 373-374
            v += mu[m];
        } else {
            v = zm[m] * scale[m] + mu[m];                                    // :563-565
        }
        if (expdiag) v = exp(v);                                             // :567-568
        mo[m] = v;
        mean_out[w * M + m] = v;
    }
    if (!cov_out) return;
    __syncthreads();
    double* co = cov_out + w * (int64_t)M * M;
    for (int e2 = tid; e2 < M * M; e2 += 256) {
        const int i = e2 / M, j = e2 % M;
        double v;
        if (!no_pca) {
            if (expdiag && i != j) v = 0.0;
            else {
                v = 0.0;
                for (int p = 0; p < P; ++p) v = fma(zv[p] * A[p * M + i], A[p * M + j], v);   // :584-586
                v += C0[i * M + j];                                                            // :587
            }
        } else {
            v = (i == j) ? zv[i] : 0.0;                                                        // :590-592
        }
        if (expdiag && i == j) {
            const double f = sqrt(v) * mo[i];                                                  // :599-600
            v = f * f;
        }
        co[e2] = v;
    }
}

int launch_obs(gpb_ctx* ctx, int64_t W, const double* estd_dev, double* mean_dev, double* cov_dev) {
    const size_t sh = (2 * ctx->P + ctx->M) * sizeof(double);
    hipLaunchKernelGGL(k_obs, dim3((unsigned)W), dim3(256), sh, ctx->stream, ctx->mean_pc,
                       cov_dev ? ctx->var_pc : nullptr, estd_dev, ctx->Wld, (int)ctx->P, (int)ctx->M, ctx->mode,
                       ctx->A, ctx->mu, ctx->scale, ctx->C0, mean_dev, cov_dev);
    GPB_HIP(hipGetLastError());
    return 0;
}

// Right-looking Cholesky of the augmented (M+1) x M lower matrix held at c (LDS or global slab),
// then ll[w] = -1/2 |v|^2 - sum log L_jj with v = last row.  Non-PD -> NaN and a counted status.
__device__ __forceinline__ void chol_aug_finish(double* c, int ld, int M, double* dg, double* __restrict__ ll,
                                                int64_t w, int accumulate, int* __restrict__ notpd) {
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    bool bad = false;
    for (int j = 0; j < M; ++j) {
        const double ajj = c[j * ld + j];
        if (!(ajj > 0.0)) bad = true;
        const double dd = sqrt(ajj);
        __syncthreads();
        for (int i = j + 1 + tid; i <= M; i += 256) c[i * ld + j] = c[i * ld + j] / dd;
        if (tid == 0) dg[j] = dd;
        __syncthreads();
        for (int i = j + 1 + ty; i <= M; i += 16) {
            const double lij = c[i * ld + j];
            const int kmax = (i < M) ? i : (M - 1);
            for (int k = j + 1 + tx; k <= kmax; k += 16) c[i * ld + k] = fma(-lij, c[k * ld + j], c[i * ld + k]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        double q = 0.0, ld_sum = 0.0;
        for (int j = 0; j < M; ++j) {
            const double v = c[M * ld + j];
            q = fma(v, v, q);
            ld_sum += log(dg[j]);
        }
        double r = -0.5 * q - ld_sum;
        if (bad) {
            r = nan("");
            atomicAdd(notpd, 1);
        }
        ll[w] = accumulate ? (ll[w] + r) : r;
    }
}

// Generic batched mvn_loglike(y, cov) (src/mcmc.py:23-65) on caller-provided dY[W,M], cov[W,M,M].
template <bool GWS>
__global__ __launch_bounds__(256) void k_mvn(const double* __restrict__ dY, const double* __restrict__ cov, int M,
                                             double* __restrict__ gws, double* __restrict__ ll,
                                             int* __restrict__ notpd) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int64_t w = blockIdx.x;
    const int tid = threadIdx.x, ld = M + 1;
    double* dg = sm;
    double* c;
    if (GWS) c = gws + w * (int64_t)(M + 1) * ld;      // separate instantiations keep LDS accesses as ds_* ops
    else     c = sm + M;
    const double* cw = cov + w * (int64_t)M * M;
    for (int e2 = tid; e2 < M * M; e2 += 256) {
        const int i = e2 / M, j = e2 % M;
        if (j <= i) c[i * ld + j] = cw[e2];
    }
    for (int m = tid; m < M; m += 256) c[M * ld + m] = dY[w * M + m];
    __syncthreads();
    chol_aug_finish(c, ld, M, dg, ll, w, 0, notpd);
}

// ------------------------------------------------------------------ fused block log-likelihood
// One workgroup per walker.  Builds dY = mean - y_exp and C = cov_model + cov_exp directly in LDS
// (or in a global scratch slab when M > 128), then factorises the augmented matrix
//        [ C   . ]            [ L    0 ]
//        [ dY^T . ]   ->      [ v^T  . ]     with  L v = dY,
// so that  -1/2 dY^T C^-1 dY - sum log L_ii = -1/2 |v|^2 - sum log L_ii   (src/mcmc.py:42-65).
template <bool GWS>
__global__ __launch_bounds__(256) void k_loglike(const double* __restrict__ mean_pc,
                                                 const double* __restrict__ var_pc, int64_t Wld, int P, int M,
                                                 int mode, const double* __restrict__ A,
                                                 const double* __restrict__ mu, const double* __restrict__ scale,
                                                 const double* __restrict__ C0, const double* __restrict__ yexp,
                                                 const double* __restrict__ Cexp, double* __restrict__ gws,
                                                 double* __restrict__ ll, int accumulate, int* __restrict__ notpd) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int64_t w = blockIdx.x;
    const int tid = threadIdx.x;
    const int ld = M + 1;
    double* zm = sm;                 // [P]
    double* zv = sm + P;             // [P]
    double* mo = sm + 2 * P;         // [M]
    double* dg = sm + 2 * P + M;     // [M] pivots
    double* c;                       // [(M+1)][ld]
    if (GWS) c = gws + w * (int64_t)(M + 1) * ld;
    else     c = sm + 2 * P + 2 * M;
    const bool no_pca = (mode == GPB_MODE_NO_PCA || mode == GPB_MODE_NO_PCA_EXPDIAG);
    const bool expdiag = (mode == GPB_MODE_EXPDIAG || mode == GPB_MODE_NO_PCA_EXPDIAG);
    for (int p = tid; p < P; p += 256) {
        zm[p] = mean_pc[(int64_t)p * Wld + w];
        zv[p] = var_pc[(int64_t)p * Wld + w];       // extra_std == 0 on this path (src/mcmc.py:205,281)
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) {
        double v;
        if (!no_pca) {
            v = 0.0;
            for (int p = 0; p < P; ++p) v = fma(zm[p], A[p * M + m], v);
            v += mu[m];
        } else {
            v = zm[m] * scale[m] + mu[m];
        }
        if (expdiag) v = exp(v);
        mo[m] = v;
        c[M * ld + m] = v - yexp[m];                // dY (src/mcmc.py:288)
    }
    __syncthreads();
    for (int e2 = tid; e2 < M * M; e2 += 256) {
        const int i = e2 / M, j = e2 % M;
        if (j > i) continue;
        double v;
        if (!no_pca) {
            if (expdiag && i != j) v = 0.0;
            else {
                v = 0.0;
                for (int p = 0; p < P; ++p) v = fma(zv[p] * A[p * M + i], A[p * M + j], v);
                v += C0[i * M + j];
            }
        } else {
            v = (i == j) ? zv[i] : 0.0;
        }
        if (expdiag && i == j) {
            const double f = sqrt(v) * mo[i];
            v = f * f;
        }
        c[i * ld + j] = v + Cexp[i * M + j];        // src/mcmc.py:290
    }
    __syncthreads();
    chol_aug_finish(c, ld, M, dg, ll, w, accumulate, notpd);
}

// ------------------------------------------------------------------ fused block log-likelihood, register-resident
// Fast path for the PCA mode with M <= 64 (every emulator of the reference's analyses): ONE WAVE per
// walker, lane i owns row i of C = sum_p var_p A_p^T A_p + C_trunc + C_exp in VGPRs.  Right-looking
// Cholesky fully unrolled: pivots and column entries are broadcast with v_readlane (-> SGPR operands of
// the v_fma_f64), no LDS traffic and no barriers inside the factorisation; the forward solve L v = dY is
// folded into the same sweep.  Rows >= M are padded with the identity (log 1 = 0, v = 0).
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// Optional fused prior box (src/mcmc.py:194-198,275-276,296-297): X == nullptr -> plain block log-likelihood.
struct BoxArgs {
    const double* X;      // [W][d] walker positions
    const double* lo;     // [d]
    const double* hi;     // [d]
    int d;
    double outside;       // -inf or -1e300
    double inside_const;  // 2 log(1e-16)
    // compacted batch (launch_compact): X == nullptr, the box is already applied; workspace row w is row cmp[4 + w] of
    // the output, rows w >= cmp[0] do not exist (their numbers are stale workspace contents and are discarded)
    const int* cmp;
};

// Optional fused k_finalize: mpart != nullptr -> the kernel sums the per-chunk mean partials and the per-row-block
// sum-of-squares partials of its walker itself, in k_finalize's order (same bits), instead of reading mean_pc / var_pc.
// One launch and one dependent pass over HBM less per log-probability batch; needs P <= 32.
struct PartArgs {
    const double* mpart;  // [nchunk][P][Wld]
    const double* spart;  // [nI64][P][Wld]
    const double* amp;    // [P]
    const double* noise;  // [P]
    int nchunk, nI64;
};

template <int MP>
__global__ __launch_bounds__(256) void k_loglike_reg(const double* __restrict__ mean_pc,
                                                     const double* __restrict__ var_pc, int64_t Wld, int64_t W, int P,
                                                     int M, const double* __restrict__ A,
                                                     const double* __restrict__ mu, const double* __restrict__ C0,
                                                     const double* __restrict__ yexp, const double* __restrict__ Cexp,
                                                     double* __restrict__ ll, int accumulate, int* __restrict__ notpd,
                                                     BoxArgs box, PartArgs part) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sC = sm;                         // [64][MP+1]  C_trunc + C_exp, identity padded
    double* sA = sm + 64 * (MP + 1);         // [P][64]     A, zero padded
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 64 * MP; e += 256) {
        const int i = e / MP, k = e % MP;
        double v = (i == k) ? 1.0 : 0.0;
        if (i < M && k < M) v = C0[i * M + k] + Cexp[i * M + k];
        sC[i * (MP + 1) + k] = v;
    }
    for (int e = tid; e < P * 64; e += 256) {
        const int p = e >> 6, i = e & 63;
        sA[e] = (i < M) ? A[p * M + i] : 0.0;
    }
    __syncthreads();
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    if (w >= W) return;                      // wave-uniform
    double a[MP];
#pragma unroll
    for (int k = 0; k < MP; ++k) a[k] = sC[lane * (MP + 1) + k];
    double y = (lane < M) ? (mu[lane] - yexp[lane]) : 0.0;
    double zmv = 0.0;                        // fused finalize: lane p = mean of GP p, lane 32 + p = its variance
    if (part.mpart) {
        const int pl = lane & 31;
        if (pl < P) {
            if (lane < 32) {
                double m = 0.0;
#pragma unroll 8
                for (int c = 0; c < part.nchunk; ++c) m += part.mpart[((int64_t)c * P + pl) * Wld + w];
                zmv = m;
            } else {
                double s = 0.0;
#pragma unroll 8
                for (int i = 0; i < part.nI64; ++i) s += part.spart[((int64_t)i * P + pl) * Wld + w];
                zmv = (part.amp[pl] + part.noise[pl]) - s;
            }
        }
    }
    for (int p = 0; p < P; ++p) {
        // wave-uniform; extra_std == 0 on this path
        const double zm = part.mpart ? readlane_f64(zmv, p) : mean_pc[(int64_t)p * Wld + w];
        const double zv = part.mpart ? readlane_f64(zmv, 32 + p) : var_pc[(int64_t)p * Wld + w];
        const double ai = sA[p * 64 + lane];
        y = fma(zm, ai, y);                                       // dY_i (src/emulator.py:559-561, src/mcmc.py:288)
        const double t = zv * ai;
#pragma unroll
        for (int k = 0; k < MP; ++k) a[k] = fma(t, sA[p * 64 + k], a[k]);   // src/emulator.py:584-587
    }
    bool bad = false;
    double q = 0.0, prod = 1.0;
    double gprod = 1.0;                               // lane g keeps the product of pivot group g (4 pivots each)
#pragma unroll
    for (int j = 0; j < MP; ++j) {
        const double ajj = readlane_f64(a[j], j);
        bad = bad || !(ajj > 0.0);
        const double rinv = rsqrt(ajj);               // 1 / L_jj
        const double lj = a[j] * rinv;                // column j of L for the lanes i > j
        const double vj = readlane_f64(y, j) * rinv;  // forward solve: v_j
        q = fma(vj, vj, q);
        prod *= ajj;                                  // log det: sum log L_jj = 1/2 log prod a_jj, 4 pivots per log
        if ((j & 3) == 3) { gprod = (lane == (j >> 2)) ? prod : gprod; prod = 1.0; }
        y = fma(-lj, vj, y);                          // meaningful for lanes i > j
#pragma unroll
        for (int k = j + 1; k < MP; ++k) {
            const double lkj = readlane_f64(lj, k);
            a[k] = fma(-lj, lkj, a[k]);               // meaningful for lanes i >= k
            if (((k - j) & 7) == 0) __builtin_amdgcn_sched_barrier(0);   // keep the SGPR broadcasts from piling up
        }
    }
    const double glog = log(gprod);                   // the MP/4 logarithms in parallel, one per lane, off the chain
    double logdet = 0.0;
#pragma unroll
    for (int g = 0; g < MP / 4; ++g) logdet = fma(0.5, readlane_f64(glog, g), logdet);   // fixed order
    bool inside = true;
    if (box.X) {                                      // strict box over the d parameters, one per lane
        bool ok = true;
        for (int k0 = 0; k0 < box.d; k0 += 64) {
            const int k = k0 + lane;
            if (k < box.d) {
                const double x = box.X[w * box.d + k];
                ok = ok && (x > box.lo[k]) && (x < box.hi[k]);
            }
        }
        inside = __all(ok);
    }
    if (lane == 0 && (!box.cmp || w < box.cmp[0])) {
        const int64_t wo = box.cmp ? box.cmp[4 + w] : w;
        double r = -0.5 * q - logdet;
        if (bad && inside) {
            r = nan("");
            atomicAdd(notpd, 1);
        }
        r = accumulate ? (ll[wo] + r) : r;
        if (box.X) r = inside ? (r + box.inside_const) : box.outside;
        else if (box.cmp) r += box.inside_const;
        ll[wo] = r;
    }
}

template <int MP>
static int launch_loglike_reg(gpb_ctx* ctx, int64_t W, double* ll_dev, bool accumulate, const BoxArgs& box,
                              const PartArgs& part) {
    const size_t sh = (64 * (MP + 1) + (size_t)ctx->P * 64) * sizeof(double);
    hipLaunchKernelGGL(k_loglike_reg<MP>, dim3((unsigned)((W + 3) / 4)), dim3(256), sh, ctx->stream, ctx->mean_pc,
                       ctx->var_pc, ctx->Wld, W, (int)ctx->P, (int)ctx->M, ctx->A, ctx->mu, ctx->C0, ctx->yexp,
                       ctx->Cexp, ll_dev, accumulate ? 1 : 0, ctx->notpd, box, part);
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ fused block log-likelihood, one workgroup per walker
// Low-latency variant of k_loglike_reg<64> for small walker batches (a rank's shard under 8-way sharding):
// the one-wave kernel is a single dependent instruction stream of ~80 KB of straight-line code (larger than
// the instruction cache) and takes ~50 us however few walkers there are.  Here 256 threads share one
// walker: thread (ty, tx) owns the elements (ty + 16a, tx + 16b), a >= b, of the lower triangle in VGPRs
// (2-D cyclic, so all threads stay busy to the last column); per column the owners publish the column and
// y_j through a double-buffered LDS line, one barrier per column.  Every element sees exactly the same
// sequence of operations as in k_loglike_reg (same build order over p, same rsqrt, same fma operands, same
// grouping of the log-determinant), so the two kernels are bit-identical and the choice by batch size
// never changes a result.
__global__ __launch_bounds__(256) void k_loglike_wg(const double* __restrict__ mean_pc,
                                                    const double* __restrict__ var_pc, int64_t Wld, int64_t W, int P,
                                                    int M, const double* __restrict__ A,
                                                    const double* __restrict__ mu, const double* __restrict__ C0,
                                                    const double* __restrict__ yexp, const double* __restrict__ Cexp,
                                                    double* __restrict__ ll, int accumulate, int* __restrict__ notpd,
                                                    BoxArgs box, PartArgs part) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sA = sm;                         // [P][64]  A, zero padded
    double* col = sm + (size_t)P * 64;       // [2][66]  column j of the trailing matrix (unscaled), y_j at [64]
    double* zl = col + 2 * 66;               // [2P]     fused finalize: mean, variance of GP p at [2p], [2p+1]
    __shared__ int s_outside;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int64_t w = blockIdx.x;
    for (int e = tid; e < P * 64; e += 256) {
        const int p = e >> 6, i = e & 63;
        sA[e] = (i < M) ? A[p * M + i] : 0.0;
    }
    if (part.mpart) {                        // k_finalize's sums for this walker, same order; means on wave 0, variances on wave 1
        const int pl = tid & 63;
        if (tid < 64 && pl < P) {
            double m = 0.0;
#pragma unroll 8
            for (int c = 0; c < part.nchunk; ++c) m += part.mpart[((int64_t)c * P + pl) * Wld + w];
            zl[2 * pl] = m;
        } else if (tid >= 64 && tid < 128 && pl < P) {
            double s = 0.0;
#pragma unroll 8
            for (int i = 0; i < part.nI64; ++i) s += part.spart[((int64_t)i * P + pl) * Wld + w];
            zl[2 * pl + 1] = (part.amp[pl] + part.noise[pl]) - s;
        }
    }
    if (tid == 0) s_outside = 0;
    __syncthreads();
    if (box.X) {                             // strict box over the d parameters
        bool ok = true;
        for (int k = tid; k < box.d; k += 256) {
            const double x = box.X[w * box.d + k];
            ok = ok && (x > box.lo[k]) && (x < box.hi[k]);
        }
        if (!ok) s_outside = 1;              // all writers store the same value
    }
    double c[4][4];                          // c[a][b], b <= a: element (ty + 16a, tx + 16b)
    double y[4];                             // dY rows ty + 16a (meaningful on the tx == 0 threads)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = ty + 16 * a;
        y[a] = (i < M) ? (mu[i] - yexp[i]) : 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int k = tx + 16 * b;
            double v = (i == k) ? 1.0 : 0.0;
            if (i < M && k < M) v = C0[i * M + k] + Cexp[i * M + k];
            c[a][b] = v;
        }
    }
    for (int p = 0; p < P; ++p) {
        // uniform; extra_std == 0 on this path
        const double zm = part.mpart ? zl[2 * p] : mean_pc[(int64_t)p * Wld + w];
        const double zv = part.mpart ? zl[2 * p + 1] : var_pc[(int64_t)p * Wld + w];
        double ak[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) ak[b] = sA[p * 64 + tx + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double ai = sA[p * 64 + ty + 16 * a];
            y[a] = fma(zm, ai, y[a]);                             // dY_i (src/emulator.py:559-561, src/mcmc.py:288)
            const double t = zv * ai;
#pragma unroll
            for (int b = 0; b <= a; ++b) c[a][b] = fma(t, ak[b], c[a][b]);   // src/emulator.py:584-587
        }
    }
    bool bad = false;
    double q = 0.0, prod = 1.0;
    double gprod = 1.0;                      // thread g keeps the product of pivot group g (4 pivots each)
    const int nblk = (M + 15) >> 4;          // identity-padded columns beyond M change nothing (pivot 1, v = 0)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (jb < nblk) {
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 16 * jb + jj;
                double* buf = col + (j & 1) * 66;
                if (tx == jj) {
#pragma unroll
                    for (int a = jb; a < 4; ++a) buf[ty + 16 * a] = c[a][jb];
                }
                if (tx == 0 && ty == jj) buf[64] = y[jb];
                __syncthreads();             // one barrier per column: the line of column j-1 is not rewritten before j+1
                const double ajj = buf[j];
                bad = bad || !(ajj > 0.0);
                const double rinv = rsqrt(ajj);                   // 1 / L_jj
                const double vj = buf[64] * rinv;                 // forward solve: v_j
                q = fma(vj, vj, q);
                prod *= ajj;                                      // sum log L_jj = 1/2 log prod a_jj, 4 pivots per log
                if ((j & 3) == 3) { gprod = (tid == (j >> 2)) ? prod : gprod; prod = 1.0; }
                double lk[4];
#pragma unroll
                for (int b = jb; b < 4; ++b) lk[b] = buf[tx + 16 * b] * rinv;
#pragma unroll
                for (int a = jb; a < 4; ++a) {
                    const double li = buf[ty + 16 * a] * rinv;    // L_ij for this thread's rows
                    y[a] = fma(-li, vj, y[a]);
#pragma unroll
                    for (int b = jb; b <= a; ++b) c[a][b] = fma(-li, lk[b], c[a][b]);
                }
            }
        }
    }
    __syncthreads();                         // the column lines are free: reuse them for the 16 group logarithms
    if (tid < 16) col[tid] = log(gprod);     // in parallel, off the factorisation's dependency chain
    __syncthreads();
    if (tid == 0) {
        double logdet = 0.0;
        for (int g = 0; g < 16; ++g) logdet = fma(0.5, col[g], logdet);      // fixed order (as k_loglike_reg)
        const bool inside = !s_outside;
        if (!box.cmp || w < box.cmp[0]) {
            const int64_t wo = box.cmp ? box.cmp[4 + w] : w;
            double r = -0.5 * q - logdet;
            if (bad && inside) {
                r = nan("");
                atomicAdd(notpd, 1);
            }
            r = accumulate ? (ll[wo] + r) : r;
            if (box.X) r = inside ? (r + box.inside_const) : box.outside;
            else if (box.cmp) r += box.inside_const;
            ll[wo] = r;
        }
    }
}

static int launch_loglike_wg(gpb_ctx* ctx, int64_t W, double* ll_dev, bool accumulate, const BoxArgs& box,
                             const PartArgs& part) {
    const size_t sh = ((size_t)ctx->P * 64 + 2 * 66 + 2 * (size_t)ctx->P) * sizeof(double);
    hipLaunchKernelGGL(k_loglike_wg, dim3((unsigned)W), dim3(256), sh, ctx->stream, ctx->mean_pc, ctx->var_pc,
                       ctx->Wld, W, (int)ctx->P, (int)ctx->M, ctx->A, ctx->mu, ctx->C0, ctx->yexp, ctx->Cexp, ll_dev,
                       accumulate ? 1 : 0, ctx->notpd, box, part);
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ low-rank block log-likelihood, one lane per walker
// C_w = C0 + A^T diag(var_w) A with the same C0 for every walker (gpb_api.hip: lowrank_setup has the algebra):
//     loglike_w = -1/2 (c_perp + v^T S^-1 v) - 1/2 (log det C0 + log det S),   S = I + R diag(var_w) R^T,  v = R m_w + v0
// a PP x PP Cholesky in one lane's registers instead of the M x M one (M = 64: 90 k flops and a 64-column
// dependency chain per walker; here ~1 k flops).  R is upper triangular, so row i of R touches p >= i only.
// Same conventions as the dense kernels: a non-positive pivot inside the box gives NaN and counts in notpd; the
// fused k_finalize sums (part) use k_finalize's order.
// k_finalize's sums for the 64 walkers of a workgroup, in its order (same bits): sum (GP p, mean or variance) is one chain of
// additions over the row chunks; the 2 P sums are dealt to the workgroup's eight waves, each with up to 32 partials in flight
// per round trip to L2.  (Four waves with 16 + 16 in flight walked three GPs x two round trips each: most of the kernel's
// 11 us at 256 walkers; with 8 of one kind a 64-walker workgroup spent 20 us waiting.)
// (13-16 GPs: the per-walker algebra needs more than the 256 registers an eight-wave workgroup leaves a wave: four waves)
template <int PP> constexpr int lr_threads() { return PP > 12 ? 256 : 512; }
template <int PP>
__device__ __forceinline__ void partial_sums(double (*smg)[PP][64], const double* __restrict__ mpart,
                                             const double* __restrict__ spart, const double* __restrict__ amp,
                                             const double* __restrict__ noise, int P, int nchunk, int nI64, int64_t Wld,
                                             int64_t w, int lane, int grp) {
    const int64_t st = (int64_t)P * Wld;
    for (int u = grp; u < 2 * P; u += lr_threads<PP>() / 64) {
        const int p = u >> 1, kind = u & 1;
        const double* src = (kind ? spart : mpart) + (int64_t)p * Wld + w;
        const int n = kind ? nI64 : nchunk;
        double a = 0.0;
        int c = 0;
        for (; c + 32 <= n; c += 32) {
            double v[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) v[t] = src[(c + t) * st];
#pragma unroll
            for (int t = 0; t < 32; ++t) a += v[t];
        }
        for (; c + 8 <= n; c += 8) {
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = src[(c + t) * st];
#pragma unroll
            for (int t = 0; t < 8; ++t) a += v[t];
        }
        for (; c < n; ++c) a += src[c * st];
        smg[kind][p][lane] = kind ? (amp[p] + noise[p]) - a : a;
    }
}

template <int PP>
__global__ __launch_bounds__(lr_threads<PP>()) void k_loglike_lowrank(const double* __restrict__ mean_pc,
                                                        const double* __restrict__ var_pc, int64_t Wld, int64_t W,
                                                        int P, const double* __restrict__ Rg,
                                                        const double* __restrict__ v0g, double cperp, double logdet0,
                                                        double* __restrict__ ll, int accumulate,
                                                        int* __restrict__ notpd, BoxArgs box, PartArgs part) {
    // 256 threads serve 64 walkers: the four waves share the fused k_finalize sums (wave q: GPs q, q+4, ...; the
    // loads are coalesced along the walkers), wave 0 then does the per-walker algebra.
    __shared__ double sR[PP][PP + 1];
    __shared__ double sv0[PP];
    __shared__ double smg[2][PP][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < PP * PP; e += lr_threads<PP>()) sR[e / PP][e % PP] = Rg[(e / PP) * 16 + (e % PP)];
    if (threadIdx.x < PP) sv0[threadIdx.x] = v0g[threadIdx.x];
    const int64_t w = (int64_t)blockIdx.x * 64 + lane;
    if (part.mpart && w < W && (!box.cmp || w < box.cmp[0]))
        partial_sums<PP>(smg, part.mpart, part.spart, part.amp, part.noise, P, part.nchunk, part.nI64, Wld, w, lane, grp);
    __syncthreads();
    if (grp != 0 || w >= W || (box.cmp && w >= box.cmp[0])) return;
    const int64_t wo = box.cmp ? box.cmp[4 + w] : w;
    double m[PP], g[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        m[p] = 0.0; g[p] = 0.0;
        if (p < P) {
            if (part.mpart) {
                m[p] = smg[0][p][lane];
                g[p] = smg[1][p][lane];
            } else {
                m[p] = mean_pc[(int64_t)p * Wld + w];
                g[p] = var_pc[(int64_t)p * Wld + w];
            }
        }
    }
    bool inside = true;
    if (box.X) {
        for (int k = 0; k < box.d; ++k) {
            const double x = box.X[w * box.d + k];
            inside = inside && (x > box.lo[k]) && (x < box.hi[k]);
        }
    }
    // S (lower triangle, registers) and v
    double S[PP][PP], v[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {
        double vi = sv0[i];
#pragma unroll
        for (int p = i; p < PP; ++p) vi = fma(sR[i][p], m[p], vi);
        v[i] = vi;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double sij = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int p = i; p < PP; ++p) sij = fma(sR[i][p] * g[p], sR[j][p], sij);
            S[i][j] = sij;
        }
    }
    // Cholesky + forward solve, column by column
    // log det S = sum of log pivots, four pivots per logarithm like the dense kernels (S >= I, so the pivots are >= 1
    // and a product cannot underflow; four of them overflow only beyond 1e77 each, i.e. never for a C0 that factorises
    // in fp64, whereas all PP <= 16 in one product overflowed from 1e19 per pivot: C0 ~ 1e-20 * I)
    double q = 0.0, prod = 1.0, logsum = 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
        const double ajj = S[j][j];
        bad = bad || !(ajj > 0.0);
        const double rinv = rsqrt(ajj);
        const double zj = v[j] * rinv;
        q = fma(zj, zj, q);
        prod *= ajj;
        if ((j & 3) == 3 || j == PP - 1) { logsum += log(prod); prod = 1.0; }
#pragma unroll
        for (int i = j + 1; i < PP; ++i) {
            const double lij = S[i][j] * rinv;
            v[i] = fma(-lij, zj, v[i]);
#pragma unroll
            for (int k = j + 1; k <= i; ++k) S[i][k] = fma(-lij, S[k][j] * rinv, S[i][k]);
        }
    }
    double r = -0.5 * (cperp + q) - 0.5 * (logdet0 + logsum);
    bad = bad || !(logsum < INFINITY);       // an overflowing pivot product is a failure, not a silent -inf
    if (bad && inside) {
        r = nan("");
        atomicAdd(notpd, 1);
    }
    r = accumulate ? (ll[wo] + r) : r;
    if (box.X) r = inside ? (r + box.inside_const) : box.outside;
    else if (box.cmp) r += box.inside_const;
    ll[wo] = r;
}

static bool lowrank_applies(const gpb_ctx* ctx) {
    return ctx->lowrank && ctx->lr_ok && ctx->mode == GPB_MODE_PCA && ctx->P <= 16 && !ctx->force_generic_mvn;
}

static int launch_loglike_lowrank(gpb_ctx* ctx, int64_t W, double* ll_dev, bool accumulate, const BoxArgs& box,
                                  const PartArgs& part) {
    const dim3 grid((unsigned)((W + 63) / 64));
#define GPB_LR(PPV)                                                                                              \
    hipLaunchKernelGGL(k_loglike_lowrank<PPV>, grid, dim3(lr_threads<PPV>()), 0, ctx->stream, ctx->mean_pc, ctx->var_pc, ctx->Wld, \
                       W, (int)ctx->P, ctx->lr_R, ctx->lr_v0, ctx->lr_cperp, ctx->lr_logdet0, ll_dev,              \
                       accumulate ? 1 : 0, ctx->notpd, box, part)
    switch (ctx->P) {                        // exact sizes: the work per walker grows with PP^3
        case 1: GPB_LR(1); break;   case 2: GPB_LR(2); break;   case 3: GPB_LR(3); break;   case 4: GPB_LR(4); break;
        case 5: GPB_LR(5); break;   case 6: GPB_LR(6); break;   case 7: GPB_LR(7); break;   case 8: GPB_LR(8); break;
        case 9: GPB_LR(9); break;   case 10: GPB_LR(10); break; case 11: GPB_LR(11); break; case 12: GPB_LR(12); break;
        case 13: GPB_LR(13); break; case 14: GPB_LR(14); break; case 15: GPB_LR(15); break; default: GPB_LR(16); break;
    }
#undef GPB_LR
    GPB_HIP(hipGetLastError());
    return 0;
}

// ---- the low-rank block log-likelihoods of ALL emulators of a chain in one launch (round 3) --------------------------------
// gpb_chain_logpost / gpb_chain_emcee_run end a batch with one k_loglike_lowrank per emulator, each adding its block onto the
// row's log-probability: 32 workgroups (2048 rows) of latency-bound work per launch, nine launches in a row for the
// reference's nine emulators.  Here a workgroup walks the emulators itself, in emuList order — the same sequence of
// additions per row as the separate launches.  PP = the largest number of GPs of any emulator of the chain (rounded up to a
// multiple of 4); an emulator with fewer runs with identity padding: its R and v0 are zero beyond P (lr_R / lr_v0 are zero
// padded), so the padded rows of S are unit rows, their pivots 1, their v 0 — every product, sum and logarithm of the exact-size
// kernel is reproduced bit for bit (x * 1 = x, x + 0 = x, log 1 = 0; the groups of four pivots per logarithm start at the same
// places).  Compacted batches only (the chain calls): rows w >= cmp[0] do not exist.
constexpr int MAX_LR_CTX = 24;
struct LrCtx {
    const double *mpart, *spart, *amp, *noise, *R, *v0;
    int* notpd;
    double cperp, logdet0;
    int P, nchunk, nI64;
};
struct LrTable { LrCtx c[MAX_LR_CTX]; int E; };

// Round 5: with `blocks` set the grid is (walker tiles, emulators): workgroup (t, e) computes emulator e's block alone and leaves
// it in blocks[e][w]; k_lowrank_sum then adds a row's blocks in emuList order — the same additions as the walk below, which ran
// nine latency-bound bodies one after the other in each of only W / 64 workgroups (57 us for nine emulators at 2048 rows).
template <int PP>
__global__ __launch_bounds__(lr_threads<PP>()) void k_loglike_lowrank_multi(const LrTable tab, int64_t Wld, int64_t W,
                                                              double* __restrict__ ll, const int* __restrict__ cmp,
                                                              double inside_const, double* __restrict__ blocks) {
    __shared__ double sR[PP][PP + 1];
    __shared__ double sv0[PP];
    __shared__ double smg[2][PP][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 64 + lane;
    const bool live = w < W && w < cmp[0];
    double total = 0.0;
    const int e_begin = blocks ? (int)blockIdx.y : 0, e_end = blocks ? (int)blockIdx.y + 1 : tab.E;
    for (int e = e_begin; e < e_end; ++e) {
        const LrCtx& c = tab.c[e];
        const int P = c.P;
        if (e > e_begin) __syncthreads();              // wave 0 is done with the previous emulator's tables
        for (int i = threadIdx.x; i < PP * PP; i += lr_threads<PP>()) sR[i / PP][i % PP] = c.R[(i / PP) * 16 + (i % PP)];
        if (threadIdx.x < PP) sv0[threadIdx.x] = c.v0[threadIdx.x];
        if (live)                                      // k_finalize's sums, in its order (as k_loglike_lowrank)
            partial_sums<PP>(smg, c.mpart, c.spart, c.amp, c.noise, P, c.nchunk, c.nI64, Wld, w, lane, grp);
        __syncthreads();
        if (grp != 0 || !live) continue;               // (uniform per wave; every wave still reaches the barriers above)
        double m[PP], g[PP];
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            m[p] = 0.0; g[p] = 0.0;
            if (p < P) { m[p] = smg[0][p][lane]; g[p] = smg[1][p][lane]; }
        }
        double S[PP][PP], v[PP];
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            double vi = sv0[i];
#pragma unroll
            for (int p = i; p < PP; ++p) vi = fma(sR[i][p], m[p], vi);
            v[i] = vi;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double sij = (i == j) ? 1.0 : 0.0;
#pragma unroll
                for (int p = i; p < PP; ++p) sij = fma(sR[i][p] * g[p], sR[j][p], sij);
                S[i][j] = sij;
            }
        }
        double q = 0.0, prod = 1.0, logsum = 0.0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const double ajj = S[j][j];
            bad = bad || !(ajj > 0.0);
            const double rinv = rsqrt(ajj);
            const double zj = v[j] * rinv;
            q = fma(zj, zj, q);
            prod *= ajj;
            if ((j & 3) == 3 || j == PP - 1) { logsum += log(prod); prod = 1.0; }
#pragma unroll
            for (int i = j + 1; i < PP; ++i) {
                const double lij = S[i][j] * rinv;
                v[i] = fma(-lij, zj, v[i]);
#pragma unroll
                for (int k = j + 1; k <= i; ++k) S[i][k] = fma(-lij, S[k][j] * rinv, S[i][k]);
            }
        }
        double r = -0.5 * (c.cperp + q) - 0.5 * (c.logdet0 + logsum);
        bad = bad || !(logsum < INFINITY);
        if (bad) {
            r = nan("");
            atomicAdd(c.notpd, 1);
        }
        if (blocks) { blocks[(int64_t)e * Wld + w] = r; return; }      // (grp 0, live: the ordered sum follows in k_lowrank_sum)
        total = e ? (total + r) : r;                   // ll = r0; ll = ll + r1; ... as the separate launches accumulate
    }
    if (grp == 0 && live && !blocks) ll[cmp[4 + w]] = total + inside_const;
}

// a row's log-likelihood = its emulators' blocks added in emuList order (ll = r0; ll = ll + r1; ...), + the constant
__global__ __launch_bounds__(256) void k_lowrank_sum(const double* __restrict__ blocks, int E, int64_t Wld, int64_t W,
                                                     double* __restrict__ ll, const int* __restrict__ cmp, double inside_const) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= W || w >= cmp[0]) return;
    double total = blocks[w];
    for (int e = 1; e < E; ++e) total = total + blocks[(int64_t)e * Wld + w];
    ll[cmp[4 + w]] = total + inside_const;
}

__global__ void k_box(const double* __restrict__ X, int64_t W, int d, const double* __restrict__ lo,
                      const double* __restrict__ hi, double outside, double inside_const, double* __restrict__ ll);

// box_* optional (X_box == nullptr: no prior box).  The register-resident kernel applies the box itself;
// the generic kernels are followed by k_box.
static bool block_kernels_apply(const gpb_ctx* ctx) {
    return ctx->mode == GPB_MODE_PCA && ctx->M <= 64 && ctx->P <= 96 && !ctx->force_generic_mvn;
}

// Small batches only: there the saved launch and dependent pass matter (a rank's shard), while at thousands of
// walkers the per-walker strided reads of the partials cost more than the coalesced k_finalize they replace
// (measured: +25 us on k_loglike_reg<64> at 2048 walkers against a 6.6 us kernel).
bool loglike_fuses_finalize(const gpb_ctx* ctx, int64_t W) {
    if (lowrank_applies(ctx)) return ctx->fuse_finalize != 0;     // the kernel's four waves share k_finalize's sums
    return block_kernels_apply(ctx) && ctx->P <= 32 && ctx->fuse_finalize && W <= ctx->mvn_wg_switch;
}

int launch_loglike(gpb_ctx* ctx, int64_t W, double* ll_dev, bool accumulate, bool from_partials, const double* X_box,
                   const double* lo_dev, const double* hi_dev, double outside, double inside_const, const int* cmp_dev) {
    const int64_t M = ctx->M, P = ctx->P;
    const BoxArgs box{X_box, lo_dev, hi_dev, (int)ctx->d, outside, inside_const, cmp_dev};
    if (cmp_dev && !(lowrank_applies(ctx) || block_kernels_apply(ctx)))
        GPB_FAIL(GPB_E_STATE, "gpb: internal: compacted batch without a block likelihood kernel");
    if (from_partials && !loglike_fuses_finalize(ctx, W)) GPB_FAIL(GPB_E_STATE, "gpb: internal: partials without a fused consumer");
    if (lowrank_applies(ctx) || block_kernels_apply(ctx)) {
        PartArgs part{nullptr, nullptr, nullptr, nullptr, 0, 0};
        if (from_partials)
            part = PartArgs{ctx->mpart, ctx->spart, ctx->amp, ctx->noise,
                            (int)((ctx->Np + KX_CHUNK - 1) / KX_CHUNK), (int)(ctx->Np / 64)};
        if (lowrank_applies(ctx)) return launch_loglike_lowrank(ctx, W, ll_dev, accumulate, box, part);
        if (M <= 8) return launch_loglike_reg<8>(ctx, W, ll_dev, accumulate, box, part);
        if (M <= 16) return launch_loglike_reg<16>(ctx, W, ll_dev, accumulate, box, part);
        if (M <= 32) return launch_loglike_reg<32>(ctx, W, ll_dev, accumulate, box, part);
        if (W <= ctx->mvn_wg_switch) return launch_loglike_wg(ctx, W, ll_dev, accumulate, box, part);   // same bits, lower latency
        return launch_loglike_reg<64>(ctx, W, ll_dev, accumulate, box, part);
    }
    const size_t small = (2 * P + 2 * M) * sizeof(double);
    const size_t mat = (size_t)(M + 1) * (M + 1) * sizeof(double);
    double* gws = nullptr;
    size_t sh = small + mat;
    if (sh > 150 * 1024) {                       // M > ~135: slab in HBM/L2 instead of LDS
        const int64_t need = W * (M + 1) * (M + 1);
        if (need > ctx->mvn_ws_cap) {
            GPB_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->mvn_ws) pool_free(ctx->mvn_ws);
            GPB_HIP(pool_malloc_t(&ctx->mvn_ws, need * sizeof(double)));
            ctx->mvn_ws_cap = need;
        }
        gws = ctx->mvn_ws;
        sh = small;
    }
    if (sh > 64 * 1024) {
        GPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_loglike<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    }
    if (gws)
        hipLaunchKernelGGL(k_loglike<true>, dim3((unsigned)W), dim3(256), sh, ctx->stream, ctx->mean_pc, ctx->var_pc,
                           ctx->Wld, (int)P, (int)M, ctx->mode, ctx->A, ctx->mu, ctx->scale, ctx->C0, ctx->yexp,
                           ctx->Cexp, gws, ll_dev, accumulate ? 1 : 0, ctx->notpd);
    else
        hipLaunchKernelGGL(k_loglike<false>, dim3((unsigned)W), dim3(256), sh, ctx->stream, ctx->mean_pc, ctx->var_pc,
                           ctx->Wld, (int)P, (int)M, ctx->mode, ctx->A, ctx->mu, ctx->scale, ctx->C0, ctx->yexp,
                           ctx->Cexp, gws, ll_dev, accumulate ? 1 : 0, ctx->notpd);
    if (X_box)
        hipLaunchKernelGGL(k_box, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, ctx->stream, X_box, W, (int)ctx->d,
                           lo_dev, hi_dev, outside, inside_const, ll_dev);
    GPB_HIP(hipGetLastError());
    return 0;
}

int launch_mvn(gpb_ctx* ctx, const double* dY_dev, const double* cov_dev, int64_t W, int64_t M, double* ll_dev) {
    const size_t mat = (size_t)(M + 1) * (M + 1) * sizeof(double);
    size_t sh = M * sizeof(double) + mat;
    double* gws = nullptr;
    if (sh > 150 * 1024) {
        const int64_t need = W * (M + 1) * (M + 1);
        if (need > ctx->mvn_ws_cap) {
            GPB_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->mvn_ws) pool_free(ctx->mvn_ws);
            GPB_HIP(pool_malloc_t(&ctx->mvn_ws, need * sizeof(double)));
            ctx->mvn_ws_cap = need;
        }
        gws = ctx->mvn_ws;
        sh = M * sizeof(double);
    }
    if (sh > 64 * 1024) {
        GPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mvn<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    }
    if (gws)
        hipLaunchKernelGGL(k_mvn<true>, dim3((unsigned)W), dim3(256), sh, ctx->stream, dY_dev, cov_dev, (int)M, gws,
                           ll_dev, ctx->notpd);
    else
        hipLaunchKernelGGL(k_mvn<false>, dim3((unsigned)W), dim3(256), sh, ctx->stream, dY_dev, cov_dev, (int)M, gws,
                           ll_dev, ctx->notpd);
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ compaction to the rows inside the prior box
// The reference evaluates the emulators for the rows inside the box only (src/mcmc.py:194-203, 275-283) and an
// all-outside batch costs it nothing (:278-279).  Here: ONE workgroup marks the rows (strict inequalities), writes
// `outside` for those outside, and gathers the others — in order — into Xc with their indices and count in cmp
// ([0] = count, [4..] = indices).  The count stays on the device: the kernels that follow are launched for the whole
// batch and those of their workgroups that find no row leave at once.  From uniform starting positions more than half of
// a stretch move's proposals (z > 1) leave a 20-dimensional box; a burnt-in ensemble hardly ever does.
// Pass 1: 256 rows per workgroup, staged through LDS with coalesced loads (a lane reading its own row touches 64
// different cache lines per instruction).  rank[w] = position of row w among the live rows of its workgroup (-1: outside),
// blockcnt[b] = live rows of workgroup b.
__global__ __launch_bounds__(256) void k_compact_mark(const double* __restrict__ X, int64_t W, int d,
                                                      const double* __restrict__ lo, const double* __restrict__ hi,
                                                      double outside, double* __restrict__ ll, int* __restrict__ rank,
                                                      int* __restrict__ blockcnt) {
    extern __shared__ double srow[];                   // [256][dt + 1], dt = min(d, 64): the columns go through in tiles
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dt = d < 64 ? d : 64, ldr = dt + 1;
    const int64_t w0 = (int64_t)blockIdx.x * 256, nrow = imin64(256, W - w0);
    const bool have = tid < nrow;
    int ok = have ? 1 : 0;
    for (int k0 = 0; k0 < d; k0 += dt) {
        const int kn = (d - k0 < dt) ? d - k0 : dt;
        if (k0) __syncthreads();
        for (int64_t e = tid; e < nrow * kn; e += 256) srow[(e / kn) * ldr + (e % kn)] = X[(w0 + e / kn) * d + k0 + (e % kn)];
        __syncthreads();
        if (have)
            for (int k = 0; k < kn; ++k) {
                const double x = srow[tid * ldr + k];
                ok &= (int)(x > lo[k0 + k]) & (int)(x < hi[k0 + k]);      // strict (src/mcmc.py:275); no short circuit
            }
    }
    const bool in = ok != 0;
    if (have && !in) ll[w0 + tid] = outside;
    const unsigned long long m = __ballot(in);
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int i = 0; i < wave; ++i) off += wsum[i];
    if (have) rank[w0 + tid] = in ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    if (tid == 0) blockcnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Pass 2: the live rows of workgroup b go to slots base_b + rank, base_b = live rows of the workgroups before it (the
// order of the rows is kept); cmp[0] = total, cmp[4 + slot] = row.
__global__ __launch_bounds__(256) void k_compact_gather(const double* __restrict__ X, int64_t W, int d,
                                                        const int* __restrict__ rank, const int* __restrict__ blockcnt,
                                                        double* __restrict__ Xc, int* __restrict__ cmp,
                                                        unsigned long long* __restrict__ rows_live,
                                                        unsigned long long* __restrict__ hint) {
    __shared__ int s_base, s_cnt;
    __shared__ int row_of[256];
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * 256, nrow = imin64(256, W - w0);
    int r;
    if (blockcnt) {
        if (tid < 64) {                                // fixed-order sum of the counts before this workgroup
            int s = 0;
            for (int b = lane; b < (int)blockIdx.x; b += 64) s += blockcnt[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) { s_base = s; s_cnt = blockcnt[blockIdx.x]; }
        }
        r = tid < nrow ? rank[w0 + tid] : -1;
    } else {
        // rank[] holds 0/1 flags (written by k_propose; batches of a few thousand rows): the live rows before this
        // workgroup are counted from the flags themselves (integers: any order), the rank inside it by ballots
        int s = 0;
        for (int64_t w = tid; w < w0; w += 256) s += rank[w];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const bool in = tid < nrow && rank[w0 + tid] != 0;
        const unsigned long long m = __ballot(in);
        __syncthreads();                               // (wsum is free: first use)
        if (lane == 0) { wsum[wave] = __popcll(m); row_of[wave] = s; }     // row_of[0..3] borrowed for the partial sums
        __syncthreads();
        int off = 0;
        for (int i = 0; i < wave; ++i) off += wsum[i];
        r = in ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        const int b4 = row_of[0] + row_of[1] + row_of[2] + row_of[3], c4 = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();                               // everyone has read row_of[0..3] before it is reused below
        if (tid == 0) { s_base = b4; s_cnt = c4; }
    }
    if (r >= 0) row_of[r] = tid;
    __syncthreads();
    const int base = s_base, cnt = s_cnt;
    if (r >= 0) cmp[4 + base + r] = (int)(w0 + tid);
    for (int64_t e = tid; e < (int64_t)cnt * d; e += 256) {
        const int s = (int)(e / d), k = (int)(e % d);
        Xc[((int64_t)base + s) * d + k] = X[(w0 + row_of[s]) * d + k];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        cmp[0] = base + cnt;
        if (hint) __hip_atomic_store(hint, ((unsigned long long)W << 32) | (unsigned long long)(base + cnt), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM);        // for the host's tile-shape rule, read without a sync
        if (rows_live) atomicAdd(rows_live, (unsigned long long)(base + cnt));
    }
}

// X_dev [W][dx]: dx = the chain's number of parameters (the GP's d unless the emulator has a parameter map).  The
// gathered rows land in ctx->cmp_X (grown on demand), indices and count in ctx->cmp_idx.
int launch_compact(gpb_ctx* ctx, const double* X_dev, int64_t W, int64_t dx, const double* lo_dev, const double* hi_dev,
                   double outside, double* ll_dev, int premarked) {
    if (W > ctx->Wcap || W >= (1ll << 31)) GPB_FAIL(GPB_E_STATE, "gpb: internal: compaction beyond the workspace");
    int rc = ensure_cmp_rows(ctx, dx);
    if (rc) return rc;
    ctx->hint_from = ctx;
    if (premarked == 2) return 0;                      // k_propose has gathered the rows as well
    const unsigned nb = (unsigned)((W + 255) / 256);
    int* rank = ctx->cmp_idx + 4 + ctx->Wcap;          // [Wcap] ranks, then [Wcap / 256 + 1] workgroup counts
    int* blockcnt = rank + ctx->Wcap;
    const size_t sh = sizeof(double) * 256 * (size_t)((dx < 64 ? dx : 64) + 1);
    if (sh > 64 * 1024 && !premarked)
        GPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_compact_mark), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    // premarked: k_propose has left 0/1 flags in rank[] and `outside` in ll (see there)
    if (!premarked)
        hipLaunchKernelGGL(k_compact_mark, dim3(nb), dim3(256), sh, ctx->stream, X_dev, W, (int)dx, lo_dev, hi_dev, outside,
                           ll_dev, rank, blockcnt);
    hipLaunchKernelGGL(k_compact_gather, dim3(nb), dim3(256), 0, ctx->stream, X_dev, W, (int)dx, rank,
                       premarked ? nullptr : blockcnt, ctx->cmp_X,
                       ctx->cmp_idx, ctx->profile ? ctx->rows_live : nullptr, ctx->live_hint);
    GPB_HIP(hipGetLastError());
    return 0;
}

// the buffer of gathered rows [Wcap][dx], grown on demand
int ensure_cmp_rows(gpb_ctx* ctx, int64_t dx) {
    if (ctx->cmp_X_cap < ctx->Wcap * dx) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->cmp_X) pool_free(ctx->cmp_X);
        ctx->cmp_X = nullptr;
        GPB_HIP(pool_malloc_t(&ctx->cmp_X, sizeof(double) * (size_t)(ctx->Wcap * dx)));
        GPB_HIP(hipMemsetAsync(ctx->cmp_X, 0, sizeof(double) * (size_t)(ctx->Wcap * dx), ctx->stream));   // rows past the count are read (not used) by the upper-bound launches
        ctx->cmp_X_cap = ctx->Wcap * dx;
    }
    return 0;
}

// true when gpb_logpost / gpb_emcee_run may evaluate the rows inside the box only (a block likelihood kernel follows)
bool compaction_applies(const gpb_ctx* ctx) {
    return ctx->compact && (lowrank_applies(ctx) || block_kernels_apply(ctx));
}

// ------------------------------------------------------------------ prior box
__global__ void k_box(const double* __restrict__ X, int64_t W, int d, const double* __restrict__ lo,
                      const double* __restrict__ hi, double outside, double inside_const,
                      double* __restrict__ ll) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= W) return;
    bool in = true;
    for (int k = 0; k < d; ++k) {
        const double x = X[w * d + k];
        in = in && (x > lo[k]) && (x < hi[k]);      // strict (src/mcmc.py:275)
    }
    ll[w] = in ? (ll[w] + inside_const) : outside;
}

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    U4 c = {c0, c1, c2, c3};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {   // 53-bit uniform in [0,1)
    const uint64_t b = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)b * (1.0 / 9007199254740992.0);
}

// ---- red/blue split ----------------------------------------------------------------------------------
// emcee's RedBlueMove shuffles which walkers form the two halves at every step (randomize_split=True,
// its default).  Here the shuffle is a keyed pseudo-random permutation pi_step of [0, n) that every
// thread (and every rank) can evaluate for a single index without communication or sorting: a 4-round
// Feistel network on 2*hb >= log2(n) bits with cycle walking.  Walker k of half h is pi(2k + h); with
// randomize = 0, pi is the identity (emcee's inds = arange(n) % 2).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct SplitPerm {
    uint32_t n, hb, k0, k1, on;
    __device__ __forceinline__ int64_t operator()(int64_t i) const {
        if (!on) return i;
        const uint32_t mask = (1u << hb) - 1u;
        uint32_t x = (uint32_t)i;
        do {
            uint32_t L = x >> hb, R = x & mask;
#pragma unroll
            for (uint32_t r = 0; r < 4; ++r) {
                const uint32_t F = mix32(R ^ (k0 + r * 0x9E3779B9u)) ^ mix32(k1 + r);
                const uint32_t t = L ^ (F & mask);
                L = R;
                R = t;
            }
            x = (L << hb) | R;
        } while (x >= n);                      // cycle walking keeps it a bijection on [0, n)
        return (int64_t)x;
    }
    // the inverse map: rounds backwards (a round takes (L, R) to (R, L ^ F(R))), cycle walking likewise
    __device__ __forceinline__ int64_t inv(int64_t i) const {
        if (!on) return i;
        const uint32_t mask = (1u << hb) - 1u;
        uint32_t x = (uint32_t)i;
        do {
            uint32_t L = x >> hb, R = x & mask;
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const uint32_t F = mix32(L ^ (k0 + (uint32_t)r * 0x9E3779B9u)) ^ mix32(k1 + (uint32_t)r);
                const uint32_t t = R ^ (F & mask);
                R = L;
                L = t;
            }
            x = (L << hb) | R;
        } while (x >= n);
        return (int64_t)x;
    }
};
__device__ __forceinline__ SplitPerm make_perm(uint64_t seed, uint32_t step, int64_t n, int hb, int randomize) {
    const U4 k = philox(seed, 0xFFFFFFFFu, step, 0u, 7u);
    return SplitPerm{(uint32_t)n, (uint32_t)hb, k.x, k.y, (uint32_t)randomize};
}

// A slot of the compacted batch for every walker of the workgroup that asks for one (`want`, set in the walker's lane t0 = 0):
// ONE atomic per workgroup — 2048 walkers taking their slots from one counter one by one serialised on the atomic's return
// (k_accept_propose 18 us at 2048 rows a batch against 8 at 256).  The order of the slots does not matter (a row's result
// does not depend on its place in the batch).  Barriers: only in workgroups whose every thread has a walker (`full`, uniform
// per workgroup); the ensemble's last, partly filled workgroup takes the slots walker by walker.  Returns the slot in the
// lane t0 = 0 that asked (-1 elsewhere).
__device__ __forceinline__ int take_slot(bool want, bool full, int* __restrict__ cmp) {
    __shared__ int s_want[32], s_base;                 // up to 1024 threads = 32 walkers per workgroup
    if (!full) return want ? atomicAdd(cmp, 1) : -1;
    const int wl = (int)(threadIdx.x >> 5), nw = (int)(blockDim.x >> 5);
    if ((threadIdx.x & 31) == 0) s_want[wl] = want ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int i = 0; i < nw; ++i) tot += s_want[i];
        s_base = tot ? atomicAdd(cmp, tot) : 0;
    }
    __syncthreads();
    if (!want) return -1;
    int off = 0;
    for (int i = 0; i < wl; ++i) off += s_want[i];
    return s_base + off;
}

// lo != nullptr (the C-driven loop over a compacted chain): the prior-box test of the rows [r0, r0 + chunk) — this
// rank's rows of the batch — is taken here, on the proposal still in registers: flags[k - r0] = 1 inside / 0 outside
// (strict inequalities, src/mcmc.py:275) and ll[k] = outside for the rows outside; k_compact_gather then ranks the
// flags itself and k_compact_mark's launch is saved (7.6 of a sharded half-step's 166 us).
__global__ void k_propose(const double* __restrict__ pos, int64_t nhalf, int d, int half, uint64_t seed,
                          uint32_t step, double a, double* __restrict__ q, double* __restrict__ factor, int hb,
                          int randomize, const double* __restrict__ lo = nullptr, const double* __restrict__ hi = nullptr,
                          double outside = 0.0, double* __restrict__ ll = nullptr, int* __restrict__ flags = nullptr,
                          int64_t r0 = 0, int64_t chunk = 0, double* __restrict__ Xc = nullptr, int* __restrict__ cmp = nullptr) {
    // 32 lanes per walker (one parameter each): these kernels sit between the log-probability batches of a
    // step, so they are organised for latency, not for thread economy — every lane redoes the walker's draws
#pragma clang fp contract(off)       // emcee's arithmetic rounds every product: no fused multiply-adds in here
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t k = gid >> 5;
    const int t0 = (int)(gid & 31);
    const bool full = ((int64_t)(blockIdx.x + 1) * blockDim.x) >> 5 <= nhalf;      // every thread of this workgroup has a walker
    if (k >= nhalf) return;
    const SplitPerm pi = make_perm(seed, step, 2 * nhalf, hb, randomize);
    const U4 r = philox(seed, (uint32_t)k, step, (uint32_t)half, 0u);
    const double u = u01(r.x, r.y);
    // emcee StretchMove.get_proposal, operation for operation:
    //   zz = ((a - 1) * u + 1) ** 2 / a ;  q = c - (c - s) * zz ;  factor = (ndim - 1) * log(zz)
    const double zs = (a - 1.0) * u + 1.0;
    const double zz = (zs * zs) / a;
    const int64_t j = (int64_t)(((uint64_t)r.z * (uint64_t)nhalf) >> 32);
    const double* s = pos + pi(2 * k + half) * d;
    const double* c = pos + pi(2 * j + (1 - half)) * d;
    int ok = 1;
    double v2[2] = {0.0, 0.0};                         // the first two parameters of this lane (all of them for d <= 64)
    int nv = 0;
    for (int t = t0; t < d; t += 32) {
        const double v = c[t] - (c[t] - s[t]) * zz;
        q[k * d + t] = v;
        if (lo) ok &= (int)(v > lo[t]) & (int)(v < hi[t]);
        if (nv < 2) v2[nv] = v;
        ++nv;
    }
    if (t0 == 0) factor[k] = (d - 1.0) * log(zz);
    if (lo) {                                          // wave-uniform; a walker's 32 lanes are one half of a wave
        const unsigned long long out = __ballot(!ok);
        const bool in = (((threadIdx.x & 32) ? (out >> 32) : out) & 0xffffffffull) == 0ull;
        const bool mine = k >= r0 && k < r0 + chunk;
        if (t0 == 0 && mine) {
            if (flags) flags[k - r0] = in ? 1 : 0;
            if (!in && ll) ll[k] = outside;
        }
        if (Xc) {
            // ... and gathers the rows inside the box itself: a slot from a counter (cmp[0], zeroed by k_accept), in
            // whatever order the walkers arrive — a row's result does not depend on its place in the batch
            int slot = take_slot(t0 == 0 && mine && in, full, cmp);
            if (slot >= 0) cmp[4 + slot] = (int)(k - r0);
            slot = __shfl(slot, (int)(threadIdx.x & 32), 64);
            if (slot >= 0) {
                if (t0 < d) Xc[(int64_t)slot * d + t0] = v2[0];
                if (t0 + 32 < d) Xc[(int64_t)slot * d + t0 + 32] = v2[1];
                // chains with more than 64 parameters (parameterTrafoPCA: the chain's ndim is the map's d_in, which only
                // the GPs' reduced d bounds): the same arithmetic again, operation for operation
                for (int t = t0 + 64; t < d; t += 32) Xc[(int64_t)slot * d + t] = c[t] - (c[t] - s[t]) * zz;
            }
        }
    }
}
// Where the accept step finds a proposal's log-probability.  Plain: lpq[slot].  Balanced sharding (see
// k_balance_gather): lpq is the all-gathered array of the ranks' padded slices, a proposal inside the box is found through
// its position g in the ordered list of all live rows — slice g / per, entry g % per — and one outside the box has
// `outside` without a memory access.
struct LpSource {
    const double* lpq;
    const int* rank_of;                // null: plain
    const int* meta;                   // [0] = live rows in the whole batch, [1] = rows per slice
    int64_t chunk;                     // slice stride in lpq
    double outside;
    __device__ __forceinline__ double at(int64_t slot) const {
        if (!rank_of) return lpq[slot];
        const int g = rank_of[slot];
        if (g < 0) return outside;
        const int per = meta[1];
        return lpq[(int64_t)(g / per) * chunk + (g % per)];
    }
};

__global__ void k_accept(double* __restrict__ pos, double* __restrict__ lp, int64_t nhalf, int d, int half,
                         uint64_t seed, uint32_t step, const double* __restrict__ q,
                         const double* __restrict__ factor, const double* __restrict__ lpq,
                         long long* __restrict__ naccept, int hb, int randomize,
                         long long* __restrict__ n_nan, int* __restrict__ cmp = nullptr,
                         unsigned long long* __restrict__ hint = nullptr, int64_t W_batch = 0,
                         unsigned long long* __restrict__ rows_live = nullptr, const int* __restrict__ rank_of = nullptr,
                         const int* __restrict__ meta = nullptr, int64_t chunk = 0, double outside = 0.0) {
    // 32 lanes per walker, all inside one wave: every lane takes the same decision from the OLD lp[idx]
    // (the load precedes lane 0's store in program order), then moves its own parameters
#pragma clang fp contract(off)
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t k = gid >> 5;
    const int t0 = (int)(gid & 31);
    if (cmp && gid == 0) {
        // the batch's kernels are done with the count of rows inside the box (stream order): report it (tile-shape
        // rule of the next launches, profile counter) and re-arm the counter for the next proposal kernel
        const int cnt = cmp[0];
        if (hint) __hip_atomic_store(hint, ((unsigned long long)W_batch << 32) | (unsigned long long)cnt, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM);
        if (rows_live) atomicAdd(rows_live, (unsigned long long)cnt);
        cmp[0] = 0;
    }
    if (k >= nhalf) return;
    const SplitPerm pi = make_perm(seed, step, 2 * nhalf, hb, randomize);
    const U4 r = philox(seed, (uint32_t)k, step, (uint32_t)half, 1u);
    const double u = u01(r.x, r.y);
    const int64_t idx = pi(2 * k + half);
    const double lpq_k = LpSource{lpq, rank_of, meta, chunk, outside}.at(k);
    // emcee raises "Probability function returned NaN" at the step it happens (emcee/ensemble.py compute_log_prob);
    // here the proposal is rejected (NaN compares false) and counted, and the host raises at its next check
    if (n_nan && t0 == 0 && lpq_k != lpq_k) atomicAdd(reinterpret_cast<unsigned long long*>(n_nan), 1ull);
    const double diff = (factor[k] + lpq_k) - lp[idx];
    const bool take = diff > log(u);                                 // emcee RedBlueMove.propose: f + nlp - lp[j] > log(rand)
    __builtin_amdgcn_wave_barrier();                                 // keep the loads above the stores below
    if (take) {
        for (int t = t0; t < d; t += 32) pos[idx * d + t] = q[k * d + t];
        if (t0 == 0) {
            lp[idx] = lpq_k;
            if (naccept) naccept[idx] += 1;
        }
    }
}

// The accept of one half-step and the proposal of the next in ONE launch (gpb_chain_emcee_run): a proposal needs the
// positions AFTER the pending accept, of its own walker and of its partner; instead of waiting for another kernel to
// have moved them, a walker group looks both walkers up — the inverse split permutation tells whether a walker is in the
// pending half and in which slot — and takes that slot's accept decision itself (same draws, same arithmetic as the
// group that owns the slot).  Accepted walkers are read from the pending proposals q_a, all others from pos, which this
// kernel writes for accepted walkers only: no read of a location another group writes.  lp is ping-ponged (lp_in is
// read by every decision, lp_out written once per walker), q / factor / lpq alternate between two sets.
struct PendingAccept {
    const double *q, *factor, *lp_in;
    LpSource lpq;
    uint64_t seed;
    uint32_t step;
    int half;
};
__device__ __forceinline__ bool accept_decision(const PendingAccept& A, int64_t slot, int64_t idx, double& lpq_k) {
#pragma clang fp contract(off)
    const U4 r = philox(A.seed, (uint32_t)slot, A.step, (uint32_t)A.half, 1u);
    const double u = u01(r.x, r.y);
    lpq_k = A.lpq.at(slot);
    const double diff = (A.factor[slot] + lpq_k) - A.lp_in[idx];
    return diff > log(u);                                            // as k_accept
}
__global__ void k_accept_propose(double* __restrict__ pos, const double* __restrict__ lp_in, double* __restrict__ lp_out,
                                 int64_t nhalf, int d, uint64_t seed, int hb, int randomize,
                                 int half_a, uint32_t step_a, const double* __restrict__ q_a,
                                 const double* __restrict__ factor_a, const double* __restrict__ lpq_a,
                                 long long* __restrict__ naccept, long long* __restrict__ n_nan, int* __restrict__ cmp_prev,
                                 unsigned long long* __restrict__ hint, int64_t W_batch,
                                 unsigned long long* __restrict__ rows_live,
                                 int half_p, uint32_t step_p, double a, double* __restrict__ q_p,
                                 double* __restrict__ factor_p, const double* __restrict__ lo,
                                 const double* __restrict__ hi, double outside, double* __restrict__ ll, int64_t r0,
                                 int64_t chunk, double* __restrict__ Xc, int* __restrict__ cmp,
                                 const int* __restrict__ rank_of_a, const int* __restrict__ meta_a, int64_t chunk_a,
                                 int* __restrict__ flags_p) {
#pragma clang fp contract(off)
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t k = gid >> 5;
    const int t0 = (int)(gid & 31);
    if (gid == 0) {                                    // as k_accept: report and re-arm the finished batch's counter
        const int cnt = cmp_prev[0];
        if (hint) __hip_atomic_store(hint, ((unsigned long long)W_batch << 32) | (unsigned long long)cnt, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM);
        if (rows_live) atomicAdd(rows_live, (unsigned long long)cnt);
        cmp_prev[0] = 0;
    }
    const bool full = ((int64_t)(blockIdx.x + 1) * blockDim.x) >> 5 <= nhalf;      // every thread of this workgroup has a walker
    if (k >= nhalf) return;
    const SplitPerm pa = make_perm(seed, step_a, 2 * nhalf, hb, randomize);
    const PendingAccept A{q_a, factor_a, lp_in, LpSource{lpq_a, rank_of_a, meta_a, chunk_a, outside}, seed, step_a, half_a};
    {   // ---- the accept of slot k (k_accept, with lp written to the other buffer)
        const int64_t idx = pa(2 * k + half_a);
        double lpq_k;
        const bool take = accept_decision(A, k, idx, lpq_k);
        if (n_nan && t0 == 0 && lpq_k != lpq_k) atomicAdd(reinterpret_cast<unsigned long long*>(n_nan), 1ull);
        if (take)
            for (int t = t0; t < d; t += 32) pos[idx * d + t] = q_a[k * d + t];
        if (t0 == 0) {
            lp_out[idx] = take ? lpq_k : lp_in[idx];
            if (take && naccept) naccept[idx] += 1;
            const int64_t other = pa(2 * k + (1 - half_a));           // the walker of the resting half with this slot
            lp_out[other] = lp_in[other];
        }
    }
    // ---- the proposal of slot k for (step_p, half_p): k_propose on the positions after the pending accept
    const SplitPerm pp = make_perm(seed, step_p, 2 * nhalf, hb, randomize);
    const U4 r = philox(seed, (uint32_t)k, step_p, (uint32_t)half_p, 0u);
    const double u = u01(r.x, r.y);
    const double zs = (a - 1.0) * u + 1.0;
    const double zz = (zs * zs) / a;
    const int64_t j = (int64_t)(((uint64_t)r.z * (uint64_t)nhalf) >> 32);
    auto current = [&](int64_t w) -> const double* {
        const int64_t y = pa.inv(w);
        if ((int)(y & 1) == half_a) {
            double unused;
            if (accept_decision(A, y >> 1, w, unused)) return q_a + (y >> 1) * d;
        }
        return pos + w * d;
    };
    const double* s = current(pp(2 * k + half_p));
    const double* c = current(pp(2 * j + (1 - half_p)));
    int ok = 1;
    double v2[2] = {0.0, 0.0};
    int nv = 0;
    for (int t = t0; t < d; t += 32) {
        const double v = c[t] - (c[t] - s[t]) * zz;
        q_p[k * d + t] = v;
        ok &= (int)(v > lo[t]) & (int)(v < hi[t]);
        if (nv < 2) v2[nv] = v;
        ++nv;
    }
    if (t0 == 0) factor_p[k] = (d - 1.0) * log(zz);
    const unsigned long long out = __ballot(!ok);
    const bool in = (((threadIdx.x & 32) ? (out >> 32) : out) & 0xffffffffull) == 0ull;
    if (flags_p) {                                     // balanced sharding: the flag of EVERY row, k_balance_gather does the rest
        if (t0 == 0) flags_p[k] = in ? 1 : 0;
        return;
    }
    const bool mine = k >= r0 && k < r0 + chunk;
    if (t0 == 0 && mine && !in) ll[k] = outside;
    int slot = take_slot(t0 == 0 && mine && in, full, cmp);
    if (slot >= 0) cmp[4 + slot] = (int)(k - r0);
    slot = __shfl(slot, (int)(threadIdx.x & 32), 64);
    if (slot >= 0) {
        if (t0 < d) Xc[(int64_t)slot * d + t0] = v2[0];
        if (t0 + 32 < d) Xc[(int64_t)slot * d + t0 + 32] = v2[1];
        for (int t = t0 + 64; t < d; t += 32) Xc[(int64_t)slot * d + t] = c[t] - (c[t] - s[t]) * zz;   // ndim > 64: as k_propose
    }
}

// Balanced sharding of a batch over the ranks (gpb_chain_emcee_run with a communicator).  A rank's contiguous share of
// the proposals holds a varying number of rows inside the prior box (256 proposals: 120 +- 8), and the step waits for the
// rank with the most — which, more often than not, needs one walker tile more than the others.  Every rank knows all
// proposals, so every rank ranks ALL live rows in order here (flags from the proposal kernel; counts of integers: any
// order) and takes the `r`-th of R equal slices of that list: rows with rank g in [r per, (r + 1) per), per =
// ceil(live / R), gathered into Xc.  The slices are padded to the collective's fixed size (per <= nhalf / R), so the
// all-gather is the one of the contiguous scheme; the accept kernels find a live row's value through rank_of (LpSource).
//   flags[nhalf] in; rank_of[nhalf] out (-1 outside the box); meta = {live, per}; cmp[0] = rows of this rank's slice,
//   cmp[4 + i] = i (the likelihood kernel's scatter list: results land densely in the send buffer)
__global__ __launch_bounds__(256) void k_balance_gather(const double* __restrict__ q, int64_t nhalf, int d,
                                                        const int* __restrict__ flags, int* __restrict__ rank_of,
                                                        int R, int r, double* __restrict__ Xc, int* __restrict__ cmp,
                                                        int* __restrict__ meta) {
    __shared__ int wsum[4], wbase[4], wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * 256;
    int before = 0, total = 0;
    for (int64_t w = tid; w < nhalf; w += 256) {
        const int f = flags[w];
        total += f;
        if (w < w0) before += f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_xor(before, o);
        total += __shfl_xor(total, o);
    }
    const bool in = w0 + tid < nhalf && flags[w0 + tid] != 0;
    const unsigned long long m = __ballot(in);
    if (lane == 0) { wsum[wave] = __popcll(m); wbase[wave] = before; wtot[wave] = total; }
    __syncthreads();
    int off = (wbase[0] + wbase[1]) + (wbase[2] + wbase[3]);
    const int live = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
    for (int i = 0; i < wave; ++i) off += wsum[i];
    const int g = in ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
    const int per = live > 0 ? (live + R - 1) / R : 1;
    const int lo = r * per, hi = min(lo + per, live);
    if (w0 + tid < nhalf) rank_of[w0 + tid] = g;
    if (g >= lo && g < hi) {
        const int64_t src = (w0 + tid) * d, dst = (int64_t)(g - lo) * d;
        for (int k = 0; k < d; ++k) Xc[dst + k] = q[src + k];
        cmp[4 + (g - lo)] = g - lo;
    }
    if (blockIdx.x == 0 && tid == 0) {
        cmp[0] = hi > lo ? hi - lo : 0;
        meta[0] = live;
        meta[1] = per;
    }
}

#ifdef GPB_DEBUG_VARIANTS
// test hooks: the generator and the draws of a (seed, step, half), for the parity tests against the oracle
__global__ void k_philox_test(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* v = in + 6 * i;
    const U4 r = philox((uint64_t)v[0] | ((uint64_t)v[1] << 32), v[2], v[3], v[4], v[5]);
    out[4 * i + 0] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}
__global__ void k_stretch_draws(int64_t nhalf, int half, uint64_t seed, uint32_t step, int hb, int randomize,
                                double* __restrict__ u_z, long long* __restrict__ jj, double* __restrict__ u_acc,
                                long long* __restrict__ perm) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= 2 * nhalf) return;
    perm[k] = make_perm(seed, step, 2 * nhalf, hb, randomize)(k);
    if (k >= nhalf) return;
    const U4 r = philox(seed, (uint32_t)k, step, (uint32_t)half, 0u);           // as k_propose
    u_z[k] = u01(r.x, r.y);
    jj[k] = (long long)(((uint64_t)r.z * (uint64_t)nhalf) >> 32);
    const U4 ra = philox(seed, (uint32_t)k, step, (uint32_t)half, 1u);          // as k_accept
    u_acc[k] = u01(ra.x, ra.y);
}

// test hook: out[i] = pi_step(i)
__global__ void k_perm(long long* __restrict__ out, int64_t n, uint64_t seed, uint32_t step, int hb) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_perm(seed, step, n, hb, 1)(i);
}
#endif  // GPB_DEBUG_VARIANTS

static int half_bits(int64_t n) {
    int b = 1;
    while ((1ll << b) < n) ++b;
    return (b + 1) / 2;
}

}  // namespace gpb

using namespace gpb;

extern "C" int gpb_box_finish(gpb_ctx* ctx, const double* X_dev, int64_t W, int64_t d, const double* lo_dev,
                              const double* hi_dev, double outside_value, double inside_const,
                              double* ll_inout_dev) {
    if (!ctx || W < 0 || d < 1 || !X_dev || !lo_dev || !hi_dev || !ll_inout_dev) return GPB_E_ARG;
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_box, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, ctx->stream, X_dev, W, (int)d,
                       lo_dev, hi_dev, outside_value, inside_const, ll_inout_dev);
    GPB_HIP(hipGetLastError());
    return 0;
}

extern "C" int gpb_stretch_propose(gpb_ctx* ctx, const double* pos_dev, int64_t nwalkers, int64_t d, int half,
                                   uint64_t seed, uint64_t step, double a, double* q_dev, double* factor_dev,
                                   int randomize_split) {
    if (!ctx || nwalkers < 2 || (nwalkers & 1) || nwalkers > (1ll << 30) || d < 1 || (half != 0 && half != 1))
        return GPB_E_ARG;
    const int64_t nh = nwalkers / 2;
    hipLaunchKernelGGL(k_propose, dim3((unsigned)((nh * 32 + 255) / 256)), dim3(256), 0, ctx->stream, pos_dev, nh, (int)d,
                       half, seed, (uint32_t)step, a, q_dev, factor_dev, half_bits(nwalkers), randomize_split ? 1 : 0,
                       (const double*)nullptr, (const double*)nullptr, 0.0, (double*)nullptr, (int*)nullptr, (int64_t)0, (int64_t)0,
                       (double*)nullptr, (int*)nullptr);
    GPB_HIP(hipGetLastError());
    return 0;
}

extern "C" int gpb_stretch_accept(gpb_ctx* ctx, double* pos_dev, double* lp_dev, int64_t nwalkers, int64_t d,
                                  int half, uint64_t seed, uint64_t step, const double* q_dev,
                                  const double* factor_dev, const double* lpq_dev, int64_t* naccept_dev,
                                  int randomize_split) {
    if (!ctx || nwalkers < 2 || (nwalkers & 1) || nwalkers > (1ll << 30) || d < 1 || (half != 0 && half != 1))
        return GPB_E_ARG;
    const int64_t nh = nwalkers / 2;
    hipLaunchKernelGGL(k_accept, dim3((unsigned)((nh * 32 + 255) / 256)), dim3(256), 0, ctx->stream, pos_dev, lp_dev, nh,
                       (int)d, half, seed, (uint32_t)step, q_dev, factor_dev, lpq_dev,
                       reinterpret_cast<long long*>(naccept_dev), half_bits(nwalkers), randomize_split ? 1 : 0,
                       reinterpret_cast<long long*>(ctx->n_nan), (int*)nullptr, (unsigned long long*)nullptr, (int64_t)0,
                       (unsigned long long*)nullptr);
    GPB_HIP(hipGetLastError());
    return 0;
}

extern "C" int gpb_stretch_nan_count(gpb_ctx* ctx, int64_t* count_host, int reset) {
    if (!ctx || !count_host) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    long long v = 0;
    GPB_HIP(hipMemcpyAsync(&v, ctx->n_nan, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    if (reset) GPB_HIP(hipMemsetAsync(ctx->n_nan, 0, sizeof(long long), ctx->stream));
    *count_host = (int64_t)v;
    return 0;
}

// ---------------------------------------------------------------------------- device-resident sampling loop
__global__ void k_store_step(const double* __restrict__ pos, const double* __restrict__ lp, double* __restrict__ chain,
                             double* __restrict__ lpchain, int64_t nw, int d) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (chain && i < nw * d) chain[i] = pos[i];
    if (lpchain && i < nw) lpchain[i] = lp[i];
}

__global__ void k_fill(double* __restrict__ x, int64_t n, double v) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// ---- chains of several emulators ---------------------------------------------------------------------
// Chain._predict concatenates the emulators' observables and the covariance is block-diagonal over them
// (src/mcmc.py:153-166), so the log-likelihood is the sum of the emulators' blocks; all emulators see the same rows of
// the same parameter space (those with a parameter map, src/emulator.py:492-551, through gpb_param_map).
namespace {
int64_t chain_ndim(const gpb_ctx* c) { return c->pmap_d_in > 0 ? c->pmap_d_in : c->d; }
// parameters of the CHAIN (a parameter map's d_in; the GPs' own d is bounded by 64 in gpb_gp_set).  The proposal kernels
// take any number; k_compact_mark stages 256 rows of it in LDS in tiles, so the bound is only a sanity limit.
constexpr int64_t MAX_CHAIN_NDIM = 512;

// the per-emulator blocks of a chain's log-likelihood, [E][Wcap] in the chain's first context (k_loglike_lowrank_multi / k_lowrank_sum)
int ensure_lr_blocks(gpb_ctx* ctx, int E) {
    if (E < 2 || !ctx->lr_split) return 0;
    const int64_t need = (int64_t)E * ctx->Wcap;
    if (ctx->lr_blocks_cap >= need) return 0;
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->lr_blocks) { pool_free(ctx->lr_blocks); ctx->lr_blocks = nullptr; }
    ctx->lr_blocks_cap = 0;
    GPB_HIP(pool_malloc_t(&ctx->lr_blocks, sizeof(double) * (size_t)need));
    ctx->lr_blocks_cap = need;
    return 0;
}

// every context usable by the compacted chain path?  (same device, stream and parameter space; likelihood installed;
// a block likelihood kernel applies)
int chain_check(gpb_ctx* const* ctxs, int E, const char* who) {
    gpb_ctx* ctx = ctxs[0];
    for (int e = 0; e < E; ++e) {
        const gpb_ctx* c = ctxs[e];
        if (!c) GPB_FAIL(GPB_E_ARG, std::string(who) + ": null context");
        if (!c->have_like) GPB_FAIL(GPB_E_STATE, std::string(who) + " before gpb_like_set");
        if (c->device != ctx->device || c->stream != ctx->stream)
            GPB_FAIL(GPB_E_STATE, std::string(who) + ": the emulators' contexts must share one device and stream");
        if (chain_ndim(c) != chain_ndim(ctx)) GPB_FAIL(GPB_E_ARG, std::string(who) + ": the emulators disagree on the number of parameters");
        if (!compaction_applies(c))
            GPB_FAIL(GPB_E_STATE, std::string(who) + ": needs the block likelihood kernels (PCA mode, M <= 64 or npc <= 16) for every emulator");
        if (c->pmap_d_in > 0 && c->pmap_d_out != c->d)
            GPB_FAIL(GPB_E_STATE, std::string(who) + ": a parameter map's output must be the GPs' input");
    }
    if (chain_ndim(ctx) > MAX_CHAIN_NDIM)
        GPB_FAIL(GPB_E_ARG, std::string(who) + ": more than 512 chain parameters");
    return 0;
}

// log-posterior of rows X[W][ndim] over all emulators, rows inside the box only (ctxs[0] owns the compaction)
int chain_rows(gpb_ctx* const* ctxs, int E, const double* X_dev, int64_t W, double* ll_dev, const double* lo_dev,
               const double* hi_dev, double outside, double inside_const, int premarked = 0, const int* cmpv = nullptr) {
    gpb_ctx* c0 = ctxs[0];
    int rc;
    for (int e = 0; e < E; ++e)
        if ((rc = ensure_wcap(ctxs[e], W))) { if (e) c0->err = ctxs[e]->err; return rc; }
    if ((rc = ensure_lr_blocks(c0, E))) return rc;
    if ((rc = launch_compact(c0, X_dev, W, chain_ndim(c0), lo_dev, hi_dev, outside, ll_dev, premarked))) return rc;
    if (!cmpv) cmpv = c0->cmp_idx;                     // (count, -, -, -, indices ...) of the rows inside the box
    // Three passes over the emulators (each kernel sees what it would see in its own emulator's sequence: same bits):
    // (1) parameter maps, then K*^T and the mean partials — ONE launch per run of emulators of equal padded size
    //     (k_kcross_multi);
    // (2) V = L^-1 K*^T with the fused sum of squares: ONE launch for each run of emulators whose designs pad to the same
    //     Np (the reference's analyses: nine emulators on one design) instead of one partly filled launch per emulator;
    // (3) the block log-likelihoods, added up in emuList order: one launch that walks the emulators (k_loglike_lowrank_multi)
    //     when every block takes the low-rank kernel, else one launch per emulator.
    const double* Xg[64];
    gpb_ctx* mapped[64];
    int nmapped = 0;
    for (int e = 0; e < E; ++e) {
        gpb_ctx* c = ctxs[e];
        Xg[e] = c0->cmp_X;
        c->hint_from = c0;
        if (c->pmap_d_in > 0) {                        // this emulator's GPs see the PCA-reduced parameters
            mapped[nmapped++] = c;
            Xg[e] = c->Xs;
        }
    }
    if (nmapped > 1 && c0->chain_batch) {              // the maps of all mapped emulators over the gathered rows: one launch
        if ((rc = launch_param_maps(mapped, nmapped, c0->cmp_X, W))) { c0->err = mapped[0]->err; return rc; }
    } else {
        for (int i = 0; i < nmapped; ++i)
            if ((rc = gpb_param_map(mapped[i], c0->cmp_X, W, mapped[i]->Xs))) { c0->err = mapped[i]->err; return rc; }
    }
    for (int e = 0; e < E;) {              // K*^T: one launch per run of emulators of equal padded size and PADDED input
        int n = 1;                                     // count (parameterTrafoPCA emulators keep 17-19 of 20 inputs each: one launch)
        while (c0->chain_batch && e + n < E && ctxs[e + n]->Np == ctxs[e]->Np && ctxs[e + n]->dpad == ctxs[e]->dpad && n < 32) ++n;
        if ((rc = launch_kcross_group(ctxs + e, Xg + e, n, W, cmpv))) { c0->err = ctxs[e]->err; return rc; }
        e += n;
    }
    for (int e = 0; e < E;) {
        int n = 1, gps = (int)ctxs[e]->P;
        while (c0->chain_batch && e + n < E && ctxs[e + n]->Np == ctxs[e]->Np && gps + (int)ctxs[e + n]->P <= GPB_MAX_MULTI_GP) {
            gps += (int)ctxs[e + n]->P;
            ++n;
        }
        if ((rc = launch_vsq(ctxs + e, n, W, cmpv))) { c0->err = ctxs[e]->err; return rc; }
        e += n;
    }
    if (c0->chain_batch && E > 1 && E <= MAX_LR_CTX) {  // all blocks by the low-rank kernel: one launch walks the emulators
        bool all = true;
        int64_t pmax = 0;
        for (int e = 0; e < E; ++e) {
            all = all && lowrank_applies(ctxs[e]) && ctxs[e]->fuse_finalize && ctxs[e]->Wld == c0->Wld;
            pmax = ctxs[e]->P > pmax ? ctxs[e]->P : pmax;
        }
        if (all) {
            LrTable tab;
            for (int e = 0; e < E; ++e) {
                const gpb_ctx* c = ctxs[e];
                tab.c[e] = LrCtx{c->mpart, c->spart, c->amp, c->noise, c->lr_R, c->lr_v0, c->notpd, c->lr_cperp, c->lr_logdet0,
                                 (int)c->P, (int)((c->Np + KX_CHUNK - 1) / KX_CHUNK), (int)(c->Np / 64)};
            }
            tab.E = E;
            // one workgroup per (walker tile, emulator) + the ordered sum, when the blocks' buffer is there (ensure_lr_blocks;
            // option key 49 = 0: the one-launch walk — the A/B, and the bit-identity test)
            double* blocks = (c0->lr_split && c0->lr_blocks && c0->lr_blocks_cap >= (int64_t)E * c0->Wld) ? c0->lr_blocks : nullptr;
            const dim3 grid((unsigned)((W + 63) / 64), blocks ? (unsigned)E : 1u);
#define GPB_LRM(PPV)                                                                                             \
    hipLaunchKernelGGL(k_loglike_lowrank_multi<PPV>, grid, dim3(lr_threads<PPV>()), 0, c0->stream, tab, c0->Wld, W, ll_dev, cmpv, inside_const, blocks)
            if (pmax <= 4) GPB_LRM(4); else if (pmax <= 8) GPB_LRM(8); else if (pmax <= 12) GPB_LRM(12); else GPB_LRM(16);
#undef GPB_LRM
            if (blocks)
                hipLaunchKernelGGL(k_lowrank_sum, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, c0->stream, blocks, E, c0->Wld, W,
                                   ll_dev, cmpv, inside_const);
            if (hipGetLastError() != hipSuccess) { c0->err = "gpb: k_loglike_lowrank_multi launch failed"; return GPB_E_HIP; }
            return 0;
        }
    }
    for (int e = 0; e < E; ++e) {
        gpb_ctx* c = ctxs[e];
        const bool fused = loglike_fuses_finalize(c, W);
        if ((!fused && (rc = launch_finalize(c, W, true))) ||
            (rc = launch_loglike(c, W, ll_dev, e > 0, fused, nullptr, nullptr, nullptr, outside,
                                 e == E - 1 ? inside_const : 0.0, cmpv))) {
            c0->err = c->err;
            return rc;
        }
    }
    return 0;
}
}  // namespace

extern "C" int gpb_chain_supported(gpb_ctx* const* ctxs, int E) {
    if (!ctxs || E < 1 || E > 64 || !ctxs[0]) return GPB_E_ARG;
    for (int e = 0; e < E; ++e) {
        const gpb_ctx* c = ctxs[e];
        if (!c) return GPB_E_ARG;
        if (!c->have_like || c->device != ctxs[0]->device || c->stream != ctxs[0]->stream ||
            chain_ndim(c) != chain_ndim(ctxs[0]) || !compaction_applies(c) || chain_ndim(c) > MAX_CHAIN_NDIM ||
            (c->pmap_d_in > 0 && c->pmap_d_out != c->d))
            return 0;
    }
    return 1;
}

extern "C" int gpb_chain_logpost(gpb_ctx* const* ctxs, int E, const double* Xs_dev, int64_t W, double* ll_dev,
                                 const double* lo_dev, const double* hi_dev, double outside_value, double inside_const) {
    if (!ctxs || E < 1 || E > 64 || !ctxs[0]) return GPB_E_ARG;
    gpb_ctx* ctx = ctxs[0];
    if (!Xs_dev || !ll_dev || !lo_dev || !hi_dev || W < 0) GPB_FAIL(GPB_E_ARG, "gpb_chain_logpost: null pointer or negative size");
    int rc = chain_check(ctxs, E, "gpb_chain_logpost");
    if (rc) return rc;
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    return chain_rows(ctxs, E, Xs_dev, W, ll_dev, lo_dev, hi_dev, outside_value, inside_const);
}

namespace {
// What gpb_chain_emcee_run decides before it enqueues anything: argument checks, the share of every batch this rank
// evaluates, which of the step's kernels are fused, and every workspace it needs — all of which can fail on ONE rank only
// (bad state, hipMalloc).  gpb_chain_emcee_prepare runs exactly this and nothing else, so that the ranks of a sharded run
// can agree that all of them are ready BEFORE any of them enqueues a collective the others would wait in.
struct EmceePlan {
    int64_t nh = 0, d = 0, chunk = 0, r0 = 0;
    int R = 1;
    bool sim = false, plain = false, fused = false, premark = false, fuse_ap = false, balanced = false;
    int pre = 0;
};

int emcee_plan(gpb_ctx* const* ctxs, int E, int64_t nwalkers, EmceePlan& pl) {
    gpb_ctx* ctx = ctxs[0];
    if (nwalkers < 2 || (nwalkers & 1) || nwalkers > (1ll << 30)) GPB_FAIL(GPB_E_ARG, "gpb_chain_emcee_run: nwalkers must be even, 2 .. 2^30");
    // one emulator without a parameter map may also run uncompacted (tune key 27 = 0, non-PCA modes): gpb_logpost's sequence
    pl.plain = E == 1 && ctx->pmap_d_in == 0 && !compaction_applies(ctx);
    int rc;
    if (pl.plain) {
        if (!ctx->have_like) GPB_FAIL(GPB_E_STATE, "gpb_emcee_run before gpb_like_set");
    } else if ((rc = chain_check(ctxs, E, "gpb_chain_emcee_run"))) {
        return rc;
    }
    GPB_HIP(hipSetDevice(ctx->device));
    const int64_t nh = nwalkers / 2, d = chain_ndim(ctx);
    int R = ctx->comm ? ctx->nranks : 1;
    const int rank = ctx->comm ? ctx->rank : 0;
    // measurement / test hook (tune keys 26, 32): behave like rank `sim_rank` of `sim_ranks` on a single GPU — evaluate
    // that rank's nh / sim_ranks rows of every batch only (the other rows keep -inf: rejected) and still issue the collective
    pl.sim = ctx->sim_ranks > 1 && R == 1;
    if (pl.sim) R = ctx->sim_ranks;
    if (nh % R) GPB_FAIL(GPB_E_ARG, "gpb_chain_emcee_run: half the ensemble must divide evenly over the ranks");
    if (pl.sim && ctx->sim_rank >= R) GPB_FAIL(GPB_E_ARG, "gpb_chain_emcee_run: tune key 32 (simulated rank) must be below key 26 (ranks)");
    pl.nh = nh; pl.d = d; pl.R = R;
    pl.chunk = nh / R;
    pl.r0 = (pl.sim ? ctx->sim_rank : rank) * pl.chunk;
    const int64_t chunk = pl.chunk;
    for (int e = 0; e < E; ++e)                        // all workspaces now: the loop holds pointers into them
        if ((rc = ensure_wcap(ctxs[e], chunk))) { if (e) ctx->err = ctxs[e]->err; return rc; }
    if (!pl.plain && (rc = ensure_lr_blocks(ctx, E))) return rc;
    // proposal workspace: two sets of q[nh][d], factor[nh], lpq[nh] (the fused accept + proposal kernel reads one set and
    // writes the other) and a second log-probability vector [nwalkers]
    if (ctx->mc_cap < 2 * nh * (d + 3)) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->mc_ws) pool_free(ctx->mc_ws);
        ctx->mc_ws = nullptr;
        ctx->mc_cap = 0;
        GPB_HIP(pool_malloc_t(&ctx->mc_ws, sizeof(double) * (size_t)(2 * nh * (d + 3))));
        ctx->mc_cap = 2 * nh * (d + 3);
    }
    pl.fused = pl.plain && loglike_fuses_finalize(ctx, chunk);
    // the gather kernel counts the flags in front of each of its workgroups itself: fine for a rank's rows of an
    // ensemble, quadratic for very large batches, which keep the marking kernel with its per-workgroup counts
    pl.premark = !pl.plain && ctx->premark && chunk <= 16384;
    // ... premark 2 (default): the proposal kernel gathers the rows as well (slots from a counter that the accept kernel
    // re-arms), no compaction kernel at all; 1: flags only, k_compact_gather follows
    pl.pre = pl.premark ? (ctx->premark >= 2 ? 2 : 1) : 0;
    // ... and with that, tune key 30 (default on): the accept of a half-step and the proposal of the next are one launch
    pl.fuse_ap = pl.pre == 2 && ctx->fuse_accept_propose;
    // sharded (or playing one rank of several): equal slices of the ordered list of ALL live rows instead of the live rows
    // of a contiguous share (k_balance_gather; tune key 36)
    // Worth its extra launch (k_balance_gather + the rank look-ups of the accept step: +11 us per half-step, measured) only
    // where the ranks' live counts straddle a walker-tile boundary often: 256 proposals per rank hold 120 +- 8 live rows, so
    // at 8 ranks three half-steps in four have a rank with a fifth 64x32 tile (+23 us, measured per tile); at 2 and 4 ranks
    // the contiguous shares rarely differ by a tile.  0 = never (default), 1 = from 8 ranks on, 2 = always.
    pl.balanced = pl.pre == 2 && R > 1 && nh <= 16384 &&
                  (ctx->balance_shards == 2 || (ctx->balance_shards == 1 && R >= 8));
    if (pl.balanced) {
        const int64_t need = 4 * nh + 2 * (4 + chunk) + 16;
        if (ctx->bal_cap < need) {
            GPB_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->bal_ws) pool_free(ctx->bal_ws);
            ctx->bal_ws = nullptr;
            ctx->bal_cap = 0;
            GPB_HIP(pool_malloc_t(&ctx->bal_ws, sizeof(int) * (size_t)need));
            ctx->bal_cap = need;
        }
    }
    if (pl.pre && (rc = ensure_cmp_rows(ctx, d))) return rc;
    return 0;
}
}  // namespace

extern "C" int gpb_chain_emcee_prepare(gpb_ctx* const* ctxs, int E, int64_t nwalkers) {
    if (!ctxs || E < 1 || E > 64 || !ctxs[0]) return GPB_E_ARG;
    EmceePlan pl;
    return emcee_plan(ctxs, E, nwalkers, pl);
}

extern "C" int gpb_chain_emcee_run(gpb_ctx* const* ctxs, int E, double* pos_dev, double* lp_dev, int64_t nwalkers,
                                   int64_t nsteps, uint64_t seed, uint64_t step0, double a, int randomize_split,
                                   const double* lo_dev, const double* hi_dev, double outside_value, double inside_const,
                                   double* chain_dev, double* lpchain_dev, int64_t* naccept_dev) {
    if (!ctxs || E < 1 || E > 64 || !ctxs[0]) return GPB_E_ARG;
    gpb_ctx* ctx = ctxs[0];
    if (!pos_dev || !lp_dev || !lo_dev || !hi_dev || nsteps < 0) GPB_FAIL(GPB_E_ARG, "gpb_chain_emcee_run: null pointer or negative size");
    EmceePlan pl;
    int rc = emcee_plan(ctxs, E, nwalkers, pl);
    if (rc) return rc;
    const int64_t nh = pl.nh, d = pl.d, chunk = pl.chunk, r0 = pl.r0;
    const int R = pl.R, pre = pl.pre;
    const bool sim = pl.sim, plain = pl.plain, fused = pl.fused, premark = pl.premark, fuse_ap = pl.fuse_ap,
               balanced = pl.balanced;
    double* qs[2] = {ctx->mc_ws, ctx->mc_ws + nh * (d + 2)};
    double* factors[2] = {qs[0] + nh * d, qs[1] + nh * d};
    double* lpqs[2] = {factors[0] + nh, factors[1] + nh};
    double* lp2 = ctx->mc_ws + 2 * nh * (d + 2);
    const int hb = half_bits(nwalkers), rnd = randomize_split ? 1 : 0;
    const dim3 g32((unsigned)((nh * 32 + 255) / 256));
    int *bal_flags[2] = {nullptr, nullptr}, *bal_rank[2] = {nullptr, nullptr}, *bal_cmp[2] = {nullptr, nullptr},
        *bal_meta[2] = {nullptr, nullptr};
    if (balanced) {
        const int64_t need = 4 * nh + 2 * (4 + chunk) + 16;
        GPB_HIP(hipMemsetAsync(ctx->bal_ws, 0, sizeof(int) * (size_t)need, ctx->stream));
        for (int b = 0; b < 2; ++b) {
            bal_flags[b] = ctx->bal_ws + b * nh;
            bal_rank[b] = ctx->bal_ws + 2 * nh + b * nh;
            bal_cmp[b] = ctx->bal_ws + 4 * nh + b * (4 + chunk);
            bal_meta[b] = ctx->bal_ws + 4 * nh + 2 * (4 + chunk) + 4 * b;
        }
    }
    if (pre) GPB_HIP(hipMemsetAsync(ctx->cmp_idx, 0, 2 * sizeof(int), ctx->stream));
    if (sim) hipLaunchKernelGGL(k_fill, dim3((unsigned)((2 * nh * (d + 2) + 255) / 256)), dim3(256), 0, ctx->stream, ctx->mc_ws,
                                2 * nh * (d + 2), -INFINITY);
    unsigned long long* const live = pre == 2 && ctx->profile ? ctx->rows_live : (unsigned long long*)nullptr;
    double* lp_cur = lp_dev;                           // fuse_ap: lp alternates between the caller's vector and lp2
    double* lp_alt = lp2;
    const int64_t nhalfsteps = 2 * nsteps;
    for (int64_t g = 0; g < nhalfsteps; ++g) {
        const int64_t n = g >> 1;
        const int half = (int)(g & 1), b = fuse_ap ? (int)(g & 1) : 0;
        const uint32_t step = (uint32_t)(step0 + (uint64_t)n);
        double *q = qs[b], *factor = factors[b], *lpq = lpqs[b];
        // the counter + index list of this batch's rows inside the box: two views one int apart, so that the fused kernel
        // can re-arm the finished batch's counter while it fills the next one's
        int* cmpv = balanced ? bal_cmp[b] : (ctx->cmp_idx ? ctx->cmp_idx + b : nullptr);
        if (!fuse_ap || g == 0) {
            if (balanced)      // flags of every row; k_balance_gather ranks them and takes this rank's slice
                hipLaunchKernelGGL(k_propose, g32, dim3(256), 0, ctx->stream, pos_dev, nh, (int)d, half, seed, step, a, q,
                                   factor, hb, rnd, lo_dev, hi_dev, outside_value, (double*)nullptr, bal_flags[b], (int64_t)0,
                                   nh, (double*)nullptr, (int*)nullptr);
            else if (premark)  // the proposal kernel also takes the prior-box test of this rank's rows
                hipLaunchKernelGGL(k_propose, g32, dim3(256), 0, ctx->stream, pos_dev, nh, (int)d, half, seed, step, a, q,
                                   factor, hb, rnd, lo_dev, hi_dev, outside_value, lpq,
                                   pre == 1 ? ctx->cmp_idx + 4 + ctx->Wcap : (int*)nullptr, r0, chunk,
                                   pre == 2 ? ctx->cmp_X : (double*)nullptr, pre == 2 ? cmpv : (int*)nullptr);
            else
                hipLaunchKernelGGL(k_propose, g32, dim3(256), 0, ctx->stream, pos_dev, nh, (int)d, half, seed, step, a, q,
                                   factor, hb, rnd, (const double*)nullptr, (const double*)nullptr, 0.0, (double*)nullptr,
                                   (int*)nullptr, (int64_t)0, (int64_t)0, (double*)nullptr, (int*)nullptr);
        }
        if (balanced)
            hipLaunchKernelGGL(k_balance_gather, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, ctx->stream, q, nh, (int)d,
                               bal_flags[b], bal_rank[b], R, (int)(r0 / chunk), ctx->cmp_X, cmpv, bal_meta[b]);
        // this rank's rows of the batch: [compaction to the rows inside the box,] per emulator K*^T + mean partials,
        // V = L^-1 K*^T with the fused sum of squares, block log-likelihood (+ prior box + constant)
        if (plain) {
            if ((rc = launch_predict(ctx, q + r0 * d, chunk, true, !fused))) return rc;
            if ((rc = launch_loglike(ctx, chunk, lpq + r0, false, fused, q + r0 * d, lo_dev, hi_dev, outside_value,
                                     inside_const)))
                return rc;
        } else if ((rc = chain_rows(ctxs, E, q + r0 * d, chunk, lpq + r0, lo_dev, hi_dev, outside_value, inside_const, pre,
                                    cmpv))) {
            return rc;
        }
        if (sim ? ctx->comm != nullptr : R > 1)                      // in place, on this stream
            if ((rc = gpb_dist_allgather(ctx, lpq + r0, lpq, chunk))) return rc;
        if (fuse_ap && g + 1 < nhalfsteps) {
            const int64_t g1 = g + 1;
            hipLaunchKernelGGL(k_accept_propose, g32, dim3(256), 0, ctx->stream, pos_dev, lp_cur, lp_alt, nh, (int)d, seed, hb,
                               rnd, half, step, q, factor, lpq, reinterpret_cast<long long*>(naccept_dev),
                               reinterpret_cast<long long*>(ctx->n_nan), cmpv, ctx->live_hint, chunk, live,
                               (int)(g1 & 1), (uint32_t)(step0 + (uint64_t)(g1 >> 1)), a, qs[1 - b], factors[1 - b], lo_dev,
                               hi_dev, outside_value, lpqs[1 - b], r0, chunk, ctx->cmp_X,
                               balanced ? bal_cmp[1 - b] : ctx->cmp_idx + (1 - b), (const int*)bal_rank[b],
                               (const int*)bal_meta[b], chunk, bal_flags[1 - b]);
            double* sw = lp_cur; lp_cur = lp_alt; lp_alt = sw;
        } else {
            hipLaunchKernelGGL(k_accept, g32, dim3(256), 0, ctx->stream, pos_dev, lp_cur, nh, (int)d, half, seed, step, q,
                               factor, lpq, reinterpret_cast<long long*>(naccept_dev), hb, rnd,
                               reinterpret_cast<long long*>(ctx->n_nan), pre == 2 ? cmpv : (int*)nullptr,
                               pre == 2 ? ctx->live_hint : (unsigned long long*)nullptr, chunk, live,
                               (const int*)bal_rank[b], (const int*)bal_meta[b], chunk, outside_value);
        }
        if (half == 1 && (chain_dev || lpchain_dev))
            hipLaunchKernelGGL(k_store_step, dim3((unsigned)((nwalkers * d + 255) / 256)), dim3(256), 0, ctx->stream, pos_dev,
                               lp_cur, chain_dev ? chain_dev + n * nwalkers * d : nullptr,
                               lpchain_dev ? lpchain_dev + n * nwalkers : nullptr, nwalkers, (int)d);
    }
    if (lp_cur != lp_dev)
        GPB_HIP(hipMemcpyAsync(lp_dev, lp_cur, sizeof(double) * (size_t)nwalkers, hipMemcpyDeviceToDevice, ctx->stream));
    GPB_HIP(hipGetLastError());
    return 0;
}


extern "C" int gpb_emcee_run(gpb_ctx* ctx, double* pos_dev, double* lp_dev, int64_t nwalkers, int64_t nsteps,
                             uint64_t seed, uint64_t step0, double a, int randomize_split, const double* lo_dev,
                             const double* hi_dev, double outside_value, double inside_const, double* chain_dev,
                             double* lpchain_dev, int64_t* naccept_dev) {
    if (!ctx) return GPB_E_ARG;
    if (ctx->pmap_d_in > 0 && !compaction_applies(ctx))
        GPB_FAIL(GPB_E_STATE, "gpb_emcee_run: an emulator with a parameter map needs the block likelihood kernels");
    gpb_ctx* one[1] = {ctx};
    return gpb_chain_emcee_run(one, 1, pos_dev, lp_dev, nwalkers, nsteps, seed, step0, a, randomize_split, lo_dev, hi_dev,
                               outside_value, inside_const, chain_dev, lpchain_dev, naccept_dev);
}

#ifdef GPB_DEBUG_VARIANTS      // test hooks (include/gpbayes_debug.h)
extern "C" int gpb_test_split_perm(gpb_ctx* ctx, int64_t n, uint64_t seed, uint64_t step, int64_t* out_dev) {
    if (!ctx || n < 2 || n > (1ll << 30) || !out_dev) return GPB_E_ARG;
    hipLaunchKernelGGL(k_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<long long*>(out_dev), n, seed, (uint32_t)step, half_bits(n));
    GPB_HIP(hipGetLastError());
    return 0;
}

extern "C" int gpb_test_philox(gpb_ctx* ctx, int64_t n, const uint32_t* in_host, uint32_t* out_host) {
    if (!ctx || n < 1 || n > (1 << 20) || !in_host || !out_host) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    uint32_t *din = nullptr, *dout = nullptr;
    GPB_HIP(hipMalloc(&din, sizeof(uint32_t) * 6 * n));
    GPB_HIP(hipMalloc(&dout, sizeof(uint32_t) * 4 * n));
    GPB_HIP(hipMemcpy(din, in_host, sizeof(uint32_t) * 6 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_philox_test, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, din, dout, n);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out_host, dout, sizeof(uint32_t) * 4 * n, hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    GPB_HIP(e);
    return 0;
}

extern "C" int gpb_test_stretch_draws(gpb_ctx* ctx, int64_t nwalkers, int half, uint64_t seed, uint64_t step,
                                      int randomize_split, double* u_z_dev, int64_t* j_dev, double* u_acc_dev,
                                      int64_t* perm_dev) {
    if (!ctx || nwalkers < 2 || (nwalkers & 1) || nwalkers > (1ll << 30) || (half != 0 && half != 1) || !u_z_dev ||
        !j_dev || !u_acc_dev || !perm_dev)
        return GPB_E_ARG;
    hipLaunchKernelGGL(k_stretch_draws, dim3((unsigned)((nwalkers + 255) / 256)), dim3(256), 0, ctx->stream,
                       nwalkers / 2, half, seed, (uint32_t)step, half_bits(nwalkers), randomize_split ? 1 : 0, u_z_dev,
                       reinterpret_cast<long long*>(j_dev), u_acc_dev, reinterpret_cast<long long*>(perm_dev));
    GPB_HIP(hipGetLastError());
    return 0;
}
#endif  // GPB_DEBUG_VARIANTS
