// gpb_cov.hip — full predictive covariance between query points, per GP:
//     cov_p = k_p(X*, X*) - V_p^T V_p ,   V_p = L_p^-1 K_p(X*, X)^T            (sk:_gpr.py:441-469)
// This is what GPR.predict(return_cov=True) returns and what GPR.sample_y draws from
// (sk:_gpr.py:498-540, src/emulator.py:608-633).  The MCMC path never needs it (only the diagonal,
// gpb_predict.hip); it exists for Emulator.sample_y and FittedGP.predict(return_cov=True), for small W.
#include "gpb_internal.h"
#include "gemm_tile.h"
#include <math.h>

namespace gpb {

template <int KIND>
__device__ __forceinline__ double shape_fn_c(double r2) {
    if (KIND == GPB_KERNEL_RBF) {
        return exp(-0.5 * r2);
    } else if (KIND == GPB_KERNEL_MATERN15) {
        const double t = sqrt(r2) * 1.7320508075688772;
        return (1.0 + t) * exp(-t);
    } else {
        const double t = sqrt(r2) * 2.23606797749979;
        return (1.0 + t + t * t / 3.0) * exp(-t);
    }
}

// V[p][n][w] = sum_{k<=n} Linv[p][n][k] KsT[p][k][w]   (materialised, unlike k_predict)
__global__ __launch_bounds__(256, 2) void k_vmat(const double* __restrict__ Linv, const double* __restrict__ KsT,
                                                 double* __restrict__ V, int64_t Np, int64_t Wld) {
    __shared__ TileLds<128> lds;
    const int p = blockIdx.z;
    const int64_t mb = (int64_t)blockIdx.y * 128, nb = (int64_t)blockIdx.x * 128;
    const int m_ext = (int)imin64(128, Np - mb);
    Acc<128> acc;
    acc_zero<128>(acc);
    gemm_tile_loop<128, false, false>(Linv + (int64_t)p * Np * Np, Np, KsT + (int64_t)p * Np * Wld, Wld, mb, nb,
                                      m_ext, 128, 0, imin64(mb + 128, Np), lds, acc);
    tile_store<128>(V + (int64_t)p * Np * Wld, Wld, mb, nb, m_ext, 128, 1.0, false, acc);
}

// cov[p][i][j] = c k(|xs_i/l - xs_j/l|) (+ sigma_n^2 on the diagonal; unit diagonal forced as sklearn does)
template <int KIND>
__global__ __launch_bounds__(256) void k_kss(const double* __restrict__ Xs, int64_t W, int d,
                                             const double* __restrict__ ls, int dpad, const double* __restrict__ amp,
                                             const double* __restrict__ noise, double* __restrict__ cov,
                                             int64_t Wc) {
    const int p = blockIdx.z;
    const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4), j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t Wld = Wc;
    if (i >= Wld || j >= Wld) return;
    double v = 0.0;
    if (i < W && j < W) {
        if (i == j) {
            v = amp[p] + noise[p];
        } else {
            double r2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double l = ls[p * dpad + k];
                const double df = Xs[i * d + k] / l - Xs[j * d + k] / l;
                r2 = fma(df, df, r2);
            }
            v = amp[p] * shape_fn_c<KIND>(r2);
        }
    }
    cov[((int64_t)p * Wld + i) * Wld + j] = v;
}

// cov[p] -= V_p^T V_p   (TN MFMA GEMM over the Np rows of V)
__global__ __launch_bounds__(256, 2) void k_cov_update(const double* __restrict__ V, double* __restrict__ cov,
                                                       int64_t Np, int64_t Wld, int64_t Wc) {
    __shared__ TileLds<128> lds;
    const int p = blockIdx.z;
    const int64_t mb = (int64_t)blockIdx.y * 128, nb = (int64_t)blockIdx.x * 128;
    const double* Vp = V + (int64_t)p * Np * Wld;
    Acc<128> acc;
    acc_zero<128>(acc);
    gemm_tile_loop<128, true, false>(Vp, Wld, Vp, Wld, mb, nb, 128, 128, 0, Np, lds, acc);
    tile_store<128>(cov + (int64_t)p * Wc * Wc, Wc, mb, nb, 128, 128, -1.0, true, acc);
}

// dst[p][i][j] (ld W) = src[p][i][j] (ld Wld)
__global__ void k_cov_pack(const double* __restrict__ src, double* __restrict__ dst, int64_t W, int64_t Wld) {
    const int p = blockIdx.z;
    const int64_t i = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < W) dst[((int64_t)p * W + i) * W + j] = src[((int64_t)p * Wld + i) * Wld + j];
}

int launch_predict_cov(gpb_ctx* ctx, const double* Xs_dev, int64_t W, double* cov_dev /*[P][W][W]*/) {
    // Wld: leading dimension of K*^T / V (workspace capacity); Wc: padded extent of this batch (cov's own ld)
    const int64_t P = ctx->P, Np = ctx->Np, Wc = round_up(W, WPAD), Wld = Wc;   // launch_predict lays the batch out with ld = Wc
    const int64_t need_v = P * Np * Wld, need_c = P * Wc * Wc;
    if (need_v > ctx->vbuf_cap) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->vbuf) pool_free(ctx->vbuf);
        GPB_HIP(pool_malloc_t(&ctx->vbuf, need_v * sizeof(double)));
        ctx->vbuf_cap = need_v;
    }
    if (need_c > ctx->covbuf_cap) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->covbuf) pool_free(ctx->covbuf);
        GPB_HIP(pool_malloc_t(&ctx->covbuf, need_c * sizeof(double)));
        ctx->covbuf_cap = need_c;
    }
    ctx->want_kst = true;                                    // (the joint covariance reads the fp64 K*^T itself: no digit planes here)
    int rc = launch_predict(ctx, Xs_dev, W, false);          // K*^T and the mean
    ctx->want_kst = false;
    if (rc) return rc;
    dim3 gv((unsigned)(Wc / 128), (unsigned)((Np + 127) / 128), (unsigned)P);
    hipLaunchKernelGGL(k_vmat, gv, dim3(256), 0, ctx->stream, ctx->Linv, ctx->KsT, ctx->vbuf, Np, Wld);
    dim3 gk((unsigned)((Wc + 15) / 16), (unsigned)((Wc + 15) / 16), (unsigned)P);
#define GPB_KSS(KIND)                                                                                        \
    hipLaunchKernelGGL(k_kss<KIND>, gk, dim3(256), 0, ctx->stream, Xs_dev, W, (int)ctx->d, ctx->ls, (int)ctx->dpad, \
                       ctx->amp, ctx->noise, ctx->covbuf, Wc)
    if (ctx->kind == GPB_KERNEL_RBF) GPB_KSS(GPB_KERNEL_RBF);
    else if (ctx->kind == GPB_KERNEL_MATERN15) GPB_KSS(GPB_KERNEL_MATERN15);
    else GPB_KSS(GPB_KERNEL_MATERN25);
#undef GPB_KSS
    dim3 gc((unsigned)(Wc / 128), (unsigned)(Wc / 128), (unsigned)P);
    hipLaunchKernelGGL(k_cov_update, gc, dim3(256), 0, ctx->stream, ctx->vbuf, ctx->covbuf, Np, Wld, Wc);
    dim3 gp((unsigned)((W + 255) / 256), (unsigned)W, (unsigned)P);
    hipLaunchKernelGGL(k_cov_pack, gp, dim3(256), 0, ctx->stream, ctx->covbuf, cov_dev, W, Wc);
    GPB_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpb
