// gpb_fit.hip — fit-side kernels: K(X,X) assembly, blocked right-looking fp64 Cholesky,
// triangular inverse by recursive block doubling, alpha = K^-1 z, log-marginal likelihood
// and its gradient.  Replaces the body of sklearn GPR.fit / log_marginal_likelihood
// (sk:_gpr.py:346-364, 537-652) that the reference calls from src/emulator.py:309-315.
//
// All matrices are padded to Np = multiple of 64 with identity blocks (gp_set_impl, gpb_api.hip: in FRONT of the design in whole
// 16-row units, the remainder behind), so that no kernel has ragged edges in N:  K_pad = diag(I, K, I)  =>  L_pad = diag(I, L, I), same for L^-1.
#include "gpb_internal.h"
#include "gemm_tile.h"
#include "fast_math.h"
#include <math.h>
#include <vector>

namespace gpb {

// ------------------------------------------------------------------ shape functions
template <int KIND>
__device__ __forceinline__ double shape_fn(double r2) {
    if (KIND == GPB_KERNEL_RBF) {
        return exp(-0.5 * r2);                                   // sk:kernels.py:1557
    } else if (KIND == GPB_KERNEL_MATERN15) {
        const double t = sqrt(r2) * 1.7320508075688772;          // sk:kernels.py:1721-1723
        return (1.0 + t) * exp(-t);
    } else {
        const double t = sqrt(r2) * 2.23606797749979;            // sk:kernels.py:1724-1726
        return (1.0 + t + t * t / 3.0) * exp(-t);
    }
}

// ------------------------------------------------------------------ design scaling
// Xsc[p][n][k] = X[n][k] / l_p[k]   (sklearn divides: sk:kernels.py:1556,1564)
__global__ void k_scale_design(const double* __restrict__ X, const double* __restrict__ ls,
                               double* __restrict__ Xsc, int64_t Np, int dpad, const GpSel sel) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (idx >= Np * dpad) return;
    const int k = idx % dpad;
    Xsc[(int64_t)p * Np * dpad + idx] = X[sel.q(p) * sel.x_stride + idx] / ls[p * dpad + k];
}

// Centred copy for the dot-product form of the cross kernel (k_kcross): Xc = X/l - mean/l row by row, and the
// rows' squared norms.  The walkers are shifted by the same muS, so the distances are those of the scaled design.
__global__ void k_center_design(const double* __restrict__ Xsc, const double* __restrict__ xmean,
                                const double* __restrict__ ls, double* __restrict__ muS, double* __restrict__ Xc,
                                double* __restrict__ dnorm, int64_t Np, int dpad, const GpSel sel) {
    const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (n >= Np) return;
    xmean += sel.q(p) * sel.xm_stride;
    const double* src = Xsc + ((int64_t)p * Np + n) * dpad;
    double* dst = Xc + ((int64_t)p * Np + n) * dpad;
    double s = 0.0;
    for (int k = 0; k < dpad; ++k) {
        const double m = xmean[k] / ls[p * dpad + k];
        if (n == 0) muS[p * dpad + k] = m;
        const double v = src[k] - m;
        dst[k] = v;
        s = fma(v, v, s);
    }
    dnorm[(int64_t)p * Np + n] = s;
}

int launch_scale_design(gpb_ctx* ctx) {
    const int64_t tot = ctx->Np * ctx->dpad;
    dim3 grid((unsigned)((tot + 255) / 256), (unsigned)ctx->P);
    hipLaunchKernelGGL(k_scale_design, grid, dim3(256), 0, ctx->stream, ctx->X, ctx->ls, ctx->Xsc,
                       ctx->Np, (int)ctx->dpad, ctx->sel());
    dim3 grid2((unsigned)((ctx->Np + 255) / 256), (unsigned)ctx->P);
    hipLaunchKernelGGL(k_center_design, grid2, dim3(256), 0, ctx->stream, ctx->Xsc, ctx->xmean, ctx->ls, ctx->muS,
                       ctx->Xc, ctx->dnorm, ctx->Np, (int)ctx->dpad, ctx->sel());
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ K(X,X)
// The difference form r^2 = sum_k (a_k - b_k)^2 on X / l, operation for operation sklearn's (pdist on X / length_scale,
// sk:kernels.py:1556,1711): the kernel of the GPs whose length scales are far below the design's extent (gpb_ctx::gpform = 1,
// choose_forms in gpb_api.hip — at the reference's Matern lower bound 1e-3 x extent the Gram form of k_kmat_mfma loses 3e-10
// in K); rounds 1-2 built every K with it (debug build: tune key 39 = 0).
// One 64x64 tile per workgroup, tiles of the lower block triangle only (the factorisation reads nothing above
// it); scaled design rows staged in LDS; HBM-write bound (4*Np^2 bytes per GP).  Diagonal: c*1 + sigma_n^2 + alpha (sk:kernels.py:1559-1560,
// 1401-1412; sk:_gpr.py:347).  Padding rows/cols (gp_set_impl: in front of the design in whole 16-row units, the rest behind): identity.
template <int KIND>
__global__ __launch_bounds__(256) void k_kmat(const double* __restrict__ Xsc, const double* __restrict__ amp,
                                              const double* __restrict__ noise, double alpha_reg,
                                              double* __restrict__ K, const GpSel sel, int64_t Np, int dpad,
                                              const int* __restrict__ form) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int p = blockIdx.z;
    const int64_t pad = pad_front(Np, sel.Nq(sel.q(p))), hi = pad + sel.Nq(sel.q(p));      // the design: rows [pad, hi)
    if (blockIdx.x > blockIdx.y) return;               // lower block triangle only: nothing reads K above it
    if (form && form[p] != 1) return;                  // a Gram-form GP: k_kmat_mfma's
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    const int ldx = dpad + 1;
    double* Xi = sm;
    double* Xj = sm + 64 * ldx;
    const double* Xp = Xsc + (int64_t)p * Np * dpad;
    for (int e = threadIdx.x; e < 64 * dpad; e += 256) {
        const int r = e / dpad, k = e % dpad;
        Xi[r * ldx + k] = Xp[(i0 + r) * dpad + k];
        Xj[r * ldx + k] = Xp[(j0 + r) * dpad + k];
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
    for (int k = 0; k < dpad; ++k) {
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = Xi[(ty + 16 * a) * ldx + k];
#pragma unroll
        for (int b = 0; b < 4; ++b) xj[b] = Xj[(tx + 16 * b) * ldx + k];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double df = xi[a] - xj[b];
                r2[a][b] = fma(df, df, r2[a][b]);
            }
    }
    const double c = amp[p], dg = amp[p] + noise[p] + alpha_reg;
    double* Kp = K + (int64_t)p * Np * Np;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
            double v;
            if (i < pad || j < pad || i >= hi || j >= hi) v = (i == j) ? 1.0 : 0.0;      // the padding (gp_set_impl): identity
            else if (i == j) v = dg;
            else v = c * shape_fn<KIND>(r2[a][b]);
            Kp[i * Np + j] = v;
        }
}

// The same matrix, organised for throughput (the GPs in the Gram form: all of them unless a length scale is extreme):
//   * a 1-D grid over the tiles of the lower block triangle only (no empty workgroups);
//   * r^2 = |a|^2 + |b|^2 - 2 a.b on the centred, length-scaled design (Xc / dnorm, as k_kcross): d multiply-adds per pair
//     instead of d subtractions + d multiply-adds, and the a.b of a 32x32 wave tile as 4 x dpad/4 fp64 MFMAs whose
//     fragments are read from the two operand blocks staged once in LDS (2 x 64 rows x dpad doubles, one barrier); the k
//     loop of the kernel above spent as many LDS cycles on operand reads as VALU cycles on arithmetic.  Centring at the design's column means keeps |a|, |b| small: the cancellation costs ~1e-16 (|a|^2 + |b|^2)
//     absolute in r^2, i.e. <= 1e-15 relative in K for length scales down to a tenth of the design's extent;
//   * the diagonal / padding logic only in the tiles that have a diagonal or padding (a uniform branch): the interior
//     tiles are straight-line code whose 16 exponentials share their constants.
// Bound: max(HBM write 4 Np^2 bytes per GP, fp64 VALU ~ Np^2 / 2 x (dpad + ~30) operations per GP).
// c-free value of one pair from the MFMA's a.b and the staged row terms.  RBF: the rows carry -|a|^2 / 2, so that
// -r^2 / 2 = a.b - (|a|^2 + |b|^2) / 2 is two additions — the bits of -0.5 * max(|a|^2 + |b|^2 - 2 a.b, 0), scaling by 2 being exact
// — one multiplication per pair less in a kernel that runs at the package's power limit.  Matern: r^2 itself is needed.
template <int KIND>
__device__ __forceinline__ double kmat_pair(double ab, double si, double sj) {
    if (KIND == GPB_KERNEL_RBF) return exp_nonpos(fmin(ab + (si + sj), 0.0));
    return shape_fn_fast<KIND>(fmax(fma(-2.0, ab, si + sj), 0.0));
}

template <int KIND, int DPAD>
__global__ __launch_bounds__(256) void k_kmat_mfma(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                                   const double* __restrict__ amp, const double* __restrict__ noise,
                                                   double alpha_reg, double* __restrict__ K, const GpSel sel, int64_t Np,
                                                   const int* __restrict__ form, const int2* __restrict__ tiles) {
    if (form && form[blockIdx.y] != 0) return;         // a difference-form GP: k_kmat's
    const int64_t pad = pad_front(Np, sel.Nq(sel.q(blockIdx.y))), hi = pad + sel.Nq(sel.q(blockIdx.y));     // the design: rows [pad, hi)
    // the tile's two operand blocks (64 design rows x DPAD each, contiguous in Xc) are staged in LDS by coalesced 16-byte
    // loads — fragment-shaped loads straight from global memory touched sixteen 32-byte pieces per instruction, 1920 cache
    // line requests per workgroup — and read back as MFMA fragments (row stride DPAD + 1 doubles: conflict-free)
    constexpr int LDX = DPAD + 1;
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
    // tile t of the lower block triangle, row by row: t = bi (bi + 1) / 2 + bj, bj <= bi — from the context's table (one scalar
    // load; solving for bi with a double sqrt and integer fix-ups in every workgroup cost 9 % of the kernel at N = 4096)
    const int2 tl = tiles[blockIdx.x];
    const int64_t bi = tl.x, bj = tl.y;
    const int64_t i0 = bi * 64, j0 = bj * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
    const double* Xp = Xc + (int64_t)p * Np * DPAD;
    const double* dn = dnorm + (int64_t)p * Np;
    {
        const d2* gi = reinterpret_cast<const d2*>(Xp + i0 * DPAD);
        const d2* gj = reinterpret_cast<const d2*>(Xp + j0 * DPAD);
#pragma unroll
        for (int e = tid; e < 64 * DPAD / 2; e += 256) {
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;       // DPAD is even: a pair never straddles two rows
            const d2 vi = gi[e], vj = gj[e];
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
            sXj[r * LDX + k] = vj.x; sXj[r * LDX + k + 1] = vj.y;
        }
        constexpr double RS = KIND == GPB_KERNEL_RBF ? -0.5 : 1.0;      // see kmat_pair
        if (tid < 64) sdi[tid] = RS * dn[i0 + tid];
        else if (tid < 128) sdj[tid - 64] = RS * dn[j0 + tid - 64];
    }
    __syncthreads();
    constexpr int KG = DPAD / 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    // A fragment: lane holds A[row lr][k lk]; B fragment: B[k lk][col lr] = Xc[col][k]
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        double fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fa[a] = sXi[(m0 + 16 * a + lr) * LDX + 4 * g + lk];
            fb[a] = sXj[(n0 + 16 * a + lr) * LDX + 4 * g + lk];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    const bool special = bi == bj || j0 < pad || i0 + 64 > hi;      // a diagonal or padding in this tile (j0 <= i0)
    if (!special) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] =
                        c * kmat_pair<KIND>(acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r], dj);
                __builtin_amdgcn_sched_barrier(0);      // four evaluations in flight, not sixteen: registers for occupancy
            }
        return;
    }
    const double dg = amp[p] + noise[p] + alpha_reg;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                double v;
                if (i < pad || j < pad || i >= hi || j >= hi) v = (i == j) ? 1.0 : 0.0;      // the padding (gp_set_impl): identity
                else if (i == j) v = dg;
                else v = c * kmat_pair<KIND>(acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r], dj);
                Kp[i * Np + j] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int KIND>
static void launch_kmat_mfma(gpb_ctx* ctx) {
    const int64_t nb = ctx->Np / 64;
    dim3 grid((unsigned)(nb * (nb + 1) / 2), (unsigned)ctx->P);
#define GPB_KM(DP)                                                                                                  \
    hipLaunchKernelGGL((k_kmat_mfma<KIND, DP>), grid, dim3(256), 0, ctx->stream, ctx->Xc, ctx->dnorm, ctx->amp,     \
                       ctx->noise, ctx->alpha_reg, ctx->K, ctx->sel(), ctx->Np, ctx->n_diff > 0 ? ctx->gpform : nullptr,     \
                       reinterpret_cast<const int2*>(ctx->kmtiles))
    switch (ctx->dpad) {
        case 8: GPB_KM(8); break;
        case 16: GPB_KM(16); break;
        case 20: GPB_KM(20); break;
        case 24: GPB_KM(24); break;
        case 32: GPB_KM(32); break;
        case 48: GPB_KM(48); break;
        default: GPB_KM(64); break;
    }
#undef GPB_KM
}

static void launch_kmat_diff(gpb_ctx* ctx, const int* form) {
    dim3 grid((unsigned)(ctx->Np / 64), (unsigned)(ctx->Np / 64), (unsigned)ctx->P);
    const size_t sh = 2 * 64 * (ctx->dpad + 1) * sizeof(double);
#define GPB_KMAT(KIND)                                                                          \
    hipLaunchKernelGGL(k_kmat<KIND>, grid, dim3(256), sh, ctx->stream, ctx->Xsc, ctx->amp,       \
                       ctx->noise, ctx->alpha_reg, ctx->K, ctx->sel(), ctx->Np, (int)ctx->dpad, form)
    if (ctx->kind == GPB_KERNEL_RBF) GPB_KMAT(GPB_KERNEL_RBF);
    else if (ctx->kind == GPB_KERNEL_MATERN15) GPB_KMAT(GPB_KERNEL_MATERN15);
    else GPB_KMAT(GPB_KERNEL_MATERN25);
#undef GPB_KMAT
}

int launch_kmat(gpb_ctx* ctx) {
    // each GP in its distance form (gpform); a launch with no GP is left out
    if (ctx->n_diff < ctx->P) {
        if (ctx->kind == GPB_KERNEL_RBF) launch_kmat_mfma<GPB_KERNEL_RBF>(ctx);
        else if (ctx->kind == GPB_KERNEL_MATERN15) launch_kmat_mfma<GPB_KERNEL_MATERN15>(ctx);
        else launch_kmat_mfma<GPB_KERNEL_MATERN25>(ctx);
    }
    if (ctx->n_diff > 0) launch_kmat_diff(ctx, ctx->n_diff < ctx->P ? ctx->gpform : nullptr);
    GPB_HIP(hipGetLastError());
    return 0;
}


// Trailing update (SYRK on MFMA): A[i][j] -= sum_{k in [c0, c0+kw)} L[i][k] L[j][k] for rows i >= r0 and
// columns j in [r0, ce), lower part only (tiles entirely above the diagonal exit).  Two-level blocking:
// inside an outer panel of `NBO` columns the update after every 64-column step touches the panel's own
// remaining columns only (kw = 64, ce = panel end); once per outer panel the whole trailing matrix is
// updated with kw = NBO.  A K=64 update moves 384 KB per 2.1 MFLOP tile and is HBM-bound; K=256 is not.
template <int T>
__global__ __launch_bounds__(256, 2) void k_syrk(double* __restrict__ K, int64_t Np, int64_t c0, int kw,
                                                 int64_t r0, int64_t ce) {
    __shared__ TileLds<T> lds;
    const int p = blockIdx.z;
    const int64_t mb = (int64_t)blockIdx.y * T, nb = (int64_t)blockIdx.x * T;       // relative to (r0, r0)
    if (mb + T <= nb) return;                          // entirely above the diagonal
    const int m_ext = (int)imin64(T, Np - r0 - mb), n_ext = (int)imin64(T, ce - r0 - nb);
    if (m_ext <= 0 || n_ext <= 0) return;
    double* Kp = K + (int64_t)p * Np * Np;
    const double* Pn = Kp + r0 * Np + c0;              // panel rows r0.., cols c0..c0+kw
    Acc<T> acc;
    acc_zero<T>(acc);
    gemm_tile_loop<T, false, true>(Pn, Np, Pn, Np, mb, nb, m_ext, n_ext, 0, kw, lds, acc);
    tile_store<T>(Kp + r0 * Np + r0, Np, mb, nb, m_ext, n_ext, -1.0, true, acc);
}

// Trailing update by the outer panel [pb, pe), K = panel width: rows >= r0, columns [r0, ce) (lower part).
void launch_syrk_range(gpb_ctx* ctx, hipStream_t stream, int64_t pb, int64_t pe, int64_t r0, int64_t ce) {
    const int64_t Np = ctx->Np;
    const int64_t ntr = (Np - r0 + 127) / 128, ntc = (ce - r0 + 127) / 128;
    const int64_t tiles128 = ntc * ntr - ntc * (ntc - 1) / 2;
    const bool small = ctx->syrk_tile == 64 || (ctx->syrk_tile == 0 && tiles128 * ctx->P < 16 * (int64_t)ctx->num_cu);
    if (small) {                                        // few 128-wide tiles: 64-wide ones fill the chip (same bits)
        const unsigned nr = (unsigned)((Np - r0 + 63) / 64), nc = (unsigned)((ce - r0 + 63) / 64);
        hipLaunchKernelGGL(k_syrk<64>, dim3(nc, nr, (unsigned)ctx->P), dim3(256), 0, stream, ctx->K, Np, pb,
                           (int)(pe - pb), r0, ce);
    } else {
        hipLaunchKernelGGL(k_syrk<128>, dim3((unsigned)ntc, (unsigned)ntr, (unsigned)ctx->P), dim3(256), 0, stream,
                           ctx->K, Np, pb, (int)(pe - pb), r0, ce);
    }
}
// End of an outer panel [pb, pe): the whole trailing matrix takes the panel's update.
void launch_syrk_panel(gpb_ctx* ctx, int64_t pb, int64_t pe) { launch_syrk_range(ctx, ctx->stream, pb, pe, pe, ctx->Np); }

int launch_potrf_fused(gpb_ctx* ctx);                  // gpb_chol.hip: two launches per 64-column step

int launch_potrf(gpb_ctx* ctx) {
    return launch_potrf_fused(ctx);
}

// ------------------------------------------------------------------ L^-1 by block doubling
// inv([[A,0],[C,B]]) = [[A^-1,0],[-B^-1 C A^-1, B^-1]].  Level hs: the hs-sized diagonal blocks
// of Linv are complete; phase 1: T = C A^-1, phase 2: X21 = -B^-1 T.  Both are NN MFMA GEMMs
// whose K range is clipped by the triangular operand.
template <int PHASE, int T_>
__global__ __launch_bounds__(256, 2) void k_trtri_level(const double* __restrict__ L, double* __restrict__ Linv,
                                                        double* __restrict__ T, int64_t Np, int64_t hs,
                                                        int ngroups) {
    __shared__ TileLds<T_> lds;
    const int p = blockIdx.y / ngroups, g = blockIdx.y % ngroups;
    const int64_t c0 = (int64_t)g * 2 * hs, r0 = c0 + hs;
    const int64_t n2 = imin64(hs, Np - r0);
    // longest K loops first: the dispatcher hands out blocks with z slowest, so z carries the index the K range
    // depends on (phase 1: k in [nb, hs), longest at nb = 0; phase 2: k in [0, mb + T), longest at the last mb).
    // In x-major order the last long tile started when the chip was already draining: 343 -> us at hs = 1024.
    const int64_t mb = (int64_t)(PHASE == 1 ? blockIdx.x : gridDim.z - 1 - blockIdx.z) * T_;
    const int64_t nb = (int64_t)(PHASE == 1 ? blockIdx.z : blockIdx.x) * T_;
    if (n2 <= 0 || mb >= n2 || nb >= hs) return;
    const int m_ext = (int)imin64(T_, n2 - mb), n_ext = (int)imin64(T_, hs - nb);
    const int64_t off = (int64_t)p * Np * Np;
    Acc<T_> acc;
    acc_zero<T_>(acc);
    if (PHASE == 1) {
        // T[r0+m][c0+n] = sum_{k>=n} L[r0+m][c0+k] * Linv[c0+k][c0+n]
        gemm_tile_loop<T_, false, false>(L + off + r0 * Np + c0, Np, Linv + off + c0 * Np + c0, Np, mb, nb, m_ext,
                                         n_ext, nb, hs, lds, acc);
        tile_store<T_>(T + off + r0 * Np + c0, Np, mb, nb, m_ext, n_ext, 1.0, false, acc);
    } else {
        // Linv[r0+m][c0+n] = -sum_{k<=m} Linv[r0+m][r0+k] * T[r0+k][c0+n]
        const int64_t k_end = imin64(mb + T_, n2);
        gemm_tile_loop<T_, false, false>(Linv + off + r0 * Np + r0, Np, T + off + r0 * Np + c0, Np, mb, nb, m_ext,
                                         n_ext, 0, k_end, lds, acc);
        tile_store<T_>(Linv + off + r0 * Np + c0, Np, mb, nb, m_ext, n_ext, -1.0, false, acc);
    }
}

int launch_trtri(gpb_ctx* ctx) {
    const int64_t Np = ctx->Np;
    ctx->slA_valid = false;                            // the digit planes of L^-1 (gpb_sliced.hip) are rebuilt on their next use
    for (int64_t hs = 64; hs < Np; hs *= 2) {
        const int ngroups = (int)((Np + 2 * hs - 1) / (2 * hs));
        // 64-wide tiles while 128-wide ones would leave the chip underfilled or badly quantised (measured faster up
        // to ~10 tiles of 128 per CU: N = 2048 4.58 -> 4.17 ms, N = 4096 17.96 -> 17.54 ms for the whole fit);
        // the k order of every element's sum does not depend on the tile size: same bits either way
        const int64_t tl128 = (hs + 127) / 128;
        const int64_t tiles128 = tl128 * tl128 * ngroups * ctx->P;
        const bool small = ctx->trtri_tile == 64 || (ctx->trtri_tile == 0 && tiles128 < 16 * (int64_t)ctx->num_cu);
        if (small) {
            const unsigned tl = (unsigned)((hs + 63) / 64);
            dim3 grid(tl, (unsigned)(ngroups * ctx->P), tl);
            hipLaunchKernelGGL((k_trtri_level<1, 64>), grid, dim3(256), 0, ctx->stream, ctx->K, ctx->Linv, ctx->T, Np,
                               hs, ngroups);
            hipLaunchKernelGGL((k_trtri_level<2, 64>), grid, dim3(256), 0, ctx->stream, ctx->K, ctx->Linv, ctx->T, Np,
                               hs, ngroups);
        } else {
            dim3 grid((unsigned)tl128, (unsigned)(ngroups * ctx->P), (unsigned)tl128);
            hipLaunchKernelGGL((k_trtri_level<1, 128>), grid, dim3(256), 0, ctx->stream, ctx->K, ctx->Linv, ctx->T,
                               Np, hs, ngroups);
            hipLaunchKernelGGL((k_trtri_level<2, 128>), grid, dim3(256), 0, ctx->stream, ctx->K, ctx->Linv, ctx->T,
                               Np, hs, ngroups);
        }
    }
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ alpha = L^-T (L^-1 z)
// y_i = sum_{k<=i} Linv[i][k] z_k : one wave per row, fixed-order shuffle reduction.
__global__ __launch_bounds__(256) void k_lower_matvec(const double* __restrict__ Linv, const double* __restrict__ z,
                                                      double* __restrict__ y, int64_t Np, const GpSel sel) {
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double* row = Linv + (int64_t)p * Np * Np + i * Np;
    const double* zp = z + (int64_t)sel.q(p) * Np;
    double s = 0.0;
    for (int64_t k = lane; k <= i; k += 64) s = fma(row[k], zp[k], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[(int64_t)p * Np + i] = s;
}
// alpha_j = sum_{i>=j} Linv[i][j] y_i : 64 columns per workgroup, rows strided over 16 waves.
constexpr int MVT_WAVES = 16;
__global__ __launch_bounds__(64 * MVT_WAVES) void k_lower_matvec_t(const double* __restrict__ Linv, const double* __restrict__ y,
                                                                  double* __restrict__ out, int64_t Np) {
    __shared__ double red[MVT_WAVES][64];
    const int p = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64, j = j0 + lane;
    const double* Lp = Linv + (int64_t)p * Np * Np;
    const double* yp = y + (int64_t)p * Np;
    // The kernel is pure load latency (a column block is Np / 64 workgroups x P: fewer than the chip has CUs up to N = 1024):
    // four independent chains per wave (rows i = j0 + wave + 16 t, chain t mod 4) and sixteen waves keep 64 row loads of
    // a column block in flight (four waves: 26 us at N = 1024, 59 at 2048).  Fixed order: reproducible.
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = j0 + wave;
    constexpr int64_t S = MVT_WAVES;
    for (; i + 3 * S < Np; i += 4 * S) {
        s0 = fma(Lp[i * Np + j], yp[i], s0);
        s1 = fma(Lp[(i + S) * Np + j], yp[i + S], s1);
        s2 = fma(Lp[(i + 2 * S) * Np + j], yp[i + 2 * S], s2);
        s3 = fma(Lp[(i + 3 * S) * Np + j], yp[i + 3 * S], s3);
    }
    for (; i < Np; i += S) s0 = fma(Lp[i * Np + j], yp[i], s0);
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0) {
        double s = red[0][lane];
#pragma unroll
        for (int w = 1; w < MVT_WAVES; ++w) s += red[w][lane];
        out[(int64_t)p * Np + j] = s;
    }
}

int launch_alpha(gpb_ctx* ctx) {
    hipLaunchKernelGGL(k_lower_matvec, dim3((unsigned)(ctx->Np / 4), (unsigned)ctx->P), dim3(256), 0, ctx->stream,
                       ctx->Linv, ctx->Z, ctx->yv, ctx->Np, ctx->sel());
    hipLaunchKernelGGL(k_lower_matvec_t, dim3((unsigned)(ctx->Np / 64), (unsigned)ctx->P), dim3(64 * MVT_WAVES), 0,
                       ctx->stream, ctx->Linv, ctx->yv, ctx->alpha, ctx->Np);
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ LML value
// lml = -1/2 z.alpha - sum log L_ii - N/2 log 2pi   (sk:_gpr.py:609-611)
__global__ __launch_bounds__(256) void k_lml_value(const double* __restrict__ L, const double* __restrict__ z,
                                                   const double* __restrict__ alpha, double* __restrict__ out,
                                                   const GpSel sel, int64_t Np) {
    __shared__ double r1[256], r2[256];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int q = sel.q(p);
    const int64_t N = sel.Nq(q);
    const double* Lp = L + (int64_t)p * Np * Np;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = tid; i < Np; i += 256) {
        s1 = fma(z[(int64_t)q * Np + i], alpha[(int64_t)p * Np + i], s1);
        s2 += log(Lp[i * Np + i]);
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[p * 4 + 0] = -0.5 * r1[0] - r2[0] - 0.5 * (double)N * 1.8378770664093453;  // log(2 pi)
        out[p * 4 + 1] = r1[0];
        out[p * 4 + 2] = r2[0];
    }
}

int launch_lml_value(gpb_ctx* ctx) {
    hipLaunchKernelGGL(k_lml_value, dim3((unsigned)ctx->P), dim3(256), 0, ctx->stream, ctx->K, ctx->Z, ctx->alpha,
                       ctx->lmlbuf, ctx->sel(), ctx->Np);
    GPB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ LML gradient
// K^-1 = L^-T L^-1 on the lower 128x128 tiles (TN MFMA GEMM, k >= m_base), into T.
// 1-D grid, GP fastest (round 5): tile t — the tiles are numbered longest K loop first — of ALL GPs is dispatched before tile t + 1
// of any; with the GP as grid.y the last GP's heaviest tiles started when the chip was already draining (as in k_chol_update).
// T_ = 128 or 64 (the tile; an element's sum runs over k in the same order either way: same bits).
template <int T_>
__global__ __launch_bounds__(256, 2) void k_kinv(const double* __restrict__ Linv, double* __restrict__ T,
                                                 int64_t Np, unsigned P) {
    __shared__ TileLds<T_> lds;
    const int p = (int)(blockIdx.x % P), t = (int)(blockIdx.x / P);
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int64_t mb = (int64_t)ti * T_, nb = (int64_t)tj * T_;
    const int m_ext = (int)imin64(T_, Np - mb), n_ext = (int)imin64(T_, Np - nb);
    const double* Lp = Linv + (int64_t)p * Np * Np;
    Acc<T_> acc;
    acc_zero<T_>(acc);
    gemm_tile_loop<T_, true, false>(Lp, Np, Lp, Np, mb, nb, m_ext, n_ext, mb, Np, lds, acc);
    tile_store<T_>(T + (int64_t)p * Np * Np, Np, mb, nb, m_ext, n_ext, 1.0, false, acc);
}

// grad_t = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta_t  (sk:_gpr.py:625-647), never
// materialising dK: each 64x64 lower tile recomputes the kernel and its d+2 derivatives
// (sk:kernels.py:1574-1580, 1762-1766, 1276-1289, 1401-1410) and reduces them on the fly.
template <int KIND, int DPAD>
__global__ __launch_bounds__(256) void k_lml_grad(const double* __restrict__ Xsc, const double* __restrict__ amp,
                                                  const double* __restrict__ noise,
                                                  const double* __restrict__ alpha, const double* __restrict__ Kinv,
                                                  double* __restrict__ gpart, const GpSel sel, int64_t Np, int d,
                                                  int ntiles) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int p = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int64_t pad = pad_front(Np, sel.Nq(sel.q(p))), hi = pad + sel.Nq(sel.q(p));
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int64_t i0 = (int64_t)ti * 64, j0 = (int64_t)tj * 64;
    constexpr int ldx = DPAD + 1;
    double* Xi = sm;
    double* Xj = sm + 64 * ldx;
    double* red = sm + 128 * ldx;            // [4 waves][DPAD+2]
    const double* Xp = Xsc + (int64_t)p * Np * DPAD;
    for (int e = tid; e < 64 * DPAD; e += 256) {
        const int r = e / DPAD, k = e % DPAD;
        Xi[r * ldx + k] = Xp[(i0 + r) * DPAD + k];
        Xj[r * ldx + k] = Xp[(j0 + r) * DPAD + k];
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const double c = amp[p];
    const double* ap = alpha + (int64_t)p * Np;
    const double* Kv = Kinv + (int64_t)p * Np * Np;
    double g[DPAD + 2];
#pragma unroll
    for (int k = 0; k < DPAD + 2; ++k) g[k] = 0.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const int li = ty + 16 * a, lj = tx + 16 * b;
            const int64_t i = i0 + li, j = j0 + lj;
            if (i < pad || j < pad || i >= hi || j > i) continue;           // (the design: rows [pad, hi), gp_set_impl)
            const double w = (ap[i] * ap[j] - Kv[i * Np + j]) * ((i == j) ? 0.5 : 1.0);
            double D[DPAD];
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < DPAD; ++k) {
                const double df = Xi[li * ldx + k] - Xj[lj * ldx + k];
                D[k] = df * df;
                r2 += D[k];
            }
            double ks, gl;   // ks: unit-amplitude kernel; gl: factor multiplying c*D_k in dK/dlog l_k
            if (i == j) { ks = 1.0; gl = 0.0; }
            else if (KIND == GPB_KERNEL_RBF) { ks = exp(-0.5 * r2); gl = ks; }
            else if (KIND == GPB_KERNEL_MATERN15) {
                const double tt = sqrt(3.0 * r2), e = exp(-tt);
                ks = (1.0 + sqrt(r2) * 1.7320508075688772) * exp(-sqrt(r2) * 1.7320508075688772);
                gl = 3.0 * e;
            } else {
                const double tt = sqrt(5.0 * r2), e = exp(-tt);
                const double t2 = sqrt(r2) * 2.23606797749979;
                ks = (1.0 + t2 + t2 * t2 / 3.0) * exp(-t2);
                gl = (5.0 / 3.0) * (tt + 1.0) * e;
            }
            g[0] = fma(w, c * ks, g[0]);
            const double wl = w * c * gl;
#pragma unroll
            for (int k = 0; k < DPAD; ++k) g[1 + k] = fma(wl, D[k], g[1 + k]);
            if (i == j) g[DPAD + 1] = fma(w, noise[p], g[DPAD + 1]);
        }
    // fixed-order reduction: lanes (shuffle tree) -> waves (LDS) -> tile partial
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < DPAD + 2; ++k) {
        double v = g[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave * (DPAD + 2) + k] = v;
    }
    __syncthreads();
    if (tid < DPAD + 2) {
        const double v = ((red[tid] + red[(DPAD + 2) + tid]) + red[2 * (DPAD + 2) + tid]) + red[3 * (DPAD + 2) + tid];
        // output slots: [0]=log c, [1..d]=log l, [d+1]=log noise
        int slot = -1;
        if (tid == 0) slot = 0;
        else if (tid <= d) slot = tid;
        else if (tid == DPAD + 1) slot = d + 1;
        if (slot >= 0) gpart[((int64_t)p * ntiles + t) * (d + 2) + slot] = v;
    }
}

// grad[p][k] = sum over the tiles' partials, in a fixed order: 8 interleaved chunks of tiles x 4 independent chains per thread
// (one thread per parameter walking all N (N + 64) / 8192 tiles in one dependent chain took 129 us at N = 2048, 0.5 ms at 4096:
// pure load latency), then the eight chunk sums in order.
__global__ __launch_bounds__(256) void k_grad_final(const double* __restrict__ gpart, double* __restrict__ grad,
                                                    int ntiles, int d) {
    __shared__ double red[8][32];
    const int p = blockIdx.x, k = blockIdx.y * 32 + (threadIdx.x & 31), c = threadIdx.x >> 5;
    const int nk = d + 2;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (k < nk) {
        const double* g = gpart + (int64_t)p * ntiles * nk + k;
        int t = c;
        for (; t + 24 < ntiles; t += 32) {
            s0 += g[(int64_t)t * nk];
            s1 += g[(int64_t)(t + 8) * nk];
            s2 += g[(int64_t)(t + 16) * nk];
            s3 += g[(int64_t)(t + 24) * nk];
        }
        for (; t < ntiles; t += 8) s0 += g[(int64_t)t * nk];
    }
    red[c][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (c == 0 && k < nk) {
        double s = red[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += red[i][threadIdx.x];
        grad[p * nk + k] = s;   // 1/2 and the symmetric factor 2 are folded into w
    }
}

template <int KIND>
static int launch_grad_kind(gpb_ctx* ctx, int ntiles, double* gfinal) {
    dim3 grid((unsigned)ntiles, (unsigned)ctx->P);
#define GPB_GRAD(DP)                                                                              \
    hipLaunchKernelGGL((k_lml_grad<KIND, DP>), grid, dim3(256), (128 * (DP + 1) + 4 * (DP + 2)) * sizeof(double), \
                       ctx->stream, ctx->Xsc, ctx->amp, ctx->noise, ctx->alpha, ctx->T, ctx->gpart, ctx->sel(),   \
                       ctx->Np, (int)ctx->d, ntiles)
    switch (ctx->dpad) {
        case 8: GPB_GRAD(8); break;
        case 16: GPB_GRAD(16); break;
        case 20: GPB_GRAD(20); break;
        case 24: GPB_GRAD(24); break;
        case 32: GPB_GRAD(32); break;
        case 48: GPB_GRAD(48); break;
        default: GPB_GRAD(64); break;
    }
#undef GPB_GRAD
    hipLaunchKernelGGL(k_grad_final, dim3((unsigned)ctx->P, (unsigned)((ctx->d + 2 + 31) / 32)), dim3(256), 0, ctx->stream,
                       ctx->gpart, gfinal, ntiles, (int)ctx->d);
    return 0;
}

int launch_lml_grad(gpb_ctx* ctx, double* grad_dev) {
    const int64_t nt128 = (ctx->Np + 127) / 128;
    // 64-wide tiles while 128-wide ones would leave the chip underfilled (the rule of launch_trtri; option key 50 forces one)
    const bool small = ctx->kinv_tile == 64 || (ctx->kinv_tile == 0 && nt128 * (nt128 + 1) / 2 * ctx->P < 16 * (int64_t)ctx->num_cu);
    if (small) {
        const int64_t nt = ctx->Np / 64;
        hipLaunchKernelGGL(k_kinv<64>, dim3((unsigned)(nt * (nt + 1) / 2 * ctx->P)), dim3(256), 0, ctx->stream, ctx->Linv, ctx->T,
                           ctx->Np, (unsigned)ctx->P);
    } else {
        hipLaunchKernelGGL(k_kinv<128>, dim3((unsigned)(nt128 * (nt128 + 1) / 2 * ctx->P)), dim3(256), 0, ctx->stream, ctx->Linv,
                           ctx->T, ctx->Np, (unsigned)ctx->P);
    }
    const int64_t nt64 = ctx->Np / 64;
    const int ntiles = (int)(nt64 * (nt64 + 1) / 2);
    const int64_t need = (int64_t)ctx->P * ntiles * (ctx->d + 2);
    if (need > ctx->gpart_cap) {
        if (ctx->gpart) pool_free(ctx->gpart);
        GPB_HIP(pool_malloc_t(&ctx->gpart, need * sizeof(double)));
        ctx->gpart_cap = need;
    }
    if (ctx->kind == GPB_KERNEL_RBF) launch_grad_kind<GPB_KERNEL_RBF>(ctx, ntiles, grad_dev);
    else if (ctx->kind == GPB_KERNEL_MATERN15) launch_grad_kind<GPB_KERNEL_MATERN15>(ctx, ntiles, grad_dev);
    else launch_grad_kind<GPB_KERNEL_MATERN25>(ctx, ntiles, grad_dev);
    GPB_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpb
