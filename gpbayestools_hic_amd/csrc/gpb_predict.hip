// gpb_predict.hip — the MCMC hot loop's GP part:  for every GP p and walker w
//     mean[p][w] = K*(x_w, X) alpha_p                     (sk:_gpr.py:443-444)
//     var[p][w]  = c_p + sigma_n,p^2 - || L_p^-1 K*(x_w, X)^T ||^2   (sk:_gpr.py:454-460, diag only:
//                                                                   src/emulator.py:573-575)
// Three kernels per batch:
//   k_kcross    K*^T[p][n][w] assembled once (coalesced along w), mean partials per 256-point chunk
//   k_predict   V = L^-1 K*^T on the fp64 matrix cores, lower-triangular K-clipping, fused
//               per-walker sum of squares — V is never written to HBM
//   k_finalize  fixed-order sums of the partials
// The reference instead forms a W x W covariance per GP (sk:_gpr.py:460) and keeps its diagonal.
// Reduction orders are fixed by (N, tile sizes) only — never by W or by the rank's shard — so a
// walker's result is bit-identical however the ensemble is split across GPUs.
#include "gpb_internal.h"
#include "gemm_tile.h"
#include "fast_math.h"
#include <math.h>

// the K*^T store of k_kcross: with the streaming hint (global_store ... nt).  K*^T is written once and read back by k_predict; its
// lines need not push the design rows and alpha out of the L2 while the kernel runs.  A/B against plain stores in one process and
// bench against bench (profiles/r04_nt_store_ab.txt): k_kcross a few us faster, k_predict unchanged (1.3152 -> 1.3136 ms),
// the step 0.4 % shorter at cfg 4 and 0.5 % at cfg 3; same bits.  (The same hint on K(X,X) makes the K build 2-11 % faster and
// the Cholesky that reads K next slower by more: not taken there.)
#define GPB_KSTAR_STORE(p, v) __builtin_nontemporal_store((v), (p))

namespace gpb {

template <int KIND>
__device__ __forceinline__ double shape_fn_p(double r2) {
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

// grid (chunk groups, P, walker tiles), 256 threads: lane = walker, the 4 waves stride the 64-point chunk's
// design points; the design row is wave-uniform (scalar loads), the walker row lives in VGPRs.
// DOT: r^2 = |a|^2 + |b|^2 - 2 a.b with a = design row / l - mu / l (Xc, its norms in dnorm) and b = walker / l -
// mu / l: d multiply-adds per pair instead of d subtractions + d multiply-adds (the kernel is bound by the fp64
// VALU; 78 -> 58 instructions per design row).  Centring at the design's column means keeps |a|, |b| small: the
// cancellation costs ~1e-16 (|a|^2 + |b|^2) absolute in r^2, <= 1e-15 relative in K* for length scales down to a
// tenth of the design's extent (2e-13 at sklearn's lower search bound); measured against the oracle the two forms
// are indistinguishable (1e-13 between them).  The form is chosen PER GP from theta alone (choose_forms, gpb_api.hip: the
// difference form — sklearn's own — for a GP with a length scale far below the design's extent, e.g. at the Matern lower
// search bound 1e-3 x extent, where the Gram form loses 3e-10 in K*), never from the batch, so a walker's result still
// does not depend on how the ensemble is split.
// LDS of a k_kcross workgroup (the kernels own it; the body below is shared by the one-emulator and the chain kernels)
template <int DPAD, int WPL>
struct KxLds {
    double red[4][64 * WPL];
    double sdn[KX_CHUNK];
    __attribute__((aligned(16))) double sbuf[64 * WPL * (DPAD + 1)];       // walker tile, then the design rows
    double sal[KX_CHUNK];
};

// SLICE (option key 51, gpb_sliced.hip): K*^T leaves the kernel as the six int8 digit planes the int8 predict kernel reads —
// plane[p][t][k / 16][walker][16 bytes] — instead of fp64 (6 bytes per element instead of 8, no second pass over the batch).  A
// wave then owns 16 CONSECUTIVE design points of the chunk (one 16-byte granule per plane and walker) instead of every fourth,
// so the four mean partials of a chunk group the points differently: the mean's last bits differ from the fp64 form's.
template <int KIND, int DPAD, bool DOT, int WPL, bool SLICE = false>
__device__ __forceinline__ void kcross_body(KxLds<DPAD, WPL>& L, const double* __restrict__ Xs, int64_t W, int d,
                                            const double* __restrict__ Xsc, const double* __restrict__ ls,
                                            const double* __restrict__ amp, const double* __restrict__ alpha,
                                            double* __restrict__ KsT, double* __restrict__ mpart, int64_t N, int64_t Np,
                                            int64_t Wld, int P, const double* __restrict__ dnorm,
                                            const double* __restrict__ muS, int chunks_per_wg,
                                            const int* __restrict__ nrows, const int p,
                                            int8_t* __restrict__ planes = nullptr, const double* __restrict__ colscale = nullptr) {
    constexpr int WT = 64 * WPL;                        // walkers per workgroup: lane l holds walkers l, l + 64, ...
    // compacted batches (gpb_logpost): only the first *nrows rows (those inside the prior box) exist; the launch
    // geometry was sized for the whole batch and the workgroups beyond them leave at once
    if (nrows) {
        W = *nrows;
        if ((int64_t)blockIdx.z * WT >= W) return;
    }
    double (*red)[WT] = L.red;
    double* sdn = L.sdn;
    double* sbuf = L.sbuf;
    double* sal = L.sal;
    double (*sx)[DPAD + 1] = reinterpret_cast<double (*)[DPAD + 1]>(sbuf);
    double* sxr = sbuf;                                 // [KX_CHUNK][DPAD] design rows / length scale
    // grid: x = chunk group, y = GP, z = walker tile — the walker tile is the SLOWEST index, so that the workgroups of a
    // compacted batch's empty walker tiles (the launch is sized for the whole batch) are dispatched after every live one
    // instead of in between (compacted ~1000 of 2048 rows: 83.5 us against 55 for a full 1024-row batch before)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = (int64_t)blockIdx.z * WT;
    const int64_t nchunk = Np / KX_CHUNK;               // Np is a multiple of KX_CHUNK: chunks are always whole
    const int64_t chunk0 = (int64_t)blockIdx.x * chunks_per_wg;
    const int64_t chunk1 = imin64(chunk0 + chunks_per_wg, nchunk);
    // walker tile / length scale, loaded coalesced and divided once per element, then WPL rows per lane;
    // the design rows are one contiguous block: coalesced into LDS, read back as broadcasts (scalar loads
    // of 64 x DPAD doubles per workgroup miss the scalar cache and serialise on L2 latency).  Every broadcast
    // feeds WPL multiply-adds (the LDS reads, not the VALU, bound the one-walker form).  A workgroup
    // walks `chunks_per_wg` chunks with its walker tile in registers (the set-up — fp64 divisions, three
    // barriers, two round trips to L2 — is as long as one chunk's arithmetic), the next chunk's rows in
    // flight meanwhile.
    constexpr int NPRE = (KX_CHUNK * DPAD + 255) / 256;
    double pre[NPRE], pal = 0.0, pdn = 0.0;             // the next chunk's rows / alpha / row norms in flight
    auto fetch = [&](int64_t chunk) {
        const double* Xp = Xsc + ((int64_t)p * Np + chunk * KX_CHUNK) * DPAD;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int e = threadIdx.x + 256 * j;
            pre[j] = (e < KX_CHUNK * DPAD) ? Xp[e] : 0.0;
        }
        if (threadIdx.x < KX_CHUNK) {
            pal = alpha[(int64_t)p * Np + chunk * KX_CHUNK + threadIdx.x];
            if (DOT) pdn = dnorm[(int64_t)p * Np + chunk * KX_CHUNK + threadIdx.x];
        }
    };
    fetch(chunk0);
    {   // all of a thread's loads first, then the divisions (one loop kept load -> divide -> store per element in order)
        constexpr int NLD = (WT * DPAD + 255) / 256;
        double xv[NLD], lv[NLD], mv[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + 256 * j, r = e / DPAD, k = e - r * DPAD;
            const bool ok = e < WT * DPAD && k < d && w0 + r < W;
            xv[j] = ok ? Xs[(w0 + r) * d + k] : 0.0;
            lv[j] = ok ? ls[p * DPAD + k] : 1.0;
            mv[j] = (DOT && ok) ? muS[p * DPAD + k] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + 256 * j, r = e / DPAD, k = e - r * DPAD;
            if (e < WT * DPAD) {
                const bool ok = k < d && w0 + r < W;
                double v = ok ? xv[j] / lv[j] : 0.0;
                if (DOT && ok) v -= mv[j];
                sx[r][k] = v;
            }
        }
    }
    __syncthreads();
    double xs[WPL][DPAD];
    double nb[WPL];                                     // |b|^2 of this lane's walkers (DOT)
#pragma unroll
    for (int u = 0; u < WPL; ++u) {
        nb[u] = 0.0;
#pragma unroll
        for (int k = 0; k < DPAD; ++k) {
            xs[u][k] = sx[lane + 64 * u][k];
            if (DOT) nb[u] = fma(xs[u][k], xs[u][k], nb[u]);
        }
    }
    const double c = amp[p];
    double* Kp = KsT + (int64_t)p * Np * Wld;
    // SLICE: x 2^47 / 2^e_c and + 2^52 leave the rounded 47-bit integer in the mantissa (K* in [0, c] <= 0.99 2^e_c)
    const double slice_sc = SLICE ? 140737488355328.0 / colscale[p] : 0.0;
    const int64_t plane_sz = (Np / 16) * Wld * 16;
    for (int64_t chunk = chunk0; chunk < chunk1; ++chunk) {
        const int64_t nbeg = chunk * KX_CHUNK;
        __syncthreads();                                // the walker tile / the previous chunk's rows are done with
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int e = threadIdx.x + 256 * j;
            if (e < KX_CHUNK * DPAD) sxr[e] = pre[j];
        }
        if (threadIdx.x < KX_CHUNK) {
            sal[threadIdx.x] = pal;
            if (DOT) sdn[threadIdx.x] = pdn;
        }
        __syncthreads();
        if (chunk + 1 < chunk1) fetch(chunk + 1);
        double msum[WPL];
#pragma unroll
        for (int u = 0; u < WPL; ++u) msum[u] = 0.0;
        unsigned pw[SLICE ? WPL : 1][SLICE ? 6 : 1][SLICE ? 4 : 1];     // SLICE: this wave's granule of every plane, per walker
        if (SLICE) {
#pragma unroll
            for (int u = 0; u < WPL; ++u)
#pragma unroll
                for (int tp = 0; tp < 6; ++tp)
#pragma unroll
                    for (int g = 0; g < 4; ++g) pw[u][tp][g] = 0u;
        }
#pragma unroll(SLICE ? KX_CHUNK / 4 : 1)
        for (int t = 0; t < KX_CHUNK / 4; ++t) {
            const int pt = SLICE ? 16 * wave + t : wave + 4 * t;      // wave-uniform
            const int64_t n = nbeg + pt;
            double kv[WPL];
#pragma unroll
            for (int u = 0; u < WPL; ++u) kv[u] = 0.0;
            if (n >= pad_front(Np, N) && n < pad_front(Np, N) + N) {      // (the other rows are the padding: zero rows of K*^T)
                const double* xr = sxr + pt * DPAD;
                double r2[WPL];
#pragma unroll
                for (int u = 0; u < WPL; ++u) r2[u] = 0.0;
                if (DOT) {
#pragma unroll
                    for (int k = 0; k < DPAD; ++k) {
                        const double a = xr[k];
#pragma unroll
                        for (int u = 0; u < WPL; ++u) r2[u] = fma(xs[u][k], a, r2[u]);
                    }
#pragma unroll
                    for (int u = 0; u < WPL; ++u) r2[u] = fmax(fma(-2.0, r2[u], sdn[pt] + nb[u]), 0.0);
                } else {
#pragma unroll
                    for (int k = 0; k < DPAD; ++k) {
                        const double a = xr[k];
#pragma unroll
                        for (int u = 0; u < WPL; ++u) {
                            const double df = xs[u][k] - a;
                            r2[u] = fma(df, df, r2[u]);
                        }
                    }
                }
                // the lane's WPL kernel values side by side (fast_math.h: the shape functions of k_kmat_mfma — K(X,X) and K* by the
                // same arithmetic — with the WPL chains interleaved: one exp at a time ran at its operations' latency)
                double sh[WPL];
                shape_fn_fast_n<KIND, WPL>(r2, sh);
#pragma unroll
                for (int u = 0; u < WPL; ++u) {
                    kv[u] = c * sh[u];
                    msum[u] = fma(sal[pt], kv[u], msum[u]);
                }
            }
            if (SLICE) {
#pragma unroll
                for (int u = 0; u < WPL; ++u) {
                    // the six signed radix-256 digits: bytes of (a + 0x808080808080) ^ 0x808080808080, byte t & 3 of word t >> 2
                    const unsigned long long a = (unsigned long long)__double_as_longlong(fma(kv[u], slice_sc, 4503599627370496.0)) & 0xfffffffffffffull;
                    const unsigned long long dg = (a + 0x808080808080ull) ^ 0x808080808080ull;
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp)
                        pw[u][tp][t >> 2] |= (unsigned)((dg >> (8 * tp)) & 0xffull) << (8 * (t & 3));
                }
            } else {
#pragma unroll
                for (int u = 0; u < WPL; ++u) GPB_KSTAR_STORE(&Kp[n * Wld + w0 + lane + 64 * u], kv[u]);
            }
        }
        if (SLICE) {
            int8_t* dst = planes + (int64_t)p * 6 * plane_sz + ((chunk * 4 + wave) * Wld + w0 + lane) * 16;
#pragma unroll
            for (int u = 0; u < WPL; ++u)
#pragma unroll
                for (int tp = 0; tp < 6; ++tp)
                    *reinterpret_cast<uint4*>(dst + tp * plane_sz + (int64_t)u * 64 * 16) = make_uint4(pw[u][tp][0], pw[u][tp][1], pw[u][tp][2], pw[u][tp][3]);
        }
        // red: written here, read by wave 0 below; the next write is behind the two barriers at the top of the loop
#pragma unroll
        for (int u = 0; u < WPL; ++u) red[wave][lane + 64 * u] = msum[u];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < WPL; ++u) {
                const int l = lane + 64 * u;
                mpart[(chunk * P + p) * Wld + w0 + l] = ((red[0][l] + red[1][l]) + red[2][l]) + red[3][l];
            }
        }
    }
}

template <int KIND, int DPAD, bool DOT, int WPL, bool SLICE = false>
__global__ __launch_bounds__(256) void k_kcross(const double* __restrict__ Xs, int64_t W, int d,
                                                const double* __restrict__ Xsc, const double* __restrict__ ls,
                                                const double* __restrict__ amp, const double* __restrict__ alpha,
                                                double* __restrict__ KsT, double* __restrict__ mpart,
                                                int64_t N, int64_t Np, int64_t Wld, int P,
                                                const double* __restrict__ dnorm, const double* __restrict__ muS,
                                                int chunks_per_wg, const int* __restrict__ nrows,
                                                const int* __restrict__ form, int8_t* __restrict__ planes = nullptr,
                                                const double* __restrict__ colscale = nullptr) {
    // form (optional): the per-GP distance form (gpb_ctx::gpform); this launch computes the GPs of ITS form, the workgroups
    // of the others leave at once (their launch is the other instantiation, on the same stream)
    if (form && form[blockIdx.y] != (DOT ? 0 : 1)) return;
    __shared__ KxLds<DPAD, WPL> lds;
    kcross_body<KIND, DPAD, DOT, WPL, SLICE>(lds, Xs, W, d, Xsc, ls, amp, alpha, KsT, mpart, N, Np, Wld, P, dnorm, muS,
                                             chunks_per_wg, nrows, (int)blockIdx.y, planes, colscale);
}

// The cross kernels of ALL emulators of a chain in one launch (round 3): grid.y runs over the GPs of every emulator of the
// group (same padded design size, same number of GP inputs); entry e of the table — a kernel argument — holds emulator e's
// pointers, its kernel family and the index of its first GP.  At a few hundred walkers nine launches of 6-8 GPs each were
// pure launch-and-latency time (9 x 12-14 us per half-step of 490 at 512 walkers); the work and the bits are unchanged.
constexpr int MAX_KX_CTX = 32;
struct KxCtx {
    const double *Xs, *Xc, *ls, *amp, *alpha, *dnorm, *muS;
    double *KsT, *mpart;
    int8_t* planes;              // SLICE form (option key 51): the context's digit planes of K*^T and their per-GP scales
    const double* colscale;
    int N, P, kind, g0, p0, d;   // g0: index of the entry's first GP in the launch; p0: that GP's index in its context;
};                               // d: the context's own input count (contexts of one launch share the padded count only)
struct KxTable { KxCtx c[MAX_KX_CTX]; int E; };

template <int DPAD, int WPL, bool SLICE = false>
__global__ __launch_bounds__(256) void k_kcross_multi(const KxTable tab, int64_t W, int d, int64_t Np, int64_t Wld,
                                                      int chunks_per_wg, const int* __restrict__ nrows) {
    __shared__ KxLds<DPAD, WPL> lds;
    const int g = (int)blockIdx.y;
    int e = 0;
    while (e + 1 < tab.E && g >= tab.c[e + 1].g0) ++e;             // uniform: the emulator this GP belongs to
    const KxCtx& c = tab.c[e];
    const int p = g - c.g0 + c.p0;
    if (c.kind == GPB_KERNEL_RBF)
        kcross_body<GPB_KERNEL_RBF, DPAD, true, WPL, SLICE>(lds, c.Xs, W, c.d, c.Xc, c.ls, c.amp, c.alpha, c.KsT, c.mpart, c.N, Np, Wld,
                                                            c.P, c.dnorm, c.muS, chunks_per_wg, nrows, p, c.planes, c.colscale);
    else if (c.kind == GPB_KERNEL_MATERN15)
        kcross_body<GPB_KERNEL_MATERN15, DPAD, true, WPL, SLICE>(lds, c.Xs, W, c.d, c.Xc, c.ls, c.amp, c.alpha, c.KsT, c.mpart, c.N, Np,
                                                                 Wld, c.P, c.dnorm, c.muS, chunks_per_wg, nrows, p, c.planes, c.colscale);
    else
        kcross_body<GPB_KERNEL_MATERN25, DPAD, true, WPL, SLICE>(lds, c.Xs, W, c.d, c.Xc, c.ls, c.amp, c.alpha, c.KsT, c.mpart, c.N, Np,
                                                                 Wld, c.P, c.dnorm, c.muS, chunks_per_wg, nrows, p, c.planes, c.colscale);
}

// 1-D grid of P * nI * nW tiles (T x T, T = 128 or 64), heaviest row blocks first; consecutive blocks
// differ in the walker tile so that the 8 XCDs (round-robin dispatch) each keep their own K*^T columns in
// L2.  The fused epilogue reduces V^2 over rows in a tree that depends only on the row index — 32-row
// chains, lane groups, then the two halves of each 64-row block — so both tile sizes, and therefore any
// sharding of the walkers, give bit-identical sums.  spart is indexed by 64-row block.
__device__ __forceinline__ void set_wave_prio(int p) {      // s_setprio takes an immediate
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

template <int T, int NW, int TN, int KB, bool PIPE = false>
__device__ __forceinline__ void predict_tile(TileLds<T, TN, KB>& lds, int p, int ib, int wt, const double* __restrict__ Linv,
                                             const double* __restrict__ KsT, double* __restrict__ spart, int64_t Np,
                                             int64_t Wld, int P, int prio_levels, unsigned* __restrict__ trace,
                                             int tri_skip = 1) {
    // trace (debug hook, normally null): one record per tile — where and when it ran (tools/gpu_tile_trace.py)
    const unsigned long long trace_t0 = trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int NI = T / 32, WN = NW / 2, TNW = TN / WN, NJ = TNW / 16;
    const int64_t mb = (int64_t)ib * T, nb = (int64_t)wt * TN;
    const int m_ext = (int)imin64(T, Np - mb);
    const int64_t k_end = imin64(mb + T, Np);
    // Resident grids: longer K loops issue first (s_setprio), so that a CU's long tile does not share the matrix
    // pipe equally until the short ones are done and then finish alone at one wave's issue rate.  The tile trace
    // (tools/gpu_tile_trace.py) shows the order taking effect; worth 1-2 % at 128-256 walkers.  Scheduling only.
    if (prio_levels > 0) set_wave_prio((4 * ib) / prio_levels);      // prio_levels = number of row blocks
    Acc<T, NW, TN> acc;
    acc_zero<T, NW, TN>(acc);
    // Np is a multiple of 64 and the batch is padded to 128 walkers: 64-row tiles never have an edge
    // L^-1 is lower triangular: the 128x128 kernel skips the all-zero 16-row m-tiles of the diagonal block
    // (gemm_tile_loop TRI: the wave rows own the m-tiles alternately, acc.v[i] = m-tile 2i + wave row).
    constexpr bool TRI = (NW == 4);      // (K-steps of 16 or 32)
    // tri_skip carries two things: bit 0 = skip the diagonal block's zero m-tiles; bits 8.. = the number of LEADING columns of
    // K*^T that are all zero — the design's padding sits in front (gp_set_impl) — rounded down to whole K-steps by the launcher:
    // a row block that starts at or behind them begins its K loop there (the products it leaves out are exact zeros: same bits)
    const int64_t kskip = ((tri_skip >> 8) / KB) * KB;
    const int64_t k_begin = kskip <= mb ? kskip : 0;
    gemm_tile_loop<T, false, false, NW, TN, KB, T == 64, TRI, PIPE>(Linv + (int64_t)p * Np * Np, Np,
                                                              KsT + (int64_t)p * Np * Wld, Wld, mb, nb, m_ext, TN, k_begin,
                                                              k_end, lds, acc, (tri_skip & 1) ? mb : k_end);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    double* red = &lds.Bs[0][0];                // [2 x 64-row blocks or wave rows][TN columns]
    if constexpr (TRI) {
        // Same reduction tree as below (32-row chains over m-tiles (2c, 2c+1), lane groups, two chains per 64-row
        // block), but a chain's two m-tiles now sit in different wave rows: wave row 0 starts every chain and
        // hands its per-lane partial to wave row 1 through LDS.
        double* part = &lds.As[0][0];           // [chain c][j][wn][lane]: 4*4*2*64 doubles = 16 KB of the A tile
        __syncthreads();                        // all waves are done reading the operand tiles
        if (wm == 0) {
#pragma unroll
            for (int c = 0; c < NI; ++c)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double v = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v = fma(acc.v[c][j][r], acc.v[c][j][r], v);
                    part[((c * NJ + j) * WN + wn) * 64 + lane] = v;
                }
        }
        __syncthreads();
        if (wm == 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double h[NI];
#pragma unroll
                for (int c = 0; c < NI; ++c) {
                    double v = part[((c * NJ + j) * WN + wn) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v = fma(acc.v[c][j][r], acc.v[c][j][r], v);
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    h[c] = v;
                }
                if (lane < 16) {
                    red[0 * TN + wn * TNW + 16 * j + lane] = h[0] + h[1];            // rows 0-63 of the tile
                    if (NI == 4) red[1 * TN + wn * TNW + 16 * j + lane] = h[NI - 2] + h[NI - 1];  // rows 64-127
                }
            }
        }
        __syncthreads();
    } else {
        double s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            double h[NI / 2];                       // one chain per 32 rows: m-tiles (2g, 2g+1)
#pragma unroll
            for (int g = 0; g < NI / 2; ++g) {
                double v = 0.0;
#pragma unroll
                for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v = fma(acc.v[i][j][r], acc.v[i][j][r], v);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                h[g] = v;
            }
            s[j] = (NI == 4) ? (h[0] + h[NI / 2 - 1]) : h[0];     // T=128: rows 0-31 + rows 32-63 of the wave's block
        }
        __syncthreads();                            // all waves are done reading the operand tiles
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) red[wm * TN + wn * TNW + 16 * j + lane] = s[j];
        }
        __syncthreads();
    }
    if (T == 128) {                             // each half of red is one 64-row block
        if (tid < 256) {
            const int half = tid >> 7, col = tid & 127;
            const int64_t blk = 2 * (int64_t)ib + half;
            if (blk * 64 < Np) spart[(blk * P + p) * Wld + nb + col] = red[half * TN + col];
        }
    } else if (TRI) {                           // one 64-row block, summed by wave row 1 above
        if (tid < TN) spart[((int64_t)ib * P + p) * Wld + nb + tid] = red[tid];
    } else {                                    // the two wave rows are the halves of one 64-row block
        if (tid < TN) spart[((int64_t)ib * P + p) * Wld + nb + tid] = red[tid] + red[TN + tid];
    }
    if (trace && tid == 0) {
        const unsigned slot = atomicAdd(&trace[0], 1u);
        if (slot < trace[1]) {                  // trace[1] = capacity in records
            unsigned* r = trace + 8 + 8 * (size_t)slot;
            r[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID: CU / SE / SIMD ...
            r[1] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID
            r[2] = (unsigned)p; r[3] = (unsigned)ib; r[4] = (unsigned)wt;
            r[5] = (unsigned)trace_t0;
            r[6] = (unsigned)__builtin_amdgcn_s_memrealtime();
            r[7] = blockIdx.x;
        }
    }
}


// ticket t of queue qx -> tile (GP p, row block ib, walker tile wt); false for the padding of xcd_mode 1
__device__ __forceinline__ bool decode_tile(int xcd_mode, unsigned t, unsigned qx, int nI, int nW, int P, int& p,
                                            int& ib, int& wt) {
    const int b = (int)(t * 8u + qx);
    int g;
    if (xcd_mode == 1) {
        const int q = b >> 3;
        g = (q / nW) * 8 + (b & 7);
        wt = q % nW;
        if (g >= nI * P) return false;
    } else {
        g = b / nW;
        wt = b - g * nW;
    }
    ib = nI - 1 - g / P;
    p = g - (g / P) * P;
    return true;
}

// Persistent launch: `gridDim.x` workgroups (a few per CU) pull tile indices from device-scope ticket
// queues in LPT order, so the triangular row blocks balance dynamically whatever the dispatcher does.
// Exit condition: every workgroup walks all eight queues once and leaves each when its ticket is past
// the queue's length; nothing spins.
template <int T, int NW, int TN, int KB, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, (T == 128 && NW == 4 ? 2 : (T == 64 && TN == 64 && NW == 4 ? 6 : 4))) void k_predict(const double* __restrict__ Linv, const double* __restrict__ KsT,
                                                     double* __restrict__ spart, int64_t Np, int64_t Wld, int P,
                                                     int nI, int nW, int xcd_mode, unsigned* __restrict__ queue,
                                                     unsigned nblocks, unsigned* __restrict__ trace, int tri_skip,
                                                     const int* __restrict__ nrows) {
    constexpr int prio_levels = 0;
    if (nrows) {
        // compacted batch: only the walker tiles that hold rows exist.  The tile list (and with it the LPT order and
        // the balance of the eight queues) is rebuilt for them; the grid was sized for the whole batch, the surplus
        // workgroups find their queues empty.
        nW = (*nrows + TN - 1) / TN;
        nblocks = (unsigned)((xcd_mode == 1 ? ((P * nI + 7) / 8) * 8 : P * nI) * nW);
    }
    __shared__ TileLds<T, TN, KB> lds;
    // The ticket lives in the padding of the last A row (never touched by the loaders, the MFMA fragment reads
    // or the epilogue's scratch): the operand tiles alone are an exact fraction of the CU's 160 KB LDS, and a
    // separate 4-byte variable would cost a whole workgroup of occupancy.
    unsigned& s_ticket = *reinterpret_cast<unsigned*>(&lds.As[KB - 1][T + 12]);      // (the row skew uses the pad up to T + 11)
    // Eight ticket queues, one per XCD label (blockIdx % 8; workgroups with equal labels share an XCD's
    // L2 under round-robin dispatch — speed only): queue x owns the tiles b = 8 t + x, i.e. a fixed set
    // of walker tiles (or row blocks) whose operand panels stay in that XCD's L2 while its workgroups
    // work through them in LPT order.  A workgroup whose own queue is empty steals from the others.
    // Tile -> queue maps (xcd_mode), all heaviest-row-block-first within a queue:
    //   0: queue = walker tile % 8   (each XCD keeps its own K*^T columns; all GPs interleaved)
    //   1: queue = (row block, GP) group % 8   (small W: each L^-1 row block is fetched by one XCD)
    //   (maps built for L2 reuse — one GP per XCD at a time, super-blocks of four row blocks — cut the fabric traffic by 20-55 %
    //   and were 1-15 % SLOWER for this fp64 kernel: the heaviest-first order over all GPs is worth more than the reuse,
    //   profiles/r03_xcd_map_ab.txt.  The int8 kernel, which IS bound by its operand feed, uses super-blocks: gpb_sliced.hip)
    const int x = blockIdx.x & 7;
    for (int s = 0; s < 8; ++s) {
        const unsigned qx = (unsigned)((x + s) & 7);
        const unsigned nq = (nblocks > qx) ? (nblocks - qx + 7u) / 8u : 0u;
        for (;;) {
            if (threadIdx.x == 0) s_ticket = atomicAdd(&queue[qx * 16], 1u);
            __syncthreads();
            const unsigned t = s_ticket;
            __syncthreads();                    // everyone has read the ticket before it is redrawn
            if (t >= nq) break;                 // uniform: this queue is exhausted
            int p, ib, wt;
            if (!decode_tile(xcd_mode, t, qx, nI, nW, P, p, ib, wt)) continue;   // padding (uniform)
            predict_tile<T, NW, TN, KB, PIPE>(lds, p, ib, wt, Linv, KsT, spart, Np, Wld, P, prio_levels, trace, tri_skip);
        }
    }
    // the last workgroup to finish re-arms the queues for the next launch (all others are past their draws)
    if (threadIdx.x == 0) {
        if (atomicAdd(&queue[128], 1u) == gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) queue[i * 16] = 0u;
            queue[128] = 0u;
        }
    }
}

// Static launch for the case that every tile has its own workgroup and all of them are co-resident (a rank's
// shard of a sharded ensemble): nothing can be balanced dynamically, so the tile is a fixed function of
// blockIdx.  The dispatcher deals an XCD's workgroups round-robin over its ncu_x CUs (measured,
// tools/micro/dispatch_probe.hip): workgroup j of an XCD runs on CU j % ncu_x.  order: 1 = weight-sorted as is,
// 2 = snake over the CUs (equal sums per CU), 3 = snake of neighbouring PAIRS (equal sums, and the two heaviest
// tiles of a CU finish together).  Same tile -> queue maps as above; a separate kernel so that each has ONE
// inlined copy of the tile body (with two copies the compiler parked the prefetch registers in scratch memory).
template <int T, int NW, int TN, int KB, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, (T == 128 && NW == 4 ? 2 : (KB == 32 ? 3 : 4))) void k_predict_static(
    const double* __restrict__ Linv, const double* __restrict__ KsT, double* __restrict__ spart, int64_t Np, int64_t Wld,
    int P, int nI, int nW, int xcd_mode, unsigned nblocks, int order, unsigned ncu_x, int prio_levels,
    unsigned* __restrict__ trace, int tri_skip, const int* __restrict__ nrows) {
    __shared__ TileLds<T, TN, KB> lds;
    if (nrows) {                                       // compacted batch: the tile list of the live walker tiles only
        nW = (*nrows + TN - 1) / TN;
        nblocks = (unsigned)((xcd_mode == 1 ? ((P * nI + 7) / 8) * 8 : P * nI) * nW);
    }
    const unsigned qx = blockIdx.x & 7u;
    const unsigned nq = (nblocks > qx) ? (nblocks - qx + 7u) / 8u : 0u;
    unsigned t = blockIdx.x >> 3;
    if (t >= nq) return;
    const unsigned k = t / ncu_x, c = t - k * ncu_x;
    if (order == 2) {
        if ((k & 1u) && (k + 1u) * ncu_x <= nq) t = k * ncu_x + (ncu_x - 1u - c);
    } else if (order == 3) {
        const unsigned n2 = (nq / (2u * ncu_x)) * (2u * ncu_x);
        if (t < n2) {
            const unsigned kp = k >> 1;
            const unsigned pi = kp * ncu_x + ((kp & 1u) ? (ncu_x - 1u - c) : c);
            t = 2u * pi + (k & 1u);
        }
    }
    int p, ib, wt;
    if (decode_tile(xcd_mode, t, qx, nI, nW, P, p, ib, wt))
        predict_tile<T, NW, TN, KB, PIPE>(lds, p, ib, wt, Linv, KsT, spart, Np, Wld, P, prio_levels, trace, tri_skip);
}


// ---------------------------------------------------------------------------------------------------------------
// Chains of several emulators (Chain.emuList: nine emulators, 63 GPs in the reference's analyses): ONE launch over the GPs
// of all emulators whose designs pad to the same Np, instead of one launch per emulator with 6-8 GPs each and a partly
// filled chip behind every one of them.  The tile list is that of a single emulator with G = sum of P_e GPs; entry g of
// the table (a kernel argument: at most 96 GPs x 32 bytes) says where GP g's L^-1, K*^T and partials live.  A tile
// computes exactly what it computes in its emulator's own launch: same bits.
constexpr int MAX_MULTI_GP = GPB_MAX_MULTI_GP;
struct PredGP {
    const double* Linv;      // the emulator's [P_e][Np][Np]
    const double* KsT;       // [P_e][Np][Wld]
    double* spart;           // [Np / 64][P_e][Wld]
    int P, p;                // GPs of the emulator, this GP's index in it
};
struct PredTable { PredGP gp[MAX_MULTI_GP]; };

template <int T, int NW, int TN, int KB, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, (T == 128 && NW == 4 ? 2 : (T == 64 && TN == 64 && NW == 4 ? 6 : 4))) void k_predict_multi(
    const PredTable tab, int64_t Np, int64_t Wld, int G, int nI, int nW, int xcd_mode, unsigned* __restrict__ queue,
    unsigned nblocks, unsigned* __restrict__ trace, int tri_skip, const int* __restrict__ nrows) {
    if (nrows) {
        nW = (*nrows + TN - 1) / TN;
        nblocks = (unsigned)((xcd_mode == 1 ? ((G * nI + 7) / 8) * 8 : G * nI) * nW);
    }
    __shared__ TileLds<T, TN, KB> lds;
    unsigned& s_ticket = *reinterpret_cast<unsigned*>(&lds.As[KB - 1][T + 12]);
    const int x = blockIdx.x & 7;
    for (int s = 0; s < 8; ++s) {                      // as k_predict: eight XCD-affine ticket queues, LPT order, stealing
        const unsigned qx = (unsigned)((x + s) & 7);
        const unsigned nq = (nblocks > qx) ? (nblocks - qx + 7u) / 8u : 0u;
        for (;;) {
            if (threadIdx.x == 0) s_ticket = atomicAdd(&queue[qx * 16], 1u);
            __syncthreads();
            const unsigned t = s_ticket;
            __syncthreads();
            if (t >= nq) break;
            int g, ib, wt;
            if (!decode_tile(xcd_mode, t, qx, nI, nW, G, g, ib, wt)) continue;
            const PredGP& e = tab.gp[g];
            predict_tile<T, NW, TN, KB, PIPE>(lds, e.p, ib, wt, e.Linv, e.KsT, e.spart, Np, Wld, e.P, 0, trace, tri_skip);
        }
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(&queue[128], 1u) == gridDim.x - 1) {
            for (int i = 0; i < 8; ++i) queue[i * 16] = 0u;
            queue[128] = 0u;
        }
    }
}

template <int T, int NW, int TN, int KB>
__global__ __launch_bounds__(64 * NW, (T == 128 && NW == 4 ? 2 : (KB == 32 ? 3 : 4))) void k_predict_static_multi(
    const PredTable tab, int64_t Np, int64_t Wld, int G, int nI, int nW, int xcd_mode, unsigned nblocks, int order,
    unsigned ncu_x, int prio_levels, unsigned* __restrict__ trace, int tri_skip, const int* __restrict__ nrows) {
    __shared__ TileLds<T, TN, KB> lds;
    if (nrows) {
        nW = (*nrows + TN - 1) / TN;
        nblocks = (unsigned)((xcd_mode == 1 ? ((G * nI + 7) / 8) * 8 : G * nI) * nW);
    }
    const unsigned qx = blockIdx.x & 7u;
    const unsigned nq = (nblocks > qx) ? (nblocks - qx + 7u) / 8u : 0u;
    unsigned t = blockIdx.x >> 3;
    if (t >= nq) return;
    const unsigned k = t / ncu_x, c = t - k * ncu_x;
    if (order == 2) {
        if ((k & 1u) && (k + 1u) * ncu_x <= nq) t = k * ncu_x + (ncu_x - 1u - c);
    } else if (order == 3) {
        const unsigned n2 = (nq / (2u * ncu_x)) * (2u * ncu_x);
        if (t < n2) {
            const unsigned kp = k >> 1;
            const unsigned pi = kp * ncu_x + ((kp & 1u) ? (ncu_x - 1u - c) : c);
            t = 2u * pi + (k & 1u);
        }
    }
    int g, ib, wt;
    if (decode_tile(xcd_mode, t, qx, nI, nW, G, g, ib, wt)) {
        const PredGP& e = tab.gp[g];
        predict_tile<T, NW, TN, KB, false>(lds, e.p, ib, wt, e.Linv, e.KsT, e.spart, Np, Wld, e.P, prio_levels, trace, tri_skip);
    }
}


__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ mpart, const double* __restrict__ spart,
                                                  const double* __restrict__ amp, const double* __restrict__ noise,
                                                  double* __restrict__ mean_pc, double* __restrict__ var_pc,
                                                  int64_t Wld, int64_t Wuse, int P, int nchunk, int nI64, int need_var) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int p = blockIdx.y;
    if (w >= Wuse) return;
    double m = 0.0;
#pragma unroll 8
    for (int c = 0; c < nchunk; ++c) m += mpart[((int64_t)c * P + p) * Wld + w];
    mean_pc[(int64_t)p * Wld + w] = m;
    if (need_var) {
        double s = 0.0;
#pragma unroll 8
        for (int i = 0; i < nI64; ++i) s += spart[((int64_t)i * P + p) * Wld + w];
        // (a NaN input row makes the mean NaN; the int8 kernel's integer sums cannot carry a NaN, so the variance takes it from there:
        // on the fp64 kernel s is NaN whenever m is, and nothing changes)
        var_pc[(int64_t)p * Wld + w] = m != m ? m : (amp[p] + noise[p]) - s;
    }
}

int ensure_wcap(gpb_ctx* ctx, int64_t W) {
    const int64_t need = round_up(W < 1 ? 1 : W, WPAD);
    if (need <= ctx->Wcap) return 0;
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    double** bufs[] = {&ctx->Xs, &ctx->estd, &ctx->KsT, &ctx->mpart, &ctx->spart, &ctx->mean_pc, &ctx->var_pc};
    for (auto b : bufs) {
        if (*b) { pool_free(*b); *b = nullptr; }
    }
    ctx->Wcap = 0;
    const int64_t P = ctx->P, Np = ctx->Np;
    const int64_t nchunk = (Np + KX_CHUNK - 1) / KX_CHUNK, nI64 = Np / 64;
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->Xs), sizeof(double) * need * ctx->d));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->estd), sizeof(double) * need));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->KsT), sizeof(double) * P * Np * need));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->mpart), sizeof(double) * nchunk * P * need));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->spart), sizeof(double) * nI64 * P * need));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->mean_pc), sizeof(double) * P * need));
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(&ctx->var_pc), sizeof(double) * P * need));
    if (ctx->cmp_idx) { pool_free(ctx->cmp_idx); ctx->cmp_idx = nullptr; }
    if (ctx->cmp_X) { pool_free(ctx->cmp_X); ctx->cmp_X = nullptr; }
    ctx->cmp_X_cap = 0;
    // [0] = number of rows inside the box, [4..] = their indices; then the compaction's scratch: ranks, workgroup counts
    GPB_HIP(pool_malloc_t(&ctx->cmp_idx, sizeof(int) * (size_t)(4 + 2 * need + need / 256 + 8)));
    ctx->Wcap = need;
    return 0;
}

template <int KIND>
static int launch_kcross_kind(gpb_ctx* ctx, const double* Xs_dev, int64_t W, int64_t Wuse, const int* nrows_dev, bool planes) {
    const int nchunk = (int)((ctx->Np + KX_CHUNK - 1) / KX_CHUNK);
    // chunks per workgroup: as many as still leave >= 4 workgroups per CU (a geometry choice: the per-chunk
    // partials and their order do not depend on it; measured, cfg 4: 1 / 2 / 4 chunks at 512 / 1024 / 2048+ walkers)
    // two walkers per lane (every LDS broadcast feeds two multiply-adds) for d <= 32 and batches that are a whole
    // number of 128-walker tiles (always: WPAD = 128)
    const int wpl = (ctx->kcross_wpl == 2 && ctx->dpad <= 32 && Wuse >= 256 && Wuse % 128 == 0) ? 2 : 1;
    int cpw = ctx->kcross_chunks;
    if (cpw <= 0) {
        // compacted batches: about half of a stretch move's proposals from a spread-out ensemble are live (the count is
        // known on the device only); sizing the chunks per workgroup for the whole batch left 2.5 workgroups per CU
        const int64_t Wgeo = nrows_dev ? round_up(Wuse / 2, 64 * wpl) : Wuse;
        const int64_t wgs1 = (Wgeo / (64 * wpl)) * nchunk * ctx->P;
        cpw = 1;
        while (cpw < 8 && wgs1 / (2 * cpw) >= 4 * (int64_t)ctx->num_cu) cpw *= 2;
    }
    // (grid.z, the walker tile, is limited to 65535: 4.19 M rows a call at one walker per lane; the host classes send long
    // inputs through in slabs of 131072 rows)
    if (Wuse / (64 * wpl) > 65535) GPB_FAIL(GPB_E_ARG, "gpb: more than 65535 walker tiles (4 million rows) in one batch: split it");
    dim3 grid((unsigned)((nchunk + cpw - 1) / cpw), (unsigned)ctx->P, (unsigned)(Wuse / (64 * wpl)));
    // two instantiations, one per distance form; each launch computes the GPs of its form (ctx->gpform, chosen from theta:
    // choose_forms) and is left out when no GP has it — the usual case is the Gram launch alone, with no form table to read
    const int* form = (ctx->n_diff > 0 && ctx->n_diff < ctx->P) ? ctx->gpform : nullptr;
    const bool gram = ctx->n_diff < ctx->P, diff = ctx->n_diff > 0;
    int8_t* const slB = planes ? ctx->slB : nullptr;
    const double* const slcs = planes ? sliced_colscale(ctx) : nullptr;
#define GPB_KX_LAUNCH(DP, DOT_, WPL_, XD)                                                                         \
    do {                                                                                                          \
        if (planes)                                                                                               \
            hipLaunchKernelGGL((k_kcross<KIND, DP, DOT_, WPL_, true>), grid, dim3(256), 0, ctx->stream, Xs_dev, W, (int)ctx->d, \
                               XD, ctx->ls, ctx->amp, ctx->alpha, ctx->KsT, ctx->mpart, ctx->N, ctx->Np, ctx->Wld,       \
                               (int)ctx->P, ctx->dnorm, ctx->muS, cpw, nrows_dev, form, slB, slcs);               \
        else                                                                                                      \
            hipLaunchKernelGGL((k_kcross<KIND, DP, DOT_, WPL_>), grid, dim3(256), 0, ctx->stream, Xs_dev, W, (int)ctx->d, \
                               XD, ctx->ls, ctx->amp, ctx->alpha, ctx->KsT, ctx->mpart, ctx->N, ctx->Np, ctx->Wld,       \
                               (int)ctx->P, ctx->dnorm, ctx->muS, cpw, nrows_dev, form, nullptr, nullptr);        \
    } while (0)
#define GPB_KX(DP)                                                                                               \
    do {                                                                                                         \
        if (wpl == 2) {                                                                                          \
            if (gram) GPB_KX_LAUNCH(DP, true, (DP <= 32 ? 2 : 1), ctx->Xc);                                      \
            if (diff) GPB_KX_LAUNCH(DP, false, (DP <= 32 ? 2 : 1), ctx->Xsc);                                    \
        } else {                                                                                                 \
            if (gram) GPB_KX_LAUNCH(DP, true, 1, ctx->Xc);                                                       \
            if (diff) GPB_KX_LAUNCH(DP, false, 1, ctx->Xsc);                                                     \
        }                                                                                                        \
    } while (0)
    switch (ctx->dpad) {
        case 8: GPB_KX(8); break;
        case 16: GPB_KX(16); break;
        case 20: GPB_KX(20); break;
        case 24: GPB_KX(24); break;
        case 32: GPB_KX(32); break;
        case 48: GPB_KX(48); break;
        default: GPB_KX(64); break;
    }
#undef GPB_KX
#undef GPB_KX_LAUNCH
    return 0;
}

// K*^T and the mean partials of a batch: Xs_dev [W][d] on the device.  Sets the leading dimension of the batch's workspaces.
int launch_kcross(gpb_ctx* ctx, const double* Xs_dev, int64_t W, const int* nrows_dev, bool allow_planes) {
    if (ctx->multi) GPB_FAIL(GPB_E_STATE, "gpb: a gpb_gp_set_multi context is fit-only (its GPs have different designs)");
    if (!ctx->factored) GPB_FAIL(GPB_E_STATE, "gpb: predict before gpb_gp_factor");
    if (W > ctx->Wcap) GPB_FAIL(GPB_E_STATE, "gpb: internal: W exceeds workspace");
    const int64_t Wuse = round_up(W, WPAD);
    // option key 51 (gpb_sliced.hip): the batch leaves this kernel as int8 digit planes for the int8 predict kernel — for EVERY
    // batch size once the rule admits the context (a walker's bits must not depend on the batch it arrives in) — unless the
    // caller needs K*^T itself (the joint covariance, gpb_gp_get) or shares the predict launch with other contexts
    const bool planes = allow_planes && !ctx->want_kst && sliced_applies(ctx);
    ctx->batch_sliced = planes;
    if (planes) {
        const int rc = sliced_prepare(ctx);
        if (rc) return rc;
    }
    // leading dimension of this batch's workspaces (K*^T, partials, per-GP means / variances): the padded batch, not the
    // capacity — with the capacity left at 4096 by an earlier call a 256-walker batch read its 64-walker tile rows
    // 32 KB apart and k_predict_static ran 13 % slower (117 -> 133 us), k_kcross 24 % (20 -> 25 us)
    ctx->Wld = Wuse;
    ctx->last_W = W;
    int rc;
    if (ctx->kind == GPB_KERNEL_RBF) rc = launch_kcross_kind<GPB_KERNEL_RBF>(ctx, Xs_dev, W, Wuse, nrows_dev, planes);
    else if (ctx->kind == GPB_KERNEL_MATERN15) rc = launch_kcross_kind<GPB_KERNEL_MATERN15>(ctx, Xs_dev, W, Wuse, nrows_dev, planes);
    else rc = launch_kcross_kind<GPB_KERNEL_MATERN25>(ctx, Xs_dev, W, Wuse, nrows_dev, planes);
    if (rc) return rc;
    GPB_HIP(hipGetLastError());
    return 0;
}

// K*^T and the mean partials for E contexts of a chain in ONE launch (k_kcross_multi): all with the same padded design size,
// the same number of GP inputs and the same batch; Xs[e] = context e's input rows (its parameter map's output, or the chain's
// gathered rows).  Falls back to one launch per context when the group does not qualify (a GP in the difference form,
// more contexts than the table holds): a context's own launch computes the same bits.
int launch_kcross_group(gpb_ctx* const* ctxs, const double* const* Xs, int E, int64_t W, const int* nrows_dev) {
    gpb_ctx* ctx = ctxs[0];
    bool ok = E > 1 && E <= MAX_KX_CTX;
    for (int e = 0; e < E; ++e)
        if (ctxs[e]->multi) GPB_FAIL(GPB_E_STATE, "gpb: a gpb_gp_set_multi context is fit-only (its GPs have different designs)");
    for (int e = 0; e < E && ok; ++e)           // (the shared launch is the Gram form's: a context with a difference-form GP takes its own)
        ok = ctxs[e]->n_diff == 0 && ctxs[e]->Np == ctx->Np && ctxs[e]->dpad == ctx->dpad && ctxs[e]->stream == ctx->stream;
    // option key 51: an emulator's bits must not depend on the company it is evaluated in.  A group whose contexts the int8 rule ALL
    // admits shares the SLICE form of this launch (digit planes out) and the int8 predict launch (launch_vsq); a mixed group falls
    // back to one launch per context, each in its own form
    int nsl = 0;
    for (int e = 0; e < E; ++e) nsl += (!ctxs[e]->want_kst && sliced_applies(ctxs[e])) ? 1 : 0;
    const bool planes = ok && nsl == E;
    if (nsl > 0 && !planes) ok = false;
    if (!ok) {
        for (int e = 0; e < E; ++e) {
            const int rc = launch_kcross(ctxs[e], Xs[e], W, nrows_dev);
            if (rc) { ctx->err = ctxs[e]->err; return rc; }
        }
        return 0;
    }
    const int64_t Wuse = round_up(W, WPAD);
    KxTable tab;
    int G = 0;
    for (int e = 0; e < E; ++e) {
        gpb_ctx* c = ctxs[e];
        if (!c->factored) GPB_FAIL(GPB_E_STATE, "gpb: predict before gpb_gp_factor");
        if (W > c->Wcap) GPB_FAIL(GPB_E_STATE, "gpb: internal: W exceeds workspace");
        c->Wld = Wuse;
        c->last_W = W;
        c->batch_sliced = planes;
        if (planes) {
            const int rc = sliced_prepare(c);
            if (rc) { ctx->err = c->err; return rc; }
        }
        tab.c[e] = KxCtx{Xs[e], c->Xc, c->ls, c->amp, c->alpha, c->dnorm, c->muS, c->KsT, c->mpart, planes ? c->slB : nullptr,
                         planes ? sliced_colscale(c) : nullptr, (int)c->N, (int)c->P, c->kind, G, 0, (int)c->d};
        G += (int)c->P;
    }
    tab.E = E;
    const int nchunk = (int)((ctx->Np + KX_CHUNK - 1) / KX_CHUNK);
    const int wpl = (ctx->kcross_wpl == 2 && ctx->dpad <= 32 && Wuse >= 256 && Wuse % 128 == 0) ? 2 : 1;
    int cpw = ctx->kcross_chunks;
    if (cpw <= 0) {                                    // as launch_kcross_kind, with the GPs of the whole group
        const int64_t Wgeo = nrows_dev ? round_up(Wuse / 2, 64 * wpl) : Wuse;
        const int64_t wgs1 = (Wgeo / (64 * wpl)) * nchunk * G;
        cpw = 1;
        while (cpw < 8 && wgs1 / (2 * cpw) >= 4 * (int64_t)ctx->num_cu) cpw *= 2;
    }
    if (Wuse / (64 * wpl) > 65535) GPB_FAIL(GPB_E_ARG, "gpb: more than 65535 walker tiles (4 million rows) in one batch: split it");
    dim3 grid((unsigned)((nchunk + cpw - 1) / cpw), (unsigned)G, (unsigned)(Wuse / (64 * wpl)));
#define GPB_KXM(DP)                                                                                              \
    do {                                                                                                         \
        if (wpl == 2 && planes)                                                                                  \
            hipLaunchKernelGGL((k_kcross_multi<DP, (DP <= 32 ? 2 : 1), true>), grid, dim3(256), 0, ctx->stream, tab, W, \
                               (int)ctx->d, ctx->Np, Wuse, cpw, nrows_dev);                                       \
        else if (wpl == 2)                                                                                       \
            hipLaunchKernelGGL((k_kcross_multi<DP, (DP <= 32 ? 2 : 1)>), grid, dim3(256), 0, ctx->stream, tab, W, \
                               (int)ctx->d, ctx->Np, Wuse, cpw, nrows_dev);                                       \
        else if (planes)                                                                                         \
            hipLaunchKernelGGL((k_kcross_multi<DP, 1, true>), grid, dim3(256), 0, ctx->stream, tab, W, (int)ctx->d, \
                               ctx->Np, Wuse, cpw, nrows_dev);                                                   \
        else                                                                                                     \
            hipLaunchKernelGGL((k_kcross_multi<DP, 1>), grid, dim3(256), 0, ctx->stream, tab, W, (int)ctx->d,    \
                               ctx->Np, Wuse, cpw, nrows_dev);                                                   \
    } while (0)
    switch (ctx->dpad) {
        case 8: GPB_KXM(8); break;
        case 16: GPB_KXM(16); break;
        case 20: GPB_KXM(20); break;
        case 24: GPB_KXM(24); break;
        case 32: GPB_KXM(32); break;
        case 48: GPB_KXM(48); break;
        default: GPB_KXM(64); break;
    }
#undef GPB_KXM
    GPB_HIP(hipGetLastError());
    return 0;
}

// V = L^-1 K*^T with the fused sum of squares for the GPs of E contexts in one launch (E = 1: an emulator's own launch;
// E > 1: the emulators of a chain whose designs pad to the same Np — see k_predict_multi).  All contexts: same Np, same
// batch (launch_kcross done), same stream.  Timing events and the unit count go to ctxs[0].
int launch_vsq(gpb_ctx* const* ctxs, int E, int64_t W, const int* nrows_dev) {
    gpb_ctx* ctx = ctxs[0];
    const int64_t Wuse = round_up(W, WPAD);
    if (E > 1) {
        // option key 51: contexts whose batch exists as int8 digit planes take their own launches (see launch_kcross_group);
        // the timing events of the whole group still go to ctxs[0]
        bool any_sliced = false, all_sliced = true;
        int gsum = 0;
        for (int e = 0; e < E; ++e) {
            any_sliced = any_sliced || ctxs[e]->batch_sliced;
            all_sliced = all_sliced && ctxs[e]->batch_sliced && ctxs[e]->Np == ctx->Np && ctxs[e]->Wld == Wuse && ctxs[e]->stream == ctx->stream;
            gsum += (int)ctxs[e]->P;
        }
        if (all_sliced && gsum <= MAX_MULTI_GP) {       // one int8 launch over the GPs of all the emulators
            int64_t ks = ctx->Np;
            for (int e = 0; e < E; ++e) ks = imin64(ks, pad_front(ctxs[e]->Np, ctxs[e]->N));
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ctx->profile) {
                GPB_HIP(hipEventCreate(&e0)); GPB_HIP(hipEventCreate(&e1));
                GPB_HIP(hipEventRecord(e0, ctx->stream));
            }
            const int rc = launch_vsq_sliced_multi(ctxs, E, W, nrows_dev, (int)ks);
            if (rc) return rc;
            if (ctx->profile) {
                GPB_HIP(hipEventRecord(e1, ctx->stream));
                ctx->prof_events.push_back({e0, e1});
                ctx->prof_gps = (double)gsum;
                if (!nrows_dev) ctx->prof_units += (double)gsum * (double)W;
                else ctx->prof_compacted = true;
            }
            GPB_HIP(hipGetLastError());
            return 0;
        }
        if (any_sliced) {
            for (int e = 0; e < E; ++e) {
                gpb_ctx* one[1] = {ctxs[e]};
                const bool prof = ctxs[e]->profile;
                ctxs[e]->profile = ctx->profile;
                const int rc = launch_vsq(one, 1, W, nrows_dev);
                if (ctxs[e] != ctx) {
                    for (auto& ev : ctxs[e]->prof_events) ctx->prof_events.push_back(ev);
                    ctxs[e]->prof_events.clear();
                    ctx->prof_units += ctxs[e]->prof_units; ctxs[e]->prof_units = 0.0;
                    ctx->prof_compacted = ctx->prof_compacted || ctxs[e]->prof_compacted; ctxs[e]->prof_compacted = false;
                    ctxs[e]->profile = prof;
                }
                if (rc) { ctx->err = ctxs[e]->err; return rc; }
            }
            return 0;
        }
    }
    int64_t Gsum = 0;
    for (int e = 0; e < E; ++e) {
        if (ctxs[e]->Np != ctx->Np || ctxs[e]->Wld != Wuse || ctxs[e]->stream != ctx->stream)
            GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq over contexts of different shape");
        Gsum += ctxs[e]->P;
    }
    const bool multi = E > 1;
    // leading all-zero rows of K*^T (the designs' padding, in front: gp_set_impl) common to every context of the launch, in whole
    // 16-deep K-steps: the tiles start their K loops behind them
    int64_t kskip = ctx->Np;
    for (int e = 0; e < E; ++e) kskip = imin64(kskip, pad_front(ctxs[e]->Np, ctxs[e]->N));
    const int tri_arg = (ctx->tri_skip ? 1 : 0) | ((int)kskip << 8);
    if (multi && Gsum > MAX_MULTI_GP) GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq: too many GPs for one table");
    const int64_t GP = Gsum;                           // GPs of the launch: what the tile counts are made of
    const int nI64 = (int)(ctx->Np / 64);
    {
        // T rows x TN walkers per tile: the LARGEST shape of which enough tiles exist to fill the chip (measured,
        // cfg 3 and cfg 4 sweeps at 128..2048 walkers, profiles/r01_tile_shape_sweep.txt): 128x128 (2 per CU, 64
        // MFMAs per wave between barriers) from 3.75 tiles per CU on, 64x128 (32 MFMAs) and 64x64 (16) from 5 per
        // CU, else 64x32.  Fewer, larger tiles leave CUs idle behind the heaviest triangular row block; more,
        // smaller ones pay more barriers and operand traffic per MFMA.
        // A compacted batch is launched for its upper bound W, but the tiles that exist are those of the rows inside
        // the prior box.  Their number is only known on the device; the last compaction the device has FINISHED left
        // its (live rows, batch rows) in pinned host memory (k_compact_gather), and that fraction — a few launches
        // old, since the host enqueues ahead — sizes the tile counts of the rule.  Any shape gives the same bits, so a
        // stale fraction costs time at worst.  (cfg 4 sharded 2-way: 1024-row batches with ~495 live rows ran 128x128
        // tiles at 2.5 per CU, 436 us; with the rule fed the live count 64x128.)
        int64_t Wsel = Wuse;
        if (nrows_dev && ctx->tile_by_live && ctx->hint_from && ctx->hint_from->live_hint) {
            const unsigned long long h = __atomic_load_n(ctx->hint_from->live_hint, __ATOMIC_RELAXED);
            const int64_t cnt = (int64_t)(h & 0xffffffffull), of = (int64_t)(h >> 32);
            if (of > 0 && cnt <= of) {
                const int64_t est = (int64_t)((double)cnt / (double)of * (double)W * 1.03) + 8;
                Wsel = est < 64 ? 64 : (est > Wuse ? Wuse : est);
            }
        }
        const int64_t tiles128 = GP * ((ctx->Np + 127) / 128) * ((Wsel + 127) / 128);
        const int64_t tiles64x128 = GP * nI64 * ((Wsel + 127) / 128), tiles64 = GP * nI64 * ((Wsel + 63) / 64);
        // Compacted batches have their own switch points (tools/gpu_shard_sim.py --walkers=.. --tune=force_tile:..,
        // profiles/r02_tile_shape_sweep_compacted.txt, cfg 4 at 190 .. 2060 live rows): the larger shape pays later —
        // 64x32 up to 9.4 tiles of 64x64 per CU (5), 64x64 up to 4.5 of 64x128 (5), 64x128 up to 9.4 of 128x128 (3.75).
        const bool cmpd = nrows_dev != nullptr && ctx->tile_by_live;
        const int64_t sw128 = cmpd ? ctx->tile_switch_c : ctx->tile_switch, swmid = cmpd ? ctx->mid_switch_c : ctx->mid_switch,
                      swnarrow = cmpd ? ctx->narrow_switch_c : ctx->narrow_switch;
        int T = 64, TN = 32;
        if (cmpd) {
            // ... counted in fractions of a walker tile: the estimate wanders by a few rows from launch to launch, and
            // a whole-tile count made a rank's ~495 live rows flip between 4 and 5 tiles of 128, i.e. between two shapes
            const double f128 = (double)(GP * ((ctx->Np + 127) / 128)) * (double)Wsel / 128.0 * 256.0 / ctx->num_cu;
            const double f64 = (double)(GP * nI64) * (double)Wsel / 64.0 * 256.0 / ctx->num_cu;
            // ... and the 64x32 / 64x64 switch point moves with the design size below N = 2048: a tile's K loop is half as
            // long at N = 1024 and its fixed cost (first loads, reduction, barriers) weighs twice as much, so the wider tile pays
            // from half as many tiles on (tools/gpu_tile_rule_sweep.py, profiles/r03_tile_rule_sweep.txt: cfg 3 at 512 live
            // rows 64x64 0.263 ms/step against 64x32 0.287; at 256 rows 64x32 0.179 against 0.199)
            const double nscale = ctx->Np < 2048 ? (double)ctx->Np / 2048.0 : 1.0;
            if (f128 >= (double)sw128) T = TN = 128;
            else if (f64 * 0.5 >= (double)swmid / nscale) TN = 128;          // (N = 1024: 64x128 from twice as many tiles on)
            else if (f64 >= (double)swnarrow * nscale) TN = 64;
        } else if (tiles128 * 256 >= sw128 * ctx->num_cu) T = TN = 128;
        else if (tiles64x128 * 256 >= swmid * ctx->num_cu) TN = 128;
        else if (tiles64 * 256 >= swnarrow * ctx->num_cu) TN = 64;
        if (ctx->force_tile == 64 || ctx->force_tile == 128) T = TN = ctx->force_tile;
        if (ctx->force_tile == 32) { T = 64; TN = 32; }
        if (ctx->force_tile == 65) { T = 64; TN = 128; }        // 64 rows x 128 walkers
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (ctx->profile) {                    // live HIP-event timing of the dominant kernel (bench.py)
            GPB_HIP(hipEventCreate(&e0));
            GPB_HIP(hipEventRecord(e0, ctx->stream));
            GPB_HIP(hipEventCreate(&e1));
        }
        const int nI = (T == 128) ? (int)((ctx->Np + 127) / 128) : nI64, nW = (int)(Wuse / TN);
        // which operand is larger decides the XCD affinity: L^-1 (P Np^2/2) or K*^T (P Np W)
        int xcd_rows = (2 * Wuse < ctx->Np) ? 1 : 0;
        if (ctx->force_xcd >= 0) xcd_rows = ctx->force_xcd;      // 0 or 1
        const int64_t ngroups = GP * nI;
        const int64_t nblocks = (xcd_rows == 1) ? ((ngroups + 7) / 8) * 8 * nW : ngroups * nW;
        // the persistent 128 x 128 launch: two workgroups per CU (their registers), or one per tile when fewer tiles exist; the
        // 64-row shapes launch static, one workgroup per tile in weight order — beyond co-residency the hardware dispatcher hands the
        // next workgroup to whichever CU frees a slot, which balances as well as ticket queues do (7-12 % faster at 384-768 walkers)
        const int64_t slots = (int64_t)ctx->num_cu * 2;
        const unsigned grid128 = (unsigned)(nblocks < slots ? nblocks : slots);
        if (ctx->batch_sliced) {
            // the int8 kernel (option key 51, gpb_sliced.hip) on the digit planes k_kcross left: same partials' layout and meaning
            if (multi) GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq: a sliced batch in a shared launch");
            const int rc = launch_vsq_sliced(ctx, W, nrows_dev, (int)kskip);
            if (rc) return rc;
        } else if (multi) {
            // one launch over the GPs of all the emulators (see k_predict_multi): the product shapes only
            PredTable tab;
            int g = 0;
            for (int e = 0; e < E; ++e)
                for (int pp = 0; pp < (int)ctxs[e]->P; ++pp, ++g)
                    tab.gp[g] = PredGP{ctxs[e]->Linv, ctxs[e]->KsT, ctxs[e]->spart, (int)ctxs[e]->P, pp};
            const int order = ctx->resident_order ? ctx->resident_order : 2;
            const int xr = xcd_rows;
            const unsigned nb_s = (unsigned)nblocks;
            if (T == 128)
                hipLaunchKernelGGL((k_predict_multi<128, 4, 128, 16, true>), dim3(grid128),
                                   dim3(256), 0, ctx->stream, tab, ctx->Np, ctx->Wld, (int)GP, nI, nW, xcd_rows,
                                   ctx->tile_counter, (unsigned)nblocks, ctx->tile_trace, tri_arg, nrows_dev);
            else if (TN == 32)
                hipLaunchKernelGGL((k_predict_static_multi<64, 4, 32, 16>), dim3(nb_s), dim3(256), 0, ctx->stream, tab, ctx->Np,
                                   ctx->Wld, (int)GP, nI, nW, xr, nb_s, order, (unsigned)(ctx->num_cu / 8),
                                   ctx->tile_priority ? nI : 0, ctx->tile_trace, tri_arg, nrows_dev);
            else if (TN == 128)
                hipLaunchKernelGGL((k_predict_static_multi<64, 4, 128, 16>), dim3(nb_s), dim3(256), 0, ctx->stream, tab, ctx->Np,
                                   ctx->Wld, (int)GP, nI, nW, xr, nb_s, order, (unsigned)(ctx->num_cu / 8),
                                   ctx->tile_priority ? nI : 0, ctx->tile_trace, tri_arg, nrows_dev);
            else
                hipLaunchKernelGGL((k_predict_static_multi<64, 4, 64, 16>), dim3(nb_s), dim3(256), 0, ctx->stream, tab, ctx->Np,
                                   ctx->Wld, (int)GP, nI, nW, xr, nb_s, order, (unsigned)(ctx->num_cu / 8),
                                   ctx->tile_priority ? nI : 0, ctx->tile_trace, tri_arg, nrows_dev);
        } else {
        // the shapes the rule can select: the persistent 128x128 tile with the fragment read-ahead (ticket queues) and the static
        // 64-row tiles (64x128, 64x64, 64x32; order 1-3, XCD map 0/1)
        if (T == 128) {
            hipLaunchKernelGGL((k_predict<128, 4, 128, 16, true>), dim3(grid128), dim3(256), 0, ctx->stream, ctx->Linv, ctx->KsT,
                               ctx->spart, ctx->Np, ctx->Wld, (int)ctx->P, nI, nW, xcd_rows, ctx->tile_counter,
                               (unsigned)nblocks, ctx->tile_trace, tri_arg, nrows_dev);
        } else {
            const int order = ctx->resident_order ? ctx->resident_order : 2;
            const int xr = xcd_rows;
#define GPB_PRED(NN)                                                                                                \
    hipLaunchKernelGGL((k_predict_static<64, 4, NN, 16>), dim3((unsigned)(xr == 1 ? ((ngroups + 7) / 8) * 8 * nW : ngroups * nW)), \
                       dim3(256), 0, ctx->stream, ctx->Linv, ctx->KsT, ctx->spart, ctx->Np, ctx->Wld, (int)ctx->P, nI, nW, xr, \
                       (unsigned)(xr == 1 ? ((ngroups + 7) / 8) * 8 * nW : ngroups * nW), order, (unsigned)(ctx->num_cu / 8), \
                       ctx->tile_priority ? nI : 0, ctx->tile_trace, tri_arg, nrows_dev)
            if (TN == 32) GPB_PRED(32);
            else if (TN == 128) GPB_PRED(128);
            else GPB_PRED(64);
        }
        }
#undef GPB_PRED
        if (ctx->profile) {
            GPB_HIP(hipEventRecord(e1, ctx->stream));
            ctx->prof_events.push_back({e0, e1});
            // GPs per timed interval (a chain: all of them; a pair of launches: both groups')
            ctx->prof_gps = (double)GP;
            if (!nrows_dev) ctx->prof_units += (double)GP * (double)W;          // compacted: counted on the device
            else ctx->prof_compacted = true;
        }
    }
    GPB_HIP(hipGetLastError());
    return 0;
}

int launch_finalize(gpb_ctx* ctx, int64_t W, bool need_var) {
    const int64_t Wuse = round_up(W, WPAD);
    const int nchunk = (int)((ctx->Np + KX_CHUNK - 1) / KX_CHUNK), nI64 = (int)(ctx->Np / 64);
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((Wuse + 255) / 256), (unsigned)ctx->P), dim3(256), 0,
                       ctx->stream, ctx->mpart, ctx->spart, ctx->amp, ctx->noise, ctx->mean_pc, ctx->var_pc,
                       ctx->Wld, Wuse, (int)ctx->P, nchunk, nI64, need_var ? 1 : 0);
    GPB_HIP(hipGetLastError());
    return 0;
}

// Xs_dev: [W][d] on the device.  Results land in ctx->mean_pc / var_pc ([P][Wcap]) when finalize is set, else in the
// partials (mpart, spart) for a consumer that sums them itself.
int launch_predict(gpb_ctx* ctx, const double* Xs_dev, int64_t W, bool need_var, bool finalize, const int* nrows_dev) {
    int rc = launch_kcross(ctx, Xs_dev, W, nrows_dev);
    if (rc) return rc;
    gpb_ctx* one[1] = {ctx};
    if (need_var && (rc = launch_vsq(one, 1, W, nrows_dev))) return rc;
    if (finalize) return launch_finalize(ctx, W, need_var);
    return 0;
}


#ifdef GPB_DEBUG_VARIANTS      // self-test and issue-rate probes (include/gpbayes_debug.h)
// ------------------------------------------------------------------ test hooks
// MODE: 0 = C = A[M,K] B[K,N]; 1 = C = A[M,K] B[N,K]^T; 2 = C = A[K,M]^T B[K,N]
template <int T, int MODE>
__global__ __launch_bounds__(256, 2) void k_test_gemm(const double* A, const double* B, double* C, int64_t M,
                                                      int64_t N, int64_t K) {
    __shared__ TileLds<T> lds;
    const int64_t mb = (int64_t)blockIdx.y * T, nb = (int64_t)blockIdx.x * T;
    const int me = (int)imin64(T, M - mb), ne = (int)imin64(T, N - nb);
    Acc<T> acc;
    acc_zero<T>(acc);
    if (MODE == 0) gemm_tile_loop<T, false, false>(A, K, B, N, mb, nb, me, ne, 0, K, lds, acc);
    else if (MODE == 1) gemm_tile_loop<T, false, true>(A, K, B, K, mb, nb, me, ne, 0, K, lds, acc);
    else gemm_tile_loop<T, true, false>(A, M, B, N, mb, nb, me, ne, 0, K, lds, acc);
    tile_store<T>(C, N, mb, nb, me, ne, 1.0, false, acc);
}

// b_trans: bits 0-1 = MODE above, bit 2 set = 64x64 tiles instead of 128x128
int launch_test_gemm(gpb_ctx* ctx, int64_t M, int64_t N, int64_t K, const double* A, const double* B, double* C,
                     int b_trans) {
    const int mode = b_trans & 3, T = (b_trans & 4) ? 64 : 128;
    dim3 grid((unsigned)((N + T - 1) / T), (unsigned)((M + T - 1) / T));
#define GPB_TG(TT, MM) hipLaunchKernelGGL((k_test_gemm<TT, MM>), grid, dim3(256), 0, ctx->stream, A, B, C, M, N, K)
    if (T == 128) { if (mode == 0) GPB_TG(128, 0); else if (mode == 1) GPB_TG(128, 1); else GPB_TG(128, 2); }
    else          { if (mode == 0) GPB_TG(64, 0);  else if (mode == 1) GPB_TG(64, 1);  else GPB_TG(64, 2); }
#undef GPB_TG
    GPB_HIP(hipGetLastError());
    return 0;
}

// issue-rate probes: one wave per SIMD (256 threads/WG, 1 WG/CU x 256 CUs x 4 rounds)
__global__ __launch_bounds__(256) void k_probe_mfma(double* out, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    const d4 s = a0 + a1 + a2 + a3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];
}
// same loop, stamped with the shader clock (s_memtime) and the 100 MHz reference (s_memrealtime)
__global__ __launch_bounds__(256) void k_probe_mfma_clk(double* out, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    const d4 s = a0 + a1 + a2 + a3;
    asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = (double)(t1 - t0);
        out[1] = (double)(r1 - r0);
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[2] = s[0];
}
__global__ __launch_bounds__(256) void k_probe_fma(double* out, int iters) {
    double a[8];
    const double x = 1.0 + threadIdx.x * 1e-12, y = 1e-9 * threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fma(a[k], x, y);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 12345.678) out[0] = s;
}
// waves 0-3 (one per SIMD) issue MFMA, waves 4-7 issue v_fma_f64: do the pipes overlap for fp64?
__global__ __launch_bounds__(512) void k_probe_both(double* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
        const d4 s = a0 + a1 + a2 + a3;
        if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];
    } else {
        double a[8];
        const double x = 1.0 + threadIdx.x * 1e-12, y = 1e-9 * threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = k;
        for (int i = 0; i < iters * 16; ++i) {   // 16*8 wave-FMAs = 128*64*2 flop vs 4 MFMA = 4*2048 flop... same order
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = fma(a[k], x, y);
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += a[k];
        if (s == 12345.678) out[1] = s;
    }
}

int launch_probe(gpb_ctx* ctx, int mode, double* tflops) {
    double* d_out = nullptr;
    GPB_HIP(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    GPB_HIP(hipEventCreate(&e0));
    GPB_HIP(hipEventCreate(&e1));
    const int iters = 4000, blocks = 1024;
    double flops = 0.0;
    for (int rep = 0; rep < 2; ++rep) {
        GPB_HIP(hipEventRecord(e0, ctx->stream));
        if (mode == 0) {
            hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(256), 0, ctx->stream, d_out, iters);
            flops = (double)blocks * 4 * iters * 4 * 2048.0;
        } else if (mode == 3 || mode == 4) {
            hipLaunchKernelGGL(k_probe_mfma_clk, dim3(mode == 3 ? 256 : 1024), dim3(256), 0, ctx->stream, d_out, iters);
        } else if (mode == 1) {
            hipLaunchKernelGGL(k_probe_fma, dim3(blocks), dim3(256), 0, ctx->stream, d_out, iters * 16);
            flops = (double)blocks * 256 * (double)iters * 16 * 8 * 2.0;
        } else {
            hipLaunchKernelGGL(k_probe_both, dim3(blocks), dim3(512), 0, ctx->stream, d_out, iters);
            flops = (double)blocks * 4 * iters * 4 * 2048.0 + (double)blocks * 256 * (double)iters * 16 * 8 * 2.0;
        }
        GPB_HIP(hipEventRecord(e1, ctx->stream));
        GPB_HIP(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    GPB_HIP(hipEventElapsedTime(&ms, e0, e1));
    *tflops = flops / (ms * 1e-3) / 1e12;
    if (mode == 3 || mode == 4) {
        double h[2] = {0, 0};
        GPB_HIP(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        // mode 3: shader cycles per MFMA with one wave per SIMD; mode 4: clock (GHz) held with 4 waves per SIMD
        *tflops = (mode == 3) ? h[0] / (4.0 * iters) : h[0] / h[1] * 0.1;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    return 0;
}

#endif  // GPB_DEBUG_VARIANTS

}  // namespace gpb
