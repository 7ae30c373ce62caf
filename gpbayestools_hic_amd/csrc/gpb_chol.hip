// gpb_chol.hip — blocked right-looking fp64 Cholesky of K + (sigma_n^2 + alpha) I for all P GPs of a context
// (sk:_gpr.py:349 `cholesky(K, lower=True)` via src/emulator.py:309-315), organised around its serial chain.
//
// The factorisation of one GP is a chain of Np/64 dependent steps (factor the 64x64 diagonal block, solve the
// column panel below it, update what the next step reads); only the P GPs run side by side.  Round 1 spent 3
// launches and ~58 us per step, 36 us of them in the diagonal-block kernel on P workgroups while the rest of the
// chip idled (profiles/r01_fit_kernel_stats_final.csv).  Here a step is TWO launches:
//
//   k_chol_trsm    L_ik = A_ik L_kk^-T for all row blocks i > k: one 64x64x64 MFMA product per workgroup, both
//                  operand tiles resident in LDS (no K-loop pipeline: one barrier), written in place.
//   k_chol_update  A_ij -= L_ik L_jk^T for the tiles of the columns j of the current outer panel (K = 64), and — in
//                  the workgroup that owns tile (k+1, k+1), dispatched first — the NEXT step's diagonal block:
//                  it is factored and inverted right there, from LDS, while the other tiles are still being updated.
//
// Once per outer panel (chol_outer columns) the whole trailing matrix is updated with K = panel width on the tile
// engine (k_syrk in gpb_fit.hip), followed by a stand-alone diagonal-block launch (k_chol_diag) for the first block of
// the next panel.
//
// potf2_inv_64 (the chain's body): 16x16 sub-blocks factored in ONE wave's registers (lane i = row i, readlane
// broadcasts, no barriers: ~140 cycles per pivot, which is the rsqrt's dependent latency), the 48/32/16 rows below
// solved by right-looking substitution (one row per lane, 16 dependent steps), the trailing part of the block and
// the assembly of the inverse (block doubling 16 -> 32 -> 64) on v_mfma_f64_16x16x4_f64 straight from LDS.
#include "gpb_internal.h"
#include "gemm_tile.h"
#include "chol_block.h"
#include <math.h>
#include <vector>

// the trailing matrix's tile of k_chol_update / k_chol_update2 is read once and written once per step: the write carries the
// streaming hint (global_store ... nt), so that it does not push the step's operand blocks — shared by a whole row / column of
// tiles — out of the L2.  A/B in one process (tools/micro/fit_ab.py, profiles/r04_nt_store_ab.txt): Cholesky 0.7-1.1 % shorter at
// N = 1024 .. 2048, unchanged at 3072 / 4096; the hint on the READ of the tile as well: -3 % at N = 1024 and 3072 but +1.6 % at 1536;
// on the read alone: +1.5 .. +3 % at N <= 1536 — not taken.
#define GPB_C_LOAD(p) (*(p))
#define GPB_C_STORE(p, v) __builtin_nontemporal_store((v), (p))

namespace gpb {

namespace {

// s.a -> L block of K (upper zeroed), s.x -> diagonal block of Linv; reports a non-positive pivot
__device__ __forceinline__ void store_diag(const CholLds& s, double* __restrict__ Kb, double* __restrict__ Xb, int64_t Np,
                                           int64_t c0, int* __restrict__ info_p, bool first = false) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + CHOL_THREADS * e;
        const int r = idx >> 5, c = (idx & 31) * 2;
        const d2 lv = {c <= r ? s.a[r][c] : 0.0, c + 1 <= r ? s.a[r][c + 1] : 0.0};
        const d2 xv = {c <= r ? s.x[r][c] : 0.0, c + 1 <= r ? s.x[r][c + 1] : 0.0};
        *reinterpret_cast<d2*>(Kb + (int64_t)r * Np + c) = lv;
        *reinterpret_cast<d2*>(Xb + (int64_t)r * Np + c) = xv;
    }
    // LAPACK dpotrf's info: the first non-positive pivot of the GP.  The first diagonal block of a factorisation also clears
    // what the previous one left (instead of a memset in front of the chain: two fill kernels, ~10 us)
    if (tid == 0) {
        if (first) *info_p = s.bad >= 0 ? (int)(c0 + s.bad + 1) : 0;
        else if (s.bad >= 0 && *info_p == 0) *info_p = (int)(c0 + s.bad + 1);
    }
}

}  // namespace

// Diagonal block kb alone (first block of an outer panel: its last update came from the panel-wide SYRK).
__global__ __launch_bounds__(CHOL_THREADS, 4) void k_chol_diag(double* __restrict__ K, double* __restrict__ Linv, int64_t Np,
                                                            int64_t kb, int* __restrict__ info) {
    __shared__ CholLds s;
    const int p = blockIdx.x;
    const int64_t c0 = kb * 64;
    double* Kb = K + (int64_t)p * Np * Np + c0 * Np + c0;
    load_tile(Kb, Np, s.a, threadIdx.x);
    potf2_inv_64(s);                                   // opens with a barrier
    store_diag(s, Kb, Linv + (int64_t)p * Np * Np + c0 * Np + c0, Np, c0, info + p, kb == 0);
}

// Column panel: L_ik = A_ik L_kk^-T for row block i = kb + 1 + blockIdx.x, in place.
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_trsm(double* __restrict__ K, const double* __restrict__ Linv,
                                                            int64_t Np, int64_t kb, unsigned P) {
    __shared__ double sa[64][LDP], sx[64][LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = (int)(blockIdx.x % P), bx = (int)(blockIdx.x / P);      // GP fastest: see k_chol_update
    const int64_t c0 = kb * 64, r0 = (kb + 1 + bx) * 64;
    double* Ap = K + (int64_t)p * Np * Np + r0 * Np + c0;
    load_tile(Ap, Np, sa, tid);
    load_tile(Linv + (int64_t)p * Np * Np + c0 * Np + c0, Np, sx, tid);
    __syncthreads();
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    mma_nt_64(sa, sx, acc, wave, lane);                // out[i][j] = sum_k A[i][k] X[j][k]
    const int m0 = (wave >> 2) * 32, n0 = (wave & 3) * 16, lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ap[(int64_t)(m0 + 16 * t + lk + 4 * r) * Np + n0 + lr] = acc[t][r];
}

// In-panel trailing update by block column kb (K = 64) + the next diagonal block.
// Tiles (i, j), kb < j < je, j <= i < nb, numbered column by column; tile 0 = (kb+1, kb+1) is the next step's
// diagonal block: its workgroup keeps the updated block in LDS, factors and inverts it, and stores L and L^-1.
__global__ __launch_bounds__(CHOL_THREADS, 4) void k_chol_update(double* __restrict__ K, double* __restrict__ Linv,
                                                              int64_t Np, int64_t kb, int64_t je,
                                                              int* __restrict__ info, unsigned P) {
    __shared__ CholLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, GP fastest: tile t of ALL GPs is dispatched before tile t + 1 of any.  With the GP as grid.y the chain's
    // workgroup (tile 0) of the last GP stood behind every tile of the GPs before it — 4464 workgroups at the first step
    // of N = 2048 with 10 GPs — and the step waited for it.
    const int p = (int)(blockIdx.x % P);
    const unsigned bx = blockIdx.x / P;
    const int64_t nb = Np / 64;
    int64_t j = kb + 1, t = bx;
    while (t >= nb - j) { t -= nb - j; ++j; }          // column j holds nb - j tiles (rows j .. nb-1)
    const int64_t i = j + t;
    if (j >= je) return;
    double* Kp = K + (int64_t)p * Np * Np;
    const int64_t c0 = kb * 64;
    load_tile(Kp + i * 64 * Np + c0, Np, s.a, tid);
    if (i != j) load_tile(Kp + j * 64 * Np + c0, Np, s.x, tid);
    const int m0 = (wave >> 2) * 32, n0 = (wave & 3) * 16, lr = lane & 15, lk = lane >> 4;
    double* Cp = Kp + i * 64 * Np + j * 64;
    d4 c[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[tt][r] = GPB_C_LOAD(&Cp[(int64_t)(m0 + 16 * tt + lk + 4 * r) * Np + n0 + lr]);
    __syncthreads();
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    mma_nt_64(s.a, i != j ? s.x : s.a, acc, wave, lane);
    if (bx != 0) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) GPB_C_STORE(&Cp[(int64_t)(m0 + 16 * tt + lk + 4 * r) * Np + n0 + lr], c[tt][r] - acc[tt][r]);
        return;
    }
    __syncthreads();                                   // every wave is done reading the operand tile in s.a
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s.a[m0 + 16 * tt + lk + 4 * r][n0 + lr] = c[tt][r] - acc[tt][r];
    potf2_inv_64(s);                                   // opens with a barrier
    store_diag(s, Cp, Linv + (int64_t)p * Np * Np + j * 64 * Np + j * 64, Np, j * 64, info + p);
}

// Trailing update by TWO block columns at once (K = 128) + the next diagonal block: the second half of a column PAIR (kb, kb + 1).
// Tiles (i, j), kb + 1 < j, j <= i < nb, numbered column by column; tile 0 = (kb + 2, kb + 2) is the next pair's first diagonal
// block (kept in LDS, factored and inverted right there, as in k_chol_update).  A tile's C is read and written ONCE for the two
// columns' updates: half the trailing-matrix traffic of two K = 64 steps — the K = 64 update moves 64 KB of C per 0.5 MFLOP and runs
// at the HBM's pace (3.7 TB/s of C traffic, matrix pipes 27 % busy: profiles/r04_fit_pmc.json).  The operand tiles of column kb + 1
// are fetched into registers while column kb's are multiplied.
__global__ __launch_bounds__(CHOL_THREADS, 4) void k_chol_update2(double* __restrict__ K, double* __restrict__ Linv,
                                                               int64_t Np, int64_t kb, int* __restrict__ info, unsigned P) {
    __shared__ CholLds s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = (int)(blockIdx.x % P);               // GP fastest: see k_chol_update
    const unsigned bx = blockIdx.x / P;
    const int64_t nb = Np / 64;
    int64_t j = kb + 2, t = bx;
    while (t >= nb - j) { t -= nb - j; ++j; }          // column j holds nb - j tiles (rows j .. nb-1)
    const int64_t i = j + t;
    double* Kp = K + (int64_t)p * Np * Np;
    const int64_t c0 = kb * 64, c1 = c0 + 64;
    const bool diag = i == j;
    load_tile(Kp + i * 64 * Np + c0, Np, s.a, tid);
    if (!diag) load_tile(Kp + j * 64 * Np + c0, Np, s.x, tid);
    TileRegs na, nx;                                   // column kb + 1's operand tiles, in flight under the first product
    gload_tile(Kp + i * 64 * Np + c1, Np, na, tid);
    if (!diag) gload_tile(Kp + j * 64 * Np + c1, Np, nx, tid);
    const int m0 = (wave >> 2) * 32, n0 = (wave & 3) * 16, lr = lane & 15, lk = lane >> 4;
    double* Cp = Kp + i * 64 * Np + j * 64;
    d4 c[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[tt][r] = GPB_C_LOAD(&Cp[(int64_t)(m0 + 16 * tt + lk + 4 * r) * Np + n0 + lr]);
    __syncthreads();
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    mma_nt_64(s.a, diag ? s.a : s.x, acc, wave, lane);
    __syncthreads();                                   // every wave is done reading column kb's operand tiles
    lstore_tile(s.a, na, tid);
    if (!diag) lstore_tile(s.x, nx, tid);
    __syncthreads();
    mma_nt_64(s.a, diag ? s.a : s.x, acc, wave, lane);  // k continues: one sum over the 128 columns of the pair
    if (bx != 0) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) GPB_C_STORE(&Cp[(int64_t)(m0 + 16 * tt + lk + 4 * r) * Np + n0 + lr], c[tt][r] - acc[tt][r]);
        return;
    }
    __syncthreads();                                   // every wave is done reading the operand tile in s.a
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s.a[m0 + 16 * tt + lk + 4 * r][n0 + lr] = c[tt][r] - acc[tt][r];
    potf2_inv_64(s);                                   // opens with a barrier
    store_diag(s, Cp, Linv + (int64_t)p * Np * Np + j * 64 * Np + j * 64, Np, j * 64, info + p);
}

// declared in gpb_fit.hip: trailing update by the panel [pb, pe): rows >= r0, columns [r0, ce), K = pe - pb
void launch_syrk_range(gpb_ctx* ctx, hipStream_t stream, int64_t pb, int64_t pe, int64_t r0, int64_t ce);

namespace {
int ensure_lookahead(gpb_ctx* ctx, size_t nev) {
    if (!ctx->side_stream) {
        // lowest priority: the far update must only fill what the chain leaves idle (at equal priority the chain's
        // 10-workgroup diagonal kernel waited 105 us instead of 16 behind the side stream's tiles)
        // lowest priority: the far update should only fill what the chain leaves idle.  (It still delays the chain:
        // once its tiles occupy every CU a 512-thread chain workgroup waits for a hole — 105-130 us instead of 16 for
        // the first diagonal kernel after a panel, priority or not; masking the side stream off a share of the CUs
        // (hipExtStreamCreateWithCUMask, 3/4 and 1/2) cost 30-50 % instead.  Net gain of the lookahead: 2-4 %.)
        int lo = 0, hi = 0;
        GPB_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        GPB_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
    }
    while (ctx->chol_events.size() < nev) {
        hipEvent_t e;
        GPB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->chol_events.push_back(e);
    }
    return 0;
}
}  // namespace

// Schedule: per outer panel [pb, pe) the chain (diagonal block, then per 64-column step k_chol_trsm + k_chol_update),
// then the panel's update of the trailing matrix with K = panel width — split in two (lookahead): the NEAR part, the
// next panel's own columns [pe, pe + NBO), stays on the chain's stream; the FAR part, columns >= pe + NBO, goes to a
// side stream and runs underneath the next panel's chain, which leaves nine tenths of the matrix cores idle (two events
// per panel; the far parts of consecutive panels are ordered by the side stream itself).
// Column PAIRS (option 47, chol_pair): every second step's update is left out and the step after it updates the trailing matrix by
// both columns at once (k_chol_update2) — per pair: column solve of kb, update of column kb + 1 alone (+ its diagonal block), column
// solve of kb + 1, K = 128 update of everything to the right (+ the next diagonal block).  One "panel" over the whole matrix: no
// k_syrk, no lookahead.  Measured (profiles/r04_fit_notes.txt): N = 2048 1303-1309 us against 1322-1340 for single steps (half the
// trailing-matrix traffic buys 2-3 %: the update is bound by the latency of a tile in its slot, not by the HBM), N = 1024 +1 %
// (the chain), N = 1536 754 against 735, N = 3072 3270 against 3300, N = 4096 6770-6940 us against 6150-6270 for panels of 256 with
// k_syrk.  Round 5: with 63 GPs per launch (the batch train_emulators factors) N = 1024 is bound by the bulk, not the chain, and pairs
// win 11-14 % (1129-1156 -> 966-1026 us; panels of 256 + k_syrk: 1050; profiles/r05_fit_notes.txt) — used for 1024 <= Np <= 3072,
// whatever the number of GPs (10 GPs at N = 1024 / 1536 pay 1-3 % for it).
static int launch_potrf_pairs(gpb_ctx* ctx) {
    const int64_t Np = ctx->Np, nb = Np / 64;
    const unsigned P = (unsigned)ctx->P;
    hipLaunchKernelGGL(k_chol_diag, dim3(P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv, Np, (int64_t)0, ctx->info);
    for (int64_t kb = 0; kb + 1 < nb; kb += 2) {
        hipLaunchKernelGGL(k_chol_trsm, dim3((unsigned)(nb - kb - 1) * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv,
                           Np, kb, P);
        // column kb + 1 alone: its nb - kb - 1 tiles, the first of which is the diagonal block (k_chol_update with je = kb + 2)
        hipLaunchKernelGGL(k_chol_update, dim3((unsigned)(nb - kb - 1) * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K,
                           ctx->Linv, Np, kb, kb + 2, ctx->info, P);
        if (kb + 2 >= nb) break;                       // column kb + 1 was the last
        hipLaunchKernelGGL(k_chol_trsm, dim3((unsigned)(nb - kb - 2) * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv,
                           Np, kb + 1, P);
        int64_t ntile = 0;
        for (int64_t j = kb + 2; j < nb; ++j) ntile += nb - j;
        hipLaunchKernelGGL(k_chol_update2, dim3((unsigned)ntile * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv, Np,
                           kb, ctx->info, P);
    }
    GPB_HIP(hipGetLastError());
    return 0;
}

int launch_potrf_fused(gpb_ctx* ctx) {
    // (the rule reads Np alone — never the number of GPs of the launch: a GP's bits must not depend on its neighbours, which is what
    // keeps train_emulators' batched searches identical to the one-emulator ones)
    // (round 6: the first 256-1536 columns in panels of 256 with the panel-wide SYRK, K = 256, and the pairs behind them: -0.5 % at
    // N = 2048 with 256 columns, slower beyond; -2.3 % at N = 3072, -1.5 % for 63 GPs at N = 1024: below the 8 % it had to bring,
    // removed — profiles/r06_fit_notes.txt)
    if (ctx->chol_outer == 0 && (ctx->chol_pair == 2 || (ctx->chol_pair == 1 && ctx->Np >= 1024 && ctx->Np <= 3072)))
        return launch_potrf_pairs(ctx);
    const int64_t Np = ctx->Np, nb = Np / 64;
    // outer panel width (multiple of 64).  0 = by size (tools/gpu_fit_timing.py, 10 GPs): up to N = 2048 ONE panel — every
    // step updates the whole trailing matrix with K = 64 and no panel-end SYRK is left (0.74 -> 0.70 ms at 1024, 2.41 ->
    // 2.28 ms at 2048: the K = 64 updates' HBM traffic is cheaper than the extra launches and the tail of a SYRK); beyond
    // that 256 with lookahead (N = 4096: 11.7 ms; 512: 12.0; 1024: 12.5; one panel: 13.8)
    const int64_t NBO = ctx->chol_outer > 0 ? ctx->chol_outer : (Np <= 2048 ? Np : 256);
    const unsigned P = (unsigned)ctx->P;
    const int64_t npanel = (Np + NBO - 1) / NBO;
    const bool look = ctx->chol_lookahead && npanel > 2;
    if (look) {
        int rc = ensure_lookahead(ctx, (size_t)(2 * npanel));
        if (rc) return rc;
    }
    int64_t ip = 0;                                    // (info: cleared by the first k_chol_diag)
    bool far_pending = false;
    for (int64_t pb = 0; pb < Np; pb += NBO, ++ip) {
        const int64_t pe = imin64(pb + NBO, Np), je = pe / 64;
        hipLaunchKernelGGL(k_chol_diag, dim3(P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv, Np, pb / 64,
                           ctx->info);
        for (int64_t kb = pb / 64; kb < je; ++kb) {
            const int64_t rem = nb - kb - 1;
            if (rem <= 0) break;
            hipLaunchKernelGGL(k_chol_trsm, dim3((unsigned)rem * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K, ctx->Linv,
                               Np, kb, P);
            if (kb + 1 < je) {
                int64_t ntile = 0;
                for (int64_t j = kb + 1; j < je; ++j) ntile += nb - j;
                hipLaunchKernelGGL(k_chol_update, dim3((unsigned)ntile * P), dim3(CHOL_THREADS), 0, ctx->stream, ctx->K,
                                   ctx->Linv, Np, kb, je, ctx->info, P);
            }
        }
        if (pe >= Np) break;
        const int64_t near_end = imin64(pe + NBO, Np);
        if (!look || near_end >= Np) {                 // no far part (or lookahead off): everything on the chain's stream
            if (far_pending) {                         // the previous far part wrote into these columns
                GPB_HIP(hipEventRecord(ctx->chol_events[2 * ip - 1], ctx->side_stream));
                GPB_HIP(hipStreamWaitEvent(ctx->stream, ctx->chol_events[2 * ip - 1], 0));
                far_pending = false;
            }
            launch_syrk_range(ctx, ctx->stream, pb, pe, pe, Np);
            continue;
        }
        // panel [pb, pe) is final: the far part may start
        GPB_HIP(hipEventRecord(ctx->chol_events[2 * ip], ctx->stream));
        if (far_pending) {                             // near part: after the previous far part, which wrote there
            GPB_HIP(hipEventRecord(ctx->chol_events[2 * ip - 1], ctx->side_stream));
            GPB_HIP(hipStreamWaitEvent(ctx->stream, ctx->chol_events[2 * ip - 1], 0));
        }
        launch_syrk_range(ctx, ctx->stream, pb, pe, pe, near_end);
        GPB_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->chol_events[2 * ip], 0));
        launch_syrk_range(ctx, ctx->side_stream, pb, pe, near_end, Np);
        far_pending = true;
    }
    if (far_pending) {                                 // cannot happen (the last panel with a trailing part has no far part)
        GPB_HIP(hipEventRecord(ctx->chol_events[0], ctx->side_stream));
        GPB_HIP(hipStreamWaitEvent(ctx->stream, ctx->chol_events[0], 0));
    }
    GPB_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpb
