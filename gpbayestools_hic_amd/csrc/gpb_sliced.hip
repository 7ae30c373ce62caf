// gpb_sliced.hip — V = L^-1 K*^T with the fused sum of squares on the INT8 matrix pipe (option key 51; round 6).
//
// What it replaces: k_predict's fp64 tiles (gpb_predict.hip; sk:_gpr.py:454-460, src/emulator.py:553,573-575) for the contexts
// the rule below admits, at every batch size (a walker's bits must not depend on the batch it arrives in).  Scheme, accuracy and costs: profiles/r06_sliced_model.txt, tools/ozaki_probe.py,
// tools/micro/sliced_probe.hip (the micro-kernel this file grew from).
//
//   * row j of L^-1 is scaled by 2^-e_j (max|row| <= 0.99 2^e_j) and rounded ONCE to a 47-bit integer, K*^T (in (0, c]) by ONE
//     power of two per GP; the six signed radix-256 digits of each (bytes of (a + 0x808080808080) XOR 0x80) are int8 planes;
//   * the digit pairs of one level ta + tb are summed EXACTLY in one int32 accumulator set by v_mfma_i32_32x32x32_i8; the 21
//     pairs of levels 5..10 are kept; the epilogue combines the six levels in fp64 (Horner from the least significant one),
//     scales, squares and reduces over rows in an order fixed by the row index: spart[64-row block][GP][walker], the layout and
//     the meaning of k_predict's output.  Integer sums are exact, so a walker's bits do not depend on tiles, batch cuts,
//     compaction or rank counts — by construction, not by ordering.
//   * accuracy: the variance c + sn2 - sum v^2 within ~1.4e-13 x (c + sn2) / var of exact arithmetic (fp64 GEMM: ~1e-15 x);
//     var >= sn2, so the RULE "every GP of the context has 1 + c / sn2 <= 128" (theta alone) keeps the 1e-10 bar with > 5x
//     margin; contexts outside it stay on the fp64 kernel.
//
// Plane layout in HBM: plane[p][t][k / 16][row][16 bytes]: the 16 k-consecutive bytes an MFMA lane takes are one granule and
// granules of consecutive rows (walkers) are contiguous, so a tile's share of a plane and k-block is ONE contiguous piece that
// LDS-DMA (global_load_lds_dwordx4) moves without staging registers into the same layout in LDS, from where ds_read_b128
// delivers fragments with no transposition and no bank conflict.
//
// Kernel: 128 x 128 block tile (128 x 64 for small batches), 8 waves (two per SIMD) of 32 rows x 64 (32) walkers (6 x 2 x 16 = 192
// accumulator registers), K-step 32, three LDS stages of 48 (36) KB with two DMA stages in flight across raw s_barriers (counted
// vmcnt: a __syncthreads() would drain them), the DMA of a step issued BEHIND its MFMAs, tiles in super-blocks (one GP x 8 row
// blocks x 4 walker tiles = the 32 workgroups of an XCD) dealt round-robin to the XCDs, heaviest first.
#include "gpb_internal.h"

namespace gpb {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* sl_lds_ptr;

constexpr int SL_D = 6, SL_LOW = 5, SL_NLEV = 2 * SL_D - 1 - SL_LOW;            // 6 digits, levels 5..10, 21 products
constexpr int SL_BM = 128;
// block tile 128 rows x (64 WTN) walkers: WTN = 2 for batches that fill the chip with 128 x 128 tiles, WTN = 1 (half the
// accumulators, 3/4 of the bytes per K-step for half the MFMAs: bound by the operand feed) for smaller ones — a rank's share of a
// sharded ensemble.  The shape never changes a bit: integer sums are exact and the epilogue's order is the row's.
template <int WTN>
struct SlGeo {
    static constexpr int BN = 64 * WTN;
    static constexpr int A_CHUNKS = SL_D * 2 * (SL_BM / 64), B_CHUNKS = SL_D * 2 * WTN, CHUNKS = A_CHUNKS + B_CHUNKS;   // 1-KB pieces per stage
    static constexpr int STAGE = CHUNKS * 1024, NSTAGE = 3;
    static constexpr int LDS = NSTAGE * STAGE + SL_BM * 8;                   // + the tile's row scales
    static constexpr int NPW = (CHUNKS + 7) / 8, REM = CHUNKS % 8;           // DMAs per wave and stage: NPW for waves < REM (all if REM = 0), else NPW - 1
};
constexpr int SL_RG = 8, SL_CG = 4;                                            // super-block: 8 row blocks x 4 walker tiles
constexpr double SL_RULE = 128.0;                                              // 1 + c / sn2 above this: fp64 kernel

// ---------------------------------------------------------------------------------------------------------------- digits
// six signed digits of a 47-bit integer: byte t of (a + 0x808080808080) XOR 0x80
__device__ __forceinline__ unsigned long long sl_digits(long long a) {
    return ((unsigned long long)(a + 0x808080808080ll)) ^ 0x808080808080ull;
}

// power-of-two exponent e with m <= 0.99 * 2^e (m > 0)
__device__ __forceinline__ int sl_exponent(double m) {
    int ex;
    const double f = frexp(m, &ex);              // m = f 2^ex, f in [0.5, 1)
    return f <= 0.99 ? ex : ex + 1;
}

// rowscale[p][j] = 2^(e_j - 14) (Horner's result is in units of the top level: 2^(e_j + e_c - 94 + 80)), rowexp the exponent itself;
// colscale[p] = 2^(e_c), colexp[p] = e_c.  One wave per row.
__global__ __launch_bounds__(256) void k_sl_rowscale(const double* __restrict__ Linv, const double* __restrict__ amp,
                                                     double* __restrict__ rowscale, int* __restrict__ rowexp,
                                                     double* __restrict__ colscale, int64_t Np, int64_t Np128) {
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= Np128) return;
    double m = 0.0;
    if (j < Np) {
        const double* row = Linv + ((int64_t)p * Np + j) * Np;
        for (int64_t k = lane; k <= j; k += 64) m = fmax(m, fabs(row[k]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    }
    if (lane == 0) {
        const int e = m > 0.0 ? sl_exponent(m) : 0;
        rowexp[(int64_t)p * Np128 + j] = e;
        rowscale[(int64_t)p * Np128 + j] = ldexp(1.0, e - 14);
        if (j == 0) colscale[p] = ldexp(1.0, sl_exponent(amp[p]));
    }
}

// planes of L^-1: thread = (row j, k-block kb) of the lower triangle; 16 doubles in, six 16-byte granules out.  The upper
// triangle and the rows behind Np are zero from the buffer's one memset.
__global__ __launch_bounds__(256) void k_sl_slice_linv(const double* __restrict__ Linv, const int* __restrict__ rowexp,
                                                       int8_t* __restrict__ planes, int64_t Np, int64_t Np128) {
    const int p = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, kb = blockIdx.y;
    if (j >= Np || kb * 16 > j) return;
    const double* src = Linv + ((int64_t)p * Np + j) * Np + kb * 16;
    const int e = rowexp[(int64_t)p * Np128 + j];
    unsigned long long dg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double x = (kb * 16 + i <= j) ? src[i] : 0.0;
        dg[i] = sl_digits(__double2ll_rn(ldexp(x, 47 - e)));
    }
    const int64_t plane = (Np / 16) * Np128 * 16;
    int8_t* dst = planes + (int64_t)p * SL_D * plane + (kb * Np128 + j) * 16;
#pragma unroll
    for (int t = 0; t < SL_D; ++t) {
        unsigned w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            w[g] = (unsigned)((dg[4 * g] >> (8 * t)) & 0xff) | (unsigned)(((dg[4 * g + 1] >> (8 * t)) & 0xff) << 8) |
                   (unsigned)(((dg[4 * g + 2] >> (8 * t)) & 0xff) << 16) | (unsigned)(((dg[4 * g + 3] >> (8 * t)) & 0xff) << 24);
        *reinterpret_cast<uint4*>(dst + t * plane) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// (the planes of K*^T are written by k_kcross itself, SLICE form: gpb_predict.hip)

// ---------------------------------------------------------------------------------------------------------------- the tile kernel
// wave-uniform: this wave's share of the DMA of one K-step (k-blocks kb0, kb0 + 1) into stage buffer `buf`
template <int WTN>
__device__ __forceinline__ void sl_dma_stage(char* lds, int buf, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                             int64_t a_plane, int64_t b_plane, int64_t Np128, int64_t Wld, int64_t mb, int64_t nb,
                                             int64_t kb0, int wave, int lane) {
    typedef SlGeo<WTN> G;
    char* base = lds + buf * G::STAGE;
#pragma unroll
    for (int c = 0; c < G::NPW; ++c) {
        const int ch = 8 * c + wave;                                    // wave-uniform
        if (c == G::NPW - 1 && G::REM != 0 && wave >= G::REM) break;
        const bool is_a = ch < G::A_CHUNKS;
        const int cb = is_a ? ch : ch - G::A_CHUNKS;
        const int per = is_a ? SL_BM / 64 : WTN;                        // 1-KB pieces per (plane, k-block)
        const int t = cb / (2 * per), q = (cb / per) & 1, h = cb % per;
        const int64_t ld = is_a ? Np128 : Wld, off = is_a ? mb : nb;
        const int8_t* plane = is_a ? Ap + t * a_plane : Bp + t * b_plane;
        const int8_t* src = plane + ((kb0 + q) * ld + off + 64 * h + lane) * 16;
        __builtin_amdgcn_global_load_lds((const void*)src, (sl_lds_ptr)(base + ch * 1024), 16, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void sl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// all of this wave's DMAs but those of its newest stage have landed (more: every one)
template <int WTN>
__device__ __forceinline__ void sl_wait_stage(bool more, bool full_count) {
    typedef SlGeo<WTN> G;
    if (!more) sl_wait_vm<0>();
    else if (full_count) sl_wait_vm<G::NPW>();
    else sl_wait_vm<G::NPW - 1>();
}

// fragments are read where they are used (the SIMD's other wave covers the LDS latency): A once, B per 32-walker n-tile
template <int WTN>
__device__ __forceinline__ void sl_mma_step(const char* lds, int buf, int wm, int wn, int lane, v16i (&acc)[SL_NLEV][WTN]) {
    typedef SlGeo<WTN> G;
    const char* base = lds + buf * G::STAGE;
    const int q = lane >> 5, r = lane & 31;
    v4i a[SL_D];
#pragma unroll
    for (int t = 0; t < SL_D; ++t) a[t] = *reinterpret_cast<const v4i*>(base + ((t * 2 + q) * SL_BM + wm * 32 + r) * 16);
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        v4i b[SL_D];
#pragma unroll
        for (int t = 0; t < SL_D; ++t)
            b[t] = *reinterpret_cast<const v4i*>(base + G::A_CHUNKS * 1024 + ((t * 2 + q) * G::BN + wn * 32 * WTN + 32 * j + r) * 16);
#pragma unroll
        for (int tb = 0; tb < SL_D; ++tb)
#pragma unroll
            for (int ta = 0; ta < SL_D; ++ta) {
                if (ta + tb < SL_LOW) continue;
                acc[ta + tb - SL_LOW][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], b[tb], acc[ta + tb - SL_LOW][j], 0, 0, 0);
            }
    }
}

// one GP of a launch: its context's planes and scales, where its partials go, its index inside the context
struct SlGP {
    const int8_t *A, *B;
    const double *rowscale, *colscale;
    double* spart;
    int P, p;
};
constexpr int SL_MAX_GP = GPB_MAX_MULTI_GP;
struct SlTable { SlGP gp[SL_MAX_GP]; };

// Static launch, one workgroup per tile slot.  Block b belongs to XCD label b % 8 (round-robin dispatch: speed only); the m-th
// block of a label works on super-block (m / 32) * 8 + label, tile m % 32 of it.  Super-blocks in order: row groups heaviest
// first, then GP, then walker group.  Slots of a ragged super-block (row blocks or walker tiles that do not exist) return at once.
// MULTI: the launch's GPs come from a table (the emulators of a chain: every GP with its own context's planes), else from ONE context.
template <int WTN, bool MULTI>
__device__ __forceinline__ void predict_sliced_body(const SlGP& G0, const SlTable* __restrict__ tab, int64_t Np, int64_t Np128, int64_t Wld,
                                                    int P, int nI, int nW, int kskip, const int* __restrict__ nrows) {
    typedef SlGeo<WTN> G;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (nrows) nW = (*nrows + G::BN - 1) / G::BN;                       // compacted batch: the live walker tiles only
    const int nG = (nI + SL_RG - 1) / SL_RG, nWG = (nW + SL_CG - 1) / SL_CG;
    const unsigned label = blockIdx.x & 7u, m = blockIdx.x >> 3;
    const unsigned sb = (m / (SL_RG * SL_CG)) * 8u + label, local = m % (SL_RG * SL_CG);
    if (sb >= (unsigned)(nG * P * nWG)) return;
    const int g = (int)(sb / (unsigned)(P * nWG)), rem = (int)(sb % (unsigned)(P * nWG));
    const int pg = rem / nWG, wg = rem % nWG;           // pg: the GP's index in the launch
    const SlGP gp = MULTI ? tab->gp[pg] : G0;
    const int p = MULTI ? gp.p : pg;                    // ... and in its context
    const int8_t* __restrict__ A = gp.A;
    const int8_t* __restrict__ B = gp.B;
    const double* __restrict__ rowscale = gp.rowscale;
    const double* __restrict__ colscale = gp.colscale;
    double* __restrict__ spart = gp.spart;
    const int Pc = MULTI ? gp.P : P;                    // GPs of the GP's own context (the stride of its partials)
    const int ib = (nG - 1 - g) * SL_RG + (SL_RG - 1 - (int)(local / SL_CG)), wt = wg * SL_CG + (int)(local % SL_CG);
    if (ib >= nI || wt >= nW) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                            // 4 x 2 waves of 32 rows x (32 WTN) walkers
    const int64_t mb = (int64_t)ib * SL_BM, nb = (int64_t)wt * G::BN;
    const bool fullc = G::REM == 0 || wave < G::REM;
    const int64_t a_plane = (Np / 16) * Np128 * 16, b_plane = (Np / 16) * Wld * 16;
    const int8_t* Ap = A + (int64_t)p * SL_D * a_plane;
    const int8_t* Bp = B + (int64_t)p * SL_D * b_plane;
    double* rs = reinterpret_cast<double*>(lds + G::NSTAGE * G::STAGE);
    if (tid < SL_BM) rs[tid] = rowscale[(int64_t)p * Np128 + mb + tid];
    // k in [k_begin, k_end): the row block's part of the triangle; the design's padding in front (all-zero rows of K*^T, whole
    // 32-deep steps of it) is left out when the row block lies behind it — the products it skips are exact zeros
    const int64_t k_end = imin64(mb + SL_BM, Np);
    const int64_t ks = (kskip / 32) * 32;
    const int64_t k_begin = ks <= mb ? ks : 0;
    const int nsteps = (int)((k_end - k_begin) / 32);
    const int64_t kb_first = k_begin / 16;
    v16i acc[SL_NLEV][WTN];
#pragma unroll
    for (int l = 0; l < SL_NLEV; ++l)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;
    sl_dma_stage<WTN>(lds, 0, Ap, Bp, a_plane, b_plane, Np128, Wld, mb, nb, kb_first, wave, lane);
    if (nsteps > 1) sl_dma_stage<WTN>(lds, 1, Ap, Bp, a_plane, b_plane, Np128, Wld, mb, nb, kb_first + 2, wave, lane);
    for (int s = 0; s < nsteps; ++s) {
        // stage s has landed for this wave (its own DMAs of stage s + 1 may still fly), every wave says so at the barrier, and
        // every wave has finished reading buffer (s + 2) % 3 = (s - 1) % 3 (its reads were consumed by the MFMAs of step s - 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sl_wait_stage<WTN>(s + 1 < nsteps, fullc);
        __builtin_amdgcn_s_barrier();
        sl_mma_step<WTN>(lds, s % 3, wm, wn, lane, acc);
        // the DMA of stage s + 2 BEHIND the step's MFMAs: a global_load_lds costs its wave ~60 cycles of issue, which then fall
        // into the time its 42 queued MFMAs drain (issued first: 0.95 -> 0.74 ms on the micro-kernel, profiles/r06_sliced_model.txt);
        // buffer (s + 2) % 3 was last read in step s - 1, which every wave has left (the barrier above)
        if (s + 2 < nsteps)
            sl_dma_stage<WTN>(lds, (s + 2) % 3, Ap, Bp, a_plane, b_plane, Np128, Wld, mb, nb, kb_first + 2 * (int64_t)(s + 2), wave, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // epilogue: levels -> fp64 (Horner from the least significant one), scale, square; a wave sums its 32 rows in the order
    // (register, lane half), the odd wave row hands its sum to the even one: one partial per 64-row block, order fixed by the row
    const double cs = colscale[p];
    double* hand = reinterpret_cast<double*>(lds);                     // [wave][WTN][32] (the stage buffers are free now)
    double sums[WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double t = (double)acc[0][j][r];
#pragma unroll
            for (int l = 1; l < SL_NLEV; ++l) t = fma(t, 1.0 / 256.0, (double)acc[l][j][r]);
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const double v = t * rs[row] * cs;
            sum = fma(v, v, sum);
        }
        sum += __shfl_xor(sum, 32);
        sums[j] = sum;
        if ((wm & 1) && lane < 32) hand[(wave * WTN + j) * 32 + lane] = sum;
    }
    __syncthreads();
    const int64_t blk = (int64_t)ib * 2 + (wm >> 1);
    if (!(wm & 1) && lane < 32 && blk * 64 < Np) {
#pragma unroll
        for (int j = 0; j < WTN; ++j)
            spart[(blk * Pc + p) * Wld + nb + wn * 32 * WTN + 32 * j + lane] = sums[j] + hand[((wave + 2) * WTN + j) * 32 + lane];
    }
}

template <int WTN>
__global__ __launch_bounds__(512, 2) void k_predict_sliced(const SlGP gp, int64_t Np, int64_t Np128, int64_t Wld, int P, int nI, int nW,
                                                           int kskip, const int* __restrict__ nrows) {
    predict_sliced_body<WTN, false>(gp, nullptr, Np, Np128, Wld, P, nI, nW, kskip, nrows);
}

// the GPs of ALL emulators of a chain in one launch (same Np, same batch): P = their number, the table says whose planes each reads
template <int WTN>
__global__ __launch_bounds__(512, 2) void k_predict_sliced_multi(const SlTable tab, int64_t Np, int64_t Np128, int64_t Wld, int P, int nI,
                                                                 int nW, int kskip, const int* __restrict__ nrows) {
    predict_sliced_body<WTN, true>(tab.gp[0], &tab, Np, Np128, Wld, P, nI, nW, kskip, nrows);
}

// ---------------------------------------------------------------------------------------------------------------- host side
// The rule (theta alone): every GP of the context has 1 + c / sn2 <= SL_RULE.  theta = [log c, log l_1..l_d, log sn2].
bool sliced_applies(const gpb_ctx* ctx) {
    if (!ctx->predict_sliced || ctx->multi || !ctx->h_theta || !ctx->have_theta) return false;
    if (ctx->tile_trace) return false;                                  // the tile trace is a hook of the fp64 kernel (debug library)
    if (ctx->predict_sliced == 2) return true;                          // test hook: the rule off (accuracy probes)
    const int64_t stride = ctx->d + 2;
    for (int64_t p = 0; p < ctx->P; ++p) {
        const double lc = ctx->h_theta[p * stride], ln = ctx->h_theta[p * stride + ctx->d + 1];
        if (!(1.0 + exp(lc - ln) <= SL_RULE)) return false;
    }
    return true;
}

void sliced_free(gpb_ctx* ctx) {
    if (ctx->slA) { pool_free(ctx->slA); ctx->slA = nullptr; }
    if (ctx->slB) { pool_free(ctx->slB); ctx->slB = nullptr; }
    if (ctx->sl_scale) { pool_free(ctx->sl_scale); ctx->sl_scale = nullptr; }
    ctx->slA_valid = false;
    ctx->slB_cap = 0;
}

const double* sliced_colscale(const gpb_ctx* ctx) {
    return reinterpret_cast<const double*>(ctx->sl_scale) + ctx->P * round_up(ctx->Np, 128);
}

// Buffers of the sliced path; the planes of L^-1 follow a new factorisation here, on first use.
int sliced_prepare(gpb_ctx* ctx) {
    const int64_t Np = ctx->Np, Np128 = round_up(Np, 128), P = ctx->P;
    const size_t a_bytes = (size_t)P * SL_D * (size_t)Np * (size_t)Np128;
    if (!ctx->slA) {
        GPB_HIP(pool_malloc_t(&ctx->slA, a_bytes));
        GPB_HIP(hipMemsetAsync(ctx->slA, 0, a_bytes, ctx->stream));       // the upper triangle and the rows behind Np: zero for good
        GPB_HIP(pool_malloc_t(&ctx->sl_scale, sizeof(double) * (size_t)(P * Np128 + P) + sizeof(int) * (size_t)(P * Np128)));
        ctx->slA_valid = false;
    }
    double* rowscale = reinterpret_cast<double*>(ctx->sl_scale);
    double* colscale = rowscale + P * Np128;
    int* rowexp = reinterpret_cast<int*>(colscale + P);
    if (!ctx->slA_valid) {
        hipLaunchKernelGGL(k_sl_rowscale, dim3((unsigned)((Np128 + 3) / 4), (unsigned)P), dim3(256), 0, ctx->stream, ctx->Linv,
                           ctx->amp, rowscale, rowexp, colscale, Np, Np128);
        hipLaunchKernelGGL(k_sl_slice_linv, dim3((unsigned)((Np + 255) / 256), (unsigned)(Np / 16), (unsigned)P), dim3(256), 0,
                           ctx->stream, ctx->Linv, rowexp, ctx->slA, Np, Np128);
        ctx->slA_valid = true;
    }
    if (ctx->slB_cap < ctx->Wcap || !ctx->slB) {
        if (ctx->slB) { GPB_HIP(hipStreamSynchronize(ctx->stream)); pool_free(ctx->slB); ctx->slB = nullptr; }
        GPB_HIP(pool_malloc_t(&ctx->slB, (size_t)P * SL_D * (size_t)Np * (size_t)ctx->Wcap));
        ctx->slB_cap = ctx->Wcap;
    }
    GPB_HIP(hipGetLastError());
    return 0;
}

// gpb_gp_get(GPB_GET_KSTAR) after a sliced batch: GP p's K*^T as the int8 kernel sees it — the 47-bit fixed-point values its digit
// planes hold (within 2^(e_c - 48) of the fp64 values) — rows [pad, pad + N) x the batch's W walkers into out[n * W + w].
int sliced_read_kstar(gpb_ctx* ctx, int64_t p, int64_t pad, int64_t N, int64_t W, double* out) {
    const int64_t Np = ctx->Np, Wld = ctx->Wld, plane = (Np / 16) * Wld * 16;
    std::vector<int8_t> h((size_t)(SL_D * plane));
    double cs = 0.0;
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    GPB_HIP(hipMemcpy(h.data(), ctx->slB + p * SL_D * plane, (size_t)(SL_D * plane), hipMemcpyDeviceToHost));
    GPB_HIP(hipMemcpy(&cs, sliced_colscale(ctx) + p, sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t n = 0; n < N; ++n)
        for (int64_t w = 0; w < W; ++w) {
            const int64_t k = pad + n;
            long long a = 0;
            for (int t = SL_D - 1; t >= 0; --t) a = a * 256 + (long long)h[(size_t)(t * plane + ((k / 16) * Wld + w) * 16 + k % 16)];
            out[n * W + w] = ldexp((double)a, -47) * cs;
        }
    return 0;
}

// V^2 partials of the CURRENT batch by the int8 kernel: k_kcross (SLICE form) has left the batch's digit planes in slB.
int launch_vsq_sliced(gpb_ctx* ctx, int64_t W, const int* nrows_dev, int kskip) {
    const int64_t Np = ctx->Np, Np128 = round_up(Np, 128), P = ctx->P, Wld = ctx->Wld;
    if (!ctx->slA || !ctx->slA_valid || !ctx->slB) GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq_sliced before sliced_prepare");
    const double* rowscale = reinterpret_cast<const double*>(ctx->sl_scale);
    const double* colscale = rowscale + P * Np128;
    // 128 x 128 tiles when enough of them exist to fill the chip (three per CU of the rows that are live, as far as the host
    // knows), 128 x 64 below that (cfg 4, one MI355X, launch us at 2048 / 1024 / 512 / 256 rows: 653 / 344 / 220 / 217 against
    // 757 / 375 / 206 / 136: gpurun_out r6_share8_e); either shape gives the same bits
    const int nI = (int)(Np128 / SL_BM);
    int64_t Wsel = Wld;
    if (nrows_dev && ctx->tile_by_live && ctx->hint_from && ctx->hint_from->live_hint) {
        const unsigned long long h = __atomic_load_n(ctx->hint_from->live_hint, __ATOMIC_RELAXED);
        const int64_t cnt = (int64_t)(h & 0xffffffffull), of = (int64_t)(h >> 32);
        if (of > 0 && cnt <= of) Wsel = imin64(Wld, (int64_t)((double)cnt / (double)of * (double)W * 1.03) + 8);
    }
    const int64_t tiles128 = P * nI * ((Wsel + 127) / 128);
    int wtn = tiles128 >= 3 * (int64_t)ctx->num_cu ? 2 : 1;
    if (ctx->force_tile == 128) wtn = 2;
    if (ctx->force_tile == 64 || ctx->force_tile == 32 || ctx->force_tile == 65) wtn = 1;
    const int nW = (int)(Wld / (64 * wtn));
    const int nG = (nI + SL_RG - 1) / SL_RG, nWG = (nW + SL_CG - 1) / SL_CG;
    const int64_t nSB = (int64_t)nG * P * nWG;
    const unsigned grid = (unsigned)(((nSB + 7) / 8) * 8 * SL_RG * SL_CG);
    static bool attr_set = false;
    if (!attr_set) {
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<2>::LDS));
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<1>::LDS));
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced_multi<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<2>::LDS));
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced_multi<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<1>::LDS));
        attr_set = true;
    }
    const SlGP gp{ctx->slA, ctx->slB, rowscale, colscale, ctx->spart, (int)P, 0};
    if (wtn == 2)
        hipLaunchKernelGGL(k_predict_sliced<2>, dim3(grid), dim3(512), SlGeo<2>::LDS, ctx->stream, gp, Np, Np128, Wld, (int)P, nI, nW,
                           kskip, nrows_dev);
    else
        hipLaunchKernelGGL(k_predict_sliced<1>, dim3(grid), dim3(512), SlGeo<1>::LDS, ctx->stream, gp, Np, Np128, Wld, (int)P, nI, nW,
                           kskip, nrows_dev);
    (void)W;
    return 0;
}

// The same for the GPs of E contexts (a chain's emulators of equal padded size, every one admitted by the rule and its batch left
// as digit planes by the shared SLICE cross launch): one table, one launch.
int launch_vsq_sliced_multi(gpb_ctx* const* ctxs, int E, int64_t W, const int* nrows_dev, int kskip) {
    gpb_ctx* ctx = ctxs[0];
    const int64_t Np = ctx->Np, Np128 = round_up(Np, 128), Wld = ctx->Wld;
    SlTable tab;
    int G = 0;
    for (int e = 0; e < E; ++e) {
        gpb_ctx* c = ctxs[e];
        if (!c->slA || !c->slA_valid || !c->slB || c->Np != Np || c->Wld != Wld)
            GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq_sliced_multi over a context that is not prepared");
        const double* rowscale = reinterpret_cast<const double*>(c->sl_scale);
        const double* colscale = rowscale + c->P * Np128;
        for (int p = 0; p < (int)c->P; ++p, ++G) {
            if (G >= SL_MAX_GP) GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_vsq_sliced_multi: too many GPs for one table");
            const int64_t a_plane = (Np / 16) * Np128 * 16, b_plane = (Np / 16) * Wld * 16;
            (void)a_plane; (void)b_plane;
            tab.gp[G] = SlGP{c->slA, c->slB, rowscale, colscale, c->spart, (int)c->P, p};
        }
    }
    const int nI = (int)(Np128 / SL_BM);
    int64_t Wsel = Wld;
    if (nrows_dev && ctx->tile_by_live && ctx->hint_from && ctx->hint_from->live_hint) {
        const unsigned long long h = __atomic_load_n(ctx->hint_from->live_hint, __ATOMIC_RELAXED);
        const int64_t cnt = (int64_t)(h & 0xffffffffull), of = (int64_t)(h >> 32);
        if (of > 0 && cnt <= of) Wsel = imin64(Wld, (int64_t)((double)cnt / (double)of * (double)W * 1.03) + 8);
    }
    const int64_t tiles128 = (int64_t)G * nI * ((Wsel + 127) / 128);
    int wtn = tiles128 >= 3 * (int64_t)ctx->num_cu ? 2 : 1;
    if (ctx->force_tile == 128) wtn = 2;
    if (ctx->force_tile == 64 || ctx->force_tile == 32 || ctx->force_tile == 65) wtn = 1;
    const int nW = (int)(Wld / (64 * wtn));
    const int nG = (nI + SL_RG - 1) / SL_RG, nWG = (nW + SL_CG - 1) / SL_CG;
    const int64_t nSB = (int64_t)nG * G * nWG;
    const unsigned grid = (unsigned)(((nSB + 7) / 8) * 8 * SL_RG * SL_CG);
    static bool attr_set = false;
    if (!attr_set) {
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced_multi<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<2>::LDS));
        GPB_HIP(hipFuncSetAttribute((const void*)k_predict_sliced_multi<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SlGeo<1>::LDS));
        attr_set = true;
    }
    if (wtn == 2)
        hipLaunchKernelGGL(k_predict_sliced_multi<2>, dim3(grid), dim3(512), SlGeo<2>::LDS, ctx->stream, tab, Np, Np128, Wld, G, nI, nW,
                           kskip, nrows_dev);
    else
        hipLaunchKernelGGL(k_predict_sliced_multi<1>, dim3(grid), dim3(512), SlGeo<1>::LDS, ctx->stream, tab, Np, Np128, Wld, G, nI, nW,
                           kskip, nrows_dev);
    return 0;
}

}  // namespace gpb
