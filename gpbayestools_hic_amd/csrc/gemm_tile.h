// gemm_tile.h — fp64 MFMA tile engine for gfx950 (CDNA4), shared by the dense kernels:
//   predict (V = L^-1 K*^T, fused sum of squares), Cholesky trailing update (SYRK),
//   triangular inverse (recursive block doubling) and K^-1 = L^-T L^-1 for the LML gradient.
//
// Block tile T x TN (128x128, 64x64, 64x32), K-step KB (16), 256 threads =
// 4 waves (2x2), wave tile T/2 x TN/2 of v_mfma_f64_16x16x4_f64 tiles (T=128: 16 accumulators of 4 f64 = 128
// VGPRs).  Operand tiles are staged in LDS k-major ([k][m], [k][n]) with a 16-double row pad so that the
// per-lane fragment reads (lane l: row k=l>>4, 16 consecutive m) are bank-conflict free for
// ds_read_b64; global loads are register-prefetched one K-step ahead.
// Measured on MI355X (profiles/r01_mfma_f64_issue_rate.txt): v_mfma_f64_16x16x4_f64 occupies the matrix pipe
// for 64 cycles, but ONE wave can issue one only every ~138 cycles, i.e. 2048 flop per 1 KiB of fragments, so
// LDS and L2 traffic are far from their limits; the design goal is only to keep the matrix pipe issuing back to
// back.  T = 64 exists for small walker batches (multi-GPU shards), where 128-wide tiles leave CUs idle behind
// the heaviest triangular row block; its waves have only 16 MFMAs between the two barriers of a 16-deep K-step
// (~1300 cycles of synchronisation per ~2200 of issue, profiles/r01_tile_trace.txt).  KB is a template parameter;
// 32-deep steps for the 64-row tiles were measured within 2.5 % of 16-deep either way and are not instantiated.
// Triangular operands (L^-1 in predict): gemm_tile_loop<..., TRI> leaves out the 16-row m-tiles of the diagonal
// block that are all zeros, the same share for every wave (see there) — with two waves per SIMD an MFMA saved in
// only one of them is an issue slot the other cannot use.
//
// v_mfma_f64_16x16x4_f64 lane maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i=l&15][k=l>>4];  B: lane l holds B[k=l>>4][j=l&15];
//   C/D: 4 f64 per lane, reg r -> row (l>>4)+4r, col l&15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpb {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;                   // default K-step
constexpr int GEMM_THREADS = 256;

// T rows of A (the m extent) by TN columns of B (the n extent); TN = T except for the narrow predict tile.
template <int T, int TN = T, int KB = BK>
struct __attribute__((aligned(16))) TileLds {
    static constexpr int LD = T + 16;   // row stride = 2*(T+16) dwords = 32 (mod 64) banks
    double As[KB][T + 16];
    double Bs[KB][TN + 16];
};                                       // KB=16: T=128: 36,864 B; T=64: 20,480 B; 64x32: 16,384 B

// Row skew of the staged tiles: element (k, x) lives at [k][x + lds_skew(k)], inside the 16-double row pad.  The row
// stride is 2 (T + 16) dwords = 0 (mod 32 banks), so the transposing stores of lstore_trans — a 16-lane group of
// ds_write_b64 (32 banks) holds four x's of FOUR k rows, 4 apart — hit the same banks four times (16 LDS-array cycles per
// instruction against the 6 of its register transfer: with 64-row tiles the LDS array, not the matrix pipe, set the
// pace).  Shifting rows k, k+4, k+8, k+12 by 0, 4, 8, 12 doubles puts the four rows on disjoint banks; for the fragment
// reads of a k-group (rows kk .. kk+3: one skew) it is a compile-time constant folded into the instruction's offset —
// no extra address registers (an XOR swizzle cost 6-14 VGPRs and pushed three predict kernels into scratch).
// Layout only: no result changes.
__device__ __forceinline__ constexpr int lds_skew(int k) { return ((k >> 2) & 3) << 2; }

// NW = waves per workgroup (4: 2x2 waves, wave tile T/2 x T/2;  8: 2x4 waves, wave tile T/2 x T/4).
template <int T, int NW = 4, int KB = BK>
struct Frag { d2 r[(KB * T / 2) / (64 * NW)]; };       // native vectors: HIP's double2 struct is copied with memcpy,
                                                        // which kept some instantiations' prefetch registers in scratch
template <int T, int NW = 4, int TN = T>
struct Acc { d4 v[T / 32][(2 * TN / NW) / 16]; };

// logical tile[k][x] = G[(k0+k)*ld + x0+x]   (k<KB, x<T); rows are contiguous in x.
// FULL = the caller guarantees x_ext == T (no edge): the loads are unconditional.
template <int T, int NW, int KB, bool FULL = false>
__device__ __forceinline__ void gload_direct(const double* __restrict__ G, int64_t ld, int64_t k0, int64_t x0,
                                             int x_ext, Frag<T, NW, KB>& f, int tid) {
    constexpr int TPR = T / 2;                 // threads per row (2 doubles each)
    constexpr int RPP = (64 * NW) / TPR;       // rows per pass
    const int row = tid / TPR, col = (tid % TPR) * 2;
    const bool ok = FULL || col < x_ext;
#pragma unroll
    for (int j = 0; j < (KB * T / 2) / (64 * NW); ++j) {
        if (ok) f.r[j] = *reinterpret_cast<const d2*>(G + (k0 + row + RPP * j) * ld + x0 + col);
        else    f.r[j] = d2{0.0, 0.0};
    }
}
template <int T, int NW, int KB>
__device__ __forceinline__ void lstore_direct(double (*S)[T + 16], const Frag<T, NW, KB>& f, int tid) {
    constexpr int TPR = T / 2, RPP = (64 * NW) / TPR;
    const int row = tid / TPR, col = (tid % TPR) * 2;
    // lds_skew(row + RPP j) = lds_skew(row) + lds_skew(RPP j): RPP is a multiple of 4 (no carry into the skew bits) and the
    // sum stays below 16 for every shape in use — a per-thread base plus a compile-time constant
    static_assert(RPP % 4 == 0 && RPP * ((KB * T / 2) / (64 * NW)) == KB, "lstore_direct: rows per pass");
    const int colk = col + lds_skew(row);
#pragma unroll
    for (int j = 0; j < (KB * T / 2) / (64 * NW); ++j)
        *reinterpret_cast<d2*>(&S[row + RPP * j][colk + lds_skew(RPP * j)]) = f.r[j];
}
// logical tile[k][x] = G[(x0+x)*ld + k0+k]   (rows of G are contiguous in k): transpose on store.
template <int T, int NW, int KB, bool FULL = false>
__device__ __forceinline__ void gload_trans(const double* __restrict__ G, int64_t ld, int64_t k0, int64_t x0,
                                            int x_ext, Frag<T, NW, KB>& f, int tid) {
    constexpr int TPX = (64 * NW) / T;         // threads per x row
    constexpr int KPT = KB / TPX;              // k's per thread
    const int x = tid / TPX, kh = (tid % TPX) * KPT;
    const bool ok = FULL || x < x_ext;
    const double* p = G + (x0 + x) * ld + k0 + kh;
#pragma unroll
    for (int j = 0; j < KPT / 2; ++j) {
        if (ok) f.r[j] = *reinterpret_cast<const d2*>(p + 2 * j);
        else    f.r[j] = d2{0.0, 0.0};
    }
}
template <int T, int NW, int KB>
__device__ __forceinline__ void lstore_trans(double (*S)[T + 16], const Frag<T, NW, KB>& f, int tid) {
    constexpr int TPX = (64 * NW) / T, KPT = KB / TPX;
    static_assert(KPT == 2 || KPT % 4 == 0, "lstore_trans: k's per thread");
    const int x = tid / TPX, kh = (tid % TPX) * KPT;
    const int xk = x + lds_skew(kh);
#pragma unroll
    for (int j = 0; j < KPT / 2; ++j) {
        // lds_skew(kh + 2 j) = lds_skew(kh) + lds_skew(2 j) (kh is a multiple of KPT; k and k + 1 share the skew)
        S[kh + 2 * j][xk + lds_skew(2 * j)] = f.r[j].x;
        S[kh + 2 * j + 1][xk + lds_skew(2 * j)] = f.r[j].y;
    }
}

template <int T, int NW = 4, int TN = T>
__device__ __forceinline__ void acc_zero(Acc<T, NW, TN>& acc) {
#pragma unroll
    for (int i = 0; i < T / 32; ++i)
#pragma unroll
        for (int j = 0; j < (2 * TN / NW) / 16; ++j) acc.v[i][j] = d4{0.0, 0.0, 0.0, 0.0};
}

// One K-step of MFMAs from the staged tiles.  The wave's m-tile i (16 rows) starts at row m0 + MS*i of the block
// tile: MS = 16, m0 = wave row * T/2 for the usual contiguous halves; MS = 32, m0 = wave row * 16 when the two wave
// rows own the 16-row m-tiles alternately (TRI below).  IMIN > 0 leaves out m-tiles 0..IMIN-1.
// PIPE (the 128x128 predict tile; tune key 37 = 0 for the plain form): the fragments of the next k-group are read from LDS before the MFMAs of
// the current one, so that their latency runs under 16 MFMAs instead of in front of them.
template <int T, int NW, int TN, int KB, int MS = 16, int IMIN = 0, bool PIPE = false>
__device__ __forceinline__ void tile_mma(const TileLds<T, TN, KB>& L, Acc<T, NW, TN>& acc, int lane, int m0,
                                         int n0) {
    constexpr int NI = T / 32, NJ = (2 * TN / NW) / 16;
    const int lr = lane & 15, lk = lane >> 4;
    if constexpr (PIPE) {
        double a[2][NI], b[2][NJ];
#pragma unroll
        for (int i = IMIN; i < NI; ++i) a[0][i] = L.As[lk][m0 + MS * i + lr + lds_skew(0)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[0][j] = L.Bs[lk][n0 + 16 * j + lr + lds_skew(0)];
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < KB) {
#pragma unroll
                for (int i = IMIN; i < NI; ++i) a[nxt][i] = L.As[kk + 4 + lk][m0 + MS * i + lr + lds_skew(kk + 4)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[nxt][j] = L.Bs[kk + 4 + lk][n0 + 16 * j + lr + lds_skew(kk + 4)];
            }
#pragma unroll
            for (int i = IMIN; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], b[cur][j], acc.v[i][j], 0, 0, 0);
            // keep the order written here: the next group's LDS reads, then this group's MFMAs.  (Spreading the reads
            // between the MFMAs, one per two, cost 28 bytes of scratch and 0.7 %; 32-deep K-steps for this tile 88 bytes
            // and 4-5 %: profiles/r02_k_predict_inner_loop.txt)
            __builtin_amdgcn_sched_group_barrier(0x100, NI - IMIN + NJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (NI - IMIN) * NJ, 0);
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < KB; kk += 4) {
        double a[NI], b[NJ];
#pragma unroll
        for (int i = IMIN; i < NI; ++i) a[i] = L.As[kk + lk][m0 + MS * i + lr + lds_skew(kk)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = L.Bs[kk + lk][n0 + 16 * j + lr + lds_skew(kk)];
#pragma unroll
        for (int i = IMIN; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc.v[i][j], 0, 0, 0);
    }
}

// acc += A[m_base.., k] * B[k, n_base..] for k in [k_begin, k_end), both multiples of KB.  The k order of the
// accumulation (ascending, four at a time inside an MFMA) does not depend on KB: results are identical.
//   A_TRANS=false: A[m][k] = Ag[(m_base+m)*lda + k]       A_TRANS=true: A[m][k] = Ag[k*lda + m_base+m]
//   B_TRANS=false: B[k][n] = Bg[k*ldb + n_base+n]         B_TRANS=true: B[k][n] = Bg[(n_base+n)*ldb + k]
// m_ext / n_ext (even, <= T) bound the valid rows / columns of this tile; the rest reads as 0.
//
// TRI: A is lower triangular and k_end - T (= tri_begin) .. k_end is the tile's diagonal block, where the 16-row
// m-tile g is all zeros from K-step g + 1 of the block on.  To let EVERY wave drop the same share of that work
// the two wave rows own the m-tiles alternately (wave row r: m-tiles r, r+2, ...; acc.v[i] = m-tile 2i + r): in the
// q-th pair of diagonal K-steps all of a wave's m-tiles i < q are zero, whichever its row, so the pattern is a
// compile-time constant per pair and the main loop stays free of control flow.  37.5 % of the diagonal block's
// MFMAs go (T = 128), the skipped products are exact zeros, the sums are unchanged.  Callers must read acc with
// the same interleaved map.
template <int T, bool A_TRANS, bool B_TRANS, int NW = 4, int TN = T, int KB = BK, bool FULL = false, bool TRI = false,
          bool PIPE = false>
__device__ __forceinline__ void gemm_tile_loop(const double* __restrict__ Ag, int64_t lda,
                                               const double* __restrict__ Bg, int64_t ldb, int64_t m_base,
                                               int64_t n_base, int m_ext, int n_ext, int64_t k_begin, int64_t k_end,
                                               TileLds<T, TN, KB>& L, Acc<T, NW, TN>& acc, int64_t tri_begin = 0) {
    static_assert(!TRI || KB == 16 || KB == 32, "TRI covers the diagonal block 32 k at a time");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int WN = NW / 2, TNW = TN / WN;   // waves along n, wave tile width
    constexpr int MS = TRI ? 32 : 16;
    const int m0 = (wave / WN) * (TRI ? 16 : T / 2), n0 = (wave % WN) * TNW;
    Frag<T, NW, KB> fa;
    Frag<TN, NW, KB> fb;
    if (k_begin < k_end) {
        if (A_TRANS) gload_direct<T, NW, KB, FULL>(Ag, lda, k_begin, m_base, m_ext, fa, tid);
        else         gload_trans<T, NW, KB, FULL>(Ag, lda, k_begin, m_base, m_ext, fa, tid);
        if (B_TRANS) gload_trans<TN, NW, KB, FULL>(Bg, ldb, k_begin, n_base, n_ext, fb, tid);
        else         gload_direct<TN, NW, KB, FULL>(Bg, ldb, k_begin, n_base, n_ext, fb, tid);
    }
    // one K-step of the pipeline: publish the prefetched operands in LDS, start the next prefetch
    auto stage = [&](int64_t k0) {
        __syncthreads();
        if (A_TRANS) lstore_direct<T, NW, KB>(L.As, fa, tid); else lstore_trans<T, NW, KB>(L.As, fa, tid);
        if (B_TRANS) lstore_trans<TN, NW, KB>(L.Bs, fb, tid); else lstore_direct<TN, NW, KB>(L.Bs, fb, tid);
        __syncthreads();
        const int64_t kn = k0 + KB;
        if (kn < k_end) {
            if (A_TRANS) gload_direct<T, NW, KB, FULL>(Ag, lda, kn, m_base, m_ext, fa, tid);
            else         gload_trans<T, NW, KB, FULL>(Ag, lda, kn, m_base, m_ext, fa, tid);
            if (B_TRANS) gload_trans<TN, NW, KB, FULL>(Bg, ldb, kn, n_base, n_ext, fb, tid);
            else         gload_direct<TN, NW, KB, FULL>(Bg, ldb, kn, n_base, n_ext, fb, tid);
        }
    };
    const int64_t k_main = TRI ? (tri_begin < k_end ? tri_begin : k_end) : k_end;
    int64_t k0 = k_begin;
    for (; k0 < k_main; k0 += KB) {
        stage(k0);
        tile_mma<T, NW, TN, KB, MS, 0, PIPE>(L, acc, lane, m0, n0);
    }
    if constexpr (TRI) {
        // the diagonal block, two K-steps at a time (its extent is a multiple of 32: Np is one of 64)
#define GPB_TRI_PAIR(Q)                                                                  \
        if (Q < T / 32 && k0 < k_end) {                                                  \
            stage(k0);                                                                   \
            tile_mma<T, NW, TN, KB, MS, (Q < T / 32 ? Q : 0), PIPE>(L, acc, lane, m0, n0);     \
            if (KB == 16) {                                                              \
                stage(k0 + KB);                                                          \
                tile_mma<T, NW, TN, KB, MS, (Q < T / 32 ? Q : 0), PIPE>(L, acc, lane, m0, n0); \
            }                                                                            \
            k0 += 32;                                                                    \
        }
        GPB_TRI_PAIR(0) GPB_TRI_PAIR(1) GPB_TRI_PAIR(2) GPB_TRI_PAIR(3)
#undef GPB_TRI_PAIR
    }
}

// (A two-stage LDS variant with one barrier per K-step was measured twice: equal at small batches, 5-6 % slower
// at 4096+ walkers — the second stage halves nothing that limits this kernel.)
// C[(m_base+row)*ldc + n_base+col] = (accumulate ? C : 0) + scale*acc   for row<m_ext, col<n_ext
template <int T>
__device__ __forceinline__ void tile_store(double* __restrict__ C, int64_t ldc, int64_t m_base, int64_t n_base,
                                           int m_ext, int n_ext, double scale, bool accumulate, const Acc<T>& acc) {
    constexpr int NI = T / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (wave >> 1) * (T / 2), n0 = (wave & 1) * (T / 2);
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * i + lk + 4 * r;
                if (row < m_ext && col < n_ext) {
                    double* p = C + (m_base + row) * ldc + n_base + col;
                    const double v = scale * acc.v[i][j][r];
                    *p = accumulate ? (*p + v) : v;
                }
            }
        }
}

}  // namespace gpb
