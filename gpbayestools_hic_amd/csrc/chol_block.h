// chol_block.h — the body of the Cholesky chain: factor and invert one 64x64 SPD block held in LDS
// (potf2_inv_64), and the one-shot 64x64x64 MFMA product the step kernels are made of.  Shared by gpb_chol.hip and
// the timing probe tools/micro/potf2_probe.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "gemm_tile.h"

namespace gpb {

constexpr int LDP = 65;            // row stride (doubles) of a 64x64 block in LDS.  The MFMA fragment reads of mma_nt_64 — lane (lr, lk)
                                   // reads [row lr][k + lk] — are merged by the compiler into ds_read2_b64 (two k-steps per instruction),
                                   // which the LDS serves in 16-lane groups with banks (a / 4) mod 32: sixteen rows at 130 dwords
                                   // = 2 (mod 32) apart are conflict-free; at 66 (132 = 4 mod 32) rows r and r + 8 met on a bank:
                                   // SQ_LDS_BANK_CONFLICT was 43 % of the step kernels' LDS cycles (profiles/r04_fit_pmc.json).
                                   // Rows are 8-byte aligned only: tiles are stored with ds_write_b64 (load_tile)
#ifndef POTF2_AHEAD
#define POTF2_AHEAD 4              // columns whose scalar multipliers are in flight ahead of their fma (potf2_inv_64)
#endif
constexpr int CHOL_THREADS = 512;  // 8 waves: two per SIMD, what it takes to keep the f64 MFMA pipe issuing back to back

struct CholLds {
    double a[64][LDP];             // product operand A, then the block being factored -> L
    double x[64][LDP];             // product operand B, then L^-1
    double tm[32][34];             // scratch of the inverse assembly (one 32x32 block)
    double rdg[64];                // reciprocals of the pivots
    int bad;                       // first non-positive pivot of this block (-1: none)
};                                 // 75,532 bytes: two workgroups per CU

// Broadcast lane K of every ROW of 16 lanes to the 16 lanes of that row: one v_mov_b64_dpp row_newbcast:K (the only DPP
// control the 64-bit ALU takes on gfx90a+).  No SGPR round trip as with v_readlane (two per double, plus the
// spills of ~30 live scalars per pivot that the register-resident factorisation of round 1 paid for), and the four
// rows of a wave work on four different 16x16 problems at once.
template <int K>
__device__ __forceinline__ double bc16(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0xf, true);
}
// bc16<K> for a K known after unrolling only
__device__ __forceinline__ double bc16v(double v, int k) {
    switch (k) {
        case 0: return bc16<0>(v);   case 1: return bc16<1>(v);   case 2: return bc16<2>(v);   case 3: return bc16<3>(v);
        case 4: return bc16<4>(v);   case 5: return bc16<5>(v);   case 6: return bc16<6>(v);   case 7: return bc16<7>(v);
        case 8: return bc16<8>(v);   case 9: return bc16<9>(v);   case 10: return bc16<10>(v); case 11: return bc16<11>(v);
        case 12: return bc16<12>(v); case 13: return bc16<13>(v); case 14: return bc16<14>(v); default: return bc16<15>(v);
    }
}
// acc += (lane K of the row's `src`) * mul in ONE instruction: v_fmac_f64 with the DPP broadcast on its first operand.
// hipcc does not fold a v_mov_b64_dpp into the fma that uses it (here every broadcast feeds two), hence inline asm.
// The caller owes the two wait states between the VALU write of `src` and a DPP read of it (hazard_dpp_src below):
// the hazard recogniser does not look inside asm statements.
template <int K>
__device__ __forceinline__ void fmac_bc16(double& acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
__device__ __forceinline__ void fmac_bc16v(double& acc, double src, double mul, int k) {
    switch (k) {
        case 0: fmac_bc16<0>(acc, src, mul); break;   case 1: fmac_bc16<1>(acc, src, mul); break;
        case 2: fmac_bc16<2>(acc, src, mul); break;   case 3: fmac_bc16<3>(acc, src, mul); break;
        case 4: fmac_bc16<4>(acc, src, mul); break;   case 5: fmac_bc16<5>(acc, src, mul); break;
        case 6: fmac_bc16<6>(acc, src, mul); break;   case 7: fmac_bc16<7>(acc, src, mul); break;
        case 8: fmac_bc16<8>(acc, src, mul); break;   case 9: fmac_bc16<9>(acc, src, mul); break;
        case 10: fmac_bc16<10>(acc, src, mul); break; case 11: fmac_bc16<11>(acc, src, mul); break;
        case 12: fmac_bc16<12>(acc, src, mul); break; case 13: fmac_bc16<13>(acc, src, mul); break;
        case 14: fmac_bc16<14>(acc, src, mul); break; default: fmac_bc16<15>(acc, src, mul); break;
    }
}
__device__ __forceinline__ void hazard_dpp_src(double& src) { asm volatile("s_nop 1" : "+v"(src)); }
// lane K's value in a scalar register pair (uniform): two v_readlane_b32, volatile so that the pass keeps them where they are
// written (hoisted by the scheduler, the scalars of a whole pivot lived at once and spilled: round 1's version)
template <int K>
__device__ __forceinline__ double readlane_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int slo, shi;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(slo) : "v"(lo), "n"(K));
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(shi) : "v"(hi), "n"(K));
    return __hiloint2double(shi, slo);
}
// acc += s * mul with the uniform s in scalar registers (plain v_fmac_f64: full rate)
__device__ __forceinline__ void fmac_s(double& acc, double s, double mul) {
    asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "s"(s), "v"(mul));
}
// pins the place of a value's definition among the volatile asm statements around it (no instruction)
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// 1/sqrt(a) to full precision: v_rsq_f64 (about 2^-26) and one third-order correction y (1 + e/2 + 3 e^2/8),
// e = 1 - a y^2.  No special cases: a <= 0 gives NaN / inf, which is what a failed pivot has to propagate.
__device__ __forceinline__ double rsqrt_nr(double a) {
    const double y = __builtin_amdgcn_rsq(a);
    const double e = fma(-(a * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// row-major global tile G[r*ld + c] (64x64) -> S[r][c]; 512 threads, 16-byte loads (ld and the tile origin are even), 8-byte
// LDS stores (odd rows of S start on an odd double)
__device__ __forceinline__ void load_tile(const double* __restrict__ G, int64_t ld, double (*S)[LDP], int tid) {
    d2 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + CHOL_THREADS * e;          // 2048 pairs
        const int r = idx >> 5, c = (idx & 31) * 2;
        v[e] = *reinterpret_cast<const d2*>(G + (int64_t)r * ld + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + CHOL_THREADS * e;
        const int r = idx >> 5, c = (idx & 31) * 2;
        S[r][c] = v[e].x;
        S[r][c + 1] = v[e].y;
    }
}

// the two halves of load_tile, for a tile that is fetched while another one is being multiplied
struct TileRegs { d2 v[4]; };
__device__ __forceinline__ void gload_tile(const double* __restrict__ G, int64_t ld, TileRegs& t, int tid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + CHOL_THREADS * e;
        const int r = idx >> 5, c = (idx & 31) * 2;
        t.v[e] = *reinterpret_cast<const d2*>(G + (int64_t)r * ld + c);
    }
}
__device__ __forceinline__ void lstore_tile(double (*S)[LDP], const TileRegs& t, int tid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + CHOL_THREADS * e;
        const int r = idx >> 5, c = (idx & 31) * 2;
        S[r][c] = t.v[e].x;
        S[r][c + 1] = t.v[e].y;
    }
}

// out[i][j] = sum_k A[i][k] * B[j][k] over a 64x64x64 block, operands in LDS.  8 waves as 2 x 4: wave (wm, wn) owns
// rows 32 wm .. +31 (two 16-row m-tiles) and columns 16 wn .. +15; acc[t] = m-tile t.
__device__ __forceinline__ void mma_nt_64(const double (*A)[LDP], const double (*B)[LDP], d4 acc[2], int wave, int lane) {
    const int m0 = (wave >> 2) * 32, n0 = (wave & 3) * 16;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 64; kk += 4) {
        const double b = B[n0 + lr][kk + lk];
        const double a0 = A[m0 + lr][kk + lk], a1 = A[m0 + 16 + lr][kk + lk];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc[1], 0, 0, 0);
    }
}

// Cholesky factor and inverse of the 64x64 SPD block in s.a (lower triangle read).  On return s.a = L and s.x = L^-1,
// both with zeros above the diagonal; s.bad = index of the first non-positive pivot or -1 (the factor is NaN from
// there on, as LAPACK's would be garbage).  All 512 threads must call it; begins and ends with a barrier.
// STAMP (probe builds only): stamps[n] = s_memtime at the phase boundaries, written by thread 0.
template <bool STAMP = false>
__device__ __forceinline__ void potf2_inv_64(CholLds& s, unsigned long long* stamps = nullptr) {
    int nstamp = 0;
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (threadIdx.x == 0) stamps[nstamp] = __builtin_amdgcn_s_memtime();
            ++nstamp;
        }
    };
    stamp();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 64 * LDP; e += CHOL_THREADS) (&s.x[0][0])[e] = 0.0;
    if (tid == 0) s.bad = -1;
    __syncthreads();
    for (int bb = 0; bb < 4; ++bb) {
        const int o = 16 * bb;
        const int nrem = 48 - o;                       // rows below this sub-block inside the 64-block
        if (wave == 0) {
            // ---- 16x16 diagonal sub-block AND the panel below it in one pass of one wave: lane l holds row o + l of the block
            //      column (lanes 0-15 the diagonal sub-block, lanes 16 .. 15 + nrem the panel), ONE register row per lane.
            //      Per pivot: a_jj from lane j, 1/sqrt (uniform), the column scaled in all rows at once; for every later
            //      column k the multiplier L_kj = lane k's l goes through a scalar register pair (two v_readlane_b32) into ONE
            //      v_fmac_f64 for the factor's and the panel's rows together.
            //      (Rounds 2-5 broadcast by DPP inside the 16-lane rows — v_fmac_f64_dpp row_newbcast, every row of 16 lanes
            //      factoring its own copy of the diagonal block, a second fma for the panel rows, the whole chain of a pivot
            //      behind the previous pivot's updates: 4.86 k cycles per pass; this form 4.34 k.  Neither is bound by the
            //      updates' issue alone (a lone wave issues v_fmac_f64 every 4.6 cycles, with a DPP operand every 4.8,
            //      v_readlane_b32 and v_mov_b64_dpp every 8: tools/micro/issue_rate_probe.hip): a pivot's skeleton — pivot to a
            //      scalar, v_rsq_f64 + correction, scaling, the first update — is ~230 cycles of mostly dependent
            //      instructions, 16 times per pass: profiles/r06_fit_notes.txt.  The same products on the same operands in
            //      the same order per accumulator as before: the same bits.)
            const bool has_row = lane < 16 + nrem;
            double a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = s.a[o + (has_row ? lane : 0)][o + k];      // (lanes without a row: a copy of row o, never stored)
            int badj = -1;
            double myrinv = 0.0;
            // The pass is a software pipeline: one wave issues in order, and pivot j + 1's chain — a_(j+1)(j+1) to a scalar,
            // v_rsq_f64, the five dependent operations of its correction, the column's scaling — needs nothing but column
            // j + 1, which is final after the FIRST update of pivot j: its seven stages are dealt out between the remaining
            // updates (volatile asm statements keep their order; pin() places the compiler's own instructions among them).
            double l, nl;
            {
                const double ajj = readlane_f64<0>(a[0]);
                if (!(ajj > 0.0)) badj = 0;
                const double rinv = rsqrt_nr(ajj);                 // 1 / L_00
                l = a[0] * rinv;                                   // column 0 of L, factor and panel rows
                myrinv = (lane == 0) ? rinv : myrinv;              // lane j keeps 1 / L_jj (no branch inside the chain)
                a[0] = l;
                nl = -l;
                pin(l); pin(nl);
            }
            static_for<0, 15>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // a[.][k] -= l_.j * L_kj for k = j + 1 .. 15; the scalar of column k + AH is fetched before column k's fma
                // (a scalar written by v_readlane_b32 reaches a vector instruction only after the scalar write-back)
                constexpr int AH = POTF2_AHEAD;
                double sm[16 + AH];                               // the multipliers L_kj, uniform: scalar register pairs
                static_for<0, AH>([&](auto ac) {
                    constexpr int k = j + 1 + decltype(ac)::value;
                    if constexpr (k < 16) sm[k] = readlane_f64<(k < 16 ? k : 15)>(l);
                });
                if constexpr (j + 1 + AH < 16) sm[j + 1 + AH] = readlane_f64<(j + 1 + AH < 16 ? j + 1 + AH : 15)>(l);
                fmac_s(a[j + 1], sm[j + 1], nl);                   // column j + 1 is final: pivot j + 1 can start
                constexpr int rest = 14 - j;                       // updates left, dealt over the seven gaps in front of the stages
                double ajj = 0.0, y = 0.0, t = 0.0, e = 0.0, v = 0.0, u = 0.0, rinv = 0.0;
                static_for<0, 7>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    constexpr int nf = rest / 7 + (st < rest % 7 ? 1 : 0);                        // 0, 1 or 2
                    constexpr int k0 = j + 2 + st * (rest / 7) + (st < rest % 7 ? st : rest % 7);
                    static_for<0, nf>([&](auto fc) {
                        constexpr int k = k0 + decltype(fc)::value;
                        if constexpr (k + AH < 16) sm[k + AH] = readlane_f64<(k + AH < 16 ? k + AH : 15)>(l);
                        fmac_s(a[k], sm[k], nl);
                    });
                    if constexpr (st == 0) {
                        ajj = readlane_f64<j + 1>(a[j + 1]);
                    } else if constexpr (st == 1) {
                        y = __builtin_amdgcn_rsq(ajj);             // rsqrt_nr(ajj), stage by stage
                        if (!(ajj > 0.0) && badj < 0) badj = j + 1;
                        pin(y);
                    } else if constexpr (st == 2) {
                        t = ajj * y;
                        pin(t);
                    } else if constexpr (st == 3) {
                        e = fma(-t, y, 1.0);
                        pin(e);
                    } else if constexpr (st == 4) {
                        v = y * e;
                        u = fma(e, 0.375, 0.5);
                        pin(v); pin(u);
                    } else if constexpr (st == 5) {
                        rinv = fma(v, u, y);
                        pin(rinv);
                    } else {
                        const double l2 = a[j + 1] * rinv;
                        myrinv = (lane == j + 1) ? rinv : myrinv;
                        a[j + 1] = l2;
                        // (pivot j's l / nl are dead here: every update of pivot j has been issued above)
                        l = l2; nl = -l2;
                        pin(l); pin(nl);
                    }
                });
            });
            if (badj >= 0 && lane == 0 && s.bad < 0) s.bad = o + badj;
            if (lane < 16) s.rdg[o + lane] = myrinv;
            if (has_row) {
#pragma unroll
                for (int k = 0; k < 16; ++k) s.a[o + lane][o + k] = (lane >= 16 || k <= lane) ? a[k] : 0.0;
            }
                }
        __syncthreads();
stamp();
        if (nrem > 0) {
            // ---- trailing update A22 -= L21 L21^T on the matrix cores: 16x16 tiles of the lower triangle, K = 16
            const int nt = nrem >> 4, ntile = nt * (nt + 1) / 2;
            for (int e = wave; e < ntile; e += CHOL_THREADS / 64) {
                int ti = 0;
                while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
                const int tj = e - ti * (ti + 1) / 2;
                const int r0 = o + 16 + 16 * ti, c0 = o + 16 + 16 * tj;
                const int lr = lane & 15, lk = lane >> 4;
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = s.a[r0 + lk + 4 * r][c0 + lr];
#pragma unroll
                for (int kk = 0; kk < 16; kk += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-s.a[r0 + lr][o + kk + lk], s.a[c0 + lr][o + kk + lk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) s.a[r0 + lk + 4 * r][c0 + lr] = acc[r];
            }
            __syncthreads();
            stamp();
        }
    }
    if (wave == 0) {
        // ---- inverses of the four 16x16 diagonal factors in ONE wave: row w of 16 lanes inverts block w.  Lane c of
        //      the row owns column c of X (solve L x = e_c) and holds row c of L; right-looking: once x_i is known every
        //      later partial sum takes its term, L_mi coming from lane m of the row by broadcast
        const int o = 16 * (lane >> 4), li = lane & 15;
        double r[16], x[16], rg[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            r[k] = s.a[o + li][o + k];                 // row li of L11 (zeros above the diagonal)
            x[k] = (k == li) ? 1.0 : 0.0;
            rg[k] = s.rdg[o + k];
        }
        // x_m -= L_mi x_i with the broadcast of L_mi folded into the fma (v_fmac_f64_dpp: (-a) b + c = a (-b) + c, bit for bit;
        // half the instructions of v_mov_b64_dpp + v_fma_f64), and x_(i+1) scaled right behind the one update it waits for
        double nxi;
        {
            const double xi = x[0] * rg[0];
            x[0] = xi;
            nxi = -xi;
        }
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            fmac_bc16v(x[i + 1], r[i], nxi, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            const double xn = x[i + 1] * rg[i + 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = i + 2; m < 16; ++m) fmac_bc16v(x[m], r[i], nxi, m);      // L_mi x_i
            x[i + 1] = xn;
            nxi = -xn;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s.x[o + k][o + li] = (k >= li) ? x[k] : 0.0;        // column li of the inverse
    }
    __syncthreads();
stamp();
    // ---- assemble L^-1 by block doubling: X21 = -X22 (L21 X11), for 16-blocks (two pairs), then 32-blocks
    {
        const int lr = lane & 15, lk = lane >> 4;
        if (wave < 2) {                                // pair `wave`: T = L21 X11 (16x16x16)
            const int q0 = 32 * wave, q1 = q0 + 16;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s.a[q1 + lr][q0 + kk + lk], s.x[q0 + kk + lk][q0 + lr], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.tm[16 * wave + lk + 4 * r][lr] = acc[r];
        }
        __syncthreads();
        stamp();
        if (wave < 2) {                                // X21 = -X22 T: rows q1 .. q1+15, columns q0 .. q0+15
            const int q0 = 32 * wave, q1 = q0 + 16;
            d4 acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4)
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(s.x[q1 + lr][q1 + kk + lk], s.tm[16 * wave + kk + lk][lr], acc2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.x[q1 + lk + 4 * r][q0 + lr] = -acc2[r];
        }
        __syncthreads();
        stamp();
        if (wave < 4) {                                // T = L21 X11 (32x32x32), one 16x16 tile per wave
            const int ti = wave >> 1, tj = wave & 1;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 32; kk += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s.a[32 + 16 * ti + lr][kk + lk], s.x[kk + lk][16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.tm[16 * ti + lk + 4 * r][16 * tj + lr] = acc[r];
        }
        __syncthreads();
        stamp();
        if (wave < 4) {                                // X21 = -X22 T
            const int ti = wave >> 1, tj = wave & 1;
            d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 32; kk += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s.x[32 + 16 * ti + lr][32 + kk + lk], s.tm[kk + lk][16 * tj + lr], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.x[32 + 16 * ti + lk + 4 * r][16 * tj + lr] = -acc[r];
        }
        __syncthreads();
        stamp();
    }
}

}  // namespace gpb
