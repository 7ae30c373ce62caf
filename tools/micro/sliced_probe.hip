// Stage 1 of the sliced-integer k_predict (round 6, VERDICT r5 "next round" item 1): the inner loop of V = L^-1 K*^T evaluated on
// the int8 matrix pipe, with the real LDS staging and the real epilogue, on synthetic digit planes, one launch shaped like cfg 4
// (10 GPs x 2048 x 2048 x 2048 walkers, lower triangle).  Numerics of the scheme: tools/ozaki_probe.py; costs on paper:
// profiles/r06_sliced_model.txt.  Standalone:
//     hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/sliced_probe tools/micro/sliced_probe.hip
//     tools/micro/_bin/sliced_probe [reps]
//
// Scheme.  Both operands are D signed radix-256 digit planes (int8) of fixed-point numbers: row j of L^-1 scaled by 2^-eA[j],
// K*^T scaled by ONE power of two per GP.  The pairs (ta, tb) of one level ta + tb are summed EXACTLY in one int32 accumulator
// set by v_mfma_i32_32x32x32_i8; levels below LOW are dropped; the epilogue combines the levels in fp64 (Horner from the least
// significant one), scales, squares and reduces over the 64 rows of a wave tile into spart[64-row block][GP][walker] — the
// layout k_predict writes.  D = 7, LOW = 6: 28 products, fp64-equivalent; D = 6, LOW = 5: 21 products, ~2^-47.
//
// Layout of a plane set in HBM: plane[p][t][k / 16][row][16 bytes] — the 16 k-consecutive bytes one MFMA lane takes are one
// 16-byte granule, granules of consecutive rows (or walkers) are contiguous: a tile's share of a plane and k-block is ONE
// contiguous piece, moved by LDS-DMA (global_load_lds_dwordx4: no staging registers) into the same layout in LDS, from where
// ds_read_b128 delivers MFMA fragments with no transposition and no bank conflict.
//
// Block = 4 waves (2 x 2), wave tile (32 WTM) x (32 WTN), K-step 32 (one MFMA deep), three LDS stages: at step s the DMA of
// step s + 3 is issued, the fragments of step s + 1 are read, the MFMAs of step s run from registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define STAMP 1      // 1: stamp the wait + barrier of every step (diagnostic build; costs a few percent)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct TileRec { int p, ib, wt, pad; };

template <int D, int WTM, int WTN>
struct Geo {
    static constexpr int BM = 64 * WTM, BN = 64 * WTN;                 // block tile (2 x 2 waves)
    static constexpr int A_CHUNKS = D * 2 * (BM / 64), B_CHUNKS = D * 2 * (BN / 64);      // 1-KB pieces per stage
    static constexpr int CHUNKS = A_CHUNKS + B_CHUNKS;
    static constexpr int STAGE_BYTES = CHUNKS * 1024;
    static constexpr int NSTAGE = 3;
    static constexpr int NPW = (CHUNKS + 3) / 4;                        // DMA instructions per wave and stage
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES + BM * 8;     // + the tile's row scales
};

// wave-uniform: issue this wave's share of the DMA of one K-step (k-blocks kb0, kb0 + 1) into stage buffer `buf`.  Branch-free
// and the same count NPW for every wave (a surplus slot repeats the stage's last piece: same bytes to the same place), so that
// one counted s_waitcnt vmcnt(NPW) retires exactly one stage and the instructions can be interleaved with the MFMAs.
template <int D, int WTM, int WTN>
__device__ __forceinline__ void dma_stage(char* lds, int buf, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                          int64_t Np, int64_t Wld, int64_t mb, int64_t nb, int64_t kb0, int wave, int lane) {
    typedef Geo<D, WTM, WTN> G;
    char* base = lds + buf * G::STAGE_BYTES;
    const int64_t a_plane = (Np / 16) * Np * 16, b_plane = (Np / 16) * Wld * 16;
#pragma unroll
    for (int c = 0; c < G::NPW; ++c) {
        int ch = 4 * c + wave;                                          // wave-uniform
        ch = ch < G::CHUNKS ? ch : G::CHUNKS - 1;
        const bool is_a = ch < G::A_CHUNKS;
        const int cb = is_a ? ch : ch - G::A_CHUNKS;
        const int per = is_a ? G::BM / 64 : G::BN / 64;                 // 1-KB pieces per (plane, k-block)
        const int t = cb / (2 * per), q = (cb / per) & 1, h = cb % per;
        const int64_t ld = is_a ? Np : Wld, off = is_a ? mb : nb;
        const int8_t* plane = is_a ? Ap + t * a_plane : Bp + t * b_plane;
        const int8_t* src = plane + ((kb0 + q) * ld + off + 64 * h + lane) * 16;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)(base + ch * 1024), 16, 0, 0);
    }
}

template <int D, int WTM, int WTN>
struct Frags {
    v4i a[WTM][D], b[WTN][D];
};

template <int D, int WTM, int WTN>
__device__ __forceinline__ void read_frags(Frags<D, WTM, WTN>& f, const char* lds, int buf, int wm, int wn, int lane) {
    typedef Geo<D, WTM, WTN> G;
    const char* base = lds + buf * G::STAGE_BYTES;
    const int q = lane >> 5, r = lane & 31;
#pragma unroll
    for (int t = 0; t < D; ++t) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
            f.a[i][t] = *reinterpret_cast<const v4i*>(base + ((t * 2 + q) * G::BM + wm * 32 * WTM + 32 * i + r) * 16);
#pragma unroll
        for (int j = 0; j < WTN; ++j)
            f.b[j][t] = *reinterpret_cast<const v4i*>(base + G::A_CHUNKS * 1024 + ((t * 2 + q) * G::BN + wn * 32 * WTN + 32 * j + r) * 16);
    }
}

template <int D, int LOW, int WTM, int WTN>
__device__ __forceinline__ void mma_step(const Frags<D, WTM, WTN>& f, v16i (&acc)[2 * D - 1 - LOW][WTM][WTN]) {
#pragma unroll
    for (int tb = 0; tb < D; ++tb)
#pragma unroll
        for (int ta = 0; ta < D; ++ta) {
            if (ta + tb < LOW) continue;
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[ta + tb - LOW][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[i][ta], f.b[j][tb], acc[ta + tb - LOW][i][j], 0, 0, 0);
        }
}

// register-lean form for the 128 x 128 block (6 levels x 4 tiles x 16 = 384 accumulator registers): fragments are read where they
// are used, B per n-tile, A per (n-tile, m-tile) — A twice per K-step
template <int D, int LOW, int WTM, int WTN>
__device__ __forceinline__ void mma_step_streamed(const char* lds, int buf, int wm, int wn, int lane, v16i (&acc)[2 * D - 1 - LOW][WTM][WTN]) {
    typedef Geo<D, WTM, WTN> G;
    const char* base = lds + buf * G::STAGE_BYTES;
    const int q = lane >> 5, r = lane & 31;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        v4i b[D];
#pragma unroll
        for (int t = 0; t < D; ++t)
            b[t] = *reinterpret_cast<const v4i*>(base + G::A_CHUNKS * 1024 + ((t * 2 + q) * G::BN + wn * 32 * WTN + 32 * j + r) * 16);
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            v4i a[D];
#pragma unroll
            for (int t = 0; t < D; ++t)
                a[t] = *reinterpret_cast<const v4i*>(base + ((t * 2 + q) * G::BM + wm * 32 * WTM + 32 * i + r) * 16);
#pragma unroll
            for (int tb = 0; tb < D; ++tb)
#pragma unroll
                for (int ta = 0; ta < D; ++ta) {
                    if (ta + tb < LOW) continue;
                    acc[ta + tb - LOW][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], b[tb], acc[ta + tb - LOW][i][j], 0, 0, 0);
                }
        }
    }
}

__device__ __forceinline__ v4i rnd_frag(int tid, int k) {
    const unsigned h = (unsigned)tid * 0x9E3779B1u + (unsigned)k * 0x85EBCA6Bu;
    return v4i{(int)(h * 0xC2B2AE35u), (int)((h ^ (h >> 15)) * 0x27D4EB2Fu), (int)(h * 77u + 12345u), (int)~(h * 0x165667B1u)};
}

template <int N>
__device__ __forceinline__ void sgb_n() {
    __builtin_amdgcn_sched_group_barrier(0x008, N, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
}
__device__ __forceinline__ void sgb_mfma_then_read(int n) {          // n is a constant after unrolling
    switch (n) {
        case 1: sgb_n<1>(); break; case 2: sgb_n<2>(); break; case 3: sgb_n<3>(); break; case 4: sgb_n<4>(); break;
        case 5: sgb_n<5>(); break; case 6: sgb_n<6>(); break; case 7: sgb_n<7>(); break; default: break;
    }
}

// The streamed step with the fragment reads placed by hand (WTN = 2): all of a and of the first n-tile's b issued back to back in
// the order of their first use (the first MFMA waits for two reads, not twelve), the second n-tile's b[t] read into the registers
// b0[t] leaves as soon as b0[t]'s MFMAs are issued, so the second half starts with its fragments there.
template <int D, int LOW, class F>
__device__ __forceinline__ void mma_step8_ordered(const char* lds, int buf, int wm, int wn, int lane, v16i (&acc)[2 * D - 1 - LOW][2], F dma) {
    typedef Geo<D, 2, 2> G;
    const char* base = lds + buf * G::STAGE_BYTES;
    const int q = lane >> 5, r = lane & 31;
    const char* pa = base + (q * G::BM + wm * 32 + r) * 16;
    const char* pb = base + G::A_CHUNKS * 1024 + (q * G::BN + wn * 64 + r) * 16;
    v4i a[D], b[D], c[D];
#pragma unroll
    for (int t = 0; t < D; ++t) {                                       // first use: MFMA group tb = t needs a[D - 1 - t .. D - 1] and b[t]
        a[D - 1 - t] = *reinterpret_cast<const v4i*>(pa + (D - 1 - t) * 2 * G::BM * 16);
        b[t] = *reinterpret_cast<const v4i*>(pb + t * 2 * G::BN * 16);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * D, 0);
#pragma unroll
    for (int tb = 0; tb < D; ++tb) {
#pragma unroll
        for (int ta = D - 1; ta >= 0; --ta) {
            if (ta + tb < LOW) continue;
            acc[ta + tb - LOW][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], b[tb], acc[ta + tb - LOW][0], 0, 0, 0);
        }
        c[tb] = *reinterpret_cast<const v4i*>(pb + 32 * 16 + tb * 2 * G::BN * 16);      // the second n-tile's plane tb
        sgb_mfma_then_read(tb + 1 - (LOW - (D - 1)));
    }
    __builtin_amdgcn_sched_barrier(0);                                  // the second n-tile's MFMAs stay behind the first's
#pragma unroll
    for (int tb = 0; tb < D; ++tb)
#pragma unroll
        for (int ta = D - 1; ta >= 0; --ta) {
            if (ta + tb < LOW) continue;
            acc[ta + tb - LOW][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], c[tb], acc[ta + tb - LOW][1], 0, 0, 0);
        }
    dma();
}

// MODE bits: 1 = DMA, 2 = fragment reads, 4 = MFMAs (7 = the real loop; the others isolate one limiter each)
template <int D, int LOW, int WTM, int WTN, int MODE, bool PF, bool SCHED>
__global__ __launch_bounds__(256, 1) void k_sliced(const int8_t* __restrict__ A, const int8_t* __restrict__ B,
                                                   const double* __restrict__ rowscale, const double* __restrict__ colscale,
                                                   double* __restrict__ spart, const TileRec* __restrict__ tiles, int64_t Np,
                                                   int64_t Wld, int P, unsigned long long* __restrict__ stamps) {
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stall = 0;
    typedef Geo<D, WTM, WTN> G;
    constexpr int NLEV = 2 * D - 1 - LOW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const TileRec tr = tiles[blockIdx.x];
    const int p = tr.p, ib = tr.ib, wt = tr.wt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t mb = (int64_t)ib * G::BM, nb = (int64_t)wt * G::BN;
    const int8_t* Ap = A + (int64_t)p * D * (Np / 16) * Np * 16;
    const int8_t* Bp = B + (int64_t)p * D * (Np / 16) * Wld * 16;
    double* rs = reinterpret_cast<double*>(lds + G::NSTAGE * G::STAGE_BYTES);
    if (tid < G::BM) rs[tid] = rowscale[(int64_t)p * Np + mb + tid];
    const int nsteps = (int)((mb + G::BM) / 32);                        // k in [0, mb + BM): the triangle's row block
    v16i acc[NLEV][WTM][WTN];
#pragma unroll
    for (int l = 0; l < NLEV; ++l)
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[l][i][j][r] = 0;
    constexpr int NPW = G::NPW;
    constexpr int NMMA = ((D * D) - (LOW * (LOW + 1)) / 2) * WTM * WTN, NRD = (WTM + WTN) * D;
    if constexpr (PF) {
        if (MODE & 1) {
            dma_stage<D, WTM, WTN>(lds, 0, Ap, Bp, Np, Wld, mb, nb, 0, wave, lane);
            if (nsteps > 1) dma_stage<D, WTM, WTN>(lds, 1, Ap, Bp, Np, Wld, mb, nb, 2, wave, lane);
            if (nsteps > 2) dma_stage<D, WTM, WTN>(lds, 2, Ap, Bp, Np, Wld, mb, nb, 4, wave, lane);
            if (nsteps > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
            else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the row scales are in LDS
        __builtin_amdgcn_s_barrier();
        Frags<D, WTM, WTN> f0, f1;
        if (MODE & 2) read_frags<D, WTM, WTN>(f0, lds, 0, wm, wn, lane);
        else {
#pragma unroll
            for (int t = 0; t < D; ++t) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) f0.a[i][t] = f1.a[i][t] = rnd_frag(tid, 2 * t + 31 * i);
#pragma unroll
                for (int j = 0; j < WTN; ++j) f0.b[j][t] = f1.b[j][t] = rnd_frag(tid, 2 * t + 1 + 37 * j);
            }
        }
        // steady state, no branch inside (one scheduling region): frags(s) in registers, stage s + 1 landed behind the counted wait
        // and the barrier, stage s + 2 in flight; DMA of stage s + 3 into the buffer frags(s) came from, fragment reads of stage
        // s + 1 and the MFMAs of step s interleaved by the scheduler directives below
        auto step_full = [&](int s, Frags<D, WTM, WTN>& cur, Frags<D, WTM, WTN>& nxt) {
            const unsigned long long w0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's reads of buffer s % 3 are done
            if (MODE & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            __builtin_amdgcn_s_barrier();
            if (STAMP) stall += __builtin_amdgcn_s_memtime() - w0;
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 1) dma_stage<D, WTM, WTN>(lds, s % 3, Ap, Bp, Np, Wld, mb, nb, 2 * (int64_t)(s + 3), wave, lane);
            if (MODE & 2) read_frags<D, WTM, WTN>(nxt, lds, (s + 1) % 3, wm, wn, lane);
            if (MODE & 4) mma_step<D, LOW, WTM, WTN>(cur, acc);
            if (SCHED && (MODE & 4)) {
                // NMMA MFMAs, NRD fragment reads, NPW DMAs: one memory instruction per MFMA while they last, DMAs first
                if (MODE & 1) {
#pragma unroll
                    for (int i = 0; i < NPW; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                if (MODE & 2) {
#pragma unroll
                    for (int i = 0; i < NRD; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, (NMMA - NPW) / NRD, 0);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMMA, 0);    // the rest
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto step_tail = [&](int s, Frags<D, WTM, WTN>& cur, Frags<D, WTM, WTN>& nxt) {
            if (s + 1 < nsteps) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (MODE & 1) {
                    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");   // stage s + 1 has landed
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                if ((MODE & 1) && s + 3 < nsteps)
                    dma_stage<D, WTM, WTN>(lds, s % 3, Ap, Bp, Np, Wld, mb, nb, 2 * (int64_t)(s + 3), wave, lane);
                if (MODE & 2) read_frags<D, WTM, WTN>(nxt, lds, (s + 1) % 3, wm, wn, lane);
            }
            if (MODE & 4) mma_step<D, LOW, WTM, WTN>(cur, acc);
        };
        int s = 0;
        for (; s + 4 < nsteps; s += 2) {
            step_full(s, f0, f1);
            step_full(s + 1, f1, f0);
        }
        for (; s + 1 < nsteps; s += 2) {
            step_tail(s, f0, f1);
            step_tail(s + 1, f1, f0);
        }
        if (s < nsteps) step_tail(s, f0, f1);
    } else {
        // fragments read inside the step: stages s + 1, s + 2 in flight while step s runs from buffer s % 3
        if (MODE & 1) {
            dma_stage<D, WTM, WTN>(lds, 0, Ap, Bp, Np, Wld, mb, nb, 0, wave, lane);
            if (nsteps > 1) dma_stage<D, WTM, WTN>(lds, 1, Ap, Bp, Np, Wld, mb, nb, 2, wave, lane);
        }
        for (int s = 0; s < nsteps; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE & 1) {
                if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");       // stage s has landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if ((MODE & 1) && s + 2 < nsteps)
                dma_stage<D, WTM, WTN>(lds, (s + 2) % 3, Ap, Bp, Np, Wld, mb, nb, 2 * (int64_t)(s + 2), wave, lane);
            if (MODE & 4) mma_step_streamed<D, LOW, WTM, WTN>(lds, s % 3, wm, wn, lane, acc);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // epilogue: levels -> fp64 (Horner from the least significant level), scale, square, sum over the wave tile's rows in a
    // fixed order (m-tile, register, then the two lane halves), one partial per 64-row block when WTM = 2
    const double cs = colscale[p];
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double t = (double)acc[0][i][j][r];
#pragma unroll
                for (int l = 1; l < NLEV; ++l) t = fma(t, 1.0 / 256.0, (double)acc[l][i][j][r]);
                const int row = wm * 32 * WTM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const double v = t * rs[row] * cs;
                sum = fma(v, v, sum);
            }
        sum += __shfl_xor(sum, 32);
        if (lane < 32) spart[(((int64_t)ib * 2 + wm) * P + p) * Wld + nb + wn * 32 * WTN + 32 * j + lane] = sum;
    }
    if (stamps && tid == 0) {        // in-kernel clock = d memtime / d memrealtime x 100 MHz (diagnostic: values go nowhere else)
        stamps[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
        stamps[3 * blockIdx.x + 2] = stall;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same block tile (128 x 64, K-step 32, three LDS stages) with EIGHT waves, two per SIMD, each owning one 32 x 32 output tile
// (7 levels x 16 = 112 accumulator registers; <= 256 registers per wave).  Why: one global_load_lds costs its wave ~60 cycles of
// issue time (MI355X_MICROARCH.md, cycle constants), during which a lone wave's SIMD issues no MFMA; with a partner wave on the
// SIMD the matrix pipe keeps going.  DMAs are dealt over the eight waves; a wave's count per stage is NPW8 or NPW8 - 1.
template <int D, int WTN>
struct Geo8 {
    typedef Geo<D, 2, WTN> G;
    static constexpr int NPW = (G::CHUNKS + 7) / 8, REM = G::CHUNKS % 8;      // waves < REM issue NPW, the others NPW - 1 (REM = 0: all NPW)
};

template <int D, int WTN>
__device__ __forceinline__ void dma_stage8(char* lds, int buf, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                           int64_t Np, int64_t Wld, int64_t mb, int64_t nb, int64_t kb0, int wave, int lane) {
    typedef Geo<D, 2, WTN> G;
    char* base = lds + buf * G::STAGE_BYTES;
    const int64_t a_plane = (Np / 16) * Np * 16, b_plane = (Np / 16) * Wld * 16;
#pragma unroll
    for (int c = 0; c < Geo8<D, WTN>::NPW; ++c) {
        const int ch = 8 * c + wave;                                    // wave-uniform
        if (c == Geo8<D, WTN>::NPW - 1 && Geo8<D, WTN>::REM != 0 && wave >= Geo8<D, WTN>::REM) break;
        const bool is_a = ch < G::A_CHUNKS;
        const int cb = is_a ? ch : ch - G::A_CHUNKS;
        const int per = is_a ? G::BM / 64 : G::BN / 64;
        const int t = cb / (2 * per), q = (cb / per) & 1, h = cb % per;
        const int64_t ld = is_a ? Np : Wld, off = is_a ? mb : nb;
        const int8_t* plane = is_a ? Ap + t * a_plane : Bp + t * b_plane;
        const int8_t* src = plane + ((kb0 + q) * ld + off + 64 * h + lane) * 16;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)(base + ch * 1024), 16, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// wait until at most `stages` of this wave's stages are still in flight
template <int D, int WTN>
__device__ __forceinline__ void wait_stages(int stages, bool full_count) {
    constexpr int NPW = Geo8<D, WTN>::NPW;
    if (stages == 0) { wait_vm<0>(); return; }
    if (full_count) { if (stages == 1) wait_vm<NPW>(); else wait_vm<2 * NPW>(); }
    else            { if (stages == 1) wait_vm<NPW - 1>(); else wait_vm<2 * (NPW - 1)>(); }
}

template <int D, int WTN>
struct Frags8 { v4i a[D], b[WTN][D]; };

template <int D, int WTN>
__device__ __forceinline__ void read_frags8(Frags8<D, WTN>& f, const char* lds, int buf, int wm, int wn, int lane) {
    typedef Geo<D, 2, WTN> G;
    const char* base = lds + buf * G::STAGE_BYTES;
    const int q = lane >> 5, r = lane & 31;
#pragma unroll
    for (int t = 0; t < D; ++t) {
        f.a[t] = *reinterpret_cast<const v4i*>(base + ((t * 2 + q) * G::BM + wm * 32 + r) * 16);
#pragma unroll
        for (int j = 0; j < WTN; ++j)
            f.b[j][t] = *reinterpret_cast<const v4i*>(base + G::A_CHUNKS * 1024 + ((t * 2 + q) * G::BN + wn * 32 * WTN + 32 * j + r) * 16);
    }
}

template <int D, int LOW, int WTN>
__device__ __forceinline__ void mma_step8(const Frags8<D, WTN>& f, v16i (&acc)[2 * D - 1 - LOW][WTN]) {
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int tb = 0; tb < D; ++tb)
#pragma unroll
            for (int ta = 0; ta < D; ++ta) {
                if (ta + tb < LOW) continue;
                acc[ta + tb - LOW][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[ta], f.b[j][tb], acc[ta + tb - LOW][j], 0, 0, 0);
            }
}

// register-lean: A fragments once, B fragments of one n-tile at a time, read where they are used.  `dma` (this wave's DMA issue
// of stage s + 2) runs at position POS: 0 = before the reads, 1 = behind the first reads (a, b0) while they fly, 2 = behind the
// first n-tile's MFMAs, 3 = waves 0-3 as 1, waves 4-7 as 2 (the two waves of a SIMD out of step)
template <int D, int LOW, int WTN, int POS, class F>
__device__ __forceinline__ void mma_step8_streamed(const char* lds, int buf, int wm, int wn, int lane, int wave, v16i (&acc)[2 * D - 1 - LOW][WTN], F dma) {
    typedef Geo<D, 2, WTN> G;
    const char* base = lds + buf * G::STAGE_BYTES;
    const int q = lane >> 5, r = lane & 31;
    if (POS == 0) dma();
    v4i a[D];
#pragma unroll
    for (int t = 0; t < D; ++t) a[t] = *reinterpret_cast<const v4i*>(base + ((t * 2 + q) * G::BM + wm * 32 + r) * 16);
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        v4i b[D];
#pragma unroll
        for (int t = 0; t < D; ++t)
            b[t] = *reinterpret_cast<const v4i*>(base + G::A_CHUNKS * 1024 + ((t * 2 + q) * G::BN + wn * 32 * WTN + 32 * j + r) * 16);
        if (j == 0 && (POS == 1 || ((POS == 3 || POS == 4 || POS == 6) && wave < 4))) dma();
        if (j == 1 && (POS == 2 || ((POS == 3 || POS == 6) && wave >= 4) || (POS == 5 && wave < 4))) dma();
#pragma unroll
        for (int tb = 0; tb < D; ++tb)
#pragma unroll
            for (int ta = 0; ta < D; ++ta) {
                if (ta + tb < LOW) continue;
                acc[ta + tb - LOW][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta], b[tb], acc[ta + tb - LOW][j], 0, 0, 0);
            }
    }
    if (WTN == 1 && POS >= 2) dma();
    if (WTN == 2 && (((POS == 4 || POS == 5) && wave >= 4) || POS == 7)) dma();      // behind all of the step's MFMAs
}

template <int D, int LOW, int MODE, bool PF, int WTN, int POS = 0>
__global__ __launch_bounds__(512, 2) void k_sliced8(const int8_t* __restrict__ A, const int8_t* __restrict__ B,
                                                    const double* __restrict__ rowscale, const double* __restrict__ colscale,
                                                    double* __restrict__ spart, const TileRec* __restrict__ tiles, int64_t Np,
                                                    int64_t Wld, int P, unsigned long long* __restrict__ stamps) {
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stall = 0;
    typedef Geo<D, 2, WTN> G;
    constexpr int NLEV = 2 * D - 1 - LOW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const TileRec tr = tiles[blockIdx.x];
    const int p = tr.p, ib = tr.ib, wt = tr.wt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                            // 4 x 2 waves of 32 x 32
    const int64_t mb = (int64_t)ib * G::BM, nb = (int64_t)wt * G::BN;
    const int8_t* Ap = A + (int64_t)p * D * (Np / 16) * Np * 16;
    const int8_t* Bp = B + (int64_t)p * D * (Np / 16) * Wld * 16;
    double* rs = reinterpret_cast<double*>(lds + G::NSTAGE * G::STAGE_BYTES);
    if (tid < G::BM) rs[tid] = rowscale[(int64_t)p * Np + mb + tid];
    const int nsteps = (int)((mb + G::BM) / 32);
    const bool fullc = Geo8<D, WTN>::REM == 0 || wave < Geo8<D, WTN>::REM;
    if (POS == 6 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    v16i acc[NLEV][WTN];
#pragma unroll
    for (int l = 0; l < NLEV; ++l)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;
    if (MODE & 1) {
        dma_stage8<D, WTN>(lds, 0, Ap, Bp, Np, Wld, mb, nb, 0, wave, lane);
        if (nsteps > 1) dma_stage8<D, WTN>(lds, 1, Ap, Bp, Np, Wld, mb, nb, 2, wave, lane);
        if (PF && nsteps > 2) dma_stage8<D, WTN>(lds, 2, Ap, Bp, Np, Wld, mb, nb, 4, wave, lane);
    }
    if constexpr (PF) {
        // fragments one step ahead in registers: at step s stage s + 1 is read, stages s + 2, s + 3 are in flight
        if (MODE & 1) wait_stages<D, WTN>(nsteps > 2 ? 2 : (nsteps > 1 ? 1 : 0), fullc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        Frags8<D, WTN> f0, f1;
        if (MODE & 2) read_frags8<D, WTN>(f0, lds, 0, wm, wn, lane);
        else {
#pragma unroll
            for (int t = 0; t < D; ++t) {
                f0.a[t] = f1.a[t] = rnd_frag(tid, 2 * t);
#pragma unroll
                for (int j = 0; j < WTN; ++j) f0.b[j][t] = f1.b[j][t] = rnd_frag(tid, 2 * t + 1 + 64 * j);
            }
        }
        auto step = [&](int s, Frags8<D, WTN>& cur, Frags8<D, WTN>& nxt) {
            if (s + 1 < nsteps) {
                const unsigned long long w0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (MODE & 1) wait_stages<D, WTN>(s + 2 < nsteps ? 1 : 0, fullc);      // stage s + 1 has landed
                __builtin_amdgcn_s_barrier();
                if (STAMP) stall += __builtin_amdgcn_s_memtime() - w0;
                if ((MODE & 1) && s + 3 < nsteps) dma_stage8<D, WTN>(lds, s % 3, Ap, Bp, Np, Wld, mb, nb, 2 * (int64_t)(s + 3), wave, lane);
                if (MODE & 2) read_frags8<D, WTN>(nxt, lds, (s + 1) % 3, wm, wn, lane);
            }
            if (MODE & 4) mma_step8<D, LOW, WTN>(cur, acc);
        };
        int s = 0;
        for (; s + 1 < nsteps; s += 2) { step(s, f0, f1); step(s + 1, f1, f0); }
        if (s < nsteps) step(s, f0, f1);
    } else {
        // fragments read in the step that uses them (the partner wave covers the LDS latency): stage s read, s + 1, s + 2 in flight
        for (int s = 0; s < nsteps; ++s) {
            const unsigned long long w0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE & 1) wait_stages<D, WTN>(s + 1 < nsteps ? 1 : 0, fullc);          // stage s has landed
            __builtin_amdgcn_s_barrier();
            if (STAMP) stall += __builtin_amdgcn_s_memtime() - w0;
            auto dma = [&]() {
                if ((MODE & 1) && s + 2 < nsteps) dma_stage8<D, WTN>(lds, (s + 2) % 3, Ap, Bp, Np, Wld, mb, nb, 2 * (int64_t)(s + 2), wave, lane);
            };
            if ((MODE & 6) != 6) dma();
            if constexpr (POS == 8 && WTN == 2) { if ((MODE & 6) == 6) mma_step8_ordered<D, LOW>(lds, s % 3, wm, wn, lane, acc, dma); }
            else if ((MODE & 6) == 6) mma_step8_streamed<D, LOW, WTN, POS>(lds, s % 3, wm, wn, lane, wave, acc, dma);
            else if (MODE & 4) {
                Frags8<D, WTN> f;
#pragma unroll
                for (int t = 0; t < D; ++t) {
                    f.a[t] = rnd_frag(tid, 2 * t);
#pragma unroll
                    for (int j = 0; j < WTN; ++j) f.b[j][t] = rnd_frag(tid, 2 * t + 1 + 64 * j);
                }
                mma_step8<D, LOW, WTN>(f, acc);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // epilogue: one partial per 64-row block = wave rows (2 b, 2 b + 1): the odd wave row hands its 32-row sum to the even one
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
            for (int l = 1; l < NLEV; ++l) t = fma(t, 1.0 / 256.0, (double)acc[l][j][r]);
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const double v = t * rs[row] * cs;
            sum = fma(v, v, sum);
        }
        sum += __shfl_xor(sum, 32);
        sums[j] = sum;
        if ((wm & 1) && lane < 32) hand[(wave * WTN + j) * 32 + lane] = sum;
    }
    __syncthreads();
    if (!(wm & 1) && lane < 32) {
#pragma unroll
        for (int j = 0; j < WTN; ++j)
            spart[(((int64_t)ib * 2 + (wm >> 1)) * P + p) * Wld + nb + wn * 32 * WTN + 32 * j + lane] = sums[j] + hand[((wave + 2) * WTN + j) * 32 + lane];
    }
    if (stamps && tid == 0) {
        stamps[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
        stamps[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
        stamps[3 * blockIdx.x + 2] = stall;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

struct Problem {
    int P; int64_t Np, W; int D;
    std::vector<int8_t> A, B; std::vector<double> rs, cs;
    int8_t *dA = nullptr, *dB = nullptr; double *drs = nullptr, *dcs = nullptr, *dsp = nullptr;
};

static void make_problem(Problem& pr, int P, int64_t Np, int64_t W, int D, bool keep_host) {
    pr.P = P; pr.Np = Np; pr.W = W; pr.D = D;
    const size_t na = (size_t)P * D * Np * Np, nb = (size_t)P * D * Np * W;
    pr.A.resize(na); pr.B.resize(nb); pr.rs.resize((size_t)P * Np); pr.cs.resize(P);
    // random digits (random data: the clock the chip holds under load depends on it); A is lower triangular: plane[t][k/16][row][k%16]
    for (size_t i = 0; i < na; i += 4) { uint32_t r = rnd(); memcpy(&pr.A[i], &r, 4); }
    for (size_t i = 0; i < nb; i += 4) { uint32_t r = rnd(); memcpy(&pr.B[i], &r, 4); }
    for (int p = 0; p < P; ++p)
        for (int t = 0; t < D; ++t)
            for (int64_t kb = 0; kb < Np / 16; ++kb)
                for (int64_t row = 0; row < Np; ++row)
                    for (int kk = 0; kk < 16; ++kk)
                        if (kb * 16 + kk > row) pr.A[((((size_t)p * D + t) * (Np / 16) + kb) * Np + row) * 16 + kk] = 0;
    for (auto& v : pr.rs) v = ldexp(1.0, (int)(rnd() % 5) - 2);
    for (auto& v : pr.cs) v = ldexp(1.0, -8 * (2 * D - 2) / 2);          // keeps the sums in range; any power of two
    CK(hipMalloc(&pr.dA, na)); CK(hipMalloc(&pr.dB, nb));
    CK(hipMalloc(&pr.drs, pr.rs.size() * 8)); CK(hipMalloc(&pr.dcs, pr.cs.size() * 8));
    CK(hipMalloc(&pr.dsp, (size_t)(Np / 64) * P * W * 8));
    CK(hipMemcpy(pr.dA, pr.A.data(), na, hipMemcpyHostToDevice)); CK(hipMemcpy(pr.dB, pr.B.data(), nb, hipMemcpyHostToDevice));
    CK(hipMemcpy(pr.drs, pr.rs.data(), pr.rs.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pr.dcs, pr.cs.data(), pr.cs.size() * 8, hipMemcpyHostToDevice));
    if (!keep_host) { pr.A.clear(); pr.A.shrink_to_fit(); pr.B.clear(); pr.B.shrink_to_fit(); }
}

static void free_problem(Problem& pr) { (void)hipFree(pr.dA); (void)hipFree(pr.dB); (void)hipFree(pr.drs); (void)hipFree(pr.dcs); (void)hipFree(pr.dsp); }

// tile orders.  map 0: heaviest row block first, GPs interleaved, walker tile fastest (queue = walker tile % 8 under round-robin
// dispatch: what k_predict does today).  map 1: super-blocks (GP, RG consecutive row blocks, CG walker tiles), RG * CG = 32 = the
// workgroups one XCD holds, heaviest first, dealt round-robin to the eight XCDs (block b -> XCD b % 8).
static std::vector<TileRec> make_tiles(int P, int nI, int nW, int map, int RG, int CG) {
    std::vector<TileRec> t;
    if (map == 2) {                                                     // every block the heaviest tile of GP 0: pure cache hits
        for (int i = 0; i < P * nI * nW; ++i) t.push_back({0, nI - 1, 0, 0});
        return t;
    }
    if (map == 0) {
        for (int ib = nI - 1; ib >= 0; --ib) for (int p = 0; p < P; ++p) for (int wt = 0; wt < nW; ++wt) t.push_back({p, ib, wt, 0});
        return t;
    }
    struct SB { int p, i0, w0; };
    std::vector<SB> sbs;
    for (int g = (nI + RG - 1) / RG - 1; g >= 0; --g)
        for (int p = 0; p < P; ++p)
            for (int w0 = 0; w0 < nW; w0 += CG) sbs.push_back({p, g * RG, w0});
    // XCD x takes super-blocks x, x + 8, ...; its workgroups are blocks x, x + 8, ... in order
    std::vector<std::vector<TileRec>> per(8);
    for (size_t s = 0; s < sbs.size(); ++s)
        for (int i = std::min(sbs[s].i0 + RG, nI) - 1; i >= sbs[s].i0; --i)
            for (int w = sbs[s].w0; w < std::min(sbs[s].w0 + CG, nW); ++w) per[s % 8].push_back({sbs[s].p, i, w, 0});
    size_t mx = 0; for (auto& v : per) mx = std::max(mx, v.size());
    for (size_t k = 0; k < mx; ++k) for (int x = 0; x < 8; ++x) t.push_back(k < per[x].size() ? per[x][k] : TileRec{-1, 0, 0, 0});
    // (padding records would need a guard in the kernel: the shapes used here divide evenly)
    for (auto& r : t) if (r.p < 0) { fprintf(stderr, "uneven super-block deal\n"); exit(1); }
    return t;
}

template <int D, int LOW, int WTM, int WTN, int MODE, bool SCHED = true>
static double run(Problem& pr, int map, int RG, int CG, int reps, const char* name, bool print = true) {
    typedef Geo<D, WTM, WTN> G;
    const int nI = (int)(pr.Np / G::BM), nW = (int)(pr.W / G::BN);
    std::vector<TileRec> tiles = make_tiles(pr.P, nI, nW, map, RG, CG);
    TileRec* dt; CK(hipMalloc(&dt, tiles.size() * sizeof(TileRec)));
    CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(TileRec), hipMemcpyHostToDevice));
    auto kern = k_sliced<D, LOW, WTM, WTN, MODE, (WTN == 1), SCHED>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    unsigned long long* dst; CK(hipMalloc(&dst, tiles.size() * 24)); CK(hipMemset(dst, 0, tiles.size() * 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles.size()), dim3(256), G::LDS_BYTES, 0, pr.dA, pr.dB, pr.drs, pr.dcs, pr.dsp, dt, pr.Np, pr.W, pr.P, dst);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles.size()), dim3(256), G::LDS_BYTES, 0, pr.dA, pr.dB, pr.drs, pr.dcs, pr.dsp, dt, pr.Np, pr.W, pr.P, dst);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> st(tiles.size() * 3);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost)); CK(hipFree(dst));
    std::vector<double> clk;
    double busy = 0, stl = 0;
    for (size_t i = 0; i < tiles.size(); ++i) {
        if (st[3 * i + 1] > 200) clk.push_back((double)st[3 * i] / (double)st[3 * i + 1] * 0.1);   // GHz
        busy += (double)st[3 * i]; stl += (double)st[3 * i + 2];
    }
    std::sort(clk.begin(), clk.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    constexpr int NPROD = (D * D) - (LOW * (LOW + 1)) / 2;              // pairs with ta + tb >= LOW (LOW <= D)
    double ksteps = 0; for (auto& t : tiles) ksteps += (double)(t.ib + 1) * G::BM / 32;
    const double ops = ksteps * G::BM * G::BN * 32.0 * 2 * NPROD;        // executed int8 multiply-adds x 2
    const double alg = (double)pr.P * pr.Np * pr.Np * pr.W;              // fp64 flops of the triangle product (k_predict's count)
    if (print)
        printf("%-34s D=%d prod=%2d tile=%3dx%-3d map=%d(%dx%d) lds=%3d KB  %8.4f ms  %7.1f TOP/s int8  %6.1f TF/s fp64-equivalent  (x%.2f of 1.311 ms)  %.2f GHz  wait+barrier %.0f%% of block time, blocks fill %.0f%% of CU time\n",
               name, D, NPROD, G::BM, G::BN, map, RG, CG, G::LDS_BYTES / 1024, ms, ops / (ms * 1e-3) / 1e12, alg / (ms * 1e-3) / 1e12, 1.311 / ms, ghz, 100.0 * stl / busy, 100.0 * busy / (ghz * 1e9 * ms * 1e-3 * 256));
    CK(hipFree(dt));
    return ms;
}

// bit-exact check of the full loop against the host: exact integer level sums, the same fp64 epilogue
template <int D, int LOW, int WTM, int WTN>
static int check() {
    typedef Geo<D, WTM, WTN> G;
    constexpr int NLEV = 2 * D - 1 - LOW;
    Problem pr; make_problem(pr, 2, 256, 256, D, true);
    run<D, LOW, WTM, WTN, 7>(pr, 0, 1, 1, 1, "check", false);
    std::vector<double> sp((size_t)(pr.Np / 64) * pr.P * pr.W);
    CK(hipMemcpy(sp.data(), pr.dsp, sp.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0; double worst = 0;
    const int64_t Np = pr.Np, W = pr.W;
    auto Aat = [&](int p, int t, int64_t row, int64_t k) { return (int)pr.A[((((size_t)p * D + t) * (Np / 16) + k / 16) * Np + row) * 16 + k % 16]; };
    auto Bat = [&](int p, int t, int64_t k, int64_t w) { return (int)pr.B[((((size_t)p * D + t) * (Np / 16) + k / 16) * W + w) * 16 + k % 16]; };
    for (int p = 0; p < pr.P; ++p)
        for (int64_t blk = 0; blk < Np / 64; ++blk)
            for (int64_t w = 0; w < W; w += 7) {                          // a sample of the walkers
                // the kernel's order: per lane half h (rows 4 h + ...), m-tile i, register r; then half 0 + half 1
                double half[2] = {0, 0};
                for (int h = 0; h < 2; ++h)
                    for (int i = 0; i < 2; ++i)
                        for (int r = 0; r < 16; ++r) {
                            const int64_t row = blk * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                            long long lev[NLEV] = {0};
                            for (int ta = 0; ta < D; ++ta) for (int tb = 0; tb < D; ++tb) {
                                if (ta + tb < LOW) continue;
                                long long sacc = 0;
                                for (int64_t k = 0; k <= row; ++k) sacc += (long long)Aat(p, ta, row, k) * Bat(p, tb, k, w);
                                lev[ta + tb - LOW] += sacc;
                            }
                            double t = (double)lev[0];
                            for (int l = 1; l < NLEV; ++l) t = fma(t, 1.0 / 256.0, (double)lev[l]);
                            const double v = t * pr.rs[(size_t)p * Np + row] * pr.cs[p];
                            half[h] = fma(v, v, half[h]);
                        }
                const double ref = half[0] + half[1], got = sp[((size_t)blk * pr.P + p) * W + w];
                if (ref != got) { ++bad; worst = std::max(worst, fabs(got - ref) / fabs(ref)); }
            }
    printf("check D=%d LOW=%d tile %dx%d: %s (%d of the sampled sums differ, worst %.2e)\n", D, LOW, G::BM, G::BN, bad ? "MISMATCH" : "bit-exact", bad, worst);
    free_problem(pr);
    return bad;
}

template <int D, int LOW, int MODE, bool PF, int WTN = 1, int POS = 0>
static double run8(Problem& pr, int map, int RG, int CG, int reps, const char* name, bool print = true) {
    typedef Geo<D, 2, WTN> G;
    const int nI = (int)(pr.Np / G::BM), nW = (int)(pr.W / G::BN);
    std::vector<TileRec> tiles = make_tiles(pr.P, nI, nW, map, RG, CG);
    TileRec* dt; CK(hipMalloc(&dt, tiles.size() * sizeof(TileRec)));
    CK(hipMemcpy(dt, tiles.data(), tiles.size() * sizeof(TileRec), hipMemcpyHostToDevice));
    auto kern = k_sliced8<D, LOW, MODE, PF, WTN, POS>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    unsigned long long* dst; CK(hipMalloc(&dst, tiles.size() * 24)); CK(hipMemset(dst, 0, tiles.size() * 24));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles.size()), dim3(512), G::LDS_BYTES, 0, pr.dA, pr.dB, pr.drs, pr.dcs, pr.dsp, dt, pr.Np, pr.W, pr.P, dst);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles.size()), dim3(512), G::LDS_BYTES, 0, pr.dA, pr.dB, pr.drs, pr.dcs, pr.dsp, dt, pr.Np, pr.W, pr.P, dst);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> st(tiles.size() * 3);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost)); CK(hipFree(dst));
    std::vector<double> clk;
    double busy = 0, stl = 0;
    for (size_t i = 0; i < tiles.size(); ++i) {
        if (st[3 * i + 1] > 200) clk.push_back((double)st[3 * i] / (double)st[3 * i + 1] * 0.1);   // GHz
        busy += (double)st[3 * i]; stl += (double)st[3 * i + 2];
    }
    std::sort(clk.begin(), clk.end());
    const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    constexpr int NPROD = (D * D) - (LOW * (LOW + 1)) / 2;
    double ksteps = 0; for (auto& t : tiles) ksteps += (double)(t.ib + 1) * G::BM / 32;
    const double ops = ksteps * G::BM * G::BN * 32.0 * 2 * NPROD;
    const double alg = (double)pr.P * pr.Np * pr.Np * pr.W;
    if (print)
        printf("%-34s D=%d prod=%2d 8 waves %3dx%-3d %s map=%d(%dx%d) lds=%3d KB  %8.4f ms  %7.1f TOP/s int8  %6.1f TF/s fp64-equivalent  (x%.2f of 1.311 ms)  %.2f GHz  wait+barrier %.0f%% of block time, blocks fill %.0f%% of CU time\n",
               name, D, NPROD, G::BM, G::BN, PF ? "prefetch" : "in-step ", map, RG, CG, G::LDS_BYTES / 1024, ms, ops / (ms * 1e-3) / 1e12, alg / (ms * 1e-3) / 1e12, 1.311 / ms, ghz, 100.0 * stl / busy, 100.0 * busy / (ghz * 1e9 * ms * 1e-3 * 256));
    CK(hipFree(dt));
    return ms;
}

template <int D, int LOW, bool PF, int WTN = 1, int POS = 0>
static int check8() {
    constexpr int NLEV = 2 * D - 1 - LOW;
    Problem pr; make_problem(pr, 2, 256, 256, D, true);
    run8<D, LOW, 7, PF, WTN, POS>(pr, 0, 1, 1, 1, "check", false);
    std::vector<double> sp((size_t)(pr.Np / 64) * pr.P * pr.W);
    CK(hipMemcpy(sp.data(), pr.dsp, sp.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0; double worst = 0;
    const int64_t Np = pr.Np, W = pr.W;
    auto Aat = [&](int p, int t, int64_t row, int64_t k) { return (int)pr.A[((((size_t)p * D + t) * (Np / 16) + k / 16) * Np + row) * 16 + k % 16]; };
    auto Bat = [&](int p, int t, int64_t k, int64_t w) { return (int)pr.B[((((size_t)p * D + t) * (Np / 16) + k / 16) * W + w) * 16 + k % 16]; };
    for (int p = 0; p < pr.P; ++p)
        for (int64_t blk = 0; blk < Np / 64; ++blk)
            for (int64_t w = 0; w < W; w += 7) {
                // the kernel's order: per 32-row wave tile i: lane half h, register r; half 0 + half 1; then tile 0 + tile 1
                double tile[2];
                for (int i = 0; i < 2; ++i) {
                    double half[2] = {0, 0};
                    for (int h = 0; h < 2; ++h)
                        for (int r = 0; r < 16; ++r) {
                            const int64_t row = blk * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                            long long lev[NLEV] = {0};
                            for (int ta = 0; ta < D; ++ta) for (int tb = 0; tb < D; ++tb) {
                                if (ta + tb < LOW) continue;
                                long long sacc = 0;
                                for (int64_t k = 0; k <= row; ++k) sacc += (long long)Aat(p, ta, row, k) * Bat(p, tb, k, w);
                                lev[ta + tb - LOW] += sacc;
                            }
                            double t = (double)lev[0];
                            for (int l = 1; l < NLEV; ++l) t = fma(t, 1.0 / 256.0, (double)lev[l]);
                            const double v = t * pr.rs[(size_t)p * Np + row] * pr.cs[p];
                            half[h] = fma(v, v, half[h]);
                        }
                    tile[i] = half[0] + half[1];
                }
                const double ref = tile[0] + tile[1], got = sp[((size_t)blk * pr.P + p) * W + w];
                if (ref != got) { ++bad; worst = std::max(worst, fabs(got - ref) / fabs(ref)); }
            }
    printf("check D=%d LOW=%d 8 waves 128x%d %s: %s (%d of the sampled sums differ, worst %.2e)\n", D, LOW, 64 * WTN, PF ? "prefetch" : "in-step", bad ? "MISMATCH" : "bit-exact", bad, worst);
    free_problem(pr);
    return bad;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const bool all = argc > 2;
    int bad = 0;
    bad += check<7, 6, 2, 1>();
    bad += check<6, 5, 2, 1>();
    bad += check8<7, 6, true>();
    bad += check8<6, 5, false, 2>();
    bad += check8<6, 5, false, 2, 8>();
    if (bad) { printf("layout or loop error: timings not taken\n"); return 1; }
    if (all) {
        Problem pr; make_problem(pr, 10, 2048, 2048, 7, false);
        run<7, 6, 2, 1, 7>(pr, 0, 1, 1, reps, "D7 28prod full loop map0");
        run<7, 6, 2, 1, 7>(pr, 1, 8, 4, reps, "D7 28prod full loop superblk 8x4");
        run<7, 6, 2, 1, 7, false>(pr, 1, 4, 8, reps, "D7 28prod full, compiler's order");
        run<7, 6, 2, 1, 4>(pr, 1, 4, 8, reps, "D7 28prod MFMA only");
        run<7, 6, 2, 1, 6>(pr, 1, 4, 8, reps, "D7 28prod MFMA + LDS reads");
        run<7, 6, 2, 1, 1>(pr, 1, 4, 8, reps, "D7 DMA only");
        run<7, 6, 2, 1, 1>(pr, 0, 1, 1, reps, "D7 DMA only map0");
        run8<7, 6, 7, true>(pr, 1, 4, 8, reps, "D7 8w full superblk 4x8");
        run8<7, 6, 7, false>(pr, 1, 4, 8, reps, "D7 8w full superblk 4x8");
        run8<7, 6, 4, true>(pr, 1, 4, 8, reps, "D7 8w MFMA only (random regs)");
        free_problem(pr);
    }
    {
        Problem pr; make_problem(pr, 10, 2048, 2048, 6, false);
        if (all) {
            run<6, 5, 2, 1, 7>(pr, 1, 2, 16, reps, "D6 21prod 128x64 superblk 2x16");
            run<6, 5, 2, 1, 4>(pr, 1, 4, 8, reps, "D6 21prod 128x64 MFMA only");
            run<6, 5, 2, 1, 1>(pr, 1, 4, 8, reps, "D6 128x64 DMA only");
            run8<6, 5, 7, false>(pr, 1, 4, 8, reps, "D6 8w full superblk 4x8");
        }
        run8<6, 5, 7, false, 2, 0>(pr, 1, 8, 4, reps, "D6 8w 128x128 DMA first (pos 0)");
        run8<6, 5, 7, false, 2, 1>(pr, 1, 8, 4, reps, "D6 8w 128x128 DMA behind reads (pos 1)");
        run8<6, 5, 7, false, 2, 2>(pr, 1, 8, 4, reps, "D6 8w 128x128 DMA mid-step (pos 2)");
        run8<6, 5, 7, false, 2, 3>(pr, 1, 8, 4, reps, "D6 8w 128x128 DMA staggered (pos 3)");
        run8<6, 5, 7, false, 2, 4>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 4 (early / end)");
        run8<6, 5, 7, false, 2, 5>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 5 (mid / end)");
        run8<6, 5, 7, false, 2, 6>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 6 (3 + setprio)");
        run8<6, 5, 7, false, 2, 7>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 7 (all at the end)");
        run8<6, 5, 7, false, 2, 8>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 8 (reads by hand)");
        run8<6, 5, 7, false, 2, 7>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 7 again");
        run8<6, 5, 7, false, 2, 8>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 8 again");
        run8<6, 5, 7, false, 2, 3>(pr, 1, 4, 8, reps, "D6 8w 128x128 pos 3 superblk 4x8");
        run8<6, 5, 4, false, 2>(pr, 1, 8, 4, reps, "D6 8w 128x128 MFMA only (random regs)");
        run8<6, 5, 1, false, 2>(pr, 1, 8, 4, reps, "D6 8w 128x128 DMA only");
        CK(hipMemset(pr.dA, 0, (size_t)pr.P * pr.D * pr.Np * pr.Np)); CK(hipMemset(pr.dB, 0, (size_t)pr.P * pr.D * pr.Np * pr.W));
        run8<6, 5, 7, false, 2, 0>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 0 on ZERO planes");
        run8<6, 5, 7, false, 2, 3>(pr, 1, 8, 4, reps, "D6 8w 128x128 pos 3 on ZERO planes");
        free_problem(pr);
    }
    return 0;
}
