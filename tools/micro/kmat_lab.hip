// kmat_lab — variants of the K(X,X) assembly kernel (k_kmat_mfma, gpb_fit.hip) side by side on synthetic operands, timed
// with HIP events; the product kernel is V0.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/kmat_lab kmat_lab.hip
// usage: kmat_lab [N=2048] [kind=0|1|2] [P=10] [reps=20]
//   V0  one 64x64 tile per workgroup (the product kernel's body)
//   V1  V0 without the stores          (diagnostic)
//   V2  V0 without the shape function  (diagnostic: stores r^2)
//   V3  strips: a workgroup keeps its row block's operand and walks TJ column blocks, the next block's rows in flight
//   V4  V3 with the table-driven exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../gpbayestools_hic_amd/csrc/fast_math.h"

typedef double d4 __attribute__((ext_vector_type(4)));
struct d2 { double x, y; };
using namespace gpb;

template <int KIND, int DPAD, int MODE>
__global__ __launch_bounds__(256) void k_v0(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                            const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np) {
    constexpr int LDX = DPAD + 1;
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
    const int64_t t = blockIdx.x;
    int64_t bi = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int64_t bj = t - bi * (bi + 1) / 2;
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
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
            const d2 vi = gi[e], vj = gj[e];
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
            sXj[r * LDX + k] = vj.x; sXj[r * LDX + k + 1] = vj.y;
        }
        if (tid < 64) sdi[tid] = dn[i0 + tid];
        else if (tid < 128) sdj[tid - 64] = dn[j0 + tid - 64];
    }
    __syncthreads();
    constexpr int KG = DPAD / 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
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
    const bool special = bi == bj || i0 + 64 > N;
    if (!special) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                    const double v = MODE == 2 ? r2 : c * shape_fn_fast<KIND>(r2);
                    if (MODE != 1 || v == 12345.678) Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                double v;
                if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                else if (i == j) v = c + 0.1;
                else v = MODE == 2 ? r2 : c * shape_fn_fast<KIND>(r2);
                if (MODE != 1 || v == 12345.678) Kp[i * Np + j] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- V5: V0 with the tile coordinates from a table and an occupancy request
template <int KIND, int DPAD, int MODE, int WPE>
__global__ __launch_bounds__(256, WPE) void k_v5(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                            const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                            const int2* __restrict__ tiles) {
    constexpr int LDX = DPAD + 1;
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
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
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
            d2 vi, vj;
            if (MODE == 3) { vi = d2{1e-3 * e, 1.0}; vj = d2{0.5, 2e-3 * e}; }      // MODE 3: no global loads
            else { vi = gi[e]; vj = gj[e]; }
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
            sXj[r * LDX + k] = vj.x; sXj[r * LDX + k + 1] = vj.y;
        }
        if (MODE == 3) { if (tid < 64) sdi[tid] = tid; else if (tid < 128) sdj[tid - 64] = tid; }
        else if (tid < 64) sdi[tid] = dn[i0 + tid];
        else if (tid < 128) sdj[tid - 64] = dn[j0 + tid - 64];
    }
    __syncthreads();
    constexpr int KG = DPAD / 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
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
            for (int b = 0; b < 2; ++b) {
                if (MODE == 4) { if (g == 0) acc[a][b] = d4{fa[a], fb[b], fa[a] + fb[b], fa[a] - fb[b]}; }      // MODE 4: no MFMA
                else acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
    }
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    const bool special = bi == bj || i0 + 64 > N;
    if (!special) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                    const double v = (MODE >= 2) ? r2 : c * shape_fn_fast<KIND>(r2);
                    if (MODE != 1 || v == 12345.678) Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                double v;
                if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                else if (i == j) v = c + 0.1;
                else v = (MODE >= 2) ? r2 : c * shape_fn_fast<KIND>(r2);
                if (MODE != 1 || v == 12345.678) Kp[i * Np + j] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- V9: V5 with non-temporal stores (MODE 0), non-temporal operand loads (MODE 5) or both (MODE 6)
template <int KIND, int DPAD, int MODE, int WPE>
__global__ __launch_bounds__(256, WPE) void k_v9(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                            const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                            const int2* __restrict__ tiles) {
    constexpr int LDX = DPAD + 1;
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
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
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
            d2 vi, vj;
            if (MODE == 3) { vi = d2{1e-3 * e, 1.0}; vj = d2{0.5, 2e-3 * e}; }      // MODE 3: no global loads
            else if (MODE >= 5) {
                vi.x = __builtin_nontemporal_load(&gi[e].x); vi.y = __builtin_nontemporal_load(&gi[e].y);
                vj.x = __builtin_nontemporal_load(&gj[e].x); vj.y = __builtin_nontemporal_load(&gj[e].y);
            } else { vi = gi[e]; vj = gj[e]; }
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
            sXj[r * LDX + k] = vj.x; sXj[r * LDX + k + 1] = vj.y;
        }
        if (MODE == 3) { if (tid < 64) sdi[tid] = tid; else if (tid < 128) sdj[tid - 64] = tid; }
        else if (tid < 64) sdi[tid] = dn[i0 + tid];
        else if (tid < 128) sdj[tid - 64] = dn[j0 + tid - 64];
    }
    __syncthreads();
    constexpr int KG = DPAD / 4;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
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
            for (int b = 0; b < 2; ++b) {
                if (MODE == 4) { if (g == 0) acc[a][b] = d4{fa[a], fb[b], fa[a] + fb[b], fa[a] - fb[b]}; }      // MODE 4: no MFMA
                else acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
    }
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    const bool special = bi == bj || i0 + 64 > N;
    if (!special) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                    const double v = (MODE >= 2) ? r2 : c * shape_fn_fast<KIND>(r2);
                    if (MODE != 5) __builtin_nontemporal_store(v, &Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr]);
                    else Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                double v;
                if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                else if (i == j) v = c + 0.1;
                else v = (MODE >= 2) ? r2 : c * shape_fn_fast<KIND>(r2);
                if (MODE != 1 || v == 12345.678) Kp[i * Np + j] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- strips ---------------------------------------------------------------------------------------------------------
// chunk c of the lower block triangle: row block bi, column blocks [TJ q, min(TJ q + TJ, bi + 1)); chunks numbered row by row
template <int KIND, int DPAD, int GROUP>
__global__ __launch_bounds__(256) void k_v3(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                            const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                            int TJ, const int2* __restrict__ chunks) {
    constexpr int LDX = DPAD + 1;
    constexpr int KG = DPAD / 4;
    constexpr int NPRE = (64 * DPAD / 2 + 255) / 256;
    __shared__ double sXi[64 * LDX], sXj[2][64 * LDX], sdi[64], sdj[2][64];
    const int p = blockIdx.y;
    const int2 ch = chunks[blockIdx.x];
    const int bi = ch.x, bj0 = ch.y, nt = min(TJ, bi + 1 - bj0);
    const int64_t i0 = (int64_t)bi * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
    const double* Xp = Xc + (int64_t)p * Np * DPAD;
    const double* dn = dnorm + (int64_t)p * Np;
    d2 pre[NPRE];
    double pdn = 0.0;
    auto fetch = [&](int64_t j0) {
        const d2* gj = reinterpret_cast<const d2*>(Xp + j0 * DPAD);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = tid + 256 * u;
            if (e < 64 * DPAD / 2) pre[u] = gj[e];
        }
        if (tid < 64) pdn = dn[j0 + tid];
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = tid + 256 * u;
            if (e < 64 * DPAD / 2) {
                const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
                sXj[buf][r * LDX + k] = pre[u].x; sXj[buf][r * LDX + k + 1] = pre[u].y;
            }
        }
        if (tid < 64) sdj[buf][tid] = pdn;
    };
    fetch((int64_t)bj0 * 64);
    {
        const d2* gi = reinterpret_cast<const d2*>(Xp + i0 * DPAD);
#pragma unroll
        for (int e = tid; e < 64 * DPAD / 2; e += 256) {
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
            const d2 vi = gi[e];
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
        }
        if (tid >= 64 && tid < 128) sdi[tid - 64] = dn[i0 + tid - 64];
    }
    stash(0);
    __syncthreads();
    double fa[KG][2];
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[g][a] = sXi[(m0 + 16 * a + lr) * LDX + 4 * g + lk];
    double di[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) di[a][r] = sdi[m0 + 16 * a + lk + 4 * r];
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    for (int jt = 0; jt < nt; ++jt) {
        const int bj = bj0 + jt, buf = jt & 1;
        const int64_t j0 = (int64_t)bj * 64;
        if (jt + 1 < nt) fetch(j0 + 64);
        d4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            double fb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = sXj[buf][(n0 + 16 * b + lr) * LDX + 4 * g + lk];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[g][a], fb[b], acc[a][b], 0, 0, 0);
        }
        const bool special = bi == bj || i0 + 64 > N;
        if (!special) {
            if (GROUP == 4) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const double dj = sdj[buf][n0 + 16 * b + lr];
                        double r2[4], sh[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) r2[r] = fmax(fma(-2.0, acc[a][b][r], di[a][r] + dj), 0.0);
                        shape_fn_fast_n<KIND, 4>(r2, sh);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = c * sh[r];
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const double dj = sdj[buf][n0 + 16 * b + lr];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double r2 = fmax(fma(-2.0, acc[a][b][r], di[a][r] + dj), 0.0);
                            Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = c * shape_fn_fast<KIND>(r2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double dj = sdj[buf][n0 + 16 * b + lr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], di[a][r] + dj), 0.0);
                        double v;
                        if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                        else if (i == j) v = c + 0.1;
                        else v = c * shape_fn_fast<KIND>(r2);
                        Kp[i * Np + j] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (jt + 1 < nt) {
            stash(buf ^ 1);
            __syncthreads();
        }
    }
}


// ---- store-only kernels: what the tile's write pattern alone costs ---------------------------------------------------
//   PAT 0: the MFMA output layout — one store instruction covers 4 rows x 16 columns (4 x 128 B)
//   PAT 1: one row x 64 columns per instruction (512 B contiguous), 8-byte stores
//   PAT 2: two rows x 64 columns per instruction, 16-byte stores (a lane holds two adjacent columns)
//   PAT 3: PAT 2 over a 32-row x 128-column tile (1 KB contiguous per row)
template <int PAT>
__global__ __launch_bounds__(256) void k_store(double* __restrict__ K, int64_t Np) {
    const int p = blockIdx.y;
    const int64_t t = blockIdx.x;
    int64_t bi = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    const int64_t bj = t - bi * (bi + 1) / 2;
    const int64_t i0 = bi * 64, j0 = bj * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Kp = K + (int64_t)p * Np * Np;
    const double v = 1.0 + 1e-9 * tid;
    if (PAT == 0) {
        const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = v + r;
    } else if (PAT == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Kp[(i0 + wave * 16 + r) * Np + j0 + lane] = v + r;
    } else if (PAT == 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            d2 w{v + r, v - r};
            *reinterpret_cast<d2*>(Kp + (i0 + wave * 16 + 2 * r + (lane >> 5)) * Np + j0 + 2 * (lane & 31)) = w;
        }
    } else {
        // 32 x 128: tile t covers rows [32 (2 bi + h)), columns 128 (bj / 2) ... only a bandwidth probe: writes the same bytes
        const int64_t ii = i0 + 32 * (bj & 1), jj = (bj >> 1) * 128;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            d2 w{v + r, v - r};
            *reinterpret_cast<d2*>(Kp + (ii + wave * 8 + r) * Np + jj + 2 * lane) = w;
        }
    }
}

// ---- V6: persistent workgroups; the operands of a workgroup's NEXT tile are loaded into registers before the current tile's
// stores are issued (a workgroup's loads queue behind the CU's stores: V5m3 above — no loads — runs at the store bound)
template <int KIND, int DPAD, int WPE>
__global__ __launch_bounds__(256, WPE) void k_v6(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                                 const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                                 const int2* __restrict__ tiles, int ntile) {
    constexpr int LDX = DPAD + 1;
    constexpr int KG = DPAD / 4;
    constexpr int NPRE = (64 * DPAD / 2 + 255) / 256;
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
    const double* Xp = Xc + (int64_t)p * Np * DPAD;
    const double* dn = dnorm + (int64_t)p * Np;
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    d2 pi[NPRE], pj[NPRE];
    double pdn = 0.0;
    auto fetch = [&](int t) {
        const int2 tl = tiles[t];
        const d2* gi = reinterpret_cast<const d2*>(Xp + (int64_t)tl.x * 64 * DPAD);
        const d2* gj = reinterpret_cast<const d2*>(Xp + (int64_t)tl.y * 64 * DPAD);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = tid + 256 * u;
            if (e < 64 * DPAD / 2) { pi[u] = gi[e]; pj[u] = gj[e]; }
        }
        if (tid < 64) pdn = dn[(int64_t)tl.x * 64 + tid];
        else if (tid < 128) pdn = dn[(int64_t)tl.y * 64 + tid - 64];
    };
    int t = blockIdx.x;
    if (t >= ntile) return;
    fetch(t);
    for (; t < ntile; t += gridDim.x) {
        const int2 tl = tiles[t];
        const int64_t bi = tl.x, bj = tl.y, i0 = bi * 64, j0 = bj * 64;
        __syncthreads();                                // the previous tile's fragment reads are done
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int e = tid + 256 * u;
            if (e < 64 * DPAD / 2) {
                const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
                sXi[r * LDX + k] = pi[u].x; sXi[r * LDX + k + 1] = pi[u].y;
                sXj[r * LDX + k] = pj[u].x; sXj[r * LDX + k + 1] = pj[u].y;
            }
        }
        if (tid < 64) sdi[tid] = pdn;
        else if (tid < 128) sdj[tid - 64] = pdn;
        __syncthreads();
        if (t + (int)gridDim.x < ntile) fetch(t + gridDim.x);
        d4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
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
        const bool special = bi == bj || i0 + 64 > N;
        if (!special) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                        Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = c * shape_fn_fast<KIND>(r2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double dj = sdj[n0 + 16 * b + lr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], sdi[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                        double v;
                        if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                        else if (i == j) v = c + 0.1;
                        else v = c * shape_fn_fast<KIND>(r2);
                        Kp[i * Np + j] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
}

// ---- V7: persistent workgroups of FIVE waves: four compute (fragment reads, MFMA, shape function, stores) and one loader that
// never stores to global memory.  gfx9's vmcnt counts loads and stores together and the two complete out of order, so a wave
// with stores in flight that needs loaded data waits for its stores to be written (the compiler's s_waitcnt vmcnt(0) in V6);
// the loader's counter holds loads only.  It keeps the NEXT tile's two operand blocks in registers (a whole tile period for
// them to arrive behind the CU's queued stores) and writes them to LDS between the two barriers of a tile.
typedef double d2v __attribute__((ext_vector_type(2)));
template <int KIND, int DPAD, int WPE, int MODE = 0>
__global__ __launch_bounds__(320, WPE) void k_v7(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                                 const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                                 const int2* __restrict__ tiles, int ntile) {
    constexpr int LDX = DPAD + 1;
    constexpr int KG = DPAD / 4;
    constexpr int NLD = DPAD / 4;                       // 2 KB (64 lanes x 4 doubles) loads per operand block of 64 x DPAD doubles
    __shared__ double sXi[64 * LDX], sXj[64 * LDX], sdi[64], sdj[64];
    const int p = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Xp = Xc + (int64_t)p * Np * DPAD;
    const double* dn = dnorm + (int64_t)p * Np;
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntile) return;
    if (wave == 4) {
        // ---------------- loader
        d4 ri[NLD], rj[NLD];
        double di = 0.0, dj = 0.0;
        auto fetch = [&](int tt) {
            const int2 tl = tiles[tt];
            const d4* gi = reinterpret_cast<const d4*>(Xp + (int64_t)tl.x * 64 * DPAD);
            const d4* gj = reinterpret_cast<const d4*>(Xp + (int64_t)tl.y * 64 * DPAD);
#pragma unroll
            for (int u = 0; u < NLD; ++u) { ri[u] = gi[lane + 64 * u]; rj[u] = gj[lane + 64 * u]; }
            di = dn[(int64_t)tl.x * 64 + lane];
            dj = dn[(int64_t)tl.y * 64 + lane];
        };
        auto stash = [&]() {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int q = 4 * (lane + 64 * u), r = q / DPAD, k = q - r * DPAD;
                double* a = sXi + r * LDX + k;
                double* b = sXj + r * LDX + k;
                a[0] = ri[u][0]; a[1] = ri[u][1]; a[2] = ri[u][2]; a[3] = ri[u][3];
                b[0] = rj[u][0]; b[1] = rj[u][1]; b[2] = rj[u][2]; b[3] = rj[u][3];
            }
            sdi[lane] = di; sdj[lane] = dj;
        };
        fetch(t);
        stash();                                        // (waits for the first tile's loads: the one exposed latency)
        if (t + G < ntile) fetch(t + G);
        __syncthreads();                                // B of the prologue: tile t is in LDS
        for (; t < ntile; t += G) {
            __syncthreads();                            // A: the compute waves have read tile t's fragments
            if (t + G < ntile) {
                stash();                                // tile t + G (loaded during the previous tile's period)
                if (t + 2 * G < ntile) fetch(t + 2 * G);
            }
            __syncthreads();                            // B
        }
        return;
    }
    // -------------------- compute waves
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
    __syncthreads();                                    // B of the prologue
    for (; t < ntile; t += G) {
        const int2 tl = tiles[t];
        const int64_t bi = tl.x, bj = tl.y, i0 = bi * 64, j0 = bj * 64;
        d4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
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
        double vi[2][4], vj[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            vj[a] = sdj[n0 + 16 * a + lr];
#pragma unroll
            for (int r = 0; r < 4; ++r) vi[a][r] = sdi[m0 + 16 * a + lk + 4 * r];
        }
        __syncthreads();                                // A: LDS may be overwritten with the next tile
        const bool special = bi == bj || i0 + 64 > N;
        if (!special) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], vi[a][r] + vj[b]), 0.0);
                        Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = MODE == 2 ? r2 : c * shape_fn_fast<KIND>(r2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], vi[a][r] + vj[b]), 0.0);
                        double v;
                        if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                        else if (i == j) v = c + 0.1;
                        else v = MODE == 2 ? r2 : c * shape_fn_fast<KIND>(r2);
                        Kp[i * Np + j] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();                                // B: the next tile is in LDS
    }
}

// ---- V8: a 128 x 128 macro tile per workgroup: both operand blocks (128 rows each) staged once, the four 64 x 64 sub-tiles one
// after the other (three on the diagonal) — half the operand loads per pair of V5
template <int KIND, int DPAD, int WPE>
__global__ __launch_bounds__(256, WPE) void k_v8(const double* __restrict__ Xc, const double* __restrict__ dnorm,
                                                 const double* __restrict__ amp, double* __restrict__ K, int64_t N, int64_t Np,
                                                 const int2* __restrict__ tiles) {
    constexpr int LDX = DPAD + 1;
    __shared__ double sXi[128 * LDX], sXj[128 * LDX], sdi[128], sdj[128];
    const int p = blockIdx.y;
    const int2 tl = tiles[blockIdx.x];
    const int64_t Bi = tl.x, Bj = tl.y;                 // 128-row blocks, Bj <= Bi
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
    const double* Xp = Xc + (int64_t)p * Np * DPAD;
    const double* dn = dnorm + (int64_t)p * Np;
    {
        const d2* gi = reinterpret_cast<const d2*>(Xp + Bi * 128 * DPAD);
        const d2* gj = reinterpret_cast<const d2*>(Xp + Bj * 128 * DPAD);
#pragma unroll
        for (int e = tid; e < 128 * DPAD / 2; e += 256) {
            const int r = (2 * e) / DPAD, k = 2 * e - r * DPAD;
            const d2 vi = gi[e], vj = gj[e];
            sXi[r * LDX + k] = vi.x; sXi[r * LDX + k + 1] = vi.y;
            sXj[r * LDX + k] = vj.x; sXj[r * LDX + k + 1] = vj.y;
        }
        if (tid < 128) sdi[tid] = dn[Bi * 128 + tid];
        else sdj[tid - 128] = dn[Bj * 128 + tid - 128];
    }
    __syncthreads();
    constexpr int KG = DPAD / 4;
    const double c = amp[p];
    double* Kp = K + (int64_t)p * Np * Np;
#pragma unroll 1
    for (int sub = 0; sub < 4; ++sub) {
        const int si = sub >> 1, sj = sub & 1;
        const int64_t bi = 2 * Bi + si, bj = 2 * Bj + sj;
        if (bj > bi) continue;                          // above the diagonal: nothing reads it
        const int64_t i0 = bi * 64, j0 = bj * 64;
        const double* xi = sXi + si * 64 * LDX;
        const double* xj = sXj + sj * 64 * LDX;
        d4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            double fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                fa[a] = xi[(m0 + 16 * a + lr) * LDX + 4 * g + lk];
                fb[a] = xj[(n0 + 16 * a + lr) * LDX + 4 * g + lk];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        const double* di = sdi + si * 64;
        const double* dj_ = sdj + sj * 64;
        const bool special = bi == bj || i0 + 64 > N;
        if (!special) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double dj = dj_[n0 + 16 * b + lr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], di[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                        Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = c * shape_fn_fast<KIND>(r2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const double dj = dj_[n0 + 16 * b + lr];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t i = i0 + m0 + 16 * a + lk + 4 * r, j = j0 + n0 + 16 * b + lr;
                        const double r2 = fmax(fma(-2.0, acc[a][b][r], di[m0 + 16 * a + lk + 4 * r] + dj), 0.0);
                        double v;
                        if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;
                        else if (i == j) v = c + 0.1;
                        else v = c * shape_fn_fast<KIND>(r2);
                        Kp[i * Np + j] = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
}

// ---- store-only with the stores spread in time, as a kernel that computes between them does: after every group of four store
// instructions a wave sleeps ~SLEEP x 64 cycles.  PAT 0: MFMA layout over 64 x 64; PAT 3: 32 x 128 tile, 1 KB row pieces;
// PAT 4: 16 x 256 tile, 2 KB row pieces (each wave: 4 rows)
template <int PAT, int SLEEP>
__global__ __launch_bounds__(256) void k_store_d(double* __restrict__ K, int64_t Np, const int2* __restrict__ tiles) {
    __shared__ double pad[2816];                      // the footprint of the real kernel: 22.5 KB of LDS per workgroup
    const int p = blockIdx.y;
    const int2 tl = tiles[blockIdx.x];
    const int64_t bi = tl.x, bj = tl.y;
    const int64_t i0 = bi * 64, j0 = bj * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Kp = K + (int64_t)p * Np * Np;
    const double v = 1.0 + 1e-9 * tid;
    if (v == 77.0) pad[tid] = v;
    if (PAT == 0) {
        const int m0 = (wave >> 1) * 32, n0 = (wave & 1) * 32, lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Kp[(i0 + m0 + 16 * a + lk + 4 * r) * Np + j0 + n0 + 16 * b + lr] = v + r;
                if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            }
    } else if (PAT == 3) {
        const int64_t ii = i0 + 32 * (bj & 1), jj = (bj >> 1) * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                d2 w{v + r, v - r};
                *reinterpret_cast<d2*>(Kp + (ii + wave * 8 + 2 * g + r) * Np + jj + 2 * lane) = w;
            }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
    } else {
        const int64_t ii = i0 + 16 * (bj & 3), jj = (bj >> 2) * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                d2 w{v + g, v - h};
                *reinterpret_cast<d2*>(Kp + (ii + wave * 4 + g) * Np + jj + 128 * h + 2 * lane) = w;
            }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
}

static int nchunks(int nb, int TJ) { int s = 0; for (int b = 0; b < nb; ++b) s += (b + TJ) / TJ; return s; }

static const char* g_only = getenv("LAB_ONLY");
static const char* g_name = "";
#define REPORT(name, ...) do { g_name = name; report(name, (g_only && !strstr(name, g_only)) ? -1.f : time_it(__VA_ARGS__)); } while (0)
template <typename F>
static float time_it(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int KIND>
static void run(int64_t N, int P, int reps) {
    constexpr int DP = 20;
    const int64_t Np = (N + 63) / 64 * 64;
    const int nb = (int)(Np / 64);
    std::vector<double> hX((size_t)P * Np * DP, 0.0), hd((size_t)P * Np, 0.0), ha(P);
    srand(1);
    for (int p = 0; p < P; ++p) {
        ha[p] = 0.8 + 0.05 * p;
        for (int64_t i = 0; i < N; ++i) {
            double s = 0.0;
            for (int k = 0; k < DP; ++k) {
                const double v = ((double)rand() / RAND_MAX - 0.5) * 1.2;      // r^2 of a few units, as a fitted design's
                hX[((size_t)p * Np + i) * DP + k] = v; s += v * v;
            }
            hd[(size_t)p * Np + i] = s;
        }
    }
    double *X, *dn, *amp, *K0, *K1;
    hipMalloc(&X, hX.size() * 8); hipMalloc(&dn, hd.size() * 8); hipMalloc(&amp, P * 8);
    hipMalloc(&K0, (size_t)P * Np * Np * 8); hipMalloc(&K1, (size_t)P * Np * Np * 8);
    hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dn, hd.data(), hd.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(amp, ha.data(), P * 8, hipMemcpyHostToDevice);
    hipMemset(K0, 0, (size_t)P * Np * Np * 8); hipMemset(K1, 0, (size_t)P * Np * Np * 8);
    const dim3 g0((unsigned)(nb * (nb + 1) / 2), (unsigned)P);
    const double bytes = 4.0 * Np * Np * P;
    const char* only = getenv("LAB_ONLY");
    auto report = [&](const char* name, float us) {
        if (us < 0) return;
        printf("N %lld kind %d  %-28s %8.1f us   %.3f of the 8 TB/s write bound\n", (long long)N, KIND, name, us, bytes / 8e12 / (us * 1e-6));
    };
    REPORT("V0 tile per workgroup", [&] { hipLaunchKernelGGL((k_v0<KIND, DP, 0>), g0, dim3(256), 0, 0, X, dn, amp, K0, N, Np); }, reps);
    REPORT("V1 V0 without stores", [&] { hipLaunchKernelGGL((k_v0<KIND, DP, 1>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np); }, reps);
    REPORT("V2 V0 without shape fn", [&] { hipLaunchKernelGGL((k_v0<KIND, DP, 2>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np); }, reps);
    REPORT("S0 stores only, MFMA layout", [&] { hipLaunchKernelGGL((k_store<0>), g0, dim3(256), 0, 0, K1, Np); }, reps);
    REPORT("S1 stores only, row x 64", [&] { hipLaunchKernelGGL((k_store<1>), g0, dim3(256), 0, 0, K1, Np); }, reps);
    REPORT("S2 stores only, 2 rows x 64, 16 B", [&] { hipLaunchKernelGGL((k_store<2>), g0, dim3(256), 0, 0, K1, Np); }, reps);
    REPORT("S3 stores only, 32 x 128, 16 B", [&] { hipLaunchKernelGGL((k_store<3>), g0, dim3(256), 0, 0, K1, Np); }, reps);
    if (getenv("LAB_STORES_ONLY")) return;
    std::vector<double> h0((size_t)Np * Np), h1((size_t)Np * Np);
    int2* dtab; hipMalloc(&dtab, sizeof(int2) * (size_t)(nb * (nb + 1) / 2 + 16));
    {
        std::vector<int2> tl;
        for (int bi = 0; bi < nb; ++bi) for (int bj = 0; bj <= bi; ++bj) tl.push_back(int2{bi, bj});
        hipMemcpy(dtab, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice);
    }
    REPORT("D0 stores MFMA layout, no sleep", [&] { hipLaunchKernelGGL((k_store_d<0, 0>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D0 stores MFMA layout, sleep 6", [&] { hipLaunchKernelGGL((k_store_d<0, 6>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D0 stores MFMA layout, sleep 12", [&] { hipLaunchKernelGGL((k_store_d<0, 12>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D3 stores 32x128, sleep 6", [&] { hipLaunchKernelGGL((k_store_d<3, 6>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D3 stores 32x128, sleep 12", [&] { hipLaunchKernelGGL((k_store_d<3, 12>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D4 stores 16x256, sleep 6", [&] { hipLaunchKernelGGL((k_store_d<4, 6>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    REPORT("D4 stores 16x256, sleep 12", [&] { hipLaunchKernelGGL((k_store_d<4, 12>), g0, dim3(256), 0, 0, K1, Np, dtab); }, reps);
    {
        const int ntile = nb * (nb + 1) / 2;
        for (int wpc : {3, 4, 5, 6, 8}) {
            char nm[64];
            const int gx = std::min(ntile, (256 * wpc + P - 1) / P);
            const dim3 g6((unsigned)gx, (unsigned)P);
            hipMemset(K1, 0, (size_t)P * Np * Np * 8);
            snprintf(nm, sizeof nm, "V6 persistent, %d workgroups/CU", wpc);
            REPORT(nm, [&] { hipLaunchKernelGGL((k_v6<KIND, DP, 4>), g6, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab, ntile); }, reps);
            if (g_only && !strstr(nm, g_only)) continue;
            hipMemcpy(h1.data(), K1 + (size_t)(P - 1) * Np * Np, h1.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h0.data(), K0 + (size_t)(P - 1) * Np * Np, h0.size() * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (int64_t i = 0; i < Np; ++i) for (int64_t j = 0; j <= i; ++j) bad += h0[i * Np + j] != h1[i * Np + j];
            printf("      V6 lower-triangle elements that are not V0's: %zu (meaningful when V0 ran in this process)\n", bad);
        }
    }
    {
        const int ntile = nb * (nb + 1) / 2;
        for (int wpc : {2, 3, 4}) {
            char nm[64];
            const int gx = std::min(ntile, (256 * wpc + P - 1) / P);
            const dim3 g7((unsigned)gx, (unsigned)P);
            hipMemset(K1, 0, (size_t)P * Np * Np * 8);
            snprintf(nm, sizeof nm, "V7 loader wave, %d workgroups/CU", wpc);
            REPORT(nm, [&] { hipLaunchKernelGGL((k_v7<KIND, DP, 4>), g7, dim3(320), 0, 0, X, dn, amp, K1, N, Np, dtab, ntile); }, reps);
            snprintf(nm, sizeof nm, "V7m2 loader wave no shape, %d workgroups/CU", wpc);
            REPORT(nm, [&] { hipLaunchKernelGGL((k_v7<KIND, DP, 4, 2>), g7, dim3(320), 0, 0, X, dn, amp, K1, N, Np, dtab, ntile); }, reps);
            snprintf(nm, sizeof nm, "V7 loader wave, %d workgroups/CU", wpc);
            if (g_only && !strstr(nm, g_only)) continue;
            hipMemcpy(h1.data(), K1 + (size_t)(P - 1) * Np * Np, h1.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h0.data(), K0 + (size_t)(P - 1) * Np * Np, h0.size() * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (int64_t i = 0; i < Np; ++i) for (int64_t j = 0; j <= i; ++j) bad += h0[i * Np + j] != h1[i * Np + j];
            printf("      V7 lower-triangle elements that are not V0's: %zu (meaningful when V0 ran in this process)\n", bad);
        }
    }
    if (nb % 2 == 0) {
        const int nb2 = nb / 2;
        std::vector<int2> tl;
        for (int bi = 0; bi < nb2; ++bi) for (int bj = 0; bj <= bi; ++bj) tl.push_back(int2{bi, bj});
        int2* dtab2; hipMalloc(&dtab2, tl.size() * sizeof(int2));
        hipMemcpy(dtab2, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice);
        const dim3 g8((unsigned)tl.size(), (unsigned)P);
        hipMemset(K1, 0, (size_t)P * Np * Np * 8);
        REPORT("V8 128x128 macro tile", [&] { hipLaunchKernelGGL((k_v8<KIND, DP, 3>), g8, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab2); }, reps);
        if (!g_only) {
            hipMemcpy(h1.data(), K1 + (size_t)(P - 1) * Np * Np, h1.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(h0.data(), K0 + (size_t)(P - 1) * Np * Np, h0.size() * 8, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (int64_t i = 0; i < Np; ++i) for (int64_t j = 0; j <= i; ++j) if ((i / 64) >= (j / 64)) bad += h0[i * Np + j] != h1[i * Np + j];
            printf("      V8 lower-triangle elements that are not V0's: %zu\n", bad);
        }
        hipFree(dtab2);
    }
    REPORT("V9 non-temporal stores", [&] { hipLaunchKernelGGL((k_v9<KIND, DP, 0, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V9m5 non-temporal loads", [&] { hipLaunchKernelGGL((k_v9<KIND, DP, 5, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V9m6 non-temporal loads and stores", [&] { hipLaunchKernelGGL((k_v9<KIND, DP, 6, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5m3 no loads, no shape (MFMA + stores)", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 3, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5m4 no MFMA, no shape (loads + stores)", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 4, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5 no shape fn (MFMA + stores)", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 2, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5 table, 5 waves/SIMD", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 0, 5>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5 table, 6 waves/SIMD", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 0, 6>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5 table, 8 waves/SIMD", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 0, 8>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    REPORT("V5 table, 8 w/SIMD, no stores", [&] { hipLaunchKernelGGL((k_v5<KIND, DP, 1, 8>), g0, dim3(256), 0, 0, X, dn, amp, K1, N, Np, dtab); }, reps);
    hipMemcpy(h0.data(), K0 + (size_t)(P - 1) * Np * Np, h0.size() * 8, hipMemcpyDeviceToHost);
    for (int TJ : {1, 2, 3, 4, 6, 8}) {
        const dim3 g3((unsigned)nchunks(nb, TJ), (unsigned)P);
        {
            std::vector<int2> ch;
            for (int bi = 0; bi < nb; ++bi) for (int q = 0; q * TJ <= bi; ++q) ch.push_back(int2{bi, q * TJ});
            hipMemcpy(dtab, ch.data(), ch.size() * sizeof(int2), hipMemcpyHostToDevice);
        }
        char nm[64];
        hipMemset(K1, 0, (size_t)P * Np * Np * 8);
        snprintf(nm, sizeof nm, "V3 strips TJ=%d", TJ);
        REPORT(nm, [&] { hipLaunchKernelGGL((k_v3<KIND, DP, 1>), g3, dim3(256), 0, 0, X, dn, amp, K1, N, Np, TJ, dtab); }, reps);
        hipMemcpy(h1.data(), K1 + (size_t)(P - 1) * Np * Np, h1.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int64_t i = 0; i < Np; ++i) for (int64_t j = 0; j <= i; ++j) bad += h0[i * Np + j] != h1[i * Np + j];
        snprintf(nm, sizeof nm, "V3 strips TJ=%d interleaved", TJ);
        REPORT(nm, [&] { hipLaunchKernelGGL((k_v3<KIND, DP, 4>), g3, dim3(256), 0, 0, X, dn, amp, K1, N, Np, TJ, dtab); }, reps);
        hipMemcpy(h1.data(), K1 + (size_t)(P - 1) * Np * Np, h1.size() * 8, hipMemcpyDeviceToHost);
        size_t bad4 = 0;
        for (int64_t i = 0; i < Np; ++i) for (int64_t j = 0; j <= i; ++j) bad4 += h0[i * Np + j] != h1[i * Np + j];
        printf("      lower-triangle elements differing from V0: %zu, %zu\n", bad, bad4);
    }
    hipFree(X); hipFree(dn); hipFree(amp); hipFree(K0); hipFree(K1);
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 2048;
    const int kind = argc > 2 ? atoi(argv[2]) : 0, P = argc > 3 ? atoi(argv[3]) : 10, reps = argc > 4 ? atoi(argv[4]) : 20;
    if (kind == 0) run<0>(N, P, reps);
    else if (kind == 1) run<1>(N, P, reps);
    else run<2>(N, P, reps);
    return 0;
}
