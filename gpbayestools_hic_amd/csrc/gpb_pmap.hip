// gpb_pmap.hip — device pre-pass for the reference's parameterTrafoPCA option (src/emulator.py:492-551):
// three groups of model parameters are replaced by the leading principal components of the functions they
// parametrise (zeta/s(T) :102-108, eta/s(mu_B) :111-117, y_loss(y_init) :120-126), each evaluated on a
// 100-point grid, standardised and projected.  The reference does this with Python double loops per
// prediction row; inside an MCMC step it would be the only host round trip, so it runs here: one workgroup
// per eight walkers, the grid functions land in LDS, one thread per (walker, output column) does the projection.
#include "gpb_internal.h"
#include <math.h>

namespace gpb {

constexpr int PMAP_GRID = 100;

// fn 0: zeta/s(T; zeta_max, T_zeta0, sigma_plus, sigma_minus) at mu_B = 0
// fn 1: eta/s(mu_B; eta_0, eta_2, eta_4)      fn 2: y_loss(y_init; yloss_2, yloss_4, yloss_6)
__device__ __forceinline__ double pmap_fn(int fn, const double* par, double g) {
    if (fn == 0) {
        const double zmax = par[0], T0 = par[1], sp = par[2], sm = par[3];
        const double Tmu = T0 - 0.15 * 0.0;
        const double sig = (g < T0) ? sm : sp;
        const double dT = g - Tmu;
        return zmax * exp(-(dT * dT) / (2.0 * (sig * sig)));
    } else if (fn == 1) {
        const double e0 = par[0], e2 = par[1], e4 = par[2];
        if (0.0 < g && g <= 0.2) return e0 + (e2 - e0) * (g / 0.2);
        if (0.2 < g && g < 0.4) return e2 + (e4 - e2) * ((g - 0.2) / 0.2);
        return e4 + 0.0 * g;
    } else {
        const double y2 = par[0], y4 = par[1], y6 = par[2];
        if (0.0 < g && g <= 2.0) return y2 * (g / 2.0);
        if (2.0 < g && g < 4.0) return y2 + (y4 - y2) * ((g - 2.0) / 2.0);
        return y4 + (y6 - y4) * ((g - 4.0) / 2.0);
    }
}

// desc[g] = {fn, col0, col1, col2, col3, npc};  tab[g] = grid | scaler mean | scaler scale | pca mean | comps[maxpc]
// PM_ROWS walkers' rows through one emulator's map (the whole workgroup; u = PM_ROWS x G x PMAP_GRID doubles of LDS).  One row
// per workgroup left 7 of 128 threads in the projection loops and paid every dependent load (descriptor, parameters, tables) once
// per row: 49 us for nine maps of 2048 rows.  Every function value and every projection is computed as before, term by term
// (k ascending): same bits whatever the grouping.
constexpr int PM_ROWS = 8;
constexpr int PM_THREADS = 256;

__device__ __forceinline__ void param_map_rows(const double* __restrict__ X, int64_t w0, int64_t W, int d_in, int d_out,
                                               const int* __restrict__ col_src, int G, int maxpc,
                                               const int* __restrict__ desc, const double* __restrict__ tab,
                                               double* __restrict__ out, double* u) {
    const int t = threadIdx.x;
    const int nrow = (int)((W - w0) < PM_ROWS ? (W - w0) : PM_ROWS);
    const int per_row = G * PMAP_GRID;
    for (int it = t; it < nrow * per_row; it += PM_THREADS) {
        const int r = it / per_row, rem = it - r * per_row, g = rem / PMAP_GRID, k = rem - g * PMAP_GRID;
        const double* x = X + (w0 + r) * d_in;
        const int* dg = desc + 6 * g;
        const double* tg = tab + (size_t)g * (4 + maxpc) * PMAP_GRID;
        double par[4];
        for (int q = 0; q < 4; ++q) par[q] = (dg[1 + q] >= 0) ? x[dg[1 + q]] : 0.0;
        const double f = pmap_fn(dg[0], par, tg[k]);
        u[it] = (f - tg[PMAP_GRID + k]) / tg[2 * PMAP_GRID + k] - tg[3 * PMAP_GRID + k];
    }
    __syncthreads();
    for (int it = t; it < nrow * d_out; it += PM_THREADS) {
        const int r = it / d_out, j = it - r * d_out;
        const double* x = X + (w0 + r) * d_in;
        const int src = col_src[j];
        double v;
        if (src >= 0) {
            v = x[src];
        } else {
            const int code = -1 - src, g = code / maxpc, c = code - g * maxpc;
            const double* comp = tab + ((size_t)g * (4 + maxpc) + 4 + c) * PMAP_GRID;
            const double* ur = u + r * per_row + g * PMAP_GRID;
            v = 0.0;
            for (int k = 0; k < PMAP_GRID; ++k) v = fma(ur[k], comp[k], v);
        }
        out[(w0 + r) * d_out + j] = v;
    }
}

__global__ __launch_bounds__(PM_THREADS) void k_param_map(const double* __restrict__ X, int64_t W, int d_in, int d_out,
                                                          const int* __restrict__ col_src, int G, int maxpc,
                                                          const int* __restrict__ desc, const double* __restrict__ tab,
                                                          double* __restrict__ out) {
    extern __shared__ double u[];            // [PM_ROWS][G][PMAP_GRID] standardised, centred function values
    param_map_rows(X, (int64_t)blockIdx.x * PM_ROWS, W, d_in, d_out, col_src, G, maxpc, desc, tab, out, u);
}

// The maps of several emulators of a chain over the same rows in ONE launch (chain_rows: nine mapped emulators were nine
// launches, one behind the other on the stream).  blockIdx.y = the emulator; every workgroup runs param_map_rows as its
// emulator's own launch would: same bits.
struct PmEntry {
    const int* col_src; const int* desc; const double* tab; double* out;
    int d_in, d_out, G, maxpc;
};
constexpr int MAX_PM_CTX = 32;
struct PmTable { PmEntry e[MAX_PM_CTX]; };

__global__ __launch_bounds__(PM_THREADS) void k_param_map_multi(const PmTable tab, const double* __restrict__ X, int64_t W) {
    extern __shared__ double u[];
    const PmEntry& m = tab.e[blockIdx.y];
    param_map_rows(X, (int64_t)blockIdx.x * PM_ROWS, W, m.d_in, m.d_out, m.col_src, m.G, m.maxpc, m.desc, m.tab, m.out, u);
}

// chain_rows' form: the maps of ctxs[0..n) (all mapped, same stream, same input rows) — one launch per 32 emulators
int launch_param_maps(gpb_ctx* const* ctxs, int n, const double* X_dev, int64_t W) {
    gpb_ctx* ctx = ctxs[0];
    for (int e0 = 0; e0 < n; e0 += MAX_PM_CTX) {
        const int m = n - e0 < MAX_PM_CTX ? n - e0 : MAX_PM_CTX;
        PmTable tab;
        int gmax = 0;
        for (int i = 0; i < m; ++i) {
            gpb_ctx* c = ctxs[e0 + i];
            if (!c->pmap_int || !c->Xs || c->stream != ctx->stream) GPB_FAIL(GPB_E_STATE, "gpb: internal: launch_param_maps over a context without a map");
            tab.e[i] = PmEntry{c->pmap_int, c->pmap_int + c->pmap_d_out, c->pmap_tab, c->Xs,
                               (int)c->pmap_d_in, (int)c->pmap_d_out, c->pmap_groups, c->pmap_maxpc};
            gmax = c->pmap_groups > gmax ? c->pmap_groups : gmax;
        }
        hipLaunchKernelGGL(k_param_map_multi, dim3((unsigned)((W + PM_ROWS - 1) / PM_ROWS), (unsigned)m), dim3(PM_THREADS),
                           (size_t)PM_ROWS * gmax * PMAP_GRID * sizeof(double), ctx->stream, tab, X_dev, W);
        GPB_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace gpb

using namespace gpb;

extern "C" int gpb_param_map_set(gpb_ctx* ctx, int64_t d_in, int64_t d_out, const int32_t* col_src,
                                 int32_t n_groups, const int32_t* group_desc, const double* tables, int32_t maxpc) {
    if (!ctx) return GPB_E_ARG;
    if (d_in < 1 || d_out < 1 || n_groups < 1 || n_groups > 8 || maxpc < 1 || !col_src || !group_desc || !tables)
        GPB_FAIL(GPB_E_ARG, "gpb_param_map_set: bad sizes or null input");
    // the map's output is the GPs' input: chain_rows writes W x d_out doubles into the [Wcap][d] staging buffer
    if (ctx->N == 0) GPB_FAIL(GPB_E_STATE, "gpb_param_map_set before gpb_gp_set");
    if (d_out != ctx->d) GPB_FAIL(GPB_E_ARG, "gpb_param_map_set: d_out must equal the GPs' number of inputs");
    for (int64_t j = 0; j < d_out; ++j) {
        const int s = col_src[j];
        if (s >= d_in || (s < 0 && (-1 - s) >= n_groups * maxpc)) GPB_FAIL(GPB_E_ARG, "gpb_param_map_set: bad column map");
    }
    for (int g = 0; g < n_groups; ++g) {
        const int32_t* dg = group_desc + 6 * g;
        if (dg[0] < 0 || dg[0] > 2 || dg[5] < 1 || dg[5] > maxpc) GPB_FAIL(GPB_E_ARG, "gpb_param_map_set: bad group");
        for (int k = 1; k <= 4; ++k)
            if (dg[k] >= d_in) GPB_FAIL(GPB_E_ARG, "gpb_param_map_set: group column out of range");
    }
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->pmap_int) { pool_free(ctx->pmap_int); ctx->pmap_int = nullptr; }
    if (ctx->pmap_tab) { pool_free(ctx->pmap_tab); ctx->pmap_tab = nullptr; }
    const size_t ni = (size_t)d_out + 6 * (size_t)n_groups;
    const size_t nt = (size_t)n_groups * (4 + maxpc) * PMAP_GRID;
    GPB_HIP(pool_malloc_t(&ctx->pmap_int, ni * sizeof(int)));
    GPB_HIP(pool_malloc_t(&ctx->pmap_tab, nt * sizeof(double)));
    GPB_HIP(hipMemcpy(ctx->pmap_int, col_src, (size_t)d_out * sizeof(int), hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->pmap_int + d_out, group_desc, 6 * (size_t)n_groups * sizeof(int), hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->pmap_tab, tables, nt * sizeof(double), hipMemcpyHostToDevice));
    ctx->pmap_d_in = d_in; ctx->pmap_d_out = d_out; ctx->pmap_groups = n_groups; ctx->pmap_maxpc = maxpc;
    return 0;
}

extern "C" int gpb_param_map(gpb_ctx* ctx, const double* X_dev, int64_t W, double* out_dev) {
    if (!ctx || !X_dev || !out_dev || W < 0) return GPB_E_ARG;
    if (!ctx->pmap_int) GPB_FAIL(GPB_E_STATE, "gpb_param_map before gpb_param_map_set");
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    const int d_out = (int)ctx->pmap_d_out;
    hipLaunchKernelGGL(k_param_map, dim3((unsigned)((W + PM_ROWS - 1) / PM_ROWS)), dim3(PM_THREADS),
                       (size_t)PM_ROWS * ctx->pmap_groups * PMAP_GRID * sizeof(double), ctx->stream, X_dev, W, (int)ctx->pmap_d_in, d_out, ctx->pmap_int, ctx->pmap_groups,
                       ctx->pmap_maxpc, ctx->pmap_int + d_out, ctx->pmap_tab, out_dev);
    GPB_HIP(hipGetLastError());
    return 0;
}
