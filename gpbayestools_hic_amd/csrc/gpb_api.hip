// gpb_api.hip — C-ABI entry points (include/gpbayes.h): state management, host<->HBM staging,
// and kernel sequencing on the context's HIP stream.  Every per-row number is produced by the kernels; the one
// piece of host arithmetic is lowrank_setup: an M x M long-double Cholesky + thin QR, once per gpb_like_set (O(M^3):
// callers that re-set the likelihood per batch pay it every time).
#include "gpb_internal.h"
#include <dlfcn.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <mutex>
#include <condition_variable>

using namespace gpb;

namespace {

// (through the buffer cache of gpb_pool.hip: contexts come and go with every training)
template <typename T>
int dev_alloc(gpb_ctx* ctx, T** p, int64_t count) {
    if (*p) {        // a buffer that is replaced goes back to the cache: nothing may still be using it (callers have synchronised
        if (ctx->side_stream) GPB_HIP(hipStreamSynchronize(ctx->side_stream));      // ctx->stream; the look-ahead stream here)
        pool_free(*p); *p = nullptr;
    }
    GPB_HIP(pool_malloc(reinterpret_cast<void**>(p), sizeof(T) * (size_t)(count > 0 ? count : 1)));
    return 0;
}
template <typename T>
void dev_free(T** p) {
    if (*p) { pool_free(*p); *p = nullptr; }
}

int pick_dpad(int64_t d) {
    const int opts[] = {8, 16, 20, 24, 32, 48, 64};     // 20: the reference's analyses have 17-20 model parameters
    for (int o : opts) if (d <= o) return o;
    return -1;
}

int ensure_out(gpb_ctx* ctx, int64_t count) {
    if (count <= ctx->out_cap) return 0;
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(&ctx->out_stage);
    GPB_HIP(pool_malloc_t(&ctx->out_stage, sizeof(double) * (size_t)count));
    ctx->out_cap = count;
    return 0;
}

// [P][Wld] -> [W][P]
__global__ void k_transpose_pw(const double* __restrict__ src, double* __restrict__ dst, int64_t W, int64_t Wld,
                               int P) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= W) return;
    for (int p = 0; p < P; ++p) dst[w * P + p] = src[(int64_t)p * Wld + w];
}

// stage caller inputs: returns device pointers for Xs (and extra_std)
int stage_inputs(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device, const double* estd,
                 const double** Xs_dev, const double** estd_dev) {
    int rc = ensure_wcap(ctx, W);
    if (rc) return rc;
    if (on_device) {
        *Xs_dev = Xs;
        if (estd_dev) *estd_dev = estd;
    } else {
        GPB_HIP(hipMemcpyAsync(ctx->Xs, Xs, sizeof(double) * W * ctx->d, hipMemcpyHostToDevice, ctx->stream));
        *Xs_dev = ctx->Xs;
        if (estd_dev) {
            if (estd) {
                GPB_HIP(hipMemcpyAsync(ctx->estd, estd, sizeof(double) * W, hipMemcpyHostToDevice, ctx->stream));
                *estd_dev = ctx->estd;
            } else {
                *estd_dev = nullptr;
            }
        }
    }
    return 0;
}

}  // namespace

extern "C" int gpb_version(void) { return GPB_VERSION; }

extern "C" int gpb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gpb_ctx_create(int device, void* stream, gpb_ctx** out) {
    if (!out) return GPB_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GPB_E_NODEV;   // no CPU fallback, by design
    if (device < 0 || device >= n) return GPB_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return GPB_E_HIP;
    gpb_ctx* ctx = new gpb_ctx();
    ctx->device = device;
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return GPB_E_HIP;
        }
        ctx->own_stream = true;
    }
    if (pool_malloc_t(&ctx->notpd, sizeof(int)) != hipSuccess ||
        hipMemsetAsync(ctx->notpd, 0, sizeof(int), ctx->stream) != hipSuccess ||
        pool_malloc_t(&ctx->rows_live, sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(ctx->rows_live, 0, sizeof(unsigned long long), ctx->stream) != hipSuccess ||
        pool_malloc_t(&ctx->n_nan, sizeof(long long)) != hipSuccess ||
        hipMemsetAsync(ctx->n_nan, 0, sizeof(long long), ctx->stream) != hipSuccess ||
        pool_malloc_t(&ctx->tile_counter, 129 * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(ctx->tile_counter, 0, 129 * sizeof(unsigned), ctx->stream) != hipSuccess) {
        delete ctx;
        return GPB_E_ALLOC;
    }
    // optional: without it the tile rule sizes compacted batches by their upper bound
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->live_hint), sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess)
        *ctx->live_hint = 0;
    else
        ctx->live_hint = nullptr, (void)hipGetLastError();
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
        ctx->num_cu = ncu;
    *out = ctx;
    return 0;
}

extern "C" int gpb_ctx_destroy(gpb_ctx* ctx) {
    if (!ctx) return GPB_E_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    // every stream that touched the buffers is idle BEFORE the first of them re-enters circulation through the buffer cache
    // (the look-ahead stream of the factorisation reads and writes K and L^-1: gpb_chol.hip)
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    gpb_dist_finalize(ctx);
    free(ctx->h_theta);
    dev_free(&ctx->lr_R); dev_free(&ctx->lr_v0); dev_free(&ctx->lr_blocks);
    dev_free(&ctx->xmean); dev_free(&ctx->muS); dev_free(&ctx->Xc); dev_free(&ctx->dnorm); dev_free(&ctx->kmtiles); dev_free(&ctx->gpN);
    dev_free(&ctx->X); dev_free(&ctx->Xsc); dev_free(&ctx->thblk);
    ctx->ls = ctx->amp = ctx->noise = nullptr; ctx->gpform = ctx->gpmap = nullptr;         // (carved out of thblk)
    if (ctx->h_thblk) (void)hipHostFree(ctx->h_thblk);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    dev_free(&ctx->Z); dev_free(&ctx->K); dev_free(&ctx->Linv); dev_free(&ctx->T); dev_free(&ctx->yv);
    gpb::sliced_free(ctx);
    dev_free(&ctx->alpha); dev_free(&ctx->apart); dev_free(&ctx->info); dev_free(&ctx->lmlbuf);
    dev_free(&ctx->gpart); dev_free(&ctx->Xs); dev_free(&ctx->estd); dev_free(&ctx->KsT); dev_free(&ctx->mpart);
    dev_free(&ctx->spart); dev_free(&ctx->mean_pc); dev_free(&ctx->var_pc); dev_free(&ctx->out_stage);
    dev_free(&ctx->vbuf); dev_free(&ctx->covbuf); dev_free(&ctx->pmap_int); dev_free(&ctx->pmap_tab);
    dev_free(&ctx->tile_trace);
    dev_free(&ctx->A); dev_free(&ctx->mu); dev_free(&ctx->scale); dev_free(&ctx->C0); dev_free(&ctx->yexp);
    dev_free(&ctx->Cexp); dev_free(&ctx->mvn_ws); dev_free(&ctx->notpd); dev_free(&ctx->tile_counter);
    dev_free(&ctx->n_nan); dev_free(&ctx->mc_ws); dev_free(&ctx->bal_ws); dev_free(&ctx->rows_live); dev_free(&ctx->cmp_idx); dev_free(&ctx->cmp_X);
    if (ctx->live_hint) (void)hipHostFree(ctx->live_hint);
    for (hipEvent_t e : ctx->chol_events) (void)hipEventDestroy(e);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

extern "C" int gpb_ctx_set_stream(gpb_ctx* ctx, void* stream) {
    if (!ctx) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = reinterpret_cast<hipStream_t>(stream);      // NULL = the legacy default stream
    return 0;
}

extern "C" int gpb_pool_trim(void) {
    pool_trim();
    return 0;
}

extern "C" int gpb_sync(gpb_ctx* ctx) {
    if (!ctx) return GPB_E_ARG;
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" const char* gpb_last_error(gpb_ctx* ctx) { return ctx ? ctx->err.c_str() : "gpb: null context"; }
extern "C" void* gpb_stream(gpb_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---------------------------------------------------------------------------- GP state
// P GPs over one design (gpb_gp_set: every X_p the same pointer, stored once) or each over its own (gpb_gp_set_multi: the GPs
// of several emulators, or the restarts of a hyper-parameter search, side by side in one batch; all padded to the same Np)
static int gp_set_impl(gpb_ctx* ctx, int64_t P, int64_t d, const int64_t* N_p, const double* const* X_p,
                       const double* const* Z_p, bool multi, int kernel_id, double alpha) {
    const int dpad = pick_dpad(d);
    if (dpad < 0) GPB_FAIL(GPB_E_ARG, "gpb_gp_set: d > 64 not supported");
    int64_t N = 0;
    for (int64_t p = 0; p < P; ++p) N = N_p[p] > N ? N_p[p] : N;
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    ctx->N = N; ctx->d = d; ctx->P = ctx->Pstore = P; ctx->dpad = dpad; ctx->kind = kernel_id; ctx->alpha_reg = alpha;
    ctx->multi = multi; ctx->subset = false;
    ctx->Np = round_up(N, NB);
    ctx->have_theta = ctx->factored = false;
    // whatever was installed for the previous GPs (observable transform sized [old P][M], likelihood block, low-rank
    // factors, parameter map) does not describe the new ones: it has to be set again
    ctx->have_transform = ctx->have_like = ctx->lr_ok = false;
    ctx->M = 0;
    dev_free(&ctx->A); dev_free(&ctx->mu); dev_free(&ctx->scale); dev_free(&ctx->C0); dev_free(&ctx->yexp);
    dev_free(&ctx->Cexp); dev_free(&ctx->lr_R); dev_free(&ctx->lr_v0); dev_free(&ctx->pmap_int); dev_free(&ctx->pmap_tab);
    ctx->h_A.clear(); ctx->h_mu.clear(); ctx->h_C0.clear();
    ctx->pmap_d_in = ctx->pmap_d_out = 0; ctx->pmap_groups = ctx->pmap_maxpc = 0;
    const int64_t Np = ctx->Np, PX = multi ? P : 1;
    // workspaces sized by (Np, P) are stale now
    dev_free(&ctx->KsT); dev_free(&ctx->mpart); dev_free(&ctx->spart); dev_free(&ctx->mean_pc);
    dev_free(&ctx->var_pc); dev_free(&ctx->Xs); dev_free(&ctx->estd); dev_free(&ctx->cmp_idx); dev_free(&ctx->cmp_X);
    ctx->Wcap = 0; ctx->cmp_X_cap = 0; ctx->last_W = 0;
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->X, PX * Np * dpad))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->Xsc, P * Np * dpad))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->xmean, PX * dpad))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->muS, P * dpad))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->Xc, P * Np * dpad))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dnorm, P * Np))) return rc;
    {   // one block: [ls P x dpad | amp P | noise P] doubles, then [gpform P | gpmap P] ints; its page-locked twin on the host
        const size_t nd = (size_t)(P * dpad + 2 * P), bytes = sizeof(double) * nd + sizeof(int) * 2 * (size_t)P;
        if ((rc = dev_alloc(ctx, &ctx->thblk, (int64_t)((bytes + 7) / 8)))) return rc;
        ctx->ls = ctx->thblk; ctx->amp = ctx->ls + P * dpad; ctx->noise = ctx->amp + P;
        ctx->gpform = reinterpret_cast<int*>(ctx->noise + P); ctx->gpmap = ctx->gpform + P;
        if (ctx->h_thblk) { (void)hipHostFree(ctx->h_thblk); ctx->h_thblk = nullptr; }
        if (ctx->h_res) { (void)hipHostFree(ctx->h_res); ctx->h_res = nullptr; }
        GPB_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_thblk), bytes, hipHostMallocDefault));
        GPB_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_res), sizeof(double) * (size_t)(P * 4 + P * (d + 2)) + sizeof(int) * (size_t)P,
                              hipHostMallocDefault));
        memset(ctx->h_thblk, 0, bytes);
        ctx->thblk_bytes = bytes;
    }
    if ((rc = dev_alloc(ctx, &ctx->Z, P * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->K, P * Np * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->Linv, P * Np * Np))) return rc;
    gpb::sliced_free(ctx);                             // the digit planes follow the new shape on their next use
    // zeroed ONCE: the factorisation writes the diagonal blocks (with zeros above the diagonal) and the blocks below
    // them, never the blocks above — and the 128-wide tiles of the predict / K^-1 products read those as zeros
    GPB_HIP(hipMemsetAsync(ctx->Linv, 0, sizeof(double) * P * Np * Np, ctx->stream));
    if ((rc = dev_alloc(ctx, &ctx->T, P * Np * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->yv, P * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->alpha, P * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->info, P))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->lmlbuf, P * 4 + P * (d + 2)))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->gpN, P))) return rc;
    const int64_t nb64 = ctx->Np / 64;
    if ((rc = dev_alloc(ctx, &ctx->kmtiles, nb64 * (nb64 + 1)))) return rc;
    free(ctx->h_theta);
    ctx->h_theta = (double*)calloc((size_t)(P * (d + 2)), sizeof(double));
    std::vector<double> xp((size_t)(PX * Np * dpad), 0.0), zp((size_t)(P * Np), 0.0), xm((size_t)(PX * dpad), 0.0);
    ctx->h_ext.assign((size_t)(PX * d), 0.0);
    ctx->h_N.assign((size_t)P, 0);
    // The padding of a design to Np = a multiple of 64 points sits IN FRONT of it (round 5), in whole 16-row units (pad_front: the
    // remainder of < 16 rows stays behind): stored row pad + i holds design point i; the padded rows are identity rows of K and L,
    // zero rows of K*^T, zero targets.  In front, not behind: the predict kernel's work per row block grows with its index
    // (triangular K loops), so padding costs its rows' share of the LIGHTEST row block instead of the heaviest, and the leading
    // all-zero K-steps are skipped outright (launch_vsq) — N = 1000 wasted 4.7 % of the launch behind, 0.6 % now.  Designs of a
    // multiple of 64 points (every BASELINE configuration) are unaffected.
    for (int64_t p = 0; p < P; ++p) {
        ctx->h_N[(size_t)p] = (int)N_p[p];
        const int64_t pad = pad_front(Np, N_p[p]);
        for (int64_t i = 0; i < N_p[p]; ++i) zp[p * Np + pad + i] = Z_p[p][i];
    }
    for (int64_t q = 0; q < PX; ++q) {
        const double* Xq = X_p[q];
        const int64_t Nq = N_p[q], pad = pad_front(Np, Nq);
        for (int64_t i = 0; i < Nq; ++i)
            for (int64_t k = 0; k < d; ++k) xp[(q * Np + pad + i) * dpad + k] = Xq[i * d + k];
        for (int64_t k = 0; k < d; ++k) {
            // column means: the centre of the Gram form (k_kcross, k_kmat_mfma); column extents: what a length scale is
            // compared with when the distance form is chosen (choose_forms)
            double sum = 0.0, lo = Xq[k], hi = Xq[k];
            for (int64_t i = 0; i < Nq; ++i) {
                const double v = Xq[i * d + k];
                sum += v; lo = fmin(lo, v); hi = fmax(hi, v);
            }
            xm[(size_t)(q * dpad + k)] = sum / (double)Nq;
            ctx->h_ext[(size_t)(q * d + k)] = hi - lo;
        }
    }
    ctx->h_form.assign((size_t)P, 0);
    ctx->h_map.clear();
    ctx->n_diff = 0;
    GPB_HIP(hipMemset(ctx->gpform, 0, sizeof(int) * P));
    {   // the tiles of the lower block triangle, row by row (k_kmat_mfma's 1-D grid: the index arithmetic — a double sqrt and
        // its integer fix-ups per workgroup — was 5-9 % of that kernel's time, tools/micro/kmat_lab.hip)
        std::vector<int> tl;
        tl.reserve((size_t)(nb64 * (nb64 + 1)));
        for (int bi = 0; bi < (int)nb64; ++bi)
            for (int bj = 0; bj <= bi; ++bj) { tl.push_back(bi); tl.push_back(bj); }
        GPB_HIP(hipMemcpy(ctx->kmtiles, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice));
    }
    GPB_HIP(hipMemcpy(ctx->gpN, ctx->h_N.data(), sizeof(int) * P, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->xmean, xm.data(), sizeof(double) * xm.size(), hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->X, xp.data(), sizeof(double) * xp.size(), hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->Z, zp.data(), sizeof(double) * zp.size(), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int gpb_gp_set(gpb_ctx* ctx, int64_t N, int64_t d, int64_t P, const double* X_host,
                          const double* Z_host, int kernel_id, double alpha) {
    if (!ctx) return GPB_E_ARG;
    if (N < 1 || d < 1 || P < 1 || !X_host || !Z_host) GPB_FAIL(GPB_E_ARG, "gpb_gp_set: bad sizes or null input");
    if (kernel_id < 0 || kernel_id > 2) GPB_FAIL(GPB_E_ARG, "gpb_gp_set: unknown kernel_id");
    std::vector<int64_t> Ns((size_t)P, N);
    std::vector<const double*> Xs((size_t)P, X_host), Zs((size_t)P);
    for (int64_t p = 0; p < P; ++p) Zs[(size_t)p] = Z_host + p * N;
    return gp_set_impl(ctx, P, d, Ns.data(), Xs.data(), Zs.data(), false, kernel_id, alpha);
}

extern "C" int gpb_gp_set_multi(gpb_ctx* ctx, int64_t P, int64_t d, const int64_t* N_host, const double* const* X_host,
                                const double* const* Z_host, int kernel_id, double alpha) {
    if (!ctx) return GPB_E_ARG;
    if (P < 1 || d < 1 || !N_host || !X_host || !Z_host) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_multi: bad sizes or null input");
    if (kernel_id < 0 || kernel_id > 2) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_multi: unknown kernel_id");
    int64_t Np = 0;
    for (int64_t p = 0; p < P; ++p) {
        if (N_host[p] < 1 || !X_host[p] || !Z_host[p]) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_multi: bad size or null design / targets");
        const int64_t np_ = round_up(N_host[p], NB);
        if (p > 0 && np_ != Np) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_multi: the designs must pad to the same multiple of 64 points");
        Np = np_;
    }
    return gp_set_impl(ctx, P, d, N_host, X_host, Z_host, true, kernel_id, alpha);
}

// The distance form of every GP, from theta and the design's extents alone — never from a batch's size or a rank's share, so a
// walker's bits do not depend on how the ensemble is split.  S_p = sum_k (extent_k / l_pk)^2 bounds |a|^2 and, for queries
// inside the design's box, |b|^2 of the centred Gram form, whose cancellation error in r^2 is ~eps (|a|^2 + |b|^2); the
// kernels' slope |dk / d r^2| is at most 1/2 (RBF), 3/2 (Matern-3/2), 5/6 (Matern-5/2).  At S <= 1024 the Gram form keeps K and K*
// within ~2e-13; at the reference's Matern lower bound (l = 1e-3 x extent: S = 1e6, src/emulator.py:292-297) it would lose
// 3e-10 in K* and 2.5e-9 in the predictive variance, and those GPs take sklearn's own difference form.
int gpb::choose_forms(gpb_ctx* ctx, bool upload) {
    const int64_t P = ctx->P, d = ctx->d;
    int ndiff = 0;
    for (int64_t p = 0; p < P; ++p) {
        const double* th = ctx->h_theta + p * (d + 2);
        const int64_t gq = ctx->multi ? (ctx->subset ? ctx->h_map[(size_t)p] : p) : 0;      // whose design
        double S = 0.0;
        for (int64_t k = 0; k < d; ++k) {
            const double q = ctx->h_ext[(size_t)(gq * d + k)] / exp(th[1 + k]);
            S += q * q;
        }
        const int f = ctx->kcross_dot == 0 ? 1 : (ctx->kcross_dot == 2 ? 0 : ((S > ctx->gram_limit || !(S == S)) ? 1 : 0));
        ctx->h_form[(size_t)p] = f;
        ndiff += f;
    }
    ctx->n_diff = ndiff;
    if (!upload) return 0;                              // gpb_gp_set_theta sends the forms with the rest of its block
    GPB_HIP(hipMemcpyAsync(ctx->gpform, ctx->h_form.data(), sizeof(int) * P, hipMemcpyHostToDevice, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int gpb_gp_set_theta(gpb_ctx* ctx, const double* theta_host) {
    if (!ctx) return GPB_E_ARG;
    if (ctx->N == 0) GPB_FAIL(GPB_E_STATE, "gpb_gp_set_theta before gpb_gp_set");
    if (!theta_host) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_theta: null theta");
    const int64_t P = ctx->P, d = ctx->d, dpad = ctx->dpad;
    for (int64_t i = 0; i < P * (d + 2); ++i)
        if (!isfinite(theta_host[i])) GPB_FAIL(GPB_E_ARG, "gpb_gp_set_theta: non-finite theta");
    memcpy(ctx->h_theta, theta_host, sizeof(double) * P * (d + 2));
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));     // the previous block's copy out of the page-locked twin is done
    int rc = choose_forms(ctx, false);
    if (rc) return rc;
    // length scales, amplitudes, noise levels, distance forms and (a subset evaluation) the slot -> GP map: ONE asynchronous copy
    // from page-locked memory, ordered in front of the kernels that read them (it was three blocking copies of pageable vectors
    // plus one for the forms and one for the map, each a blit kernel and a wait)
    const int64_t Ps = ctx->Pstore;                  // the block's layout is the stored GP count's
    double* hl = ctx->h_thblk;
    double* ha = hl + Ps * dpad;
    double* hn = ha + Ps;
    int* hf = reinterpret_cast<int*>(hn + Ps);
    int* hm = hf + Ps;
    for (int64_t p = 0; p < P; ++p) {
        const double* th = theta_host + p * (d + 2);
        ha[p] = exp(th[0]);
        for (int64_t k = 0; k < dpad; ++k) hl[p * dpad + k] = k < d ? exp(th[1 + k]) : 1.0;
        hn[p] = exp(th[d + 1]);
        hf[p] = ctx->h_form[(size_t)p];
        if (ctx->subset) hm[p] = ctx->h_map[(size_t)p];
    }
    GPB_HIP(hipMemcpyAsync(ctx->thblk, ctx->h_thblk, ctx->thblk_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_scale_design(ctx))) return rc;
    ctx->have_theta = true;
    ctx->factored = false;
    return 0;
}

// LAPACK's info of slot p in the design's own numbering: the factorisation reports the first non-positive pivot's 1-based index in
// the PADDED matrix, whose padding sits in front (gp_set_impl)
static int info_unpadded(const gpb_ctx* ctx, int64_t p, int info) {
    if (info <= 0) return info;
    const int64_t q = ctx->multi ? (ctx->subset ? ctx->h_map[(size_t)p] : p) : 0;
    const int64_t Nq = ctx->multi ? ctx->h_N[(size_t)q] : ctx->N;
    return info - (int)pad_front(ctx->Np, Nq);
}

// K build, blocked Cholesky, [triangular inverse, alpha]: enqueued, nothing read back
static int factor_enqueue(gpb_ctx* ctx, bool need_inverse) {
    if (!ctx->have_theta) GPB_FAIL(GPB_E_STATE, "gpb_gp_factor before gpb_gp_set_theta");
    GPB_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = launch_kmat(ctx))) return rc;
    if ((rc = launch_potrf(ctx))) return rc;
    if (need_inverse) {
        if ((rc = launch_trtri(ctx))) return rc;
        if ((rc = launch_alpha(ctx))) return rc;
    }
    return 0;
}

static int factor_impl(gpb_ctx* ctx, int* info_host, bool need_inverse) {
    int rc = factor_enqueue(ctx, need_inverse);
    if (rc) return rc;
    std::vector<int> info((size_t)ctx->P, 0);
    GPB_HIP(hipMemcpyAsync(info.data(), ctx->info, sizeof(int) * ctx->P, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    int first = 0;
    for (int64_t p = 0; p < ctx->P; ++p) {
        info[p] = info_unpadded(ctx, p, info[p]);
        if (info_host) info_host[p] = info[p];
        if (info[p] != 0 && first == 0) first = info[p];
    }
    return first;
}

extern "C" int gpb_gp_factor(gpb_ctx* ctx, int* info_host) {
    if (!ctx) return GPB_E_ARG;
    int rc = factor_impl(ctx, info_host, true);
    if (rc < 0) return rc;
    // a failed GP leaves NaNs in L, L^-1 and alpha: the predict / likelihood entry points refuse to run on them
    ctx->factored = rc == 0;
    if (rc > 0) ctx->err = "gpb_gp_factor: kernel matrix not positive definite (see info); no factorisation installed";
    return rc;
}

extern "C" int gpb_gp_get(gpb_ctx* ctx, int what, double* out_host) {
    if (!ctx || !out_host) return GPB_E_ARG;
    if (ctx->N == 0) GPB_FAIL(GPB_E_STATE, "gpb_gp_get before gpb_gp_set");
    const int64_t N = ctx->N, Np = ctx->Np, P = ctx->P;
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    // the design's padding sits in front (gp_set_impl): GP p's own block starts at row / column Np - N_p.  A gpb_gp_set_multi
    // context's GPs may differ in size: each lands in the top-left corner of its N x N output block (N = the largest design),
    // the rest of the block being the identity (zero for alpha), as the padded device matrices read
    auto Nof = [&](int64_t p) { return ctx->multi ? (int64_t)ctx->h_N[(size_t)p] : N; };
    if (what == GPB_GET_ALPHA) {
        for (int64_t p = 0; p < P; ++p) {
            const int64_t Nq = Nof(p);
            for (int64_t i = Nq; i < N; ++i) out_host[p * N + i] = 0.0;
            GPB_HIP(hipMemcpy(out_host + p * N, ctx->alpha + p * Np + pad_front(Np, Nq), sizeof(double) * Nq, hipMemcpyDeviceToHost));
        }
        return 0;
    }
    if (what == GPB_GET_FORM) {
        for (int64_t p = 0; p < P; ++p) out_host[p] = (double)ctx->h_form[(size_t)p];
        return 0;
    }
    if (what == GPB_GET_KSTAR) {                       // KsT is [P][Np][Wld], walker fastest: transposed on the host
        const int64_t W = ctx->last_W, Wld = ctx->Wld, pad = pad_front(Np, N);
        if (W <= 0 || !ctx->KsT) GPB_FAIL(GPB_E_STATE, "gpb_gp_get(GPB_GET_KSTAR): no batch has been evaluated");
        std::vector<double> tmp((size_t)(N * W));
        for (int64_t p = 0; p < P; ++p) {
            if (ctx->batch_sliced) {                   // option 51: the batch left k_kcross as int8 digit planes: what they hold
                const int rc = gpb::sliced_read_kstar(ctx, p, pad, N, W, tmp.data());
                if (rc) return rc;
            } else
            GPB_HIP(hipMemcpy2D(tmp.data(), sizeof(double) * W, ctx->KsT + (p * Np + pad) * Wld, sizeof(double) * Wld,
                                sizeof(double) * W, (size_t)N, hipMemcpyDeviceToHost));
            for (int64_t w = 0; w < W; ++w)
                for (int64_t n = 0; n < N; ++n) out_host[(p * W + w) * N + n] = tmp[(size_t)(n * W + w)];
        }
        return 0;
    }
    const double* src = (what == GPB_GET_K || what == GPB_GET_L) ? ctx->K : (what == GPB_GET_LINV ? ctx->Linv : nullptr);
    if (!src) GPB_FAIL(GPB_E_ARG, "gpb_gp_get: unknown selector");
    for (int64_t p = 0; p < P; ++p) {
        const int64_t Nq = Nof(p), pad = pad_front(Np, Nq);
        double* o = out_host + p * N * N;
        if (Nq < N) {
            for (int64_t i = 0; i < N * N; ++i) o[i] = 0.0;
            for (int64_t i = Nq; i < N; ++i) o[i * N + i] = 1.0;
        }
        GPB_HIP(hipMemcpy2D(o, sizeof(double) * N, src + p * Np * Np + pad * Np + pad, sizeof(double) * Np,
                            sizeof(double) * Nq, (size_t)Nq, hipMemcpyDeviceToHost));
    }
    if (what == GPB_GET_L) {        // upper blocks hold scratch from the trailing updates: zero them
        for (int64_t p = 0; p < P; ++p)
            for (int64_t i = 0; i < N; ++i)
                for (int64_t j = i + 1; j < N; ++j) out_host[p * N * N + i * N + j] = 0.0;
    }
    return 0;
}

static int lml_impl(gpb_ctx* ctx, const double* theta_host, double* lml_host, double* grad_host, int* info_host) {
    int rc = gpb_gp_set_theta(ctx, theta_host);
    if (rc) return rc;
    if ((rc = factor_enqueue(ctx, true))) return rc;
    const int64_t P = ctx->P, d = ctx->d;
    if ((rc = launch_lml_value(ctx))) return rc;
    double* gdev = ctx->lmlbuf + P * 4;
    if (grad_host) {
        if ((rc = launch_lml_grad(ctx, gdev))) return rc;
    }
    const size_t nbuf = (size_t)(P * 4 + P * (d + 2));
    double* buf = ctx->h_res;
    int* info = reinterpret_cast<int*>(ctx->h_res + (size_t)(ctx->Pstore * 4 + ctx->Pstore * (d + 2)));
    GPB_HIP(hipMemcpyAsync(buf, ctx->lmlbuf, sizeof(double) * nbuf, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipMemcpyAsync(info, ctx->info, sizeof(int) * P, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t p = 0; p < P; ++p) {
        const bool bad = info[p] != 0;
        lml_host[p] = bad ? -INFINITY : buf[p * 4];                  // sk:_gpr.py:588-589
        if (grad_host)
            for (int64_t k = 0; k < d + 2; ++k) grad_host[p * (d + 2) + k] = bad ? 0.0 : buf[P * 4 + p * (d + 2) + k];
        if (info_host) info_host[p] = info_unpadded(ctx, p, info[p]);
    }
    ctx->factored = true;
    for (int64_t p = 0; p < P; ++p) if (info[p] != 0) ctx->factored = false;     // see gpb_gp_factor
    return 0;
}

extern "C" int gpb_gp_lml(gpb_ctx* ctx, const double* theta_host, double* lml_host, double* grad_host,
                          int* info_host) {
    if (!ctx || !theta_host || !lml_host) return GPB_E_ARG;
    return lml_impl(ctx, theta_host, lml_host, grad_host, info_host);
}

// n of the stored GPs, evaluated in the first n slots of the workspaces: the kernels that read a GP's inputs (design, column
// means, targets, design size) look the stored GP up through gpmap; everything a launch produces is indexed by slot.  A GP's
// numbers do not depend on its slot or on the other GPs of the launch (per-GP kernels, fixed-order reductions).
extern "C" int gpb_gp_lml_subset(gpb_ctx* ctx, int64_t n, const int32_t* gp_index, const double* theta_host,
                                 double* lml_host, double* grad_host, int* info_host) {
    if (!ctx || !gp_index || !theta_host || !lml_host) return GPB_E_ARG;
    if (ctx->N == 0) GPB_FAIL(GPB_E_STATE, "gpb_gp_lml_subset before gpb_gp_set");
    if (n < 1 || n > ctx->Pstore) GPB_FAIL(GPB_E_ARG, "gpb_gp_lml_subset: bad subset size");
    ctx->h_map.assign((size_t)n, 0);
    for (int64_t a = 0; a < n; ++a) {
        if (gp_index[a] < 0 || gp_index[a] >= ctx->Pstore) GPB_FAIL(GPB_E_ARG, "gpb_gp_lml_subset: GP index out of range");
        ctx->h_map[(size_t)a] = gp_index[a];
    }
    GPB_HIP(hipSetDevice(ctx->device));
    ctx->P = n;                                        // (the slot -> GP map goes up with theta: gpb_gp_set_theta's block)
    ctx->subset = true;
    const int rc = lml_impl(ctx, theta_host, lml_host, grad_host, info_host);
    ctx->P = ctx->Pstore;
    ctx->subset = false;
    // the workspaces hold the subset's factorisation in their first slots and theta is the subset's: nothing to predict from
    ctx->have_theta = ctx->factored = false;
    return rc;
}

extern "C" int gpb_gp_predict(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device, double* mean,
                              double* var) {
    if (!ctx || !Xs || !mean || W < 0) return GPB_E_ARG;
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    const double* Xs_dev;
    int rc = stage_inputs(ctx, Xs, W, on_device, nullptr, &Xs_dev, nullptr);
    if (rc) return rc;
    if ((rc = launch_predict(ctx, Xs_dev, W, var != nullptr))) return rc;
    const int64_t P = ctx->P;
    dim3 grid((unsigned)((W + 255) / 256));
    if (on_device) {
        hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->mean_pc, mean, W, ctx->Wld, (int)P);
        if (var) hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->var_pc, var, W, ctx->Wld, (int)P);
        GPB_HIP(hipGetLastError());
        return 0;
    }
    if ((rc = ensure_out(ctx, 2 * W * P))) return rc;
    hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->mean_pc, ctx->out_stage, W, ctx->Wld, (int)P);
    GPB_HIP(hipMemcpyAsync(mean, ctx->out_stage, sizeof(double) * W * P, hipMemcpyDeviceToHost, ctx->stream));
    if (var) {
        hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->var_pc, ctx->out_stage + W * P, W, ctx->Wld, (int)P);
        GPB_HIP(hipMemcpyAsync(var, ctx->out_stage + W * P, sizeof(double) * W * P, hipMemcpyDeviceToHost, ctx->stream));
    }
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int gpb_gp_predict_cov(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device, double* mean,
                                  double* cov) {
    if (!ctx || !Xs || !mean || !cov || W < 0) return GPB_E_ARG;
    if (W == 0) return 0;
    if (W > 8192) GPB_FAIL(GPB_E_ARG, "gpb_gp_predict_cov: W > 8192 (the W x W covariance is for small batches)");
    GPB_HIP(hipSetDevice(ctx->device));
    const double* Xs_dev;
    int rc = stage_inputs(ctx, Xs, W, on_device, nullptr, &Xs_dev, nullptr);
    if (rc) return rc;
    const int64_t P = ctx->P;
    dim3 grid((unsigned)((W + 255) / 256));
    if (on_device) {
        if ((rc = launch_predict_cov(ctx, Xs_dev, W, cov))) return rc;
        hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->mean_pc, mean, W, ctx->Wld, (int)P);
        GPB_HIP(hipGetLastError());
        return 0;
    }
    if ((rc = ensure_out(ctx, W * P + P * W * W))) return rc;
    double* dm = ctx->out_stage;
    double* dc = dm + W * P;
    if ((rc = launch_predict_cov(ctx, Xs_dev, W, dc))) return rc;
    hipLaunchKernelGGL(k_transpose_pw, grid, dim3(256), 0, ctx->stream, ctx->mean_pc, dm, W, ctx->Wld, (int)P);
    GPB_HIP(hipMemcpyAsync(mean, dm, sizeof(double) * W * P, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipMemcpyAsync(cov, dc, sizeof(double) * P * W * W, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------- emulator transform
extern "C" int gpb_emu_set_transform(gpb_ctx* ctx, int mode, int64_t M, const double* A_host, const double* mu_host,
                                     const double* cov_trunc_host, const double* scale_host) {
    if (!ctx) return GPB_E_ARG;
    if (ctx->N == 0) GPB_FAIL(GPB_E_STATE, "gpb_emu_set_transform before gpb_gp_set");
    if (mode < 0 || mode > 3 || M < 1 || !mu_host) GPB_FAIL(GPB_E_ARG, "gpb_emu_set_transform: bad arguments");
    const bool no_pca = (mode == GPB_MODE_NO_PCA || mode == GPB_MODE_NO_PCA_EXPDIAG);
    if (!no_pca && !A_host) GPB_FAIL(GPB_E_ARG, "gpb_emu_set_transform: PCA mode needs A");
    if (no_pca && (M != ctx->P || !scale_host)) GPB_FAIL(GPB_E_ARG, "gpb_emu_set_transform: no-PCA mode needs M == P and scale");
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t P = ctx->P;
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->A, P * M))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->mu, M))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->scale, M))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->C0, M * M))) return rc;
    std::vector<double> zeros((size_t)(M * M > P * M ? M * M : P * M), 0.0), ones((size_t)M, 1.0);
    GPB_HIP(hipMemcpy(ctx->A, A_host ? A_host : zeros.data(), sizeof(double) * P * M, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->mu, mu_host, sizeof(double) * M, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->scale, scale_host ? scale_host : ones.data(), sizeof(double) * M, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->C0, cov_trunc_host ? cov_trunc_host : zeros.data(), sizeof(double) * M * M, hipMemcpyHostToDevice));
    ctx->mode = mode; ctx->M = M; ctx->have_transform = true; ctx->have_like = false;
    ctx->lr_ok = false;
    ctx->h_A.assign(A_host ? A_host : zeros.data(), (A_host ? A_host : zeros.data()) + P * M);
    ctx->h_mu.assign(mu_host, mu_host + M);
    ctx->h_C0.assign(cov_trunc_host ? cov_trunc_host : zeros.data(), (cov_trunc_host ? cov_trunc_host : zeros.data()) + M * M);
    return 0;
}

// Low-rank form of the block likelihood (PCA mode, src/emulator.py:584-587 + src/mcmc.py:23-65): the covariance of
// every walker is C = C0 + A^T D A with the SAME C0 = C_trunc + C_exp and a walker-dependent D = diag(var_p) of
// rank P << M.  With C0 = L0 L0^T and the thin QR factorisation L0^-1 A^T = Q R (M x P, P x P):
//     L0^-1 C L0^-T = I + Q (R D R^T) Q^T
//     log det C = log det C0 + log det S,                 S = I_P + R D R^T
//     y^T C^-1 y = |(I - Q Q^T) L0^-1 y|^2 + v^T S^-1 v,  v = Q^T L0^-1 y = R m + v0
// because y = A^T m + (mu - yexp) and L0^-1 A^T m = Q R m lies in the span of Q: the first term is a constant of
// the emulator.  No difference of large numbers anywhere (unlike the Woodbury identity); against 40-digit
// arithmetic the form is as accurate as the dense Cholesky (1e-15 relative).  Per walker this is a P x P Cholesky
// instead of an M x M one.  Returns false (dense kernels stay in charge) when C0 is not positive definite or A is
// rank deficient.
static bool lowrank_setup(gpb_ctx* ctx, const double* yexp, const double* cexp) {
    const int64_t M = ctx->M, P = ctx->P;
    if (ctx->mode != GPB_MODE_PCA || P > 16 || P > M) return false;
    std::vector<long double> L((size_t)(M * M), 0.0L);
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j <= i; ++j) {                         // symmetrised lower triangle of C0
            const double a = ctx->h_C0[i * M + j] + cexp[i * M + j], b = ctx->h_C0[j * M + i] + cexp[j * M + i];
            L[i * M + j] = 0.5L * ((long double)a + (long double)b);
        }
    long double logdet = 0.0L;
    for (int64_t j = 0; j < M; ++j) {                              // Cholesky, column by column
        long double dsum = L[j * M + j];
        for (int64_t k = 0; k < j; ++k) dsum -= L[j * M + k] * L[j * M + k];
        if (!(dsum > 0.0L)) return false;
        const long double ljj = sqrtl(dsum);
        L[j * M + j] = ljj;
        logdet += 2.0L * logl(ljj);
        for (int64_t i = j + 1; i < M; ++i) {
            long double v = L[i * M + j];
            for (int64_t k = 0; k < j; ++k) v -= L[i * M + k] * L[j * M + k];
            L[i * M + j] = v / ljj;
        }
    }
    auto fwd = [&](std::vector<long double>& x) {                  // x <- L0^-1 x
        for (int64_t i = 0; i < M; ++i) {
            long double v = x[i];
            for (int64_t k = 0; k < i; ++k) v -= L[i * M + k] * x[k];
            x[i] = v / L[i * M + i];
        }
    };
    std::vector<std::vector<long double>> Q((size_t)P, std::vector<long double>((size_t)M));
    std::vector<long double> R((size_t)(P * P), 0.0L);
    for (int64_t j = 0; j < P; ++j) {                              // column j of L0^-1 A^T, then Gram-Schmidt twice
        std::vector<long double>& q = Q[j];
        for (int64_t i = 0; i < M; ++i) q[i] = ctx->h_A[j * M + i];
        fwd(q);
        long double norm0 = 0.0L;
        for (int64_t i = 0; i < M; ++i) norm0 += q[i] * q[i];
        for (int pass = 0; pass < 2; ++pass)
            for (int64_t i = 0; i < j; ++i) {
                long double r = 0.0L;
                for (int64_t k = 0; k < M; ++k) r += Q[i][k] * q[k];
                for (int64_t k = 0; k < M; ++k) q[k] -= r * Q[i][k];
                R[i * P + j] += r;
            }
        long double nrm = 0.0L;
        for (int64_t i = 0; i < M; ++i) nrm += q[i] * q[i];
        if (!(nrm > 1e-24L * norm0) || !(norm0 > 0.0L)) return false;      // A (numerically) rank deficient
        nrm = sqrtl(nrm);
        R[j * P + j] = nrm;
        for (int64_t i = 0; i < M; ++i) q[i] /= nrm;
    }
    std::vector<long double> w((size_t)M);
    for (int64_t i = 0; i < M; ++i) w[i] = (long double)ctx->h_mu[i] - (long double)yexp[i];
    fwd(w);
    std::vector<long double> v0((size_t)P, 0.0L);
    for (int64_t j = 0; j < P; ++j)
        for (int64_t k = 0; k < M; ++k) v0[j] += Q[j][k] * w[k];
    long double cperp = 0.0L;
    for (int64_t k = 0; k < M; ++k) {
        long double t = w[k];
        for (int64_t j = 0; j < P; ++j) t -= Q[j][k] * v0[j];
        cperp += t * t;
    }
    double Rh[256], vh[16];
    for (int i = 0; i < 256; ++i) Rh[i] = 0.0;
    for (int i = 0; i < 16; ++i) vh[i] = 0.0;
    for (int64_t i = 0; i < P; ++i) {
        vh[i] = (double)v0[i];
        for (int64_t j = i; j < P; ++j) Rh[i * 16 + j] = (double)R[i * P + j];
    }
    if (dev_alloc(ctx, &ctx->lr_R, 256) || dev_alloc(ctx, &ctx->lr_v0, 16)) return false;
    if (hipMemcpy(ctx->lr_R, Rh, sizeof(Rh), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemcpy(ctx->lr_v0, vh, sizeof(vh), hipMemcpyHostToDevice) != hipSuccess) return false;
    ctx->lr_cperp = (double)cperp;
    ctx->lr_logdet0 = (double)logdet;
    return true;
}

extern "C" int gpb_emu_predict(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device, const double* extra_std,
                               double* mean, double* cov) {
    if (!ctx || !Xs || !mean || W < 0) return GPB_E_ARG;
    if (!ctx->have_transform) GPB_FAIL(GPB_E_STATE, "gpb_emu_predict before gpb_emu_set_transform");
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    const double *Xs_dev, *estd_dev;
    int rc = stage_inputs(ctx, Xs, W, on_device, extra_std, &Xs_dev, &estd_dev);
    if (rc) return rc;
    if ((rc = launch_predict(ctx, Xs_dev, W, cov != nullptr))) return rc;
    const int64_t M = ctx->M;
    if (on_device) return launch_obs(ctx, W, estd_dev, mean, cov);
    const int64_t need = W * M + (cov ? W * M * M : 0);
    if ((rc = ensure_out(ctx, need))) return rc;
    double* dm = ctx->out_stage;
    double* dc = cov ? ctx->out_stage + W * M : nullptr;
    if ((rc = launch_obs(ctx, W, estd_dev, dm, dc))) return rc;
    GPB_HIP(hipMemcpyAsync(mean, dm, sizeof(double) * W * M, hipMemcpyDeviceToHost, ctx->stream));
    if (cov) GPB_HIP(hipMemcpyAsync(cov, dc, sizeof(double) * W * M * M, hipMemcpyDeviceToHost, ctx->stream));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------- likelihood block
extern "C" int gpb_like_set(gpb_ctx* ctx, const double* yexp_host, const double* cov_exp_host) {
    if (!ctx) return GPB_E_ARG;
    if (!ctx->have_transform) GPB_FAIL(GPB_E_STATE, "gpb_like_set before gpb_emu_set_transform");
    if (!yexp_host || !cov_exp_host) GPB_FAIL(GPB_E_ARG, "gpb_like_set: null input");
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t M = ctx->M;
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->yexp, M))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->Cexp, M * M))) return rc;
    GPB_HIP(hipMemcpy(ctx->yexp, yexp_host, sizeof(double) * M, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(ctx->Cexp, cov_exp_host, sizeof(double) * M * M, hipMemcpyHostToDevice));
    ctx->lr_ok = lowrank_setup(ctx, yexp_host, cov_exp_host);
    ctx->have_like = true;
    return 0;
}

extern "C" int gpb_loglike(gpb_ctx* ctx, const double* Xs, int64_t W, int on_device, double* ll, int accumulate,
                           int* n_notpd_host) {
    if (!ctx || !Xs || !ll || W < 0) return GPB_E_ARG;
    if (!ctx->have_like) GPB_FAIL(GPB_E_STATE, "gpb_loglike before gpb_like_set");
    if (W == 0) { if (n_notpd_host) *n_notpd_host = 0; return 0; }
    GPB_HIP(hipSetDevice(ctx->device));
    const double* Xs_dev;
    int rc = stage_inputs(ctx, Xs, W, on_device, nullptr, &Xs_dev, nullptr);
    if (rc) return rc;
    if (n_notpd_host) GPB_HIP(hipMemsetAsync(ctx->notpd, 0, sizeof(int), ctx->stream));
    const bool fused = loglike_fuses_finalize(ctx, W);
    if ((rc = launch_predict(ctx, Xs_dev, W, true, !fused))) return rc;
    if (on_device) {
        if ((rc = launch_loglike(ctx, W, ll, accumulate != 0, fused))) return rc;
    } else {
        if ((rc = ensure_out(ctx, W))) return rc;
        if (accumulate) GPB_HIP(hipMemcpyAsync(ctx->out_stage, ll, sizeof(double) * W, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_loglike(ctx, W, ctx->out_stage, accumulate != 0, fused))) return rc;
        GPB_HIP(hipMemcpyAsync(ll, ctx->out_stage, sizeof(double) * W, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n_notpd_host) {
        GPB_HIP(hipMemcpyAsync(n_notpd_host, ctx->notpd, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        GPB_HIP(hipStreamSynchronize(ctx->stream));
    } else if (!on_device) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int gpb_logpost(gpb_ctx* ctx, const double* Xs_dev, int64_t W, double* ll_dev, int accumulate,
                           const double* lo_dev, const double* hi_dev, double outside_value, double inside_const) {
    if (!ctx || !Xs_dev || !ll_dev || !lo_dev || !hi_dev || W < 0) return GPB_E_ARG;
    if (!ctx->have_like) GPB_FAIL(GPB_E_STATE, "gpb_logpost before gpb_like_set");
    if (W == 0) return 0;
    GPB_HIP(hipSetDevice(ctx->device));
    int rc = ensure_wcap(ctx, W);
    if (rc) return rc;
    const bool fused = loglike_fuses_finalize(ctx, W);
    if (!accumulate && compaction_applies(ctx)) {
        // the rows inside the prior box only, as the reference does (src/mcmc.py:194-203, 275-283); no host round trip
        if ((rc = launch_compact(ctx, Xs_dev, W, ctx->d, lo_dev, hi_dev, outside_value, ll_dev))) return rc;
        if ((rc = launch_predict(ctx, ctx->cmp_X, W, true, !fused, ctx->cmp_idx))) return rc;
        return launch_loglike(ctx, W, ll_dev, false, fused, nullptr, nullptr, nullptr, outside_value, inside_const,
                              ctx->cmp_idx);
    }
    if ((rc = launch_predict(ctx, Xs_dev, W, true, !fused))) return rc;
    return launch_loglike(ctx, W, ll_dev, accumulate != 0, fused, Xs_dev, lo_dev, hi_dev, outside_value, inside_const);
}

extern "C" int gpb_mvn_loglike(gpb_ctx* ctx, const double* dY, const double* cov, int64_t W, int64_t M,
                               int on_device, double* ll, int* n_notpd_host) {
    if (!ctx || !dY || !cov || !ll || W < 0 || M < 1) return GPB_E_ARG;
    if (W == 0) { if (n_notpd_host) *n_notpd_host = 0; return 0; }
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipMemsetAsync(ctx->notpd, 0, sizeof(int), ctx->stream));
    int rc;
    if (on_device) {
        if ((rc = launch_mvn(ctx, dY, cov, W, M, ll))) return rc;
    } else {
        const int64_t need = W * M + W * M * M + W;
        if ((rc = ensure_out(ctx, need))) return rc;
        double* dy = ctx->out_stage; double* dc = dy + W * M; double* dl = dc + W * M * M;
        GPB_HIP(hipMemcpyAsync(dy, dY, sizeof(double) * W * M, hipMemcpyHostToDevice, ctx->stream));
        GPB_HIP(hipMemcpyAsync(dc, cov, sizeof(double) * W * M * M, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_mvn(ctx, dy, dc, W, M, dl))) return rc;
        GPB_HIP(hipMemcpyAsync(ll, dl, sizeof(double) * W, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n_notpd_host) GPB_HIP(hipMemcpyAsync(n_notpd_host, ctx->notpd, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (n_notpd_host || !on_device) GPB_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---------------------------------------------------------------------------- RCCL (lazy dlopen)
namespace {
struct NcclId { char internal[128]; };
typedef int (*fn_getuid)(NcclId*);
typedef int (*fn_init)(void**, int, NcclId, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
struct Rccl {
    void* h = nullptr;
    fn_getuid getuid = nullptr; fn_init init = nullptr; fn_allgather allgather = nullptr; fn_destroy destroy = nullptr;
    bool load() {
        if (h) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        getuid = (fn_getuid)dlsym(h, "ncclGetUniqueId");
        init = (fn_init)dlsym(h, "ncclCommInitRank");
        allgather = (fn_allgather)dlsym(h, "ncclAllGather");
        destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
        return getuid && init && allgather && destroy;
    }
} g_rccl;
constexpr int NCCL_DOUBLE = 8;   // ncclFloat64
}  // namespace

extern "C" int gpb_dist_available(void) { return g_rccl.load() ? 1 : 0; }

extern "C" int gpb_dist_uid(void* uid128_host) {
    if (!uid128_host) return GPB_E_ARG;
    if (!g_rccl.load()) return GPB_E_RCCL;
    NcclId id;
    if (g_rccl.getuid(&id) != 0) return GPB_E_RCCL;
    memcpy(uid128_host, &id, sizeof(id));
    return 0;
}

extern "C" int gpb_dist_init(gpb_ctx* ctx, int rank, int nranks, const void* uid128_host) {
    if (!ctx || !uid128_host || nranks < 1 || rank < 0 || rank >= nranks) return GPB_E_ARG;
    if (!g_rccl.load()) GPB_FAIL(GPB_E_RCCL, "gpb_dist_init: cannot load librccl");
    GPB_HIP(hipSetDevice(ctx->device));
    NcclId id;
    memcpy(&id, uid128_host, sizeof(id));
    void* comm = nullptr;
    if (g_rccl.init(&comm, nranks, id, rank) != 0) GPB_FAIL(GPB_E_RCCL, "gpb_dist_init: ncclCommInitRank failed");
    ctx->comm = comm; ctx->rank = rank; ctx->nranks = nranks;
    return 0;
}

#ifdef GPB_DEBUG_VARIANTS
// ---- loopback group (test hook, debug build only): R contexts of ONE process, each with its own stream and driven by its own
// host thread, as the ranks of a communicator.  The one-GPU build box cannot form an RCCL communicator of more than one rank
// (RCCL refuses two ranks on a device), so without this the R > 1 form of gpb_chain_emcee_run — per-rank row shares, offsets
// into the gathered vector, accept steps fed by the other ranks' log-probabilities — never runs as a whole.  The all-gather is
// emulated on the ranks' streams: every rank records an event behind its producers, the host threads meet, every rank's stream
// waits for every peer's event and copies the peer's block, records a second event behind its copies, the threads meet again,
// and every stream waits for every peer's second event (a peer may still be reading this rank's block).  Same stream
// semantics as ncclAllGather: in order on the caller's stream, asynchronous to the host after the call.
// A rank whose HIP call fails still keeps BOTH appointments of the call (its peers would wait for it for good otherwise) and
// marks the group aborted: every member's current and later calls then return GPB_E_STATE.  Releasing the group wakes whoever
// still waits and frees it only when nobody is inside a meeting.
struct LoopGroup {
    int R = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, inside = 0;
    bool aborted = false;
    unsigned long long gen = 0;
    std::vector<const double*> send;
    std::vector<hipEvent_t> ready, done;
    std::vector<gpb_ctx*> members;
    bool meet() {                                     // false: the group was aborted (by a failing rank or by its release)
        std::unique_lock<std::mutex> lk(m);
        ++inside;
        const unsigned long long g = gen;
        if (++arrived == R) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || aborted; });
        --inside;
        if (aborted) cv.notify_all();                 // (a release waiting for `inside` to drain)
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

static int loop_allgather(gpb_ctx* ctx, const double* send_dev, double* recv_dev, int64_t count) {
    LoopGroup* G = ctx->loop;
    const int r = ctx->rank, R = G->R;
    hipError_t e = hipEventRecord(G->ready[r], ctx->stream);
    G->send[r] = send_dev;
    if (e != hipSuccess) G->abort();
    bool ok = G->meet() && e == hipSuccess;
    for (int q = 0; q < R && ok; ++q) {
        if (q != r) e = hipStreamWaitEvent(ctx->stream, G->ready[q], 0);
        double* dst = recv_dev + (int64_t)q * count;
        if (e == hipSuccess && dst != G->send[q])
            e = hipMemcpyAsync(dst, G->send[q], sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, ctx->stream);
        ok = e == hipSuccess;
    }
    if (ok) ok = (e = hipEventRecord(G->done[r], ctx->stream)) == hipSuccess;
    if (!ok) G->abort();                              // before the second appointment, which this rank still keeps
    ok = G->meet() && ok;
    for (int q = 0; q < R && ok; ++q)
        if (q != r) ok = (e = hipStreamWaitEvent(ctx->stream, G->done[q], 0)) == hipSuccess;
    if (!ok) {
        if (e != hipSuccess) { ctx->err = std::string("loopback all-gather: ") + hipGetErrorString(e); return GPB_E_HIP; }
        GPB_FAIL(GPB_E_STATE, "loopback all-gather: the group was aborted (a member failed or the group was released)");
    }
    return 0;                                       // (the next call's first meeting keeps a fast rank from re-recording early)
}

static void loop_free(LoopGroup* G) {
    for (hipEvent_t e : G->ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : G->done) if (e) (void)hipEventDestroy(e);
    delete G;
}

extern "C" int gpb_debug_loopback_group(gpb_ctx* const* ctxs, int R) {
    if (!ctxs || R < 1 || R > 64) return GPB_E_ARG;
    for (int r = 0; r < R; ++r)
        if (!ctxs[r] || ctxs[r]->comm || ctxs[r]->loop || ctxs[r]->device != ctxs[0]->device) return GPB_E_ARG;
    if (hipSetDevice(ctxs[0]->device) != hipSuccess) return GPB_E_HIP;      // single-device test hook: the events live there
    LoopGroup* G = new LoopGroup;
    G->R = R;
    G->send.assign((size_t)R, nullptr);
    G->ready.assign((size_t)R, nullptr); G->done.assign((size_t)R, nullptr);
    for (int r = 0; r < R; ++r) {
        if (hipEventCreateWithFlags(&G->ready[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&G->done[r], hipEventDisableTiming) != hipSuccess) {
            loop_free(G);                           // (with the events created so far)
            return GPB_E_HIP;
        }
        G->members.push_back(ctxs[r]);
    }
    for (int r = 0; r < R; ++r) {
        ctxs[r]->loop = G; ctxs[r]->rank = r; ctxs[r]->nranks = R;
        ctxs[r]->comm = (void*)G;                   // "a communicator is installed": what the step loop asks
    }
    return 0;
}

// wake whoever waits in a meeting and wait until nobody is inside one: only then may the group's mutex go
static void loop_drain(LoopGroup* G) {
    std::unique_lock<std::mutex> lk(G->m);
    G->aborted = true;
    G->cv.notify_all();
    G->cv.wait(lk, [&] { return G->inside == 0; });
}
// a member leaves (its context is being destroyed without a release of the group): the group forgets it and goes with its last member
static void loop_detach(gpb_ctx* ctx) {
    LoopGroup* G = ctx->loop;
    bool any = false;
    for (gpb_ctx*& c : G->members) {
        if (c == ctx) c = nullptr;
        any = any || c != nullptr;
    }
    ctx->loop = nullptr; ctx->comm = nullptr; ctx->rank = 0; ctx->nranks = 1;
    if (!any) { loop_drain(G); loop_free(G); }
    else G->abort();                                // the others can no longer complete a collective
}

extern "C" int gpb_debug_loopback_release(gpb_ctx* ctx) {
    if (!ctx || !ctx->loop) return GPB_E_ARG;
    LoopGroup* G = ctx->loop;
    loop_drain(G);
    for (gpb_ctx* c : G->members)
        if (c) { (void)hipStreamSynchronize(c->stream); c->loop = nullptr; c->comm = nullptr; c->rank = 0; c->nranks = 1; }
    loop_free(G);
    return 0;
}
#endif  // GPB_DEBUG_VARIANTS

extern "C" int gpb_dist_allgather(gpb_ctx* ctx, const double* send_dev, double* recv_dev, int64_t count) {
    if (!ctx || !send_dev || !recv_dev || count < 0) return GPB_E_ARG;
#ifdef GPB_DEBUG_VARIANTS
    if (ctx->loop) return loop_allgather(ctx, send_dev, recv_dev, count);
#endif
    if (!ctx->comm) GPB_FAIL(GPB_E_STATE, "gpb_dist_allgather before gpb_dist_init");
    if (g_rccl.allgather(send_dev, recv_dev, (size_t)count, NCCL_DOUBLE, ctx->comm, ctx->stream) != 0)
        GPB_FAIL(GPB_E_RCCL, "ncclAllGather failed");
    return 0;
}

extern "C" int gpb_dist_finalize(gpb_ctx* ctx) {
    if (!ctx) return GPB_E_ARG;
#ifdef GPB_DEBUG_VARIANTS
    if (ctx->loop) { loop_detach(ctx); return 0; }  // (normally the group is released as a whole: gpb_debug_loopback_release)
#endif
    if (ctx->comm && g_rccl.destroy) { g_rccl.destroy(ctx->comm); ctx->comm = nullptr; }
    return 0;
}

#ifdef GPB_DEBUG_VARIANTS
// ---------------------------------------------------------------------------- test hooks
extern "C" int gpb_test_gemm(gpb_ctx* ctx, int64_t M, int64_t N, int64_t K, const double* A_host,
                             const double* B_host, double* C_host, int b_trans) {
    if (!ctx || !A_host || !B_host || !C_host) return GPB_E_ARG;
    if (M < 2 || N < 2 || K < 16 || (K % 16) || (M % 2) || (N % 2)) GPB_FAIL(GPB_E_ARG, "gpb_test_gemm: K%16, M%2, N%2");
    GPB_HIP(hipSetDevice(ctx->device));
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    GPB_HIP(hipMalloc(&dA, sizeof(double) * M * K));
    GPB_HIP(hipMalloc(&dB, sizeof(double) * K * N));
    GPB_HIP(hipMalloc(&dC, sizeof(double) * M * N));
    GPB_HIP(hipMemcpy(dA, A_host, sizeof(double) * M * K, hipMemcpyHostToDevice));
    GPB_HIP(hipMemcpy(dB, B_host, sizeof(double) * K * N, hipMemcpyHostToDevice));
    int rc = launch_test_gemm(ctx, M, N, K, dA, dB, dC, b_trans);
    if (rc == 0) {
        GPB_HIP(hipStreamSynchronize(ctx->stream));
        GPB_HIP(hipMemcpy(C_host, dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    return rc;
}

#endif  // GPB_DEBUG_VARIANTS

// Launch-geometry knobs (never change a result); the key list is documented with the declaration in
// include/gpbayes_debug.h and mirrored by GPEngine.tune() in engine.py.
extern "C" int gpb_debug_has_variants(void) {
#ifdef GPB_DEBUG_VARIANTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int gpb_ctx_option(gpb_ctx* ctx, int key, int value) {
    if (!ctx) return GPB_E_ARG;
#ifndef GPB_DEBUG_VARIANTS
    // the measurement hooks (26, 32: one rank's share of a sharded step on a single GPU) exist in the debug build
    // (libgpbayes_debug.so, -DGPB_DEBUG_VARIANTS) and are refused here
    if ((key == 26 && value != 0) || (key == 32 && value != 0))
        GPB_FAIL(GPB_E_ARG, "gpb_ctx_option: this value selects a hook of the debug build only");
#endif
    switch (key) {
        case 0: if (value < -1 || value > 1) return GPB_E_ARG; ctx->force_xcd = value; break;
        case 4: if (value < 0 || value % 64) return GPB_E_ARG; ctx->chol_outer = value; break;
        case 5: if (value < 1 || value > 3) return GPB_E_ARG; ctx->resident_order = value; break;
        case 7: if (value < 0) return GPB_E_ARG; ctx->narrow_switch = value; break;
        case 8: if (value < 0) return GPB_E_ARG; ctx->mvn_wg_switch = value; break;
        case 9: if (value != 64 && value != 128) return GPB_E_ARG; ctx->chol_inner_tile = value; break;
        case 10: if (value < 0 || value > 1) return GPB_E_ARG; ctx->tile_priority = value; break;
        case 17: if (value < 0 || value > 1) return GPB_E_ARG; ctx->tri_skip = value; break;
        case 18:
            if (value < 0 || value > 2) return GPB_E_ARG;
            ctx->kcross_dot = value;
            if (ctx->have_theta) {                  // the forms follow at once; K(X,X) takes them at the next factorisation
                const int rc = choose_forms(ctx);
                if (rc) return rc;
            }
            break;
        case 19: if (value < 0 || value > 64) return GPB_E_ARG; ctx->kcross_chunks = value; break;
        case 20: if (value < 1 || value > 2) return GPB_E_ARG; ctx->kcross_wpl = value; break;
        case 23: if (value < 0 || value > 1) return GPB_E_ARG; ctx->lowrank = value; break;
        case 22: if (value < 0) return GPB_E_ARG; ctx->mid_switch = value; break;
        case 11: if (value < 0 || value > 1) return GPB_E_ARG; ctx->fuse_finalize = value; break;
        case 12: if (value != 0 && value != 64 && value != 128) return GPB_E_ARG; ctx->trtri_tile = value; break;
        case 14: if (value != 0 && value != 64 && value != 128) return GPB_E_ARG; ctx->syrk_tile = value; break;
        case 25: if (value < 0 || value > 1) return GPB_E_ARG; ctx->chol_lookahead = value; break;
        case 26: if (value < 0 || value > 64) return GPB_E_ARG; ctx->sim_ranks = value; break;
        case 27: if (value < 0 || value > 1) return GPB_E_ARG; ctx->compact = value; break;
        case 28: if (value < 0 || value > 1) return GPB_E_ARG; ctx->tile_by_live = value; break;
        case 29: if (value < 0 || value > 2) return GPB_E_ARG; ctx->premark = value; break;
        case 30: if (value < 0 || value > 1) return GPB_E_ARG; ctx->fuse_accept_propose = value; break;
        case 32: if (value < 0 || value > 63) return GPB_E_ARG; ctx->sim_rank = value; break;
        case 36: if (value < 0 || value > 2) return GPB_E_ARG; ctx->balance_shards = value; break;
        case 40: if (value < 0 || value > 1) return GPB_E_ARG; ctx->chain_batch = value; break;
        case 42:        // force the predict tile: 0 = by rule, 128 / 64 / 32 (= 64 x 32) / 65 (= 64 x 128) — all product shapes, same bits
            if (value != 0 && value != 32 && value != 64 && value != 65 && value != 128) return GPB_E_ARG;
            ctx->force_tile = value;
            break;
        case 43: if (value < 0 || value > 1) return GPB_E_ARG; ctx->force_generic_mvn = value != 0; break;
        case 47: if (value < 0 || value > 2) return GPB_E_ARG; ctx->chol_pair = value; break;
        case 49: if (value < 0 || value > 1) return GPB_E_ARG; ctx->lr_split = value; break;
        case 50: if (value != 0 && value != 64 && value != 128) return GPB_E_ARG; ctx->kinv_tile = value; break;
        case 51:        // V = L^-1 K*^T on the int8 matrix pipe (gpb_sliced.hip): 0 = never (default), 1 = where the rule admits, 2 = rule off
            if (value < 0 || value > 2) return GPB_E_ARG;
            ctx->predict_sliced = value;
            break;
        case 44: if (value < 0) return GPB_E_ARG; ctx->tile_switch = value > 0 ? value : 960; break;
        case 33: if (value < 0) return GPB_E_ARG; ctx->tile_switch_c = value; break;
        case 34: if (value < 0) return GPB_E_ARG; ctx->mid_switch_c = value; break;
        case 35: if (value < 0) return GPB_E_ARG; ctx->narrow_switch_c = value; break;
        default: return GPB_E_ARG;
    }
    return 0;
}

// Measurement hook: enqueue ONE piece of the fit on the context's stream (0 = K(X,X) assembly, 1 = blocked Cholesky of the
// K that is there, 2 = triangular inverse, 3 = alpha) so that bench.py and the profiling tools can time and profile the
// pieces apart.  The factorisation state is left invalid (call gpb_gp_factor afterwards).
extern "C" int gpb_profile_fit_piece(gpb_ctx* ctx, int piece) {
    if (!ctx) return GPB_E_ARG;
    if (!ctx->have_theta) GPB_FAIL(GPB_E_STATE, "gpb_profile_fit_piece before gpb_gp_set_theta");
    GPB_HIP(hipSetDevice(ctx->device));
    ctx->factored = false;
    switch (piece) {
        case 0: return launch_kmat(ctx);
        case 1: return launch_potrf(ctx);
        case 2: return launch_trtri(ctx);
        case 3: return launch_alpha(ctx);
        default: return GPB_E_ARG;
    }
}



extern "C" int gpb_profile_enable(gpb_ctx* ctx, int on) {
    if (!ctx) return GPB_E_ARG;
    ctx->profile = on != 0;
    return 0;
}

extern "C" int gpb_profile_read(gpb_ctx* ctx, int64_t* launches, double* total_ms, double* units) {
    if (!ctx || !launches || !total_ms || !units) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    for (auto& ev : ctx->prof_events) {
        float t = 0.f;
        GPB_HIP(hipEventElapsedTime(&t, ev.first, ev.second));
        ms += t;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (ctx->prof_compacted) {                       // compacted launches: the rows they evaluated were counted on the device
        unsigned long long live = 0;
        GPB_HIP(hipMemcpy(&live, ctx->rows_live, sizeof(live), hipMemcpyDeviceToHost));
        GPB_HIP(hipMemset(ctx->rows_live, 0, sizeof(live)));
        ctx->prof_units += (ctx->prof_gps > 0.0 ? ctx->prof_gps : (double)ctx->P) * (double)live;
        ctx->prof_compacted = false;
    }
    *launches = (int64_t)ctx->prof_events.size();
    *total_ms = ms;
    *units = ctx->prof_units;
    ctx->prof_events.clear();
    ctx->prof_units = 0.0;
    return 0;
}

#ifdef GPB_DEBUG_VARIANTS
// Debug hook: per-tile placement/timing records of k_predict (where the dispatcher put each tile, when it ran).
// capacity > 0 arms the trace (and clears it), capacity == 0 disarms.  read copies up to max_records records of
// 8 uint32 {HW_ID, XCC_ID, gp, row block, walker tile, t_start, t_end (100 MHz ticks), blockIdx} and re-arms.
extern "C" int gpb_debug_tile_trace(gpb_ctx* ctx, int64_t capacity) {
    if (!ctx || capacity < 0 || capacity > (1 << 22)) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    dev_free(&ctx->tile_trace);
    if (capacity == 0) return 0;
    const size_t n = 8 + 8 * (size_t)capacity;
    GPB_HIP(pool_malloc_t(&ctx->tile_trace, n * sizeof(unsigned)));
    GPB_HIP(hipMemset(ctx->tile_trace, 0, n * sizeof(unsigned)));
    const unsigned cap = (unsigned)capacity;
    GPB_HIP(hipMemcpy(ctx->tile_trace + 1, &cap, sizeof(unsigned), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int gpb_debug_tile_trace_read(gpb_ctx* ctx, uint32_t* records_host, int64_t max_records, int64_t* n_out) {
    if (!ctx || !records_host || !n_out || max_records < 0) return GPB_E_ARG;
    if (!ctx->tile_trace) GPB_FAIL(GPB_E_STATE, "gpb_debug_tile_trace_read: trace not armed");
    GPB_HIP(hipSetDevice(ctx->device));
    GPB_HIP(hipStreamSynchronize(ctx->stream));
    unsigned head[2] = {0, 0};
    GPB_HIP(hipMemcpy(head, ctx->tile_trace, sizeof(head), hipMemcpyDeviceToHost));
    int64_t n = head[0] < head[1] ? head[0] : head[1];
    if (n > max_records) n = max_records;
    if (n > 0) GPB_HIP(hipMemcpy(records_host, ctx->tile_trace + 8, (size_t)n * 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    GPB_HIP(hipMemset(ctx->tile_trace, 0, sizeof(unsigned)));      // count = 0: re-armed
    *n_out = n;
    return 0;
}

extern "C" int gpb_probe_fp64(gpb_ctx* ctx, int mode, double* tflops_out) {
    if (!ctx || !tflops_out || mode < 0 || mode > 4) return GPB_E_ARG;
    GPB_HIP(hipSetDevice(ctx->device));
    return launch_probe(ctx, mode, tflops_out);
}
#endif  // GPB_DEBUG_VARIANTS
