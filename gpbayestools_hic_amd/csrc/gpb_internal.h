// gpb_internal.h — context layout and launch prototypes shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
#include "../../include/gpbayes.h"
#include "../../include/gpbayes_debug.h"

namespace gpb {

constexpr int NB = 64;          // Cholesky diagonal block / padding granule of N
constexpr int WPAD = 128;       // walker-batch padding granule (GEMM tile width)
constexpr int KX_CHUNK = 64;    // design points per kcross workgroup (mean partial granule)
constexpr int MAX_D = 64;

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
__host__ __device__ inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

}  // namespace gpb

// Which stored GP the compact slot `a` of a fit launch stands for, and that GP's design size.  A context normally holds P GPs
// over ONE design (gpb_gp_set: map = n = nullptr); gpb_gp_set_multi gives every GP its own design (the GPs of several
// emulators, or the restarts of a hyper-parameter search, in one batch: n = per-GP sizes, all padded to the same Np) and
// gpb_gp_lml_subset evaluates a subset of the stored GPs in the first slots of the workspaces (map = slot -> stored GP).
struct GpSel {
    const int* map;
    const int* n;
    int N;
    int64_t x_stride, xm_stride;   // per-GP strides of the design / its column means (0: one design for all GPs)
    __device__ __forceinline__ int q(int a) const { return map ? map[a] : a; }
    __device__ __forceinline__ int64_t Nq(int qq) const { return n ? (int64_t)n[qq] : (int64_t)N; }
};

// Where a design of N points sits among the Np = multiple of 64 stored rows: rows [pad_front, pad_front + N).  The padding goes in
// FRONT in whole 16-row units — the predict tiles' K-step: those leading zero rows of K*^T are skipped outright and their rows of V
// fall into the lightest row block — and the remainder (< 16 rows) behind.
__host__ __device__ __forceinline__ int64_t pad_front(int64_t Np, int64_t N) { return ((Np - N) / 16) * 16; }

struct LoopGroup;
struct gpb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // ---- GP state -------------------------------------------------------------
    int64_t N = 0, Np = 0, d = 0, dpad = 0, P = 0;
    int kind = 0;
    double alpha_reg = 0.0;
    bool have_theta = false, factored = false;
    double* h_theta = nullptr;     // host [P][d+2]
    bool multi = false;            // gpb_gp_set_multi: every GP has its own design (fit-only context: no predict / likelihood)
    int64_t Pstore = 0;            // GPs stored (P is the number a launch covers: Pstore, or the subset's size inside gpb_gp_lml_subset)
    int* gpN = nullptr;            // device [Pstore] design points per GP (multi)
    int* gpmap = nullptr;          // device [Pstore] slot -> stored GP of the current subset evaluation
    bool subset = false;           // inside gpb_gp_lml_subset
    std::vector<int> h_N;          // host [Pstore]
    std::vector<int> h_map;        // host [P] of the current subset
    GpSel sel() const {
        return GpSel{subset ? gpmap : nullptr, multi ? gpN : nullptr, (int)N, multi ? Np * dpad : 0, multi ? dpad : 0};
    }
    double* X = nullptr;           // [Np][dpad]   raw design (pad rows/cols zero); multi: [Pstore][Np][dpad]
    double* Xsc = nullptr;         // [P][Np][dpad] design / length_scale_p
    double* xmean = nullptr;       // [dpad] column means of the design (0 in the padding); multi: [Pstore][dpad]
    double* muS = nullptr;         // [P][dpad] xmean / length_scale_p
    double* Xc = nullptr;          // [P][Np][dpad] Xsc - muS: centred scaled design (dot-product form of k_kcross)
    double* dnorm = nullptr;       // [P][Np] squared norms of the rows of Xc
    // Distance form per GP (chosen from theta alone: gpb_gp_set_theta -> choose_forms).  0 = Gram form r^2 = |a|^2 + |b|^2 - 2 a.b
    // on the centred design (k_kcross<DOT>, k_kmat_mfma), 1 = difference form sum ((a_k - b_k))^2 on X / l as sklearn's cdist
    // computes it (sk:kernels.py:1556,1564,1711-1716).  The Gram form's cancellation costs ~eps (|a|^2 + |b|^2) absolute in
    // r^2: a GP whose S = sum_k (extent_k / l_k)^2 exceeds gram_limit takes the difference form.
    std::vector<double> h_ext;     // host [d] column extents (max - min) of the design; multi: [Pstore][d]
    std::vector<int> h_form;       // host [P]
    int* gpform = nullptr;         // device [P]
    int* kmtiles = nullptr;        // device [Np/64 (Np/64 + 1) / 2][2]: (row block, column block) of tile t of the lower block triangle (k_kmat_mfma)
    int n_diff = 0;                // GPs in the difference form
    double gram_limit = 1024.0;    // S above this: difference form (r^2 within ~1e-13 absolute, K within ~2e-13, below it)
    double* ls = nullptr;          // [P][dpad]    length scales (1 in pad columns)
    double* amp = nullptr;         // [P] c
    double* noise = nullptr;       // [P] sigma_n^2
    double* Z = nullptr;           // [P][Np]
    double* K = nullptr;           // [P][Np][Np]  K, overwritten by L (lower) in gp_factor
    double* Linv = nullptr;        // [P][Np][Np]
    // sliced-integer predict (gpb_sliced.hip, option key 51): int8 digit planes of L^-1 (made on first use after a factorisation)
    // and of the current K*^T batch, row / column scales
    int predict_sliced = 0;        // 0 = fp64 kernel always; 1 = the int8 kernel where its rule admits the context; 2 = rule off (tests)
    int8_t* slA = nullptr;         // [P][6][Np/16][Np128][16]
    int8_t* slB = nullptr;         // [P][6][Np/16][Wcap][16] (leading dimension of a batch: Wld)
    double* sl_scale = nullptr;    // rowscale [P][Np128] | colscale [P] | rowexp (int) [P][Np128]
    bool slA_valid = false;
    int64_t slB_cap = 0;
    bool batch_sliced = false;     // the current batch's K*^T exists as digit planes (launch_kcross), not as fp64
    bool want_kst = false;         // the caller of launch_kcross needs the fp64 K*^T itself (joint covariance)
    double* T = nullptr;           // [P][Np][Np]  workspace (trtri / K^-1)
    double* yv = nullptr;          // [P][Np]      L^-1 z
    double* alpha = nullptr;       // [P][Np]      K^-1 z
    double* apart = nullptr;       // [Np/256][P][Np] alpha partials
    int* info = nullptr;           // [P]
    double* lmlbuf = nullptr;      // [P][4]
    double* gpart = nullptr;       // gradient partials
    int64_t gpart_cap = 0;

    // ---- predict workspace ------------------------------------------------------
    int64_t Wcap = 0;              // padded capacity (multiple of WPAD)
    int64_t last_W = 0;            // rows of the most recent K*^T batch (gpb_gp_get GPB_GET_KSTAR)
    int64_t Wld = 0;               // leading dimension of the current batch's workspaces (set by launch_predict: the padded batch)
    double* Xs = nullptr;          // [Wcap][d] staged inputs (when caller passes host memory)
    double* estd = nullptr;        // [Wcap]
    double* KsT = nullptr;         // [P][Np][Wcap]
    double* mpart = nullptr;       // [Np/KX_CHUNK][P][Wcap]
    double* spart = nullptr;       // [Np/64][P][Wcap]  sum-of-squares partials per 64-row block
    double* mean_pc = nullptr;     // [P][Wcap]
    double* var_pc = nullptr;      // [P][Wcap]
    unsigned long long* live_hint = nullptr;   // pinned host memory: (batch rows << 32) | live rows of the last finished compaction
    gpb_ctx* hint_from = nullptr;  // whose live_hint sizes this context's tile rule (the chain's first emulator compacts)
    int fuse_accept_propose = 1;   // tune key 30: accept of a half-step + proposal of the next in one launch (needs premark 2)
    int balance_shards = 0;        // tune key 36: sharded C loop takes equal slices of the ordered live-row list (1: from 8 ranks on, 2: always; default off)
    int* bal_ws = nullptr;         // its flags / ranks / scatter lists
    int64_t bal_cap = 0;
    int sim_rank = 0;              // tune key 32: which rank of sim_ranks the measurement hook plays
    int tile_by_live = 1;          // tune key 28
    int premark = 2;               // tune key 29: the C-driven loop's proposal kernel takes the prior-box test (1) and gathers the rows inside (2)
    double* cmp_X = nullptr;       // the rows of the current batch inside the prior box, gathered in order [Wcap][chain ndim]
    int64_t cmp_X_cap = 0;
    int* cmp_idx = nullptr;        // compaction of a log-posterior batch to the rows inside the prior box: [0] = count, [4..] = row indices
    int compact = 1;               // gpb_logpost / gpb_emcee_run evaluate the rows inside the box only (the reference: src/mcmc.py:278-283)
    unsigned long long* rows_live = nullptr;   // device counter: rows evaluated by compacted launches while profiling
    bool prof_compacted = false;
    double* vbuf = nullptr;        // [P][Np][Wcap] V = L^-1 K*^T (covariance path only)
    int64_t vbuf_cap = 0;
    double* covbuf = nullptr;      // [P][Wc][Wc]
    int64_t covbuf_cap = 0;
    double* out_stage = nullptr;   // staging for host outputs
    int64_t out_cap = 0;

    // ---- emulator transform / likelihood ----------------------------------------
    int mode = 0;
    int64_t M = 0;
    bool have_transform = false, have_like = false;
    // host copies of the observable transform (PCA modes) for the low-rank form of the likelihood
    std::vector<double> h_A, h_mu, h_C0;
    // low-rank likelihood (gpb_like.hip, k_loglike_lowrank): C = C0 + A^T D A with C0 = C_trunc + C_exp fixed
    double* lr_R = nullptr;        // [16][16] upper-triangular R of  L0^-1 A^T = Q R  (zero padded)
    double* lr_v0 = nullptr;       // [16]     Q^T L0^-1 (mu - yexp)
    double lr_cperp = 0.0;         // |(I - Q Q^T) L0^-1 (mu - yexp)|^2
    double lr_logdet0 = 0.0;       // log det C0
    bool lr_ok = false;
    int lowrank = 1;               // use it when it applies (tune key 23)
    int lr_split = 1;              // option key 49: a chain's block likelihoods as one workgroup per (walker tile, emulator) + an ordered sum
    double* lr_blocks = nullptr;   // [E][Wcap] the emulators' blocks of a chain's batch (chain's first context)
    int64_t lr_blocks_cap = 0;
    double* A = nullptr;           // [P][M]
    double* mu = nullptr;          // [M]
    double* scale = nullptr;       // [M]
    double* C0 = nullptr;          // [M][M] cov_trunc (zeros if absent)
    double* yexp = nullptr;        // [M]
    double* Cexp = nullptr;        // [M][M]
    double* mvn_ws = nullptr;      // global fallback for M > 128: [Wcap][M][M]
    int64_t mvn_ws_cap = 0;
    int* notpd = nullptr;          // device counter
    long long* n_nan = nullptr;    // device counter: NaN log-probabilities seen by the stretch move's accept step
    double* mc_ws = nullptr;       // gpb_emcee_run: proposals q[nh][d], factor[nh], log-probabilities lpq[nh]
    int64_t mc_cap = 0;
    int sim_ranks = 0;             // measurement hook: gpb_emcee_run evaluates 1/sim_ranks of every batch (one rank's share)
    int num_cu = 256;               // multiprocessor count of the device
    int64_t narrow_switch = 1280;   // 64x64 tiles when at least this many of them exist per 256 CUs, else 64x32
    int chain_batch = 1;            // tune key 40: a chain's emulators of equal padded size share one predict launch
    unsigned* tile_counter = nullptr;   // 8 ticket queues (stride 16) + done counter [128]; re-armed by the kernel
    int64_t chol_outer = 0;        // outer panel width of the two-level blocked Cholesky (0 = chosen by size, gpb_chol.hip)
    int chol_pair = 1;             // option key 47: column pairs — every second trailing update by two columns at once (k_chol_update2):
                                   // 1 = where measured faster (1024 <= Np <= 3072), 2 = always, 0 = never
    int chol_lookahead = 1;        // far part of a panel's trailing update on a side stream, under the next panel's chain
    // ls / amp / noise / gpform / gpmap are carved out of ONE device block (thblk) so that a new theta goes up in ONE asynchronous
    // copy from its page-locked twin (h_thblk) — round 5: an LML evaluation made 6 blocking copies and 5 synchronisations
    double* thblk = nullptr;
    double* h_thblk = nullptr;
    size_t thblk_bytes = 0;
    double* h_res = nullptr;       // page-locked: [lmlbuf doubles | info ints] of an LML evaluation's read-back
    hipStream_t side_stream = nullptr;
    std::vector<hipEvent_t> chol_events;
    int syrk_tile = 0;              // tile of the end-of-panel trailing updates (0 = by fill, 64, 128)
    int trtri_tile = 0;            // tile of the triangular-inverse levels (0 = by fill, 64, 128)
    int kinv_tile = 0;             // option key 50: tile of K^-1 = L^-T L^-1 (LML gradient; 0 = by fill, 64, 128)
    int chol_inner_tile = 64;      // tile of the K=64 trailing updates inside an outer panel (64 or 128)
    int resident_order = 2;         // static 64-row predict launches: order of the tiles over the CUs, 1-3 (2 = snake)
    unsigned* tile_trace = nullptr; // debug hook: [count, capacity, pad x6][capacity][8] records of k_predict tiles
    int tile_priority = 1;         // k_predict: wave priority by K-loop length (s_setprio)
    int kcross_chunks = 0;         // 64-row chunks of the design per k_kcross workgroup (0 = by grid size)
    int kcross_wpl = 2;            // walkers per lane of k_kcross (1 or 2)
    int kcross_dot = 1;            // tune key 18: distance form of k_kcross / K(X,X): 1 = per GP by theta (see gpform), 0 = every GP the
                                   // difference form, 2 = every GP the Gram form (tests: what the rule protects against)
    int tri_skip = 1;              // k_predict: skip the all-zero half of the diagonal block's second half
    int force_tile = 0;           // test hook: 0 = auto, 64 / 128 / 32 (= 64x32) force the k_predict tile
    int force_xcd = -1;             // tuning hook: -1 auto, 0 / 1 = k_predict XCD affinity by walker tile / row block
    int64_t tile_switch = 960;      // use 128x128 tiles when at least this many of them exist per 256 CUs (measured)
    int64_t mid_switch = 1280;      // else 64x128 tiles when at least this many of THEM exist, else 64x64 / 64x32
    int64_t tile_switch_c = 2400, mid_switch_c = 1150, narrow_switch_c = 2400;   // the same for compacted batches (tune keys 33-35)
    bool force_generic_mvn = false; // test hook: bypass the register-resident MVN fast path
    int fuse_finalize = 1;          // block log-likelihood kernels sum the predict partials themselves (P <= 32)
    int64_t mvn_wg_switch = 768;   // batches up to this size use one workgroup per walker (32 < M <= 64)

    // ---- parameterTrafoPCA input map (gpb_pmap.hip) ------------------------------------
    int* pmap_int = nullptr;       // col_src[d_out] | group descriptors [G][6]
    double* pmap_tab = nullptr;    // [G][4 + maxpc][100]
    int64_t pmap_d_in = 0, pmap_d_out = 0;
    int pmap_groups = 0, pmap_maxpc = 0;

    // ---- profiling (HIP events around k_predict) ------------------------------------
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    double prof_units = 0.0;       // (GP, walker) pairs processed by the timed launches
    double prof_gps = 0.0;         // GPs per timed launch (a chain's batched launch: those of all its emulators)

    // ---- RCCL ---------------------------------------------------------------------
    void* comm = nullptr;
    int rank = 0, nranks = 1;
    struct LoopGroup* loop = nullptr;   // test hook (gpb_debug_loopback_group): R contexts of ONE process as the ranks of a communicator
};

#define GPB_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return GPB_E_HIP;                                                           \
        }                                                                               \
    } while (0)

#define GPB_FAIL(code, msg)                                                             \
    do {                                                                                \
        ctx->err = (msg);                                                               \
        return (code);                                                                  \
    } while (0)

namespace gpb {
// cache of freed device buffers (gpb_pool.hip): every pointer from pool_malloc must go back through pool_free
hipError_t pool_malloc(void** p, size_t bytes);
void pool_free(void* p);
void pool_trim();
template <typename T>
inline hipError_t pool_malloc_t(T** p, size_t bytes) { return pool_malloc(reinterpret_cast<void**>(p), bytes); }
// fit side (gpb_fit.hip)
int launch_scale_design(gpb_ctx* ctx);
int choose_forms(gpb_ctx* ctx, bool upload = true);    // gpb_api.hip: per-GP distance form from h_theta and the design's extents
int launch_kmat(gpb_ctx* ctx);
int launch_potrf(gpb_ctx* ctx);
int launch_trtri(gpb_ctx* ctx);
int launch_alpha(gpb_ctx* ctx);
int launch_lml_value(gpb_ctx* ctx);
int launch_lml_grad(gpb_ctx* ctx, double* grad_host);
// predict side (gpb_predict.hip)
int ensure_wcap(gpb_ctx* ctx, int64_t W);
// finalize = false leaves the mean / variance as partials (mpart, spart) for a consumer that sums them itself
// nrows_dev (device int, optional): the batch was compacted, only its first *nrows_dev rows are live
int launch_predict(gpb_ctx* ctx, const double* Xs_dev, int64_t W, bool need_var, bool finalize = true,
                   const int* nrows_dev = nullptr);
// its three phases, for callers that batch the middle one over several contexts (chains of emulators)
int launch_kcross(gpb_ctx* ctx, const double* Xs_dev, int64_t W, const int* nrows_dev, bool allow_planes = true);
int launch_kcross_group(gpb_ctx* const* ctxs, const double* const* Xs, int E, int64_t W, const int* nrows_dev);
int launch_param_maps(gpb_ctx* const* ctxs, int n, const double* X_dev, int64_t W);      // gpb_pmap.hip
int launch_vsq(gpb_ctx* const* ctxs, int E, int64_t W, const int* nrows_dev);
int launch_finalize(gpb_ctx* ctx, int64_t W, bool need_var);
// gpb_sliced.hip: the int8 form of launch_vsq's 128 x 128 launch for ONE context (rule: sliced_applies)
bool sliced_applies(const gpb_ctx* ctx);
int sliced_prepare(gpb_ctx* ctx);                    // buffers, and the planes of L^-1 after a new factorisation
const double* sliced_colscale(const gpb_ctx* ctx);   // device [P]: the power-of-two scale of each GP's K*^T digits
int launch_vsq_sliced(gpb_ctx* ctx, int64_t W, const int* nrows_dev, int kskip);
int launch_vsq_sliced_multi(gpb_ctx* const* ctxs, int E, int64_t W, const int* nrows_dev, int kskip);   // the GPs of E contexts, one launch
void sliced_free(gpb_ctx* ctx);
int sliced_read_kstar(gpb_ctx* ctx, int64_t p, int64_t pad, int64_t N, int64_t W, double* out);
constexpr int GPB_MAX_MULTI_GP = 96;      // GPs one batched launch can address (its table is a kernel argument)
int launch_predict_cov(gpb_ctx* ctx, const double* Xs_dev, int64_t W, double* cov_dev);
// likelihood (gpb_like.hip)
int launch_obs(gpb_ctx* ctx, int64_t W, const double* estd_dev, double* mean_dev, double* cov_dev);
// true when launch_loglike will take the block log-likelihood kernels that sum the partials themselves
bool loglike_fuses_finalize(const gpb_ctx* ctx, int64_t W);
// cmp_dev (optional): the batch was compacted by launch_compact: row w of the workspace is row cmp_dev[4 + w] of ll_dev
int launch_loglike(gpb_ctx* ctx, int64_t W, double* ll_dev, bool accumulate, bool from_partials,
                   const double* X_box = nullptr, const double* lo_dev = nullptr, const double* hi_dev = nullptr,
                   double outside = 0.0, double inside_const = 0.0, const int* cmp_dev = nullptr);
// ll[row] = outside for the rows outside the open box; the rows inside are gathered into ctx->cmp_X (in order), their
// indices and count into ctx->cmp_idx
int launch_compact(gpb_ctx* ctx, const double* X_dev, int64_t W, int64_t dx, const double* lo_dev, const double* hi_dev,
                   double outside, double* ll_dev, int premarked = 0);
int ensure_cmp_rows(gpb_ctx* ctx, int64_t dx);
int launch_mvn(gpb_ctx* ctx, const double* dY_dev, const double* cov_dev, int64_t W, int64_t M, double* ll_dev);
bool compaction_applies(const gpb_ctx* ctx);
// test hooks
int launch_test_gemm(gpb_ctx* ctx, int64_t M, int64_t N, int64_t K, const double* A, const double* B,
                     double* C, int b_trans);
int launch_probe(gpb_ctx* ctx, int mode, double* tflops);
}  // namespace gpb
