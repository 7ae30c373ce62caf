/*
 * gpbayes_debug.h — test and measurement hooks of libgpbayes_debug.so ONLY.
 *
 * NOT part of the drop-in boundary and NOT in the product library: libgpbayes.so exports include/gpbayes.h and nothing
 * else (tests/test_cabi_load.py pins both export lists).  The debug build is the same sources compiled with
 * -DGPB_DEBUG_VARIANTS: every entry point of gpbayes.h, plus the hooks below for tests/ (parity of internal pieces against
 * the oracle, the R > 1 step loop on one GPU) and tools/ (A/B measurements), plus the measured-and-rejected kernel variants
 * behind their gpb_ctx_option keys.
 */
#ifndef GPBAYES_DEBUG_H
#define GPBAYES_DEBUG_H

#include "gpbayes.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sampler internals (tests/test_gpu_sampler_step.py) ----------------------------- */
/* test hook: out_dev[i] = pi_step(i), the keyed permutation that shuffles the red/blue split */
GPB_API int gpb_test_split_perm(gpb_ctx* ctx, int64_t n, uint64_t seed, uint64_t step, int64_t* out_dev);
/* test hook: Philox4x32-10 on the device for n (key, counter) pairs: in_host[n][6] = k0, k1, c0, c1, c2, c3;
 * out_host[n][4].  Checked against Random123's known-answer vectors and the oracle's restatement. */
GPB_API int gpb_test_philox(gpb_ctx* ctx, int64_t n, const uint32_t* in_host, uint32_t* out_host);
/* test hook: every random number gpb_stretch_propose / gpb_stretch_accept use for (seed, step, half): member k of
 * the half gets u_z[k] (stretch factor draw), j[k] (index into the complementary half), u_acc[k] (accept draw);
 * perm[nwalkers] is the split permutation of the step (identity when randomize_split = 0).  Device outputs. */
GPB_API int gpb_test_stretch_draws(gpb_ctx* ctx, int64_t nwalkers, int half, uint64_t seed, uint64_t step,
                           int randomize_split, double* u_z_dev /*[nw/2]*/, int64_t* j_dev /*[nw/2]*/,
                           double* u_acc_dev /*[nw/2]*/, int64_t* perm_dev /*[nw]*/);

/* ---- micro-benchmarks / self-tests (device) --------------------------------------- */
/* C[M,N] = A*B through the f64 MFMA tile engine (K%16==0).  b_trans bits 0-1: 0 = A[M,K] B[K,N],
 * 1 = A[M,K] B[N,K]^T, 2 = A[K,M]^T B[K,N]; bit 2: 64x64 tiles instead of 128x128. */
GPB_API int gpb_test_gemm(gpb_ctx* ctx, int64_t M, int64_t N, int64_t K,
                  const double* A_host, const double* B_host, double* C_host, int b_trans);
/* gpb_ctx_option keys that exist in this build only (the product library returns GPB_E_ARG for them):
 * 2 = waves per predict tile (4 or 8); 5 = 0: ticket queues for a co-resident predict grid; 21 = 64-row predict tiles launch
 * static only when co-resident (0); 24 = Cholesky schedule (0: round 1's three launches per 64-column step); 26 / 32 =
 * gpb_emcee_run behaves like rank `32` of `26` ranks on a single GPU (evaluates that rank's share of every batch, still issues
 * the collective of a one-rank communicator): one rank's share of a sharded step, tools/gpu_shard_sim.py; 37 = 0: the 128x128
 * predict tile without the fragment read-ahead; 38 = folded pairs of predict row blocks; 39 = 0: K(X,X) by the difference-form
 * kernel for every GP; 41 = 64-row predict tiles staged by LDS-DMA from a k-major copy of L^-1; 46 = fusion probe: every predict tile
 * releases its partial sums and takes a ticket of its walker tile (1), and the last tile of a walker tile reads them back as the block
 * likelihood would (2) — what folding the likelihood into the predict kernel would add (tools/gpu_fusion_probe.py; results unchanged). */
/* test hook: make R contexts of ONE process (each with its own stream, each driven by its own host thread) the ranks 0 .. R-1 of
 * a loopback communicator: gpb_dist_allgather / the in-stream all-gathers of gpb_chain_emcee_run are then emulated on the
 * ranks' streams (events + device copies; the host threads meet inside the call, so every rank must make the same calls
 * concurrently).  What a one-GPU box can run of the R > 1 step loop — everything but the RCCL wire.  Release the group
 * (any member) before destroying its contexts. */
GPB_API int gpb_debug_loopback_group(gpb_ctx* const* ctxs, int R);
GPB_API int gpb_debug_loopback_release(gpb_ctx* ctx);
/* debug hook: per-tile placement and timing of the predict kernel.  capacity > 0 arms (and clears) a trace of
 * that many records, 0 disarms; read copies up to max_records records of 8 uint32 {HW_ID register, XCC_ID,
 * GP, row block, walker tile, start, end (100 MHz ticks), blockIdx} to the host and re-arms. */
GPB_API int gpb_debug_tile_trace(gpb_ctx* ctx, int64_t capacity);
GPB_API int gpb_debug_tile_trace_read(gpb_ctx* ctx, uint32_t* records_host, int64_t max_records, int64_t* n_out);
/* issue-rate probe: returns measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64
 * (mode 0), v_fma_f64 (mode 1) or both co-issued (mode 2); mode 3: shader cycles per MFMA (one wave
 * per SIMD); mode 4: shader clock in GHz held during the dense MFMA loop. */
GPB_API int gpb_probe_fp64(gpb_ctx* ctx, int mode, double* tflops_out);

#ifdef __cplusplus
}
#endif
#endif /* GPBAYES_DEBUG_H */
