/*
 * gpbayes_debug.h — test, tuning and measurement hooks of libgpbayes.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a reference interface.  The boundary a
 * maintainer binds is include/gpbayes.h; these entry points exist for tests/ (parity of internal
 * pieces against the oracle), tools/ (A/B measurements) and bench.py (HIP-event timing of the
 * dominant kernel).  None of them changes a result of the product path: the tuning keys select
 * launch geometry only.
 */
#ifndef GPBAYES_DEBUG_H
#define GPBAYES_DEBUG_H

#include "gpbayes.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sampler internals (tests/test_gpu_sampler_step.py) ----------------------------- */
/* test hook: out_dev[i] = pi_step(i), the keyed permutation that shuffles the red/blue split */
int gpb_test_split_perm(gpb_ctx* ctx, int64_t n, uint64_t seed, uint64_t step, int64_t* out_dev);
/* test hook: Philox4x32-10 on the device for n (key, counter) pairs: in_host[n][6] = k0, k1, c0, c1, c2, c3;
 * out_host[n][4].  Checked against Random123's known-answer vectors and the oracle's restatement. */
int gpb_test_philox(gpb_ctx* ctx, int64_t n, const uint32_t* in_host, uint32_t* out_host);
/* test hook: every random number gpb_stretch_propose / gpb_stretch_accept use for (seed, step, half): member k of
 * the half gets u_z[k] (stretch factor draw), j[k] (index into the complementary half), u_acc[k] (accept draw);
 * perm[nwalkers] is the split permutation of the step (identity when randomize_split = 0).  Device outputs. */
int gpb_test_stretch_draws(gpb_ctx* ctx, int64_t nwalkers, int half, uint64_t seed, uint64_t step,
                           int randomize_split, double* u_z_dev /*[nw/2]*/, int64_t* j_dev /*[nw/2]*/,
                           double* u_acc_dev /*[nw/2]*/, int64_t* perm_dev /*[nw]*/);

/* ---- micro-benchmarks / self-tests (device) --------------------------------------- */
/* C[M,N] = A*B through the f64 MFMA tile engine (K%16==0).  b_trans bits 0-1: 0 = A[M,K] B[K,N],
 * 1 = A[M,K] B[N,K]^T, 2 = A[K,M]^T B[K,N]; bit 2: 64x64 tiles instead of 128x128. */
int gpb_test_gemm(gpb_ctx* ctx, int64_t M, int64_t N, int64_t K,
                  const double* A_host, const double* B_host, double* C_host, int b_trans);
/* test/tuning hook: force the tile of the predict kernel (0 = automatic, 64, 128, 32 = 64 rows x 32
 * walkers, 65 = 64 rows x 128 walkers) and, when switch_tiles > 0, the number of 128x128 tiles per 256 CUs
 * from which the automatic choice uses them. */
int gpb_debug_force_tile(gpb_ctx* ctx, int tile, int64_t switch_tiles);
/* tuning hook for launch geometry (never changes results): key 0 = XCD affinity of the predict kernel
 * (-1 auto, 0 by walker tile, 1 by row block, 2 by GP, 3 by (GP, four row blocks) super-block: least
 * fabric traffic); 1 = persistent 64-tile workgroups per CU;
 * 2 = waves per tile (4 or 8); 3 = persistent workgroups per CU of the 128-tile 8-wave variant;
 * 4 = outer panel width of the blocked Cholesky (0, default: by size — one panel up to N = 2048, 256 beyond); 5 = tile order when every predict tile has its own
 * co-resident workgroup (0 ticket queues, 1 sorted, 2 snake over the CUs, 3 snake of pairs);
 * 6 = persistent 64x32-tile workgroups per CU;
 * 7 = 64x64 predict tiles when at least this many of them exist per 256 CUs, else 64x32; 8 = largest batch whose block log-likelihood
 * (PCA mode, 32 < M <= 64) runs one workgroup per walker instead of one wave per walker;
 * 9 = tile (64 or 128) of the K=64 trailing updates inside an outer Cholesky panel;
 * 10 = wave priority of predict tiles by K-loop length (0/1); 11 = the block log-likelihood kernels sum the
 * predict partials themselves instead of a separate finalize launch (0/1); 12 / 14 = tile of the
 * triangular-inverse levels / of the end-of-panel Cholesky updates (0 = by fill, 64, 128);
 * 13 = co-resident workgroups per CU assumed when choosing the static predict launch (0 = built-in table);
 * 16 = persistent 64x128-tile workgroups per CU; 17 = leave out the all-zero m-tiles of the predict kernel's
 * diagonal blocks (1, default) or multiply them like any other (0: A/B measurements);
 * 18 = cross-kernel distances as |a|^2 + |b|^2 - 2 a.b on centred coordinates (1, default) or as d differences (0);
 * 19 = 64-row chunks of the design per cross-kernel workgroup (0 = by grid size); 20 = walkers per lane there (1, 2);
 * 21 = 64-row predict tiles always launch static (1, default) or only when co-resident (0);
 * 22 = 64x128 predict tiles when at least this many of them exist per 256 CUs;
 * 23 = low-rank form of the block log-likelihood when it applies (1, default) or the dense M x M kernels (0);
 * 24 = Cholesky schedule: 1 (default) two launches per 64-column step with the next diagonal block fused into the
 * update, 0 = round 1's three launches per step; 25 = lookahead (1, default): the far part of a panel's trailing update
 * runs on a side stream underneath the next panel's chain; 26 = gpb_emcee_run behaves like one rank of `value` on a single
 * GPU (evaluates the first 1/value of every batch, still issues the collective of a one-rank communicator): measurement
 * of a rank's share of a sharded step (tools/gpu_shard_sim.py); 27 = gpb_logpost / gpb_emcee_run evaluate the rows inside
 * the prior box only (1, default) or every row (0: A/B measurements; same results); 28 = the tile-shape rule of a
 * compacted batch counts the tiles of the LIVE rows, estimated from the last finished compaction (1, default), or of
 * the whole batch (0); same results either way; 29 = gpb_chain_emcee_run's proposal kernel also takes the prior-box test of
 * the rank's rows (1: saves the marking kernel's launch) and gathers the rows inside the box (2, default: no compaction
 * kernel at all), or leaves both to the compaction's own kernels (0); 30 = with 29 at 2, the accept of a half-step and the
 * proposal of the next one are one launch (1, default) or two (0).  All of these give the same ensemble.  32 = which rank
 * of key 26's simulated ranks the hook plays (default 0); 33 / 34 / 35 = the tile-shape rule's switch points (128x128, 64x128,
 * 64x64; tiles per 256 CUs) for compacted batches; 36 = a sharded gpb_chain_emcee_run gives every rank an equal slice of
 * the ordered list of all rows inside the box (2: always, 1, default: from 8 ranks on) or the rows inside the box of a
 * contiguous share (0); 37 = the 128x128 predict tile reads the next k-group's LDS fragments before the current group's
 * MFMAs (1, default) or as the compiler orders them (0); 38 = the 64x32 / 64x64 predict tiles run as folded pairs of row
 * blocks, one equal-length K loop per workgroup (k_predict_fold: 1) or one tile per workgroup (0, default: measured, the
 * fold is not faster); 39 = K(X,X) by k_kmat_mfma (1, default: dot-product form, a.b on the matrix cores) or k_kmat (0);
 * 40 = the emulators of a chain whose designs pad to the same size share ONE predict launch (1, default) or launch one
 * after the other (0); set on the chain's first context; same results; 41 = the 64-row predict tiles stage their operands
 * by LDS-DMA from a k-major copy of L^-1 (1: debug build, measured 0-5 % slower, same bits) or through registers (0, default). */
int gpb_debug_tune(gpb_ctx* ctx, int key, int value);
/* 1 when the library was built with -DGPB_DEBUG_VARIANTS (libgpbayes_debug.so: every measured-and-rejected kernel variant
 * behind its tune key, for the sweeps in tools/ and the variant tests), 0 for the product library, whose gpb_debug_tune
 * refuses the values that would select such a variant (waves 8, ticket queues for 64-row tiles, the 128x128 tile without the
 * read-ahead, folded tiles, LDS-DMA tiles, difference-form distances, the earlier K-build kernel, round 1's Cholesky schedule). */
int gpb_debug_has_variants(void);
/* test hook: make R contexts of ONE process (each with its own stream, each driven by its own host thread) the ranks 0 .. R-1 of
 * a loopback communicator: gpb_dist_allgather / the in-stream all-gathers of gpb_chain_emcee_run are then emulated on the
 * ranks' streams (events + device copies; the host threads meet inside the call, so every rank must make the same calls
 * concurrently).  What a one-GPU box can run of the R > 1 step loop — everything but the RCCL wire.  Release the group
 * (any member) before destroying its contexts. */
int gpb_debug_loopback_group(gpb_ctx* const* ctxs, int R);
int gpb_debug_loopback_release(gpb_ctx* ctx);
/* measurement hook: enqueue one piece of gpb_gp_factor alone (0 = K(X,X) assembly, 1 = Cholesky, 2 = triangular inverse,
 * 3 = alpha) on the context's stream; leaves the context without a valid factorisation (call gpb_gp_factor afterwards) */
int gpb_debug_fit_piece(gpb_ctx* ctx, int piece);
/* debug hook: per-tile placement and timing of the predict kernel.  capacity > 0 arms (and clears) a trace of
 * that many records, 0 disarms; read copies up to max_records records of 8 uint32 {HW_ID register, XCC_ID,
 * GP, row block, walker tile, start, end (100 MHz ticks), blockIdx} to the host and re-arms. */
int gpb_debug_tile_trace(gpb_ctx* ctx, int64_t capacity);
int gpb_debug_tile_trace_read(gpb_ctx* ctx, uint32_t* records_host, int64_t max_records, int64_t* n_out);
/* test hook: route gpb_loglike through the generic LDS/HBM Cholesky instead of the register-resident
 * fast path (PCA mode, M <= 64) so that both implementations can be checked against each other. */
int gpb_debug_force_generic_mvn(gpb_ctx* ctx, int on);
/* HIP-event timing of the dominant kernel (k_predict: V = L^-1 K*^T + sum of squares) on the
 * context's stream.  read: number of timed launches, their summed duration, and the (GP, walker)
 * pairs they processed; resets the counters. */
int gpb_profile_enable(gpb_ctx* ctx, int on);
int gpb_profile_read(gpb_ctx* ctx, int64_t* launches, double* total_ms, double* units);
/* measurement hook: one step of gpb_chain_emcee_run (same arguments; pos / lp are copied, not advanced) as `reps` plain
 * calls and as `reps` replays of its HIP graph; milliseconds per step each.  The replay repeats one step index: it
 * measures launch overhead, it does not sample. */
int gpb_debug_graph_probe(gpb_ctx* const* ctxs, int E, const double* pos_dev, const double* lp_dev, int64_t nwalkers,
                          uint64_t seed, double a, const double* lo_dev, const double* hi_dev, double outside_value,
                          double inside_const, int reps, double* ms_plain, double* ms_graph);
/* issue-rate probe: returns measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64
 * (mode 0), v_fma_f64 (mode 1) or both co-issued (mode 2); mode 3: shader cycles per MFMA (one wave
 * per SIMD); mode 4: shader clock in GHz held during the dense MFMA loop. */
int gpb_probe_fp64(gpb_ctx* ctx, int mode, double* tflops_out);

#ifdef __cplusplus
}
#endif
#endif /* GPBAYES_DEBUG_H */
