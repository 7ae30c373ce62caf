#!/usr/bin/env python3
"""
bench.py — headline benchmark: MCMC steps/s x walkers (walker log-posterior evaluations per second)
on BASELINE config 4 (2048 design points x 20 parameters x 64 observables, 10 GPs, 4096 walkers).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one emcee-equivalent stretch-move step of the whole ensemble = two half-ensemble
log-posterior batches through the HIP engine (propose -> GP predict -> fused MVN -> accept), all
resident in HBM.  With N > 1 the 4096 walkers are sharded over the ranks (strong scaling) and each
log-probability batch ends in one RCCL all-gather.  Prints ONE JSON line on rank 0.

The timed ensemble is BURNT-IN: walkers in a small ball around theta* (what the reference's run_mcmc holds after
re-seeding at its best points, src/mcmc.py:392-405), so that every proposal row lies inside the prior box and is
evaluated (asserted: >= 95 % over the timed region) — `value` counts only work that was done.  The run from walkers
spread uniformly over the box, where about half of the proposals leave the box and cost nothing (here as in the
reference, src/mcmc.py:275-283), is reported beside it as extras.uniform_start.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # public MI355X fp64 matrix figure; the microarch guide lists no fp64 row


def cpu_baseline(info, Xw, what):
    """The reference CPU path restated by the oracle, timed on the host cores on the half-ensemble batch `Xw` —
    proposal rows of the GPU run itself, so both sides see the same inside / outside mix (the reference, too, evaluates only the
    rows inside the box: src/mcmc.py:275-283) — in BOTH modes of the contract:
      faithful  what the reference really executes: the full W x W predictive covariance per GP over the rows inside the box, as
                sklearn forms it (sk:_gpr.py:454-460), then a Python loop of per-row dpotrf / dpotrs (src/mcmc.py:23-65, 293);
                this is `value`;
      lean      the same numbers without the waste: the diagonal of the variance only, one batched Cholesky for the MVN block.
    Each mode: whole half-ensemble calls, the median of >= 5 per-call rates with min / max; thread settings as the process sees them."""
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    d, P = info["d"], info["P"]
    kind = O.KIND_NAMES[{"RBF": "RBF", "Matern": "Matern", "Matern25": "Matern25"}[info["kernel_type"]]]
    oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], P, kind).fit(synth.fixed_theta(d, P))
    nrows = Xw.shape[0]
    inside = float(np.mean(np.all((Xw > info["lo"]) & (Xw < info["hi"]), axis=1)))
    yexp = info["yexp"]
    cexp = np.diag((0.05 * np.abs(yexp)) ** 2)

    def leg(faithful, budget_s, max_calls):
        rates, lp, t_start = [], None, time.time()
        while True:
            t0 = time.time()
            lp = O.log_prob(Xw, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e, faithful=faithful), yexp, cexp,
                            batched=not faithful)
            rates.append(nrows / (time.time() - t0))
            if len(rates) >= max_calls or (len(rates) >= 5 and time.time() - t_start >= budget_s):
                break
        r = sorted(rates)
        return {"value": r[len(r) // 2], "min": r[0], "max": r[-1], "calls": len(r), "seconds": time.time() - t_start}, lp

    faithful, lp = leg(True, 14.0, 7)
    lean, lp_lean = leg(False, 4.0, 9)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    threads = {"nproc": os.cpu_count(), "sched_affinity": cores,
               "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "OPENBLAS_NUM_THREADS": os.environ.get("OPENBLAS_NUM_THREADS"),
               "MKL_NUM_THREADS": os.environ.get("MKL_NUM_THREADS")}
    try:
        from threadpoolctl import threadpool_info
        threads["blas"] = [{k: i.get(k) for k in ("internal_api", "num_threads", "version")} for i in threadpool_info()]
    except Exception:
        pass
    lean["agrees_with_faithful"] = float(np.max(np.abs(lp_lean - lp) / np.maximum(np.abs(lp), 1e-300)))
    return {"value": faithful["value"], "unit": "walker-evals/s", "cores": cores, "kind": "port",
            "statistic": "median of per-call rates", "min": faithful["min"], "max": faithful["max"], "calls": faithful["calls"],
            "lean": dict(lean, unit="walker-evals/s", what="diag-only variance + batched MVN (oracle/gp_oracle.py): same numbers, no W x W covariance, no per-row LAPACK loop"),
            "threads": threads,
            "rows_inside_box_fraction": inside,
            "sample": f"{faithful['calls']} faithful + {lean['calls']} lean log_posterior calls on {what} ({nrows} rows, {inside:.3f} of them inside the "
                      f"prior box; faithful = W x W covariance per GP + per-row LAPACK MVN as the reference runs it), "
                      f"{faithful['seconds']:.1f} s + {lean['seconds']:.1f} s, numpy/scipy threaded BLAS"}, lp


def extras(chain4, emu4, info4, sustain_s=5.5, only_sustained=False):
    """The other two BASELINE metrics, measured on the same box (N=1 only): GP predict points/s on
    BASELINE config 2 (1024 design pts x 15 params, 10 GPs, 10 000 test points) and the fixed-theta fit
    (K build + Cholesky + L^-1 + alpha) at the design sizes of configs 2, 4 and 5, as Cholesky-equivalent GF/s
    (P N^3/3 flops / time) and with its own roofline block."""
    import torch
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain, flops_per_walker

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        t_heat = time.perf_counter()                      # clocks up before the timed repetitions (see the step loop's pre-heat)
        while time.perf_counter() - t_heat < 0.05:
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = {}
    # the dominant kernel on FULL batches (every row inside the prior box, as in a burnt-in ensemble): 2048-row
    # log-posterior calls of the timed configuration.  The timed region itself starts from walkers spread uniformly
    # over the box, where about half of every half-ensemble's proposals fall outside it and are not evaluated, so its
    # k_predict launches run on ~950-row batches with a partly filled last tile.
    eng4 = emu4._engine_ready()
    Xin = torch.as_tensor(synth.walkers(info4["W"], info4["d"], seed=synth.SEED + 9), device="cuda")
    lp = torch.empty(info4["W"], dtype=torch.float64, device="cuda")
    for _ in range(3):
        chain4.log_prob_device(Xin, lp)
    torch.cuda.synchronize()
    eng4.profile(True)
    for _ in range(10):
        chain4.log_prob_device(Xin, lp)
    n_l, ms_l, units_l = eng4.profile_read()
    eng4.profile(False)
    tf = units_l / n_l * float(info4["N"]) ** 2 / (ms_l / n_l * 1e-3) / 1e12
    out["k_predict_full_batch_cfg4"] = {"rows": info4["W"], "launches": n_l, "avg_launch_ms": ms_l / n_l, "achieved": tf,
                                        "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TFLOPS, "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                                        "what": "k_predict on 2048-row batches with every row inside the prior box "
                                                "(algorithmic flops N^2 per (GP, row))"}
    # The SUSTAINED rate of the headline's loop (a production chain is thousands of steps, src/mcmc.py:372-412; the headline's
    # timed region is 20 steps behind a 100-step pre-heat: a burst).  >= 5.5 s of continuous stretch-move steps through
    # gpb_chain_emcee_run — propose, 2048-row log-posterior batch, accept, twice per step — in blocks of 40 steps; after each
    # block the ensemble goes back to the same burnt-in ball (two device copies), so that over the whole run every proposal
    # row lies inside the prior box and every block does the same work.  Per block: wall time and the HIP-event time of its
    # k_predict launches; the whole-loop rate counts everything between the first block's start and the last block's end.
    from gpbayestools_hic_amd.sampler import StretchSampler
    nw4, blk = 2 * info4["W"], 40
    ball4 = float(min(1e-3, max(1e-13, 10.0 ** (-3.0 - 0.16 * (blk + 3)))))
    X04 = synth.walkers_ball(nw4, info4["xstar"], ball4, lo=info4["lo"], hi=info4["hi"])
    ss = StretchSampler(chain4, nw4, seed=2468)
    assert ss._resident_engine() is not None
    ss.run(X04, 0, status=10 ** 9, store=False)
    snap = ss._snapshot()
    ss.run(None, blk, status=10 ** 9, store=False)
    ss._restore(snap)
    torch.cuda.synchronize()
    blocks = []
    eng4.profile(True)
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        ss.run(None, blk, status=10 ** 9, store=False)          # (ends in one synchronisation: the NaN counter's read-out)
        ss._restore(snap, keep_counter=True)                    # the ensemble only: every block draws fresh proposals from the ball
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_l, ms_l, u_l = eng4.profile_read()
        blocks.append((t1 - t0, ms_l / max(n_l, 1), u_l / max(n_l, 1), n_l))
        if t1 - t_start >= sustain_s or len(blocks) >= 40000:
            break
    t_all = time.perf_counter() - t_start
    eng4.profile(False)
    del ss
    fr = lambda b: b[2] * float(info4["N"]) ** 2 / (b[1] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
    us = [b[1] * 1e3 for b in blocks]
    rows_in = sum(b[2] * b[3] for b in blocks) / (sum(b[3] for b in blocks) * info4["P"] * (nw4 // 2))
    tail = blocks[-max(len(blocks) // 4, 1):]
    out["sustained_cfg4"] = {
        "seconds": t_all, "steps": blk * len(blocks), "blocks": len(blocks), "steps_per_block": blk, "walkers": nw4,
        "walker_evals_per_s": nw4 * blk * len(blocks) / t_all,
        "ms_per_step": t_all / (blk * len(blocks)) * 1e3,
        "ms_per_step_first_block": blocks[0][0] / blk * 1e3, "ms_per_step_last_quarter": sum(b[0] for b in tail) / (blk * len(tail)) * 1e3,
        "rows_inside_box_fraction": rows_in,
        "k_predict_us_per_launch": {"first_block": us[0], "last_block": us[-1], "min": min(us), "max": max(us),
                                    "median": sorted(us)[len(us) // 2],
                                    "by_block": [round(u, 1) for u in us[::max(len(us) // 24, 1)]]},
        "k_predict_frac_of_peak": {"first_block": fr(blocks[0]), "last_block": fr(blocks[-1]),
                                   "last_quarter": sum(fr(b) for b in tail) / len(tail)},
        "what": "continuous stretch-move steps of the headline configuration through gpb_chain_emcee_run for >= %.1f s, in blocks " % sustain_s +
                "of 40 steps from the same burnt-in ball, each block with fresh random draws (the step counter runs on; rows_inside_box_fraction "
                "says how many proposal rows were evaluated); whole-loop "
                "rate = walkers x steps / wall time over all blocks incl. the two device copies that reset the ensemble; "
                "k_predict: HIP-event time per launch, by block (algorithmic flops N^2 per (GP, row))"}
    # The int8 variant of the dominant kernel (option key 51, csrc/gpb_sliced.hip; off by default: the headline above runs the
    # fp64 kernel).  Same full batches, same ball: time per launch, fp64-equivalent TF/s, fraction of the DENSE int8 MFMA peak
    # (2 x the bf16 figure: 5.0 POP/s) on the 21 digit products it executes, deviation of the log-posterior and of the
    # variances from the fp64 path, and the step loop with it switched on.
    try:
        INT8_PEAK_TOPS = 5000.0
        Xh512 = Xin[:512].cpu().numpy()
        m64, v64 = eng4.predict(Xh512)
        eng4.tune("predict_sliced", 1)
        lp_s = torch.empty_like(lp)
        for _ in range(3):
            chain4.log_prob_device(Xin, lp_s)
        torch.cuda.synchronize()
        eng4.profile(True); eng4.profile_read()
        for _ in range(10):
            chain4.log_prob_device(Xin, lp_s)
        n_s, ms_s, units_s = eng4.profile_read()
        eng4.profile(False)
        ms8, v8 = eng4.predict(Xh512)
        fin = torch.isfinite(lp)
        t_s = ms_s / max(n_s, 1) * 1e-3
        alg = units_s / max(n_s, 1) * float(info4["N"]) ** 2
        ss8 = StretchSampler(chain4, nw4, seed=2468)
        ss8.run(X04, 0, status=10 ** 9, store=False)
        snap8 = ss8._snapshot()
        ss8.run(None, blk, status=10 ** 9, store=False)
        ss8._restore(snap8, keep_counter=True)
        torch.cuda.synchronize()
        tb = []
        for _ in range(5):
            t0 = time.perf_counter()
            ss8.run(None, blk, status=10 ** 9, store=False)
            ss8._restore(snap8, keep_counter=True)
            torch.cuda.synchronize()
            tb.append((time.perf_counter() - t0) / blk)
        del ss8
        tb.sort()
        out["k_predict_sliced_cfg4"] = {
            "rows": info4["W"], "launches": n_s, "avg_launch_ms": t_s * 1e3,
            "fp64_kernel_avg_launch_ms": ms_l / n_l, "speedup_vs_fp64_kernel": (ms_l / n_l) / (t_s * 1e3),
            "fp64_equivalent_tflops": alg / t_s / 1e12,
            "int8_tops_on_21_products": 21.0 * alg / t_s / 1e12, "int8_peak_tops": INT8_PEAK_TOPS,
            "frac_of_int8_peak": 21.0 * alg / t_s / 1e12 / INT8_PEAK_TOPS,
            "max_rel_dev_log_posterior_vs_fp64_path": float(torch.max(torch.abs(lp_s[fin] - lp[fin]) / torch.abs(lp[fin]))),
            "max_rel_dev_variance_vs_fp64_path": float(np.max(np.abs(v8 - v64) / np.abs(v64))),
            "max_rel_dev_mean_vs_fp64_path": float(np.max(np.abs(ms8 - m64)) / np.max(np.abs(m64))),
            "step_loop_ms_per_step": tb[len(tb) // 2] * 1e3, "step_loop_walker_evals_per_s": nw4 / tb[len(tb) // 2],
            "fp64_step_loop_ms_per_step": out["sustained_cfg4"]["ms_per_step"],
            "what": "V = L^-1 K*^T on v_mfma_i32_32x32x32_i8: six signed 8-bit digit planes per operand, the 21 digit products of "
                    "levels 5..10 summed exactly in int32, combined in fp64 (profiles/r06_sliced_model.txt); K*^T leaves k_kcross as "
                    "digit planes (no fp64 copy, no second pass); rule: every GP of the context has 1 + c / sn2 <= 128, else the fp64 "
                    "kernel.  avg_launch_ms: HIP events round the predict launch of 2048-row batches (every row inside the box); "
                    "step loop: 5 blocks of 40 stretch-move steps of the headline configuration, median"}
        eng4.tune("predict_sliced", 0)
    except Exception as e:      # noqa: BLE001  (a variant's failure must not cost the line)
        out["k_predict_sliced_cfg4"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            eng4.tune("predict_sliced", 0)
        except Exception:
            pass
    if only_sustained:
        return out

    # cfg 4's second shape ("also report P = 64 via perform_no_PCA", src/emulator.py:562-565,589-592): one GP per
    # observable — 64 GPs of N = 2048, 6.4 x the predict work per walker, the dense 64 x 64 likelihood kernels instead of the
    # low-rank form.  Same walkers, same ball, the same C-driven step loop.
    try:
        chain64, emu64, info64 = build_chain(4, no_pca=True)
        eng64 = emu64._engine_ready()
        lp64 = torch.empty(info64["W"], dtype=torch.float64, device="cuda")
        for _ in range(2):
            chain64.log_prob_device(Xin, lp64)
        torch.cuda.synchronize()
        t_batch = timed(lambda: chain64.log_prob_device(Xin, lp64), 5)
        eng64.profile(True); eng64.profile_read()
        for _ in range(5):
            chain64.log_prob_device(Xin, lp64)
        n6, ms6, u6 = eng64.profile_read()
        eng64.profile(False)
        X064 = synth.walkers_ball(nw4, info64["xstar"], ball4, lo=info64["lo"], hi=info64["hi"])
        s64 = StretchSampler(chain64, nw4, seed=2468)
        s64.run(X064, 0, status=10 ** 9, store=False)
        snap64 = s64._snapshot()
        s64.run(None, 4, status=10 ** 9, store=False)
        s64._restore(snap64, keep_counter=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s64.run(None, 10, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / 10
        tf6 = u6 / max(n6, 1) * float(info64["N"]) ** 2 / (ms6 / max(n6, 1) * 1e-3) / 1e12
        out["cfg4_nopca_p64"] = {
            "gps": info64["P"], "observables": info64["M"], "walkers": nw4,
            "ms_per_step": t_step * 1e3, "walker_evals_per_s": nw4 / t_step,
            "batch_2048_rows_ms": t_batch * 1e3,
            "k_predict_avg_launch_ms": ms6 / max(n6, 1), "k_predict_tflops": tf6, "k_predict_frac": tf6 / FP64_MFMA_PEAK_TFLOPS,
            "outside_predict_launch_ms_per_batch": t_batch * 1e3 - ms6 / max(n6, 1),
            "tflops_algorithmic": flops_per_walker(info64["N"], info64["d"], info64["P"], info64["M"], "RBF") * nw4 / t_step / 1e12,
            "what": "cfg 4 with perform_no_PCA: 64 GPs (one per observable) of N = 2048, 4096 walkers; ms_per_step over 10 steps of "
                    "the C-driven loop from the burnt-in ball; k_predict: HIP events round the 64-GP launch of a 2048-row batch; "
                    "outside_predict_launch = K*^T (k_kcross) + the dense 64 x 64 likelihood kernel (k_loglike_reg/wg) + the rest "
                    "of one 2048-row log-posterior batch"}
        del s64, chain64, emu64, eng64
    except Exception as e:      # noqa: BLE001
        out["cfg4_nopca_p64"] = {"error": "%s: %s" % (type(e).__name__, e)}

    _, emu2, info2 = build_chain(2)
    eng2 = emu2._engine_ready()
    Xs = torch.as_tensor(synth.walkers(10000, info2["d"]), device="cuda")
    t = timed(lambda: eng2.predict(Xs), 5)
    out["gp_predict_cfg2"] = {"points": 10000, "gps": info2["P"], "ms": t * 1e3, "points_per_s": 10000 / t,
                              "what": "mean + variance of all 10 GPs per point, inputs/outputs resident in HBM"}
    Xh = synth.walkers(10000, info2["d"])
    ths = []
    for _ in range(4):          # the first call also fills the page-locked host cache the 82 MB result comes from
        t0 = time.perf_counter(); res = emu2.predict(Xh, return_cov=True, extra_std=0.0); ths.append(time.perf_counter() - t0)
        del res
    th = sorted(ths[1:])[1]
    out["emulator_predict_cfg2_host"] = {"points": 10000, "ms": th * 1e3, "points_per_s": 10000 / th, "first_call_ms": ths[0] * 1e3,
                                         "what": "Emulator.predict(return_cov=True): numpy in, mean[W,32] + cov[W,32,32] out (PCIe "
                                                 "inclusive; median of three calls after the first)"}
    def fit_entry(eng, Nn, Pp, kernel):
        """fit at fixed theta: K build + blocked Cholesky + L^-1 + alpha for all GPs (gpb_gp_factor).  Roofline block:
        algorithmic flops = P (N^3/3 [Cholesky] + N^3/3 [triangular inverse]) against the fp64 MFMA peak; the chain of
        N/64 dependent diagonal-block steps (k_chol_update / k_chol_trsm, gpb_chol.hip) is what bounds it."""
        tf = timed(lambda: eng.factor(), 5)
        both = 2.0 * Pp * Nn ** 3 / 3.0
        return {"N": Nn, "gps": Pp, "kernel": kernel, "ms": tf * 1e3,
                "cholesky_equiv_gflops": Pp * Nn ** 3 / 3 / tf / 1e9,
                "roofline": {"bound": "mfma", "kernel": "k_chol_update + k_chol_trsm (chain), k_syrk, k_trtri_level",
                             "flops": both, "achieved": both / tf / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": both / tf / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "frac_cholesky_only": Pp * Nn ** 3 / 3 / tf / 1e12 / FP64_MFMA_PEAK_TFLOPS},
                "what": "K build + blocked Cholesky + L^-1 + alpha for all GPs; flops = P (N^3/3 + N^3/3)"}

    for tag, emu, info in (("cfg2", emu2, info2), ("cfg4", emu4, info4)):
        out[f"fit_fixed_theta_{tag}"] = fit_entry(emu._engine_ready(), info["N"], info["P"], info["kernel"])
    from gpbayestools_hic_amd import GPEngine
    c5 = synth.CONFIGS[5]
    eng5 = GPEngine(torch.cuda.current_device())
    eng5.set_data(synth.lhs(c5["N"], c5["d"]), np.random.default_rng(1).standard_normal((c5["P"], c5["N"])),
                  c5["kernel"], 0.1)
    eng5.set_theta(synth.fixed_theta(c5["d"], c5["P"]))
    out["fit_fixed_theta_cfg5"] = fit_entry(eng5, c5["N"], c5["P"], c5["kernel"])

    # the regime train_emulators itself factors in: the 63 GPs of nine emulators (each over its own 1000-point design, all padded
    # to Np = 1024) side by side on ONE gpb_gp_set_multi context — 63 matrices per launch of the Cholesky chain
    # (src/emulator.py:309-315 for each of the nine data sets, examples/EmulatorTraining.ipynb:124-138); flops counted at N = 1000
    nb_gp, nb_N, nb_d = 63, 1000, 20
    engb = GPEngine(torch.cuda.current_device())
    rngb = np.random.default_rng(7)
    engb.set_data_multi([synth.lhs(nb_N, nb_d, seed=synth.SEED + 300 + i % 9) for i in range(nb_gp)],
                        [rngb.standard_normal(nb_N) for _ in range(nb_gp)], "RBF", 0.1)
    engb.set_theta(synth.fixed_theta(nb_d, nb_gp))
    out["fit_fixed_theta_batch63"] = fit_entry(engb, nb_N, nb_gp, "RBF")
    out["fit_fixed_theta_batch63"]["what"] += ("; 63 GPs over nine different 1000-point designs on one gpb_gp_set_multi context "
                                                "(the batch train_emulators factors per lock-step round), Np = 1024")
    engb.close()

    def k_build_entry(eng, Nn, dd, Pp, kernel):
        """K(X,X) + (noise + alpha) I of all GPs (k_kmat_mfma): the lower block triangle is all the factorisation reads.
        Bound: the slower of the HBM write of 4 N^2 bytes per GP and the fp64 vector work of N^2 / 2 pairs at
        F_pair = 3d + 3 (RBF) / 3d + 10 (Matern-5/2) flops."""
        t = timed(lambda: eng.fit_piece("kmat"), 20)
        eng.factor()
        byts = 4.0 * Nn * Nn * Pp
        fpair = 3 * dd + 3 if kernel == "RBF" else 3 * dd + 10
        flops = 0.5 * Nn * Nn * fpair * Pp
        t_hbm, t_valu = byts / 8e12, flops / (FP64_MFMA_PEAK_TFLOPS * 1e12)
        bound = "hbm" if t_hbm >= t_valu else "fp64 valu"
        return {"N": Nn, "d": dd, "gps": Pp, "kernel": kernel, "us": t * 1e6,
                "roofline": {"bound": bound, "kernel": "k_kmat_mfma", "bytes": byts, "flops": flops,
                             "achieved": byts / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": max(t_hbm, t_valu) / t,
                             "frac_note": "time at the binding limit (max of 4 N^2 P bytes at 8 TB/s and N^2/2 F_pair P "
                                          "flops at 78.6 TF/s) / measured time",
                             "valu_tflops_algorithmic": flops / t / 1e12},
                "what": "kernel-matrix assembly for all GPs, lower block triangle (what the Cholesky reads)"}

    out["k_build_cfg5"] = k_build_entry(eng5, c5["N"], c5["d"], c5["P"], c5["kernel"])
    eng5.close()
    out["k_build_cfg4"] = k_build_entry(eng4, info4["N"], info4["d"], info4["P"], info4["kernel"])

    # log-marginal likelihood + gradient of all GPs at one theta (what every L-BFGS-B iteration of the hyper-parameter
    # search costs: sk:_gpr.py:537-652): K build, Cholesky, L^-1, alpha, K^-1 = L^-T L^-1, the d + 2 derivative reductions
    th4 = synth.fixed_theta(info4["d"], info4["P"])
    t = timed(lambda: eng4.lml(th4), 5)
    fpair4 = 3 * info4["d"] + 3 if info4["kernel"] == "RBF" else 3 * info4["d"] + 10
    fl = info4["P"] * (float(info4["N"]) ** 3 + float(info4["N"]) ** 2 * (fpair4 + 2 * (info4["d"] + 2)))
    fl_exec = 4.0 / 3.0 * info4["N"] ** 3 * info4["P"]
    out["lml_grad_cfg4"] = {"N": info4["N"], "d": info4["d"], "gps": info4["P"], "ms": t * 1e3,
                            "roofline": {"bound": "mfma", "kernel": "gpb_gp_lml: k_kmat_mfma, Cholesky chain, k_trtri_level, k_kinv, k_lml_grad",
                                         "flops": fl, "achieved": fl / t / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": fl / t / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                         "frac_executed_flops": fl_exec / t / 1e12 / FP64_MFMA_PEAK_TFLOPS},
                            "what": "one LML + gradient evaluation of all GPs incl. the download of value and gradient; "
                                    "algorithmic flops per SURVEY 8(d) = P (N^3 + N^2 (F_pair + 2 (d + 2))) (Cholesky + inverse counted "
                                    "as N^3); frac_executed_flops counts what the path executes, P 4/3 N^3 (N^3/3 Cholesky + N^3/3 "
                                    "triangular inverse + 2 N^3/3 K^-1 = L^-T L^-1)"}
    emu4._engine_ready().set_theta(emu4.thetas_)
    emu4._engine_ready().factor()

    # BASELINE config 3 (1 x MI355X: 1024 design pts x 15 params, emcee stretch move with 1024 walkers, src/mcmc.py:372-412):
    # the resident step loop on a burnt-in ensemble, every proposal row evaluated, as the headline is measured
    from gpbayestools_hic_amd.sampler import StretchSampler
    chain3, emu3, info3 = build_chain(3)
    eng3 = emu3._engine_ready()
    nw3, warm3, nst3 = 2 * info3["W"], 3, 40
    ball3 = float(min(1e-3, max(1e-13, 10.0 ** (-3.0 - 0.16 * (warm3 + nst3)))))
    X03 = synth.walkers_ball(nw3, info3["xstar"], ball3, lo=info3["lo"], hi=info3["hi"])
    heat = StretchSampler(chain3, nw3, seed=4242)
    heat.run(X03, 100, status=10 ** 9, store=False)
    del heat
    s3 = StretchSampler(chain3, nw3, seed=12345)
    assert s3._resident_engine() is not None
    s3.run(X03, warm3, status=10 ** 9, store=False)
    eng3.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s3.run(None, nst3, status=10 ** 9, store=False)
    torch.cuda.synchronize()
    dt3 = (time.perf_counter() - t0) / nst3
    n_l, ms_l, u_l = eng3.profile_read()
    eng3.profile(False)
    inside3 = u_l / (n_l * info3["P"] * (nw3 // 2)) if n_l else None
    mf3 = flops_per_walker(info3["N"], info3["d"], info3["P"], info3["M"], info3["kernel"])
    tfk3 = u_l / n_l * float(info3["N"]) ** 2 / (ms_l / n_l * 1e-3) / 1e12 if n_l else None
    out["cfg3_step"] = {"N": info3["N"], "d": info3["d"], "gps": info3["P"], "observables": info3["M"], "walkers": nw3,
                        "steps": nst3, "warmup": warm3, "ms_per_step": dt3 * 1e3, "walker_evals_per_s": nw3 / dt3,
                        "rows_inside_box_fraction": inside3, "acceptance_fraction": float(s3.acceptance_fraction.mean()),
                        "mflop_per_walker_algorithmic": mf3 / 1e6,
                        "tflops_algorithmic": mf3 * nw3 * (inside3 or 1.0) / dt3 / 1e12,
                        "frac_of_peak_whole_step": mf3 * nw3 * (inside3 or 1.0) / dt3 / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                        "roofline": {"bound": "mfma", "kernel": "k_predict (512-row batches)", "achieved": tfk3,
                                     "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": tfk3 / FP64_MFMA_PEAK_TFLOPS if tfk3 else None, "launches": n_l,
                                     "avg_launch_ms": ms_l / max(n_l, 1)},
                        "what": "BASELINE config 3: stretch-move steps of 1024 walkers (two 512-row log-posterior batches each) through "
                                "gpb_chain_emcee_run, burnt-in ensemble, behind 100 untimed pre-heat steps; algorithmic flops per SURVEY "
                                "8(d) (11.05 MF per walker); roofline = k_predict's N^2 per (GP, row) over its HIP-event times"}
    del s3
    # the same steps with the int8 predict kernel (option key 51; 512-row batches: its 128x64 tiles)
    try:
        emu3.set_predict_arithmetic("int8")
        s3 = StretchSampler(chain3, nw3, seed=12345)
        s3.run(X03, warm3, status=10 ** 9, store=False)
        eng3.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s3.run(None, nst3, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        dt3s = (time.perf_counter() - t0) / nst3
        n_s, ms_s, _u = eng3.profile_read()
        eng3.profile(False)
        del s3
        out["cfg3_step"]["int8_predict"] = {"ms_per_step": dt3s * 1e3, "walker_evals_per_s": nw3 / dt3s,
                                            "speedup_vs_fp64_path": dt3 / dt3s, "k_predict_sliced_avg_launch_ms": ms_s / max(n_s, 1)}
        emu3.set_predict_arithmetic("fp64")
    except Exception as e:      # noqa: BLE001
        out["cfg3_step"]["int8_predict"] = {"error": "%s: %s" % (type(e).__name__, e)}
    emu3._engine.close()

    # BASELINE config 5 (pocoMC: 8192 particles, Matern-5/2, 4096-pt design): the call pocoMC makes,
    # log_likelihood(X[8192, d], finite=True) with vectorize=True (src/mcmc.py:798-805) — device-resident and through numpy
    chain5, emu5, info5 = build_chain(5)
    eng5c = emu5._engine_ready()
    W5 = info5["W"]
    Xh5 = synth.walkers(W5, info5["d"], seed=synth.SEED + 11)             # pocoMC's particles lie inside the prior box
    Xd5 = torch.as_tensor(Xh5, device="cuda")
    lp5 = torch.empty(W5, dtype=torch.float64, device="cuda")
    for _ in range(2):
        chain5.log_prob_device(Xd5, lp5, outside=-1e300)
    torch.cuda.synchronize()
    t5 = timed(lambda: chain5.log_prob_device(Xd5, lp5, outside=-1e300), 5)
    eng5c.profile(True)                                   # (the HIP events of five more batches: the roofline block's launches)
    for _ in range(5):
        chain5.log_prob_device(Xd5, lp5, outside=-1e300)
    n_l, ms_l, u_l = eng5c.profile_read()
    eng5c.profile(False)
    ths = []
    for _ in range(3):
        t0 = time.perf_counter(); ll5 = chain5.log_likelihood(Xh5, finite=True); ths.append(time.perf_counter() - t0)
    assert ll5.shape == (W5,) and np.all(np.isfinite(ll5)) and np.all(ll5 > -1e299)
    mf5 = flops_per_walker(info5["N"], info5["d"], info5["P"], info5["M"], info5["kernel"])
    tfk5 = u_l / n_l * float(info5["N"]) ** 2 / (ms_l / n_l * 1e-3) / 1e12 if n_l else None
    out["cfg5_batch"] = {"N": info5["N"], "d": info5["d"], "gps": info5["P"], "observables": info5["M"], "kernel": info5["kernel"],
                         "rows_per_batch": W5, "ms_per_batch": t5 * 1e3, "batches_per_s": 1.0 / t5, "rows_per_s": W5 / t5,
                         "host_call_ms_per_batch": sorted(ths)[1] * 1e3, "host_call_rows_per_s": W5 / sorted(ths)[1],
                         "mflop_per_row_algorithmic": mf5 / 1e6, "tflops_algorithmic": mf5 * W5 / t5 / 1e12,
                         "frac_of_peak_whole_batch": mf5 * W5 / t5 / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "roofline": {"bound": "mfma", "kernel": "k_predict (8192-row batches, N = 4096)", "achieved": tfk5,
                                      "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tfk5 / FP64_MFMA_PEAK_TFLOPS if tfk5 else None, "launches": n_l,
                                      "avg_launch_ms": ms_l / max(n_l, 1)},
                         "what": "BASELINE config 5: Chain.log_likelihood(X[8192, 20], finite=True) batches as pocoMC calls them "
                                 "(vectorize=True), every row inside the prior box; ms_per_batch: rows and results resident in HBM; "
                                 "host_call: numpy in, numpy out (PCIe inclusive, median of three); algorithmic flops per SURVEY 8(d) "
                                 "(171.0 MF per row)"}
    # the same batches with the int8 predict kernel (option key 51)
    try:
        emu5.set_predict_arithmetic("int8")
        lp5s = torch.empty(W5, dtype=torch.float64, device="cuda")
        for _ in range(2):
            chain5.log_prob_device(Xd5, lp5s, outside=-1e300)
        torch.cuda.synchronize()
        t5s = timed(lambda: chain5.log_prob_device(Xd5, lp5s, outside=-1e300), 5)
        dev5 = float(torch.max(torch.abs(lp5s - lp5) / torch.abs(lp5)).item())
        out["cfg5_batch"]["int8_predict"] = {"ms_per_batch": t5s * 1e3, "rows_per_s": W5 / t5s, "speedup_vs_fp64_path": t5 / t5s,
                                             "max_rel_dev_log_posterior_vs_fp64_path": dev5}
        emu5.set_predict_arithmetic("fp64")
    except Exception as e:      # noqa: BLE001
        out["cfg5_batch"]["int8_predict"] = {"error": "%s: %s" % (type(e).__name__, e)}
    emu5._engine.close()

    # training of the emulators of such a chain with their full hyper-parameter searches (src/emulator.py:286-315 for each of the
    # nine data sets, examples/EmulatorTraining.ipynb:124-138): one after the other against train_emulators, which runs the 63
    # searches in ONE lock-step batch (gpb_gp_set_multi / gpb_gp_lml_subset) — same theta*, bit for bit
    from gpbayestools_hic_amd import Emulator
    from gpbayestools_hic_amd.emulator import train_emulators
    import tempfile

    def nine(wd):
        pf = os.path.join(wd, "par.txt")
        synth.write_parameter_file(pf, np.zeros(20), np.ones(20))
        emus = []
        for i in range(9):
            Xi = synth.lhs(1000, 20, seed=synth.SEED + 100 + i)
            tp = os.path.join(wd, "t%d.pkl" % i)
            synth.write_training_pickle(tp, Xi, synth.observables(Xi, 60, seed=synth.SEED + 200 + i), 0.01)
            emus.append(Emulator(training_set_path=tp, parameter_file=pf, npc=6 + i % 3))
        return emus
    seq, tog = nine(tempfile.mkdtemp(prefix="gpb_tr_")), nine(tempfile.mkdtemp(prefix="gpb_tr_"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for em in seq:
        em.trainEmulatorAutoMask()
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    train_emulators(tog)
    torch.cuda.synchronize()
    t_tog = time.perf_counter() - t0
    out["train_nine_emulators"] = {"emulators": 9, "design_points_each": 1000, "params": 20, "gps": sum(em._ngp for em in seq),
                                   "one_after_the_other_s": t_seq, "train_emulators_s": t_tog, "speedup": t_seq / t_tog,
                                   "theta_identical": bool(all(np.array_equal(a.thetas_, b.thetas_) for a, b in zip(seq, tog))),
                                   "what": "full hyper-parameter searches (L-BFGS-B on the device log-marginal likelihood, no restarts) of nine "
                                           "emulators: Emulator.trainEmulatorAutoMask one after the other / train_emulators (all 63 GPs in one "
                                           "lock-step batch, converged searches leaving it); scaler + PCA on the host included"}
    for em in seq + tog:
        em._engine.close()

    # a chain at the size of the reference's real analyses: nine emulators with their own designs, kernels and numbers of
    # GPs over one 20-parameter space, 540 observables, block-diagonal covariance (src/mcmc.py:153-166,
    # examples/RunBayesianAnalysis.ipynb:35-48), 4096 walkers, the whole step loop in gpb_chain_emcee_run
    from gpbayestools_hic_amd.workload import build_multi_chain
    specs = [(1000, 60, 6 + i % 3, ("RBF", "Matern25", "RBF")[i % 3]) for i in range(9)]
    mchain, memus, minfo = build_multi_chain(specs, 20)
    nwm, nst = 4096, 10
    gps = sum(sp[2] for sp in specs)
    nine = {"emulators": 9, "design_points_each": 1000, "params": 20, "observables": mchain.nobs, "gps": gps, "walkers": nwm}
    for tag, X0m in (("burnt_in", synth.walkers_ball(nwm, minfo["xstar"], 1e-8)), ("uniform_start", synth.walkers(nwm, 20))):
        heat = StretchSampler(mchain, nwm, seed=6)         # untimed pre-heat on a scratch ensemble, as for the headline
        heat.run(X0m, 40, status=10 ** 9, store=False)
        del heat
        sm = StretchSampler(mchain, nwm, seed=5)
        assert sm._resident_engine() is not None
        sm.run(X0m, 3, status=10 ** 9, store=False)
        for e in memus:
            e._engine_ready().profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sm.run(None, nst, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / nst
        kms, nl, live_rows = 0.0, 0, 0.0
        for i, (e, sp) in enumerate(zip(memus, specs)):
            n_l, ms_l, u_l = e._engine_ready().profile_read()
            e._engine_ready().profile(False)
            if i == 0:                        # the chain's first emulator owns the compaction and its live-row counter,
                u0, n0 = u_l, n_l             # and the timing events of a launch that covers several emulators
            kms += ms_l
            nl += n_l
        # all nine designs pad to Np = 1024, so ONE predict launch per half-step covers the 63 GPs (k_predict_multi) and the
        # first context counts units for all of them; launched one by one (tune chain_batch 0) it counts its own GPs only
        live_rows = u0 / (gps if nl == n0 else specs[0][2])
        units = gps * live_rows * 1000.0 ** 2      # (GP, row) pairs x N^2: the algorithmic count (the kernel multiplies Np = 1024)
        nine[tag] = {"ms_per_step": dtm * 1e3, "walker_evals_per_s": nwm / dtm,
                     "rows_inside_box_fraction": live_rows / (nwm * nst),
                     "tflops_algorithmic_step": gps * live_rows / nst * 1000.0 ** 2 / dtm / 1e12,
                     "k_predict_launches_per_step": nl / nst, "k_predict_ms_per_step": kms / nst,
                     "k_predict_tflops": units / (kms * 1e-3) / 1e12 if kms else None,
                     "k_predict_frac_of_peak": units / (kms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if kms else None}
        del sm
    # the same chain with the int8 predict kernel switched on (option key 51; every emulator here is inside its rule and takes
    # its own K*^T and predict launches: the shared launches are fp64's)
    try:
        for e in memus:
            e._engine_ready().tune("predict_sliced", 1)
        X0m = synth.walkers_ball(nwm, minfo["xstar"], 1e-8)
        sm = StretchSampler(mchain, nwm, seed=5)
        sm.run(X0m, 3, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sm.run(None, nst, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / nst
        del sm
        nine["burnt_in_int8_predict"] = {"ms_per_step": dts * 1e3, "walker_evals_per_s": nwm / dts,
                                         "speedup_vs_fp64_path": nine["burnt_in"]["ms_per_step"] / (dts * 1e3)}
        for e in memus:
            e._engine_ready().tune("predict_sliced", 0)
    except Exception as e:      # noqa: BLE001
        nine["burnt_in_int8_predict"] = {"error": "%s: %s" % (type(e).__name__, e)}
    nine["what"] = ("stretch-move steps of a nine-emulator chain through gpb_chain_emcee_run; k_predict figures: algorithmic "
                    "flops N^2 = 1000^2 per evaluated (GP, row) over the HIP-event times of the predict launches (one per half-step for all nine "
                    "emulators: k_predict_multi)")
    out["nine_emulator_chain"] = nine
    for e in memus:
        e._engine.close()
    return out


class Watchdog:
    """A stalled rank says where, and a failed line is printed instead of a hang.  `arm(seconds, phase)` (re)starts the clock; when
    it runs out the process writes to stderr (a) every Python thread's stack, (b) for every OS thread of the process its name,
    scheduler state, the kernel function it sleeps in and the system call it is inside (/proc/self/task/*/{comm,stat,wchan,
    syscall}: what a stack of Python frames cannot say when the wait is inside libamdhip64 or librccl), then exits with 124 so
    that the launcher ends the other ranks.  The watcher is a Python thread (torch's copies and collectives, ctypes calls and
    time.sleep all release the GIL); behind it faulthandler's own C thread fires 15 s later in case the GIL was never released.
    No HIP call is made from either."""

    def __init__(self, tag):
        import threading
        self.tag, self.deadline, self.phase = tag, None, "start"
        self._cv = threading.Condition()
        self._t = threading.Thread(target=self._watch, daemon=True, name="bench-watchdog")
        self._t.start()

    def arm(self, seconds, phase):
        import faulthandler
        with self._cv:
            self.phase = phase
            self.deadline = None if seconds <= 0 else time.monotonic() + seconds
            self._cv.notify()
        faulthandler.cancel_dump_traceback_later()
        if seconds > 0:
            faulthandler.dump_traceback_later(seconds + 15.0, exit=True)

    def _watch(self):
        with self._cv:
            while True:
                if self.deadline is None:
                    self._cv.wait()
                    continue
                left = self.deadline - time.monotonic()
                if left > 0:
                    self._cv.wait(left)
                    continue
                break
        self.dump()
        os._exit(124)

    def dump(self, out=None):
        import faulthandler
        out = out or sys.stderr
        out.write("\n==== bench.py watchdog: %s made no progress in phase '%s' ====\n" % (self.tag, self.phase))
        out.flush()
        faulthandler.dump_traceback(file=out, all_threads=True)
        out.write("---- OS threads of pid %d: tid comm state wchan syscall ----\n" % os.getpid())
        try:
            tids = sorted(os.listdir("/proc/self/task"), key=int)
        except OSError:
            tids = []
        for tid in tids:
            vals = []
            for name in ("comm", "stat", "wchan", "syscall"):
                try:
                    with open("/proc/self/task/%s/%s" % (tid, name)) as f:
                        v = f.read().strip()
                    if name == "stat":              # "pid (comm) S ...": the state letter follows the closing bracket
                        v = v[v.rindex(")") + 2:][:1]
                except OSError as e:
                    v = "<%s>" % e.strerror
                vals.append(v or "-")
            out.write("%s %s %s %s [%s]\n" % (tid, vals[0], vals[1], vals[2], vals[3]))
        out.flush()


def spawn_ranks(n, argv):
    """`python3 bench.py --gpus N` typed bare (no torchrun around it, WORLD_SIZE unset): start the N ranks as ONE child process
    tree — python -m torch.distributed.run on this very file with the same arguments, rendezvous on 127.0.0.1 — relay rank 0's
    single JSON line to stdout (everything else the ranks print goes to stderr), and return the child's exit code.  This parent
    never imports torch and never opens the GPU: nothing is exec'ed over a process that has, and the box's count of processes
    holding the card is the N ranks alone."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GPB_BENCH_SPAWNED="1")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    lines = 0
    for raw in p.stdout:
        ln = raw.decode(errors="replace")
        is_line = False
        if ln.startswith("{"):
            try:
                is_line = "metric" in json.loads(ln)
            except ValueError:
                pass
        if is_line:
            lines += 1
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the {n} ranks ended with status 0 but printed {lines} result lines (expected 1)", file=sys.stderr)
        rc = 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preheat", type=int, default=100, help="untimed steps on a scratch ensemble before the warm-up (clock ramp)")
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--walkers", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-uniform", action="store_true", help="skip the second timed run from the uniform start")
    ap.add_argument("--cpu-rows", type=int, default=None)
    ap.add_argument("--sustain", type=float, default=5.5, help="seconds of continuous steps behind extras.sustained_cfg4")
    ap.add_argument("--only-sustained", action="store_true", help="of the extras, run the sustained-rate block alone")
    ap.add_argument("--sliced", action="store_true",
                    help="NOT the headline: the whole run with the int8 predict kernel (option key 51, csrc/gpb_sliced.hip) switched on; "
                         "the line says so in config.variant and prices its roofline against the dense int8 MFMA peak")
    args = ap.parse_args()
    if args.sliced:
        os.environ["GPB_PREDICT_SLICED"] = "1"
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # The launch contract.  Under a launcher (torchrun sets WORLD_SIZE) this process is one rank and WORLD_SIZE must be --gpus.
    # Bare with --gpus N > 1 it becomes the launcher's parent (spawn_ranks) BEFORE torch is imported or the GPU touched.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if int(env_world or "1") != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks: refusing to time a job "
              f"whose size is not the one asked for", file=sys.stderr)
        sys.exit(2)

    # GPB_BENCH_WATCHDOG=N: seconds a phase may take; 0 = off.  A multi-rank run has it on by default, at 240 s — and the whole run
    # has 570 s (the driver's limit for a bench line is 600 s): start-up gets up to 1.6 N (the first `import torch` on a fresh box
    # takes 1-2 minutes, longer with eight ranks paging the same files in), set-up N / 2, each timed run 3 N / 8, all cut to what
    # is left of the total — so a stalled N-rank line ends as a FAILED line with every rank's stacks, never as a hang, and a slow
    # but healthy start is not shot.
    wd_s = float(os.environ.get("GPB_BENCH_WATCHDOG", "240" if args.gpus > 1 else "0"))
    t_begin = time.monotonic()
    dog = Watchdog("rank %s of %d" % (os.environ.get("RANK", "0"), args.gpus))

    def phase(frac, name):
        left = 2.375 * wd_s - (time.monotonic() - t_begin)          # 570 s at the default
        dog.arm(max(min(frac * wd_s, left), 1.0) if wd_s > 0 else 0.0, name)
    phase(1.6, "start-up (import torch, rendezvous)")
    if wd_s > 0:
        import faulthandler, signal
        faulthandler.register(signal.SIGTERM, chain=True)
    import torch
    import torch.distributed as dist
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain, flops_per_walker

    rank, world, local = init_from_env()
    assert world == args.gpus
    torch.cuda.set_device(local)
    # GPB_BENCH_ONE_RANK_SHARDED=1 (with --gpus 1): the SHARDED branch of this file over a one-rank RCCL communicator — every
    # nccl-only line (replicate, try_direct, the in-stream all-gather, the self-check, the probe) on the one GPU a build box
    # has; only the wire between ranks is missing (tests/test_gpu_bench_ranks.py).  Off, N = 1 is the plain single-GPU line.
    sharded = world > 1
    if world == 1 and os.environ.get("GPB_BENCH_ONE_RANK_SHARDED") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        sharded = True
    if (sharded and os.environ.get("GPB_BENCH_DEFAULT_STREAM") != "1") or os.environ.get("GPB_BENCH_OWN_STREAM") == "1":
        # ranks of a sharded run work on a stream of their own (torch's streams are non-blocking ones): the kernels, the copies
        # and the in-stream ncclAllGather of the C ABI all go where torch's current stream is, and nothing of the step loop
        # inherits the legacy default stream's implicit synchronisation with other streams of the process (RCCL's own included)
        torch.cuda.set_stream(torch.cuda.Stream(device=local))
    phase(0.5, "set-up (training, replication, self-checks)")
    chain, emu, info = build_chain(args.config, device=local)
    N, d, M, P = info["N"], info["d"], info["M"], info["P"]
    nwalkers = args.walkers or 2 * info["W"]
    sharding = WalkerSharding() if sharded else None
    if sharding is not None:
        # every rank has just trained the same emulator on the same synthetic data — but the host part of a training (scaler
        # + PCA: an N x M SVD in numpy) rounds differently under a different BLAS threading, so the replicas are rank 0's
        # (broadcast of its fitted host state; each rank rebuilds its factorisation from it), and StretchSampler.run has all
        # ranks prove their state digests equal before the first step
        sharding.replicate(chain)
        info["yexp"] = chain.expdata[0].copy()
    sampler = StretchSampler(chain, nwalkers, seed=12345, sharding=sharding, device=local)
    # burnt-in start: a ball around theta*, small enough that the stretch move (which grows a concentrated ensemble by
    # about 1.4x per step towards the posterior's width) keeps every proposal inside the prior box over warm-up + timed steps
    ball = float(min(1e-3, max(1e-13, 10.0 ** (-3.0 - 0.16 * (args.warmup + args.steps)))))
    X0 = synth.walkers_ball(nwalkers, info["xstar"], ball, lo=info["lo"], hi=info["hi"])
    X0_uniform = synth.walkers(nwalkers, d)
    eng = emu._engine_ready()
    direct_why = None
    if sharding is not None and dist.get_backend() == "nccl" and os.environ.get("GPB_DIST_DIRECT", "1") != "0":
        # ncclAllGather from the C ABI on the kernels' own stream (8 us less per collective than the hop through
        # torch's communicator stream); self-checked against torch.distributed, which stays the exchange if not.
        direct_why = sharding.try_direct(eng)

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    # Which loop drives the steps: gpb_emcee_run (the C ABI enqueues every kernel and, when sharded, the in-stream
    # all-gather) or the host-driven loop (Python enqueues; the all-gather through torch.distributed).  A build box has
    # one GPU, so the sharded form of the C loop runs for the first time on the multi-GPU node: it is checked here
    # against the host-driven loop on a short run from the spread-out start (rows inside AND outside the box; every rank
    # must reproduce the same ensemble) and dropped on ALL ranks if any rank disagrees or raises.  StretchSampler.run
    # itself lets the ranks agree that gpb_chain_emcee_prepare succeeded everywhere before any of them enqueues a
    # collective.
    loop = "gpb_chain_emcee_run" if sampler._resident_engine() is not None else "host-driven"
    if sharded and loop == "gpb_chain_emcee_run":
        ok = True
        try:
            ref = StretchSampler(chain, nwalkers, seed=777, sharding=sharding, device=local)
            ref._resident_engine = lambda: None
            a = ref.run(X0_uniform, 2, status=10 ** 9, store=False)
            tst = StretchSampler(chain, nwalkers, seed=777, sharding=sharding, device=local)
            ok = bool(np.array_equal(a, tst.run(X0_uniform, 2, status=10 ** 9, store=False)))
        except Exception as e:          # noqa: BLE001
            print(f"rank {rank}: gpb_chain_emcee_run self-check raised {type(e).__name__}: {e}", file=sys.stderr)
            ok = False
        if not all_ranks(ok):
            sampler._resident_engine = lambda: None
            loop = "host-driven (gpb_chain_emcee_run failed its self-check against it)"

    def timed_run(X_start, seed):
        smp = sampler if seed is None else StretchSampler(chain, nwalkers, seed=seed, sharding=sharding, device=local)
        if seed is not None and loop.startswith("host-driven"):
            smp._resident_engine = lambda: None
        # untimed pre-heat on a scratch sampler (its own ensemble; the same number of steps on every rank): the clocks and —
        # sharded — RCCL's channels are up before the W warm-up steps, whatever W is.  (A configuration measured first after an
        # idle gap read 3-5 % slow at sub-millisecond steps: tools/gpu_tile_rule_sweep.py.)
        heat = StretchSampler(chain, nwalkers, seed=4242, sharding=sharding, device=local)
        if loop.startswith("host-driven"):
            heat._resident_engine = lambda: None
        if args.preheat > 0:
            heat.run(X_start, args.preheat, status=10 ** 9, store=False)
        del heat
        smp.run(X_start, args.warmup, status=10 ** 9, store=False)
        eng.profile(True)
        barrier()
        t0 = time.perf_counter()
        smp.run(None, args.steps, status=10 ** 9, store=False)      # continues from the resident state
        barrier()
        dt = time.perf_counter() - t0
        launches, kms, units = eng.profile_read()
        eng.profile(False)
        tt = torch.tensor([dt, units], dtype=torch.float64, device="cuda")
        if sharded:
            mx, sm = tt.clone(), tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            return smp, float(mx[0].item()), launches, kms, units, float(sm[1].item())
        return smp, dt, launches, kms, units, units

    phase(0.375, "timed run (pre-heat, warm-up, timed steps; burnt-in start)")
    _, dt, launches, kms, units, units_all = timed_run(X0, None)
    acc = float(sampler.acceptance_fraction.mean())
    inside_frac = (units_all / (launches * P * (nwalkers // 2))) if launches else None      # all ranks' rows
    # The headline is measured on a burnt-in ensemble (>= 0.95 of the proposal rows inside the prior box and evaluated).  The
    # stretch move grows a ball by ~1.45x per step and this posterior is as wide as the box in ten of its twenty directions
    # (10 GPs), so beyond ~60 warm-up + timed steps no start keeps every row inside.  Then the line is still printed, but
    # `value` counts EVALUATED walkers only (= value_evaluated; never rows that cost nothing) and says so.
    degraded = inside_frac is not None and inside_frac < 0.95      # the same number on every rank
    if degraded and rank == 0:
        print(f"bench.py: only {inside_frac:.3f} of the timed region's proposal rows lay inside the prior box "
              f"(--steps {args.steps} --warmup {args.warmup}: more than ~60 steps in all); `value` counts evaluated "
              f"walkers only", file=sys.stderr)
    devices_used = [local]
    if sharded:         # which GPU each rank ran on, as the communicator's ranks report it
        dv = [None] * world
        dist.all_gather_object(dv, int(local))
        devices_used = dv
    consistent = None
    if sharded:         # replicated RNG + gathered log-probabilities: every rank must hold the same ensemble
        chk = torch.stack([sampler.pos.sum(), sampler.lp.sum()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        consistent = bool(torch.equal(lo, hi))
    import hashlib
    checksum = hashlib.sha256(sampler.pos.cpu().numpy().tobytes() + sampler.lp.cpu().numpy().tobytes()).hexdigest()[:16]
    # the proposal rows of the half-step that FOLLOWS the timed region (the sampler's own proposal kernel on the final
    # ensemble): what the CPU baseline is timed on — the same inside / outside mix as the timed batches
    Xnext = torch.empty((nwalkers // 2, d), dtype=torch.float64, device=sampler.pos.device)
    fnext = torch.empty(nwalkers // 2, dtype=torch.float64, device=sampler.pos.device)
    from gpbayestools_hic_amd import _native as nat
    eng._ck(eng.lib.gpb_stretch_propose(eng.h, nat.ptr(sampler.pos), nwalkers, d, 0, sampler.seed, sampler._step_counter,
                                        sampler.a, nat.ptr(Xnext), nat.ptr(fnext), sampler.randomize_split))
    Xnext = Xnext.cpu().numpy()
    # the same loop from walkers spread uniformly over the box (round 1/2's timed region): about half of the proposals
    # leave the 20-dimensional box and are not evaluated
    uni = None
    if not args.no_uniform:
        phase(0.375, "timed run from the uniform start")
        su, dtu, lu, kmsu, unitsu, units_all_u = timed_run(X0_uniform, 4242)
        uni = {"value": nwalkers * args.steps / dtu, "unit": "walker-evals/s (proposals outside the box counted, not evaluated)",
               "value_evaluated": units_all_u / P / dtu, "ms_per_step": dtu / args.steps * 1e3,
               "rows_inside_box_fraction": (units_all_u / (lu * P * (nwalkers // 2))) if lu else None,
               "acceptance_fraction": float(su.acceptance_fraction.mean()),
               "k_predict_avg_launch_ms": kmsu / max(lu, 1),
               "k_predict_frac_of_peak": (unitsu / max(lu, 1) * float(N) * float(N) / (kmsu / max(lu, 1) * 1e-3) / 1e12
                                          / FP64_MFMA_PEAK_TFLOPS) if lu else None,
               "what": "the same step loop started from walkers uniform in the prior box (SURVEY 8d's walkers)"}
        del su

    # the wire, measured: what ONE all-gather of a batch's shares costs on the kernels' stream (twice per step) — the number the
    # expected 8-GPU rate hangs on; a failure here must not cost the line that was just timed
    wire = None
    if sharded:
        phase(0.25, "all-gather probe")
        try:
            us = sharding.time_allgather(max(nwalkers // 2 // world, 1))
            wire = {"us_per_allgather": us, "doubles_per_rank": max(nwalkers // 2 // world, 1), "collectives_per_step": 2,
                    "share_of_step": (2.0 * us * 1e-3 / (dt / args.steps * 1e3)) if us else None,
                    "what": "200 back-to-back in-place all-gathers of a half-step's shares on the kernels' stream, HIP events, max over "
                            "ranks (None under gloo: the rehearsal stages through the host)"}
        except Exception as e:      # noqa: BLE001
            wire = {"us_per_allgather": None, "error": "%s: %s" % (type(e).__name__, e)}
    if world == 1:
        dog.arm(0.0, "result line")                      # (extras and the CPU baseline: N = 1 only, minutes)
    else:
        phase(0.25, "result line")
    if rank == 0:
        value = nwalkers * args.steps / dt if not degraded else units_all / P / dt
        alg_flops_per_launch = units / max(launches, 1) * float(N) * float(N)     # N^2 per (GP, walker): the trsm term
        achieved = alg_flops_per_launch / (kms / max(launches, 1) * 1e-3) / 1e12 if launches else None
        gflop_step = flops_per_walker(N, d, P, M, info["kernel"]) * nwalkers / 1e9
        out = {
            "metric": "MCMC steps/s x walkers (walker log-posterior evaluations per second, whole job)",
            "value": value, "unit": "walker-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {N} design pts x {d} params x {M} observables, "
                                   f"{P} GPs ({info['kernel']}), {nwalkers} walkers, stretch move, "
                                   f"fixed hyper-parameters, burnt-in ensemble (ball of relative radius {ball:.1e} "
                                   f"around theta*)"
                                   + ("; the ensemble reached the box inside the timed region: value = evaluated walkers only"
                                      if degraded else ""), "walkers": nwalkers,
                       "parallelism": f"walker-shard x{world}" + ("" if world > 1 else " (one-rank RCCL rehearsal)") if sharded else "single GPU",
                       "step_loop": loop,
                       "ranks": dist.get_world_size() if sharded else 1,
                       "stream": "torch's default stream" if torch.cuda.current_stream().cuda_stream == 0 else "a non-blocking stream per rank",
                       "devices_used": devices_used,
                       "launched_by": "bench.py itself (bare --gpus N: child torch.distributed.run)"
                                      if os.environ.get("GPB_BENCH_SPAWNED") == "1" else
                                      ("an outer launcher (WORLD_SIZE set)" if world > 1 else "single process"),
                       "untimed_preheat": f"{args.preheat} steps of the same loop on a scratch ensemble before the W warm-up steps (clocks, RCCL channels)",
                       "allgather": None if not sharded else (
                           "gpb_dist_allgather (ncclAllGather on the kernel stream)" if sharding.direct is not None
                           else "torch.distributed " + dist.get_backend()
                                + (" (direct path not used: %s)" % direct_why if direct_why else ""))},
            "acceptance_fraction": acc, "ranks_hold_identical_ensemble": consistent, "ensemble_checksum": checksum,
            # proposals outside the prior box cost nothing, here as in the reference (src/mcmc.py:275-283): share of the
            # timed region's proposal rows that lay inside the box and were evaluated, and the rate counted on those alone
            "rows_inside_box_fraction": inside_frac,
            "value_evaluated": units_all / P / dt,
            "gflop_per_step_algorithmic": gflop_step,
            "tflops_algorithmic": gflop_step / (dt / args.steps) / 1e3 * (inside_frac or 1.0),
            "gflop_per_step_algorithmic_note": "SURVEY 8(d)'s figure for 4096 evaluated walkers; tflops_algorithmic = that x "
                                               "rows_inside_box_fraction / ms_per_step (<= the 78.6 TF/s fp64 peak)",
            "roofline": {"bound": "mfma", "kernel": "k_predict (V = L^-1 K*^T, fused sum of squares)",
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / FP64_MFMA_PEAK_TFLOPS) if achieved else None, "traffic": None,
                         "launches": launches, "avg_launch_ms": kms / max(launches, 1)},
        }
        if args.sliced:
            tops = 21.0 * alg_flops_per_launch / (kms / max(launches, 1) * 1e-3) / 1e12 if launches else None
            out["config"]["variant"] = ("int8 predict kernel switched on (bench.py --sliced, option key 51): NOT the default path; "
                                        "variance within ~1e-11 of the fp64 kernel, see extras.k_predict_sliced_cfg4 of the default run")
            out["roofline"] = {"bound": "mfma", "kernel": "k_predict_sliced (V = L^-1 K*^T on v_mfma_i32_32x32x32_i8, 21 digit products, fused sum of squares)",
                               "achieved": tops, "peak": 5000.0, "unit": "TOP/s (int8, dense)", "frac": (tops / 5000.0) if tops else None,
                               "fp64_equivalent_tflops": achieved, "traffic": None, "launches": launches, "avg_launch_ms": kms / max(launches, 1)}
            out["dtype"] = "f64 (V = L^-1 K*^T through int8 digit planes, exact int32 sums, fp64 combine)"
        # HBM-side traffic of the dominant kernel: PMC counters need rocprofv3, so the per-launch figure comes
        # from the committed summary of the same command (profiles/r05_pmc_traffic.json), when it matches.
        # (the newest profiles/rNN_pmc_traffic.json whose workload AND kernel-source fingerprint match this build: a profile older
        # than the kernel is not quoted — roofline.traffic_note says why)
        try:
            import glob
            import hashlib
            key = "k_predict_sliced" if args.sliced else "k_predict"
            srcs = ["gpb_sliced.hip"] if args.sliced else ["gpb_predict.hip", "gemm_tile.h"]
            fp = hashlib.sha256(b"".join(open(os.path.join(ROOT, "gpbayestools_hic_amd", "csrc", f), "rb").read() for f in srcs)).hexdigest()[:16]
            note = "no profiles/r*_pmc_traffic.json for this workload"
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                with open(path) as f:
                    pmc = json.load(f)
                wl = pmc["workload"]
                if key not in pmc or (wl["config"], wl["N"], wl["P"], wl["W_per_launch"]) != (args.config, N, P, nwalkers // 2) \
                        or world != 1 or not wl.get("burnt_in"):
                    continue
                if pmc[key].get("source_sha16") != fp:
                    note = "%s was collected on an older build of %s (fingerprint %s, now %s): not quoted" % (
                        os.path.basename(path), "+".join(srcs), pmc[key].get("source_sha16"), fp)
                    continue
                out["roofline"]["traffic"] = pmc[key]["bytes_per_launch_corrected"]
                out["roofline"]["traffic_unit"] = "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % os.path.basename(path)
                out["roofline"]["algorithmic_bytes"] = pmc[key]["algorithmic_bytes_per_launch"]
                note = None
                break
            if note:
                out["roofline"]["traffic_note"] = note
        except Exception as e:      # noqa: BLE001
            out["roofline"]["traffic_note"] = "%s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_extras:
            out["extras"] = extras(chain, emu, info, args.sustain, args.only_sustained)
        if uni is not None:
            out.setdefault("extras", {})["uniform_start"] = uni
        if wire is not None:
            out.setdefault("extras", {})["allgather_probe"] = wire
        if world == 1 and not args.no_cpu_baseline:
            rows = min(args.cpu_rows or Xnext.shape[0], Xnext.shape[0])
            cb, lp_cpu = cpu_baseline(info, np.ascontiguousarray(Xnext[:rows]),
                                      "the proposal rows of the half-step that follows the GPU's timed region "
                                      "(gpb_stretch_propose on the final ensemble)")
            lp_gpu = chain.log_posterior(Xnext[:rows])
            fin = np.isfinite(lp_cpu)
            cb["max_rel_diff_vs_gpu"] = float(np.max(np.abs(lp_gpu[fin] - lp_cpu[fin]) / np.abs(lp_cpu[fin]))) if fin.any() else None
            cb["same_rows_outside_box"] = bool(np.array_equal(fin, np.isfinite(lp_gpu)))
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
