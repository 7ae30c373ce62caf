#!/usr/bin/env python3
"""The sliced-integer predict kernel (option key 51, csrc/gpb_sliced.hip) against the fp64 kernel and the oracle on one GP set:
variance deviation, bit-stability across batch cuts / compaction-free tile counts, and the time of both kernels.
usage: gpu_sliced_check.py [N=2048] [W=2048] [P=10] [d=20] [kernel=RBF] [sn2=0.05] [c=1.0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def main():
    a = sys.argv[1:]
    N = int(a[0]) if len(a) > 0 else 2048
    W = int(a[1]) if len(a) > 1 else 2048
    P = int(a[2]) if len(a) > 2 else 10
    d = int(a[3]) if len(a) > 3 else 20
    kind = a[4] if len(a) > 4 else "RBF"
    sn2 = float(a[5]) if len(a) > 5 else 0.05
    c = float(a[6]) if len(a) > 6 else 1.0
    rng = np.random.default_rng(3)
    X = synth.lhs(N, d)
    Z = np.sin(X @ rng.standard_normal((d, P))).T + 0.05 * rng.standard_normal((P, N))
    th = synth.fixed_theta(d, P, noise=sn2)
    th[:, 0] = np.log(c) + 0.05 * np.arange(P)
    Xs = rng.random((W, d))
    Xs[: W // 4] = 0.5 + 0.02 * rng.standard_normal((W // 4, d))          # a ball, as a burnt-in ensemble
    Xs[W // 4: W // 4 + min(32, N)] = X[:min(32, N)]                          # at design points: smallest variances
    eng = GPEngine(0)
    eng.set_data(X, Z, kind, alpha=0.1); eng.set_theta(th); eng.factor()
    eng.force_tile(128)
    m0, v0 = eng.predict(Xs)
    eng.tune("predict_sliced", 1)
    m1, v1 = eng.predict(Xs)
    rel = np.abs(v1 - v0) / np.abs(v0)
    print("N=%d W=%d P=%d d=%d %s sn2=%g c=%g: sliced vs fp64 kernel: max rel dev of variance %.3e (median %.3e), mean identical: %s, min var %.3e"
          % (N, W, P, d, kind, sn2, c, rel.max(), np.median(rel), bool(np.array_equal(m0, m1)), v0.min()))
    if rel.max() == 0.0:
        print("  (identical: the rule kept this context on the fp64 kernel)")
    # oracle on a sample of the rows
    idx = np.r_[0:8, W // 4: W // 4 + 8, W - 8: W]
    kid = O.KIND_NAMES[kind]
    worst64 = worstS = 0.0
    for p in range(min(P, 3)):
        L, al = O.gp_factor(X, Z[p], th[p], kid, 0.1)
        mo, vo = O.gp_predict(Xs[idx], X, th[p], L, al, kid)
        worst64 = max(worst64, float(np.max(np.abs(v0[idx, p] - vo) / vo)))
        worstS = max(worstS, float(np.max(np.abs(v1[idx, p] - vo) / vo)))
    print("  against the oracle (24 rows x %d GPs): fp64 kernel %.2e, sliced %.2e (bar 1e-10)" % (min(P, 3), worst64, worstS))
    # a walker's bits must not depend on the batch: halves, an odd cut, a small batch
    ok = True
    for lo, hi in ((0, W // 2), (W // 2, W), (3, W - 5), (0, 130)):
        if hi - lo < 1:
            continue
        _, vv = eng.predict(Xs[lo:hi])
        same = np.array_equal(vv, v1[lo:hi])
        ok = ok and same
        if not same:
            big = eng.tune  # noqa: F841
            print("  batch [%d, %d): bits differ (max rel %.2e)%s" % (lo, hi, np.max(np.abs(vv - v1[lo:hi]) / v1[lo:hi]),
                  " — a batch this small runs the fp64 tiles" if hi - lo < 1024 else ""))
    print("  bits independent of the batch cut (where the sliced kernel runs): %s" % ok)
    # timing: predict launches with events (gpb_profile_*): 20 launches each
    import torch
    Xd = torch.as_tensor(Xs, device="cuda:0")
    for mode, name in ((0, "fp64"), (1, "sliced")):
        eng.tune("predict_sliced", mode)
        for _ in range(3):
            eng.predict(Xd)
        eng.profile(True); eng.profile_read()
        for _ in range(20):
            eng.predict(Xd)
        eng.sync()
        n, ms, u = eng.profile_read()
        eng.profile(False)
        print("  %-7s predict launch: %.4f ms (%d launches; %.1f TF/s fp64-equivalent)" % (name, ms / max(n, 1), n, P * N * N * W / (ms / max(n, 1) * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
