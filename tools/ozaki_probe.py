#!/usr/bin/env python3
"""Stage 0 of the sliced-integer k_predict (round 6): CPU-side accuracy of evaluating V = L^-1 K*^T on the int8 matrix pipe.

The scheme the kernel would run (tools/micro/sliced_probe.hip is its inner loop, profiles/r06_sliced_model.txt its costs):
  * row j of L^-1 is scaled by 2^-e_j (smallest power of two with max|row| <= 0.99 * 2^e_j) and rounded ONCE to a D-digit
    fixed-point integer a = rint(A * 2^(8 D - 1 - e_j)), |a| < 2^(8 D - 1); its D signed radix-256 digits (each -128..127:
    (a + 0x80..80) byte-wise XOR 0x80) are the int8 slices.  That rounding is the scheme's only inexact step on A.
  * K*^T is positive and bounded by the kernel's amplitude c: ONE scale 2^e_c per GP (no column maxima, no extra pass), the
    same fixed-point rounding, the same digits.
  * digit planes are multiplied exactly (int8 x int8 -> int32 on v_mfma_i32_16x16x64_i8) and all pairs of one LEVEL
    l = ta + tb are added in one int32 accumulator (exact: (l + 1) * N * 2^14 < 2^31 for N <= 2^14); the levels below
    `lowest kept level` are dropped; the kept levels are combined in fp64 by Horner's rule from the smallest up and scaled.
Everything but the two fixed-point roundings, the dropped levels and the fp64 combine is exact, so a walker's bits do not depend
on tiles, batch cuts or rank counts.

Prints, per case and per (D, kept products): the error of the variance c + sn2 - sum_j V_jw^2 against 80-bit arithmetic, next
to the error of the fp64 routes the tests compare today (explicit inverse GEMM = the device path, triangular solve = the oracle).
The bar is 1e-10 relative on the variance (tests/, tools/gpu_parity_soak.py).  numpy / scipy only, no GPU.
    python tools/ozaki_probe.py [quick]"""
import sys

import numpy as np
from scipy.linalg import cholesky, solve_triangular


def lhs(N, d, rng):
    return (np.argsort(rng.random((N, d)), axis=0) + rng.random((N, d))) / N


def shape_fn(kind, A, B, ls):
    d2 = ((A[:, None, :] / ls - B[None, :, :] / ls) ** 2).sum(-1)
    if kind == "RBF":
        return np.exp(-0.5 * d2)
    r = np.sqrt(d2)
    if kind == "Matern15":
        t = np.sqrt(3.0) * r
        return (1 + t) * np.exp(-t)
    t = np.sqrt(5.0) * r
    return (1 + t + t * t / 3) * np.exp(-t)


def digits_fast(M, e, D):
    """the same digits by float arithmetic only (vectorised; exact: every intermediate is an integer below 2^53 for D <= 6,
    and for D = 7 the top digit is split off first)"""
    x = np.ldexp(M, (8 * D - 1) - e)                                   # |x| < 2^(8 D - 1)
    if D == 7:
        top = np.rint(np.ldexp(x, -48))                                # most significant digit, round to nearest
        rest = x - np.ldexp(top, 48)                                   # |rest| <= 2^47 (exact: x has 53 bits)
        lo = digits_fast_int(np.rint(rest), 6, allow_top_carry=True)
        # a carry out of the 6 low digits (top low digit would be 128) goes into `top`
        lo, carry = lo
        return lo + [top + carry]
    return digits_fast_int(np.rint(x), D)


def digits_fast_int(a, D, allow_top_carry=False):
    planes = []
    r = a.copy()
    for t in range(D):
        q = np.floor((r + 128.0) / 256.0)                              # digit = r - 256 q in [-128, 127]
        planes.append(r - 256.0 * q)
        r = q
    if allow_top_carry:
        return planes, r
    assert np.all(r == 0), "top digit overflow: the 0.99 headroom rule was violated"
    return planes


def sliced_V(Linv, KsT, c, D, lowest_level):
    """V by the scheme above.  Linv [N, N] lower triangular, KsT [N, W] in (0, c]."""
    N = Linv.shape[0]
    mx = np.max(np.abs(Linv), axis=1)
    eA = np.ceil(np.log2(mx / 0.99)).astype(int)                       # max <= 0.99 * 2^e
    eC = int(np.ceil(np.log2(c / 0.99)))
    Arows = digits_fast(Linv / np.ldexp(1.0, eA)[:, None], 0, D)      # scaling by a power of two is exact
    Bd = digits_fast(KsT, eC, D)
    nprod = 0
    acc = None
    for lev in range(lowest_level, 2 * D - 1):                         # Horner from the smallest kept level up, in fp64
        S = np.zeros((N, KsT.shape[1]))
        for ta in range(D):
            tb = lev - ta
            if 0 <= tb < D:
                S += Arows[ta] @ Bd[tb]                                # exact: integers below 2^53
                nprod += 1
        assert np.max(np.abs(S)) < 2 ** 31, "int32 level accumulator would overflow"
        acc = S if acc is None else acc * (1.0 / 256.0) + S            # what the epilogue does (fma per level)
    # acc is in units of 256^(2 D - 2) * 2^-(8 D - 1) * 2^-(8 D - 1)
    V = np.ldexp(acc, 8 * (2 * D - 2) - 2 * (8 * D - 1)) * np.ldexp(1.0, eA)[:, None] * np.ldexp(1.0, eC)
    return V, nprod


def one_case(name, X, Xs, kind, c, ls, sn2, alpha, configs, out):
    N = X.shape[0]
    K = c * shape_fn(kind, X, X, ls)
    K[np.diag_indices(N)] = c + sn2 + alpha
    L = cholesky(K, lower=True)
    Linv = solve_triangular(L, np.eye(N), lower=True)
    KsT = c * shape_fn(kind, X, Xs, ls)                                # [N, W]
    # 80-bit reference of what the REFERENCE computes: forward substitution on the fp64 factor L
    Lq = L.astype(np.longdouble)
    Vq = np.zeros(KsT.shape, dtype=np.longdouble)
    Bq = KsT.astype(np.longdouble)
    for j in range(N):
        Vq[j] = (Bq[j] - Lq[j, :j] @ Vq[:j]) / Lq[j, j]
    var_q = np.longdouble(c + sn2) - (Vq ** 2).sum(0)
    def err(V):
        var = (c + sn2) - (np.asarray(V, dtype=np.float64) ** 2).sum(0)
        return float(np.max(np.abs(var - var_q) / var_q))
    V_tr = solve_triangular(L, KsT, lower=True)                       # the oracle's route (sk:_gpr.py:454)
    V_gemm = Linv @ KsT                                                # the device's fp64 route today
    row = dict(case=name, N=N, W=Xs.shape[0], kind=kind, cond="%.1e" % np.linalg.cond(K), cancel="%.0f" % float((c + sn2) / var_q.min()),
               rowmax="%.1e" % np.abs(Linv).max(), trsm64="%.1e" % err(V_tr), gemm64="%.1e" % err(V_gemm))
    for (D, low) in configs:
        V, nprod = sliced_V(Linv, KsT, c, D, low)
        row["D%d/%dprod" % (D, nprod)] = "%.1e" % err(V)
        out.setdefault("D%d/%dprod" % (D, nprod), []).append(err(V))
    out.setdefault("gemm64", []).append(err(V_gemm))
    out.setdefault("trsm64", []).append(err(V_tr))
    print(row, flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    rng = np.random.default_rng(6)
    # (D digits per operand, lowest kept level): D = 6 keeps levels 5..10 = 21 products (the 2^-48 scheme), levels 4..10 = 26;
    # D = 7 keeps levels 6..12 = 28 products (2^-56: fp64-equivalent), levels 7..12 = 21 products of 7-digit operands
    configs = [(5, 4), (6, 5), (6, 4), (7, 7), (7, 6)]
    out = {}
    # 1. the bench's GP (cfg 4 timing hyper-parameters: c = 1, l = 1.5, sn2 = 0.05, alpha = 0.1), cut to N = 768 for the 80-bit loop
    N, d, W = (256, 20, 64) if quick else (768, 20, 128)
    X = lhs(N, d, rng); Xs = rng.random((W, d))
    one_case("cfg4-like", X, Xs, "RBF", 1.0, np.full(d, 1.5), 0.05, 0.1, configs, out)
    # 2. a burnt-in ensemble: walkers in a small ball (what the headline runs on)
    Xs = 0.5 + 0.01 * rng.standard_normal((W, d))
    one_case("cfg4-ball", X, Xs, "RBF", 1.0, np.full(d, 1.5), 0.05, 0.1, configs, out)
    # 3. the worst the reference's search box allows: noise at its lower bound 1e-2 (src/emulator.py:302-305), large amplitude,
    #    long length scales (smooth kernel, worst conditioned K), query points AT design points (smallest variance)
    for kind in ("RBF", "Matern15", "Matern25"):
        Nn = 200 if quick else 500
        X = lhs(Nn, 5, rng)
        Xs = np.vstack([X[:32] + 1e-6 * rng.standard_normal((32, 5)), rng.random((32, 5))])
        one_case("worst-box", X, Xs, kind, float(np.exp(3.0)), np.full(5, 8.0), 1e-2, 0.1, configs, out)
    # 4. the soak's distribution (tools/gpu_parity_soak.py): random shapes, the three families, edge length scales
    kinds = ["RBF", "Matern15", "Matern25"]
    for cse in range(6 if quick else 24):
        Nn = int(rng.choice([33, 64, 100, 129, 200, 257, 400]))
        d = int(rng.choice([1, 2, 3, 5, 8, 15, 20, 31]))
        kind = kinds[int(rng.integers(0, 3))]
        edge = rng.random() < 0.25
        lo_l = 0.1 if kind == "RBF" else (1e-3 if edge else 0.05)
        ls = rng.uniform(0.5, 3.0, d) if not edge else np.where(rng.random(d) < 0.2, lo_l, rng.uniform(0.5, 3.0, d))
        c = float(np.exp(rng.uniform(-0.5, 0.7))); sn2 = float(rng.uniform(0.02, 0.2))
        X = lhs(Nn, d, rng); Xs = rng.random((40, d))
        one_case("soak%02d%s" % (cse, "e" if edge else ""), X, Xs, kind, c, ls, sn2, 0.1, configs, out)
    print("\nworst relative error of the variance over all cases (bar: 1e-10):")
    for k, v in out.items():
        print("  %-12s max %.2e   median %.2e" % (k, max(v), float(np.median(v))))


if __name__ == "__main__":
    main()
