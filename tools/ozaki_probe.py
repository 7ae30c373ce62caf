#!/usr/bin/env python3
"""CPU-side accuracy probe for an Ozaki-type int8 evaluation of V = L^-1 K*^T ("what comes next"): rows of L^-1 and
columns of K*^T are scaled by powers of two, split into s signed 7-bit slices (error-free), slice pairs (i, j) with
i + j < s are multiplied exactly (integers) and summed in fp64.  Prints the error of V and of the predictive variance
c - sum v^2 against 80-bit arithmetic for s = 4 .. 9 on a synthetic GP of cfg 4's kind.  numpy only, no GPU.
    python tools/ozaki_probe.py [N=512] [W=128]"""
import sys

import numpy as np


def slices(M, axis, s):
    """M scaled per row (axis=1) / column (axis=0) to |.| < 1, then s slices of 7 bits: M ~ scale * sum_k S_k 2^(-7 (k + 1))"""
    mx = np.max(np.abs(M), axis=axis, keepdims=True)
    e = np.ceil(np.log2(np.where(mx > 0, mx, 1.0))) + 1
    scale = 2.0 ** e
    R = M / scale
    out = []
    for k in range(s):
        R = R * 128.0
        S = np.rint(R)                      # |S| <= 64 after the first (|R| < 1/2 * 128), fits int8
        R = R - S
        out.append(S.astype(np.int64))
    return scale, out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    rng = np.random.default_rng(0)
    d = 20
    X = rng.random((N, d)); Xs = rng.random((W, d))
    ell = 1.2
    def rbf(A, B):
        d2 = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1) / ell ** 2
        return np.exp(-0.5 * d2)
    K = rbf(X, X) + 0.03 * np.eye(N)
    L = np.linalg.cholesky(K)
    Linv = np.linalg.inv(L)
    Ks = rbf(X, Xs)                                     # K*^T [N, W]
    V_ref = (Linv.astype(np.longdouble) @ Ks.astype(np.longdouble))
    var_ref = (1.0 + 0.03) - (V_ref ** 2).sum(0)
    V64 = Linv @ Ks
    var64 = (1.0 + 0.03) - (V64 ** 2).sum(0)
    print("fp64 reference: max |dV| / max|V| = %.2e, max rel err of var = %.2e (min var %.3e)" %
          (float(np.max(np.abs(V64 - V_ref)) / np.max(np.abs(V_ref))), float(np.max(np.abs(var64 - var_ref) / var_ref)), float(var_ref.min())))
    for s in range(4, 10):
        sa, A = slices(Linv, 1, s)
        sb, B = slices(Ks, 0, s)
        V = np.zeros((N, W), dtype=np.longdouble)
        npairs = 0
        for i in range(s):
            for j in range(s - i):
                V += (A[i] @ B[j]).astype(np.longdouble) * np.longdouble(2.0) ** (-7 * (i + j + 2))
                npairs += 1
        V = V * sa * sb
        var = (1.0 + 0.03) - (V ** 2).sum(0)
        print("s = %d (%2d int8 products): max |dV| / max|V| = %.2e, max rel err of var = %.2e" %
              (s, npairs, float(np.max(np.abs(V - V_ref)) / np.max(np.abs(V_ref))), float(np.max(np.abs(var - var_ref) / var_ref))))


if __name__ == "__main__":
    main()
