import numpy as np
from decimal import Decimal, getcontext
getcontext().prec = 60
ld = np.longdouble
h = Decimal(2).ln() / 2
def g_exact(r):          # (exp(r) - 1 - r) / r^2 in 60 digits
    r = Decimal(r)
    if abs(r) < Decimal('1e-12'):
        return Decimal(1)/2 + r/6 + r*r/24
    return (r.exp() - 1 - r) / (r*r)
def fit(ntail):          # tail degree ntail: total degree ntail + 2
    n = ntail + 1
    k = np.arange(n)
    u = np.cos(np.pi * (2*k + 1) / (2*n))              # Chebyshev nodes in [-1, 1]
    hh = float(h) * (1 + 1e-3)                       # a hair wider than ln2/2 (two-term reduction leaves |r| slightly above)
    rs = [Decimal(float(x)) * Decimal(hh) for x in u]
    gv = np.array([ld(str(g_exact(r))) for r in rs], dtype=ld)
    # weight: relative error of P = 1 + r + r^2 g  -> error r^2 dg / exp(r); interpolation at Chebyshev nodes is near-minimax for absolute error of g
    V = np.vander(np.array(u, dtype=ld), n, increasing=True).astype(ld)
    # solve in long double via numpy (falls back to double for linalg) -> do Gaussian elimination by hand in longdouble
    A = V.copy(); b = gv.copy()
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]; b[[i, p]] = b[[p, i]]
        for j in range(i+1, n):
            f = A[j, i] / A[i, i]
            A[j] -= f * A[i]; b[j] -= f * b[i]
    c = np.zeros(n, dtype=ld)
    for i in range(n-1, -1, -1):
        c[i] = (b[i] - np.dot(A[i, i+1:], c[i+1:])) / A[i, i]
    # c are coefficients in u = r / hh  -> monomial in r
    cr = np.array([c[i] / ld(hh)**i for i in range(n)], dtype=ld)
    return [float(x) for x in cr]
def maxerr(coef, taylor=False):
    worst = Decimal(0)
    N = 4001
    for i in range(N):
        r = Decimal(-1) * h + (2*h) * Decimal(i) / Decimal(N-1)
        rf = Decimal(float(r))
        q = Decimal(0)
        for cc in reversed(coef):
            q = q * rf + Decimal(cc)
        P = 1 + rf + rf*rf*q
        e = abs(P / rf.exp() - 1)
        worst = max(worst, e)
    return worst
import math
tay13 = [1.0/math.factorial(k) for k in range(2, 14)]
print('taylor-13 truncation', maxerr(tay13))
for nt in (8, 9, 10):
    c = fit(nt)
    print('tail degree', nt, 'total', nt+2, 'max rel err (exact arithmetic, double coefficients):', maxerr(c))
    print('  ', ', '.join('%.20e' % x for x in c))
