"""Coefficients of exp_nonpos() in bhmm_amd/csrc/estep_sweep.hpp.

exp(r) = 1 + r + r^2 q(r) on |r| <= 0.3475 (half of ln 2 plus slack for the rounding of
k = rint(x log2 e)); q of degree 9 by Remez-style exchange on the relative error of the full
expression, in 60-digit arithmetic, then rounded to double and re-checked as rounded.
"""
import mpmath as mp

mp.mp.dps = 60
A = mp.mpf('0.3475')
DEG = 9


def g(r):
    return (mp.e ** r - 1 - r) / (r * r) if r != 0 else mp.mpf(1) / 2


def fit():
    n = DEG + 2
    # start from Chebyshev nodes, nudged off 0 where the weight is singular
    xs = [A * mp.cos(mp.pi * (2 * i + 1) / (2 * n)) + mp.mpf('1e-3') for i in range(n)]
    for _ in range(12):
        # solve sum_j c_j x^j + (-1)^i E w(x) = g(x), weight w = exp(x)/x^2 (relative error of exp)
        M = mp.matrix(n, n)
        b = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for j in range(DEG + 1):
                M[i, j] = x ** j
            M[i, DEG + 1] = (-1) ** i * mp.e ** x / (x * x)
            b[i] = g(x)
        sol = mp.lu_solve(M, b)
        c = [sol[j] for j in range(DEG + 1)]

        def err(x):
            return (1 + x + x * x * mp.polyval(c[::-1], x)) / mp.e ** x - 1
        # new extrema by dense search between sign changes
        grid = [-A + 2 * A * k / 4000 for k in range(4001)]
        vals = [err(x) for x in grid]
        ext = []
        for k in range(1, 4000):
            if abs(vals[k]) >= abs(vals[k - 1]) and abs(vals[k]) >= abs(vals[k + 1]):
                ext.append((grid[k], vals[k]))
        ext = [(grid[0], vals[0])] + ext + [(grid[-1], vals[-1])]
        # keep alternating largest
        pick = []
        for x, v in ext:
            if pick and (v > 0) == (pick[-1][1] > 0):
                if abs(v) > abs(pick[-1][1]):
                    pick[-1] = (x, v)
            else:
                pick.append((x, v))
        while len(pick) > n:
            if abs(pick[0][1]) < abs(pick[-1][1]):
                pick.pop(0)
            else:
                pick.pop()
        if len(pick) < n:
            break
        xs = [p[0] for p in pick]
    return c


c = fit()
cd = [float(x) for x in c]
worst = 0
for k in range(20001):
    x = -A + 2 * A * mp.mpf(k) / 20000
    p = mp.mpf(0)
    for cj in cd[::-1]:
        p = p * x + mp.mpf(cj)
    v = abs((1 + x + x * x * p) / mp.e ** x - 1)
    worst = max(worst, v)
print("max relative error of the rounded polynomial (exact arithmetic): %.3e" % float(worst))
for j, cj in enumerate(cd):
    print("c%d = %s  // %s" % (j + 2, repr(cj), cj.hex()))
