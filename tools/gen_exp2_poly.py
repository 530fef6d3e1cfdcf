"""Coefficients of gauss_pdf() in bhmm_amd/csrc/estep_sweep.hpp.

2^f = sum_j c_j f^j on |f| <= 1/2 (f = v - rint(v) is exact there, no slack needed), minimax in the
relative error by Remez exchange in 60-digit arithmetic, rounded to double and re-checked as
rounded.  The kernel evaluates the polynomial in w = -f / 4096 (an exact rescaling: the
coefficients printed for the kernel are c_j (-4096)^j).
usage: python tools/gen_exp2_poly.py [degree]
"""
import sys
import mpmath as mp

mp.mp.dps = 60
A = mp.mpf('0.5')
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 11
SCALE = -4096


def target(x):
    return mp.mpf(2) ** x


def fit():
    n = DEG + 2
    xs = [A * mp.cos(mp.pi * (2 * i + 1) / (2 * n)) for i in range(n)]
    c = None
    for _ in range(15):
        M = mp.matrix(n, n)
        b = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for j in range(DEG + 1):
                M[i, j] = x ** j
            M[i, DEG + 1] = (-1) ** i * target(x)   # relative error
            b[i] = target(x)
        sol = mp.lu_solve(M, b)
        c = [sol[j] for j in range(DEG + 1)]

        def err(x):
            return mp.polyval(c[::-1], x) / target(x) - 1
        grid = [-A + 2 * A * k / 4000 for k in range(4001)]
        vals = [err(x) for x in grid]
        ext = [(grid[0], vals[0])]
        for k in range(1, 4000):
            if abs(vals[k]) >= abs(vals[k - 1]) and abs(vals[k]) >= abs(vals[k + 1]):
                ext.append((grid[k], vals[k]))
        ext.append((grid[-1], vals[-1]))
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
    worst = max(worst, abs(p / target(x) - 1))
print("degree %d: max relative error of the rounded polynomial (exact arithmetic): %.3e" % (DEG, float(worst)))
for j, cj in enumerate(cd):
    s = cj * float(SCALE) ** j          # exact: a power of two
    print("    %-28s // c%-2d = %s" % (s.hex() + ",", j, repr(cj)))
