"""Pure-Python restatement of the counter-based generator of bhmm_amd/csrc/host_model.cpp (TEST
INFRASTRUCTURE): uniform, ziggurat normal, Marsaglia-Tsang gamma, beta, chi-square, Dirichlet.
Python's math module calls the same C libm, so a draw sequence is reproduced to the last bit and
the algebra around the draws (Gibbs emission / Dirichlet updates) can be checked exactly."""
import math

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _zig_tables():
    C, r, v = 128, 3.442619855899, 9.91256303526217e-3
    X = [0.0] * (C + 1)
    f = math.exp(-0.5 * r * r)
    X[0] = v / f
    X[1] = r
    X[C] = 0.0
    for i in range(2, C):
        X[i] = math.sqrt(-2.0 * math.log(v / X[i - 1] + f))
        f = math.exp(-0.5 * X[i] * X[i])
    R = [X[i + 1] / X[i] for i in range(C)]
    return X, R


_ZX, _ZR = _zig_tables()


class Rng(object):
    def __init__(self, seed, stream):
        self.key = mix64(mix64(seed) + GOLDEN * (stream + 1))
        self.ctr = 0

    def bits(self):
        self.ctr += 1
        return mix64(self.key + GOLDEN * self.ctr)

    def u01(self):
        return float(self.bits() >> 11) * (1.0 / 9007199254740992.0)

    def u01_open(self):
        return (float(self.bits() >> 12) + 0.5) * (1.0 / 4503599627370496.0)

    def normal(self):
        r = 3.442619855899
        while True:
            w = self.bits()
            i = w & 127
            u = float(w >> 11) * (1.0 / 4503599627370496.0) - 1.0
            if abs(u) < _ZR[i]:
                return u * _ZX[i]
            if i == 0:
                while True:
                    x = math.log(self.u01_open()) / r
                    y = math.log(self.u01_open())
                    if not (-2.0 * y < x * x):
                        break
                return x - r if u < 0.0 else r - x
            x = u * _ZX[i]
            f0 = math.exp(-0.5 * (_ZX[i] * _ZX[i] - x * x))
            f1 = math.exp(-0.5 * (_ZX[i + 1] * _ZX[i + 1] - x * x))
            if f1 + self.u01() * (f0 - f1) < 1.0:
                return x

    def gamma(self, k):
        if not k > 0.0:
            return 0.0
        if k < 1.0:
            g = self.gamma(k + 1.0)
            return g * math.pow(self.u01_open(), 1.0 / k)
        d = k - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        while True:
            x = self.normal()
            v = 1.0 + c * x
            if v <= 0.0:
                continue
            v = v * v * v
            u = self.u01_open()
            x2 = x * x
            if u < 1.0 - 0.0331 * x2 * x2:
                return d * v
            if math.log(u) < 0.5 * x2 + d * (1.0 - v + math.log(v)):
                return d * v

    def chisquare(self, df):
        return 2.0 * self.gamma(0.5 * df)

    def dirichlet(self, alpha, out):
        g = [self.gamma(a) if a > 0.0 else 0.0 for a in alpha]
        tot = 0.0
        for a, x in zip(alpha, g):
            if a > 0.0:
                tot += x
        for i, a in enumerate(alpha):
            if a > 0.0:
                out[i] = g[i] / tot
        return out
