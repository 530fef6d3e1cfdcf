"""Test infrastructure: the reference's recursions in 80-bit arithmetic (see estep_longdouble)."""
import numpy as np


def estep_longdouble(A, pi, pobs_list):
    """The reference's recursions (_hidden.c:16-183) on the reference's own double-precision pobs,
    carried out in 80-bit arithmetic (exponent range 2^+-16383): what the reference computes minus
    the rounding of its denormal intermediate sums."""
    L = np.longdouble
    A = A.astype(L); pi = pi.astype(L)
    n = A.shape[0]
    C = np.zeros((n, n), dtype=L)
    logL = []
    for pobs in pobs_list:
        p = pobs.astype(L)
        T = p.shape[0]
        alpha = np.empty((T, n), dtype=L); ll = L(0)
        a = pi * p[0]; c = a.sum(); ll += np.log(c); alpha[0] = a / c
        for t in range(1, T):
            a = (alpha[t - 1] @ A) * p[t]; c = a.sum(); ll += np.log(c); alpha[t] = a / c
        beta = np.full(n, L(1) / n)
        for t in range(T - 2, -1, -1):
            x = alpha[t][:, None] * A * (p[t + 1] * beta)[None, :]
            C += x / x.sum()
            beta = A @ (p[t + 1] * beta); beta /= beta.sum()
        logL.append(float(ll))
    return np.array(logL), C.astype(np.float64)


def hidden_longdouble(A, pobs, pi):
    """forward / backward / gamma / transition counts of one trajectory (_hidden.c:16-183, rows
    normalised by their sums like the reference's) in 80-bit arithmetic."""
    L = np.longdouble
    A = A.astype(L); pi = pi.astype(L); p = pobs.astype(L)
    T, n = p.shape
    alpha = np.empty((T, n), dtype=L); beta = np.empty((T, n), dtype=L)
    a = pi * p[0]; c = a.sum(); ll = np.log(c); alpha[0] = a / c
    for t in range(1, T):
        a = (alpha[t - 1] @ A) * p[t]; c = a.sum(); ll += np.log(c); alpha[t] = a / c
    beta[T - 1] = L(1) / n
    for t in range(T - 2, -1, -1):
        b = A @ (p[t + 1] * beta[t + 1]); beta[t] = b / b.sum()
    g = alpha * beta; g /= g.sum(axis=1, keepdims=True)
    C = np.zeros((n, n), dtype=L)
    for t in range(T - 1):
        x = alpha[t][:, None] * A * (p[t + 1] * beta[t + 1])[None, :]
        C += x / x.sum()
    return float(ll), alpha.astype(np.float64), beta.astype(np.float64), g.astype(np.float64), C.astype(np.float64)
