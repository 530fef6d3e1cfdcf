"""Test infrastructure (uses the oracle).  Replay one case of tests/sweeps/stress_hidden.py verbosely:
python tests/sweeps/hidden_case.py SEED CASE"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import bhmm_amd.hidden as hidden
from oracle import oracle as orc
from ld_reference import hidden_longdouble
hidden.set_implementation("hip")
rng = np.random.default_rng(int(sys.argv[1]))
want = int(sys.argv[2])
for case in range(want + 1):
    n = int(rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 24, 33, 64]))
    T = int(rng.choice([1, 2, 7, 50, 700, 5000]))
    A = rng.random((n, n)) + rng.choice([0.0, 3.0]) * np.eye(n)
    if n > 1 and rng.random() < 0.3:
        mask = rng.random((n, n)) < 0.5
        np.fill_diagonal(mask, True)
        A *= mask
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    spread = rng.choice([1.0, 30.0, 200.0, 720.0])
    pobs = np.exp(-spread * rng.random((T, n)))
    if rng.random() < 0.3:
        z = rng.random((T, n)) < 0.3
        z[np.arange(T), rng.integers(0, n, T)] = False
        pobs[z] = 0.0
    if case < want:
        # the sweep draws u = rng.random(T) after the GPU calls of every case it does not skip
        with np.errstate(all="ignore"):
            lr, ar = orc.forward(A, pobs, pi); br = orc.backward(A, pobs); Cr = orc.transition_counts(ar, br, A, pobs)
        if np.isfinite(lr) and np.all(np.isfinite(ar)) and np.all(np.isfinite(br)) and np.all(np.isfinite(Cr)):
            rng.random(T)
        continue
    print("case", case, "n", n, "T", T, "spread", spread, "A", A, "pi", pi)
    with np.errstate(all="ignore"):
        lr, ar = orc.forward(A, pobs, pi); br = orc.backward(A, pobs)
        l_ld, a_ld, b_ld, g_ld, C_ld = hidden_longdouble(A, pobs, pi)
    lg, ag = hidden.forward(A, pobs, pi)
    bg = hidden.backward(A, pobs)
    gg = hidden.state_probabilities(ag, bg)
    f = lambda x: np.asarray(x, dtype=np.float64)
    print("logL gpu %r ref %r 80-bit %r" % (lg, lr, float(l_ld)))
    for name, g_, r_, l_ in (("alpha", ag, ar, f(a_ld)), ("beta", bg, br, f(b_ld)), ("gamma", gg, orc.gamma(ar, br), f(g_ld))):
        nf = ~np.isfinite(g_)
        print(name, "non-finite gpu entries:", int(nf.sum()), "first bad row", int(np.argmax(nf.any(axis=1))) if nf.any() else None,
              "| row sums min/max", np.nanmin(g_.sum(axis=1)), np.nanmax(g_.sum(axis=1)))
        with np.errstate(all="ignore"):
            rel = np.abs(g_ - l_) / np.abs(l_)
        rel[~np.isfinite(rel)] = 0
        rel[np.abs(l_) < 1e-250] = 0
        t, i = np.unravel_index(np.argmax(rel), rel.shape)
        print("   worst vs 80-bit at t=%d state %d: rel %.3e gpu %r ref %r 80-bit %r; rows off by > 1e-9: %d" % (
            t, i, rel[t, i], g_[t, i], r_[t, i], l_[t, i], int((rel.max(axis=1) > 1e-9).sum())))
        for tt in range(max(0, t - 2), min(T, t + 3)):
            print("     t", tt, "gpu", g_[tt], "ref", r_[tt], "ld", l_[tt], "pobs", pobs[tt])
    nf = ~np.isfinite(ag)
    if nf.any():
        t = int(np.argmax(nf.any(axis=1)))
        print("first non-finite alpha row", t)
        for tt in range(max(0, t - 6), min(T, t + 3)):
            print("     t", tt, "gpu", ag[tt], "ref", ar[tt], "ld", f(a_ld)[tt], "pobs", pobs[tt])
    sys.exit(0)
    bg = hidden.backward(A, pobs)
    with np.errstate(all="ignore"):
        rel = np.abs(bg - br) / np.abs(br)
    rel[~np.isfinite(rel)] = 0
    t, i = np.unravel_index(np.argmax(rel), rel.shape)
    print("worst beta at t=%d state %d: gpu %r ref %r 80-bit %r" % (t, i, bg[t, i], br[t, i], float(b_ld[t, i])))
    for tt in range(max(0, t - 2), min(T, t + 4)):
        print(" t", tt, "gpu", bg[tt], "ref", br[tt], "ld", np.asarray(b_ld[tt], dtype=float), "pobs", pobs[tt])
    print("number of rows off by > 1e-9:", int((rel.max(axis=1) > 1e-9).sum()), "first", int(np.argmax(rel.max(axis=1) > 1e-9)))
