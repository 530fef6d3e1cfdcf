"""Test infrastructure (uses the oracle).  Random parity sweep of the single-trajectory kernel API (bhmm_amd.hidden: forward, backward,
state_probabilities, transition_counts, viterbi, sample_path on caller-supplied pobs) against the
oracle: 1..100 states, lengths 1..5000, dense and sparse A, pobs rows with exact zeros and with
entries spread over hundreds of decades.  The values are compared with the reference's recursions in
80-bit arithmetic (tests/ld_reference.py): where relative weights leave the double range (below
1e-308 of the row) the double-precision reference loses states for good -- an exact zero stays zero
under a transition matrix that does not refill it -- while the chunk-parallel kernels carry separate
exponents; paths are compared with the oracle's where its rows agree with the 80-bit ones.
usage: python tests/sweeps/stress_hidden.py [seed [cases]]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import bhmm_amd.hidden as hidden
from oracle import oracle as orc
from ld_reference import hidden_longdouble
hidden.set_implementation("hip")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = unreliable = 0
for case in range(ncase):
    n = int(rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 24, 33, 64, 65, 100]))
    T = int(rng.choice([1, 2, 7, 50, 700, 5000]))
    if n > 64 and T > 700:
        T = 700  # (the any-N family: one workgroup per trajectory)
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
    tag = "case %d: n=%d T=%d spread=%g" % (case, n, T, spread)
    try:
        with np.errstate(all="ignore"):
            lr, ar = orc.forward(A, pobs, pi)
            br = orc.backward(A, pobs)
            Cr = orc.transition_counts(ar, br, A, pobs)
            vr = orc.viterbi(A, pobs, pi)
        if not (np.isfinite(lr) and np.all(np.isfinite(ar)) and np.all(np.isfinite(br)) and np.all(np.isfinite(Cr))):
            continue
        lg, ag = hidden.forward(A, pobs, pi)
        bg = hidden.backward(A, pobs)
        gg = hidden.state_probabilities(ag, bg)
        Cg = hidden.transition_counts(ag, bg, A, pobs)
        vg = hidden.viterbi(A, pobs, pi)
        u = rng.random(T)
        try:
            sg = hidden.sample_path(ag, A, pobs, u=u)
            sr = orc.sample_path(ar, A, u=u)
        except Exception:
            sg = sr = None
        with np.errstate(all="ignore"):
            l_ld, a_ld, b_ld, g_ld, C_ld = hidden_longdouble(A, pobs, pi)
        ref_ok = (np.allclose(ar, a_ld, rtol=1e-6, atol=1e-250) and np.allclose(br, b_ld, rtol=1e-6, atol=1e-250)
                  and np.isclose(lr, l_ld, rtol=1e-9))
        if ref_ok:
            # values against the 80-bit recursion (a denormal emission entry costs the double-precision
            # reference a few 1e-9 of a beta row, seed 557 case 242; the kernels are closer to the
            # 80-bit values than it is), paths against the reference's
            f = lambda x: np.asarray(x, dtype=np.float64)
            # entry by entry: a row behind a denormal emission entry may carry the reference's rounding
            # (seed 4002 case 1395: kernels == reference there, 3e-9 off the 80-bit value) while other
            # rows of the same call are the 80-bit ones
            either = lambda g, l, r: bool(np.all(np.isclose(g, l, rtol=1e-9, atol=1e-250) | np.isclose(g, r, rtol=1e-9, atol=1e-250)))
            checks = {"logL": np.isclose(lg, float(l_ld), rtol=1e-11, atol=1e-11) or np.isclose(lg, lr, rtol=1e-11, atol=1e-11),
                      "alpha": either(ag, f(a_ld), ar),
                      "beta": either(bg, f(b_ld), br),
                      # gamma and the counts are functions of the rows they are GIVEN: the reference's
                      # routines on the very same rows (the kernels' alpha may be the double-precision
                      # reference's and their beta the 80-bit one's -- each within its own check)
                      "gamma": np.allclose(gg, orc.gamma(ag, bg), rtol=1e-9, atol=1e-250),
                      "C": np.allclose(Cg, orc.transition_counts(ag, bg, A, pobs), rtol=1e-8, atol=1e-12),
                      "C end to end": np.allclose(Cg, f(C_ld), rtol=1e-6, atol=1e-9) or np.allclose(Cg, Cr, rtol=1e-6, atol=1e-9),
                      "viterbi": np.array_equal(vg, vr),
                      "sample": sg is not None and np.array_equal(sg, sr)}
        else:
            # the double-precision reference has lost states the 80-bit recursion keeps (relative
            # weights below 1e-308 that matter again later: reducible / sparse A): no parity claim
            # there, the outputs only have to be finite and normalised
            unreliable += 1
            checks = {"finite": bool(np.isfinite(lg) and np.all(np.isfinite(ag)) and np.all(np.isfinite(bg))),
                      "alpha rows sum to one": np.allclose(ag.sum(axis=1), 1.0, rtol=1e-12)}
        for k, ok in checks.items():
            if not ok:
                bad += 1
                print("MISMATCH", k, tag)
                if os.environ.get("VERBOSE") and k in ("alpha", "beta", "gamma"):
                    g_, r_ = {"alpha": (ag, ar), "beta": (bg, br), "gamma": (gg, orc.gamma(ar, br))}[k]
                    with np.errstate(all="ignore"):
                        rel = np.abs(g_ - r_) / np.abs(r_)
                    rel[~np.isfinite(rel)] = 0
                    t, i = np.unravel_index(np.argmax(rel), rel.shape)
                    print("   worst at t=%d state %d: gpu %r ref %r (row gpu %s ref %s, pobs row %s)" % (t, i, g_[t, i], r_[t, i], g_[t], r_[t], pobs[t]))
                    print("   non-finite gpu entries:", int((~np.isfinite(g_)).sum()))
    except Exception as e:  # noqa
        if os.environ.get("SAVE"):
            np.savez(os.path.join(os.environ["SAVE"], "hidden_case_%s_%d.npz" % (sys.argv[1] if len(sys.argv) > 1 else "3", case)), A=A, pi=pi, pobs=pobs, u=u, ag=ag, ar=ar)
        if os.environ.get("VERBOSE"):
            import traceback; traceback.print_exc()
            print("   alpha gpu finite:", bool(np.all(np.isfinite(ag))), "row sums min/max", ag.sum(1).min(), ag.sum(1).max())
            print("   alpha ref row sums min/max", ar.sum(1).min(), ar.sum(1).max())
        bad += 1
        print("EXCEPTION", tag, repr(e)[:200])
print("stress_hidden: %d cases, %d failures (%d cases where the double-precision reference itself is off the 80-bit recursion)" % (ncase, bad, unreliable))
sys.exit(1 if bad else 0)
