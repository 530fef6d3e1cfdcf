"""Test infrastructure (uses the oracle).  One engine, many data sets: the context is re-used across random problems of changing kind, state
count (1..40), alphabet, trajectory count and length, chunk length -- every E-step (with and without
gamma rows), Viterbi and sampled path against the oracle.  Catches state that survives
set_observations (calibration, fallback flags, buffers sized for an earlier problem).
usage: python tests/sweeps/stress_reuse.py [seed [cases]]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
eng = Engine(0)
for case in range(ncase):
    n = int(rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 17, 33, 40]))
    kind = "gaussian" if rng.random() < 0.5 else "discrete"
    K = int(rng.integers(1, 9))
    lens = [int(x) for x in rng.integers(1, int(rng.choice([30, 500, 6000])), K)]
    chunk = int(rng.choice([0, 0, 5, 32, 200]))
    A = rng.random((n, n)) + rng.choice([0.0, 3.0]) * np.eye(n); A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        par = (np.sort(rng.normal(0, 3, n)), rng.uniform(0.4, 2.0, n)); M = 0
        obs = [rng.normal(0, 3, T) for T in lens]
    else:
        M = int(rng.choice([2, 30, 64, 1300]))
        par = (rng.dirichlet(np.ones(M) * 0.5, size=n) * 0.98 + 0.02 / M, None)
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lens]
    tag = "case %d: %s n=%d M=%d K=%d lens=%s chunk=%d" % (case, kind, n, M, K, lens, chunk)
    try:
        ref = orc.estep(kind, obs, A, pi, *par, want_gamma=True)
        eng.set_observations(kind, obs, n, nsymbols=M, chunk=chunk)
        for sg in (False, True, False):
            res = eng.estep(A, pi, *par, store_gamma=sg)
            ok = np.allclose(res.logL_k, ref["logL"], rtol=1e-10, atol=1e-10) and np.allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10) \
                and np.allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10)
            if sg:
                for k in range(K):
                    ok = ok and np.allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-12)
            if not ok:
                bad += 1
                print("ESTEP MISMATCH", tag, "gamma" if sg else "")
        pobs = [orc.pobs_gaussian(o, *par) if kind == "gaussian" else orc.pobs_discrete(o, par[0]) for o in obs]
        vp = eng.viterbi(A, pi, *par)
        if not all(np.array_equal(p, orc.viterbi(A, po, pi)) for p, po in zip(vp, pobs)):
            bad += 1
            print("VITERBI MISMATCH", tag)
        u = [rng.random(T) for T in lens]
        sp = eng.sample_paths(A, pi, *par, u=u)[0]
        if not all(np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu)) for p, po, uu in zip(sp, pobs, u)):
            bad += 1
            print("SAMPLE MISMATCH", tag)
    except Exception as e:  # noqa
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300])
eng.close()
print("stress_reuse: %d cases, %d failures" % (ncase, bad))
sys.exit(1 if bad else 0)
