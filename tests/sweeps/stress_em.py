"""Test infrastructure (uses the oracle).  Random sweep of whole estimations: MaximumLikelihoodEstimator on the HIP engine against the same
run on the oracle-backed CPU engine double (tests/oracle_engine.py): likelihood history, parameters,
Viterbi paths -- gaussian and discrete, 2..6 states, reversible or not, a few trajectories, with and
without an initial model.  usage: python tests/sweeps/stress_em.py [seed [cases]]"""
import os, sys, warnings
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import bhmm_amd
from oracle_engine import OracleEngine
warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(ncase):
    n = int(rng.integers(2, 7))
    kind = "gaussian" if rng.random() < 0.6 else "discrete"
    K = int(rng.integers(1, 6))
    lens = [int(x) for x in rng.integers(50, int(rng.choice([300, 3000])), K)]
    rev = bool(rng.random() < 0.5)
    A = rng.random((n, n)) + 4 * np.eye(n); A /= A.sum(axis=1, keepdims=True)
    # generate data from a random model
    if kind == "gaussian":
        mu, sig = np.sort(rng.normal(0, 4, n)), rng.uniform(0.4, 1.2, n)
    else:
        M = int(rng.choice([3, 10, 40]))
        B = rng.dirichlet(np.ones(M) * 0.3, size=n)
    obs = []
    for T in lens:
        s = np.empty(T, dtype=int); s[0] = rng.integers(0, n)
        for t in range(1, T):
            s[t] = rng.choice(n, p=A[s[t - 1]])
        obs.append(rng.normal(mu[s], sig[s]) if kind == "gaussian" else np.array([rng.choice(M, p=B[x]) for x in s]))
    tag = "case %d: %s n=%d K=%d lens=%s reversible=%s" % (case, kind, n, K, lens, rev)
    try:
        A0 = rng.random((n, n)) + 2 * np.eye(n); A0 /= A0.sum(axis=1, keepdims=True)
        pi0 = np.full(n, 1.0 / n)
        if kind == "gaussian":
            init = bhmm_amd.gaussian_hmm(pi0, A0, mu + rng.normal(0, 0.3, n), sig * rng.uniform(0.8, 1.3, n))
        else:
            B0 = 0.7 * B + 0.3 / M
            init = bhmm_amd.discrete_hmm(pi0, A0, B0 / B0.sum(axis=1, keepdims=True))
        kw = dict(initial_model=init, reversible=rev, accuracy=1e-4, maxit=12)
        lag = int(rng.choice([1, 1, 2, 3, 5]))
        lobs = bhmm_amd.lag_observations(obs, lag) if lag > 1 else obs   # GPU: views cut on the device
        tag += " lag=%d" % lag
        ref = bhmm_amd.MaximumLikelihoodEstimator(lobs, n, engine_factory=OracleEngine, **kw)
        est = bhmm_amd.MaximumLikelihoodEstimator(lobs, n, **kw)
        try:
            href = ref.fit()
        except (RuntimeError, AssertionError) as e_ref:
            # a state collapsed onto a single observation (gaussian.py:271-272 raises there), or the
            # log-likelihood left the floating-point range (maximum_likelihood.py:385 asserts): the run
            # on the HIP engine has to end the same way
            try:
                est.fit()
                bad += 1
                print("MISMATCH", tag, "the oracle-engine run raised", repr(e_ref)[:80], "the HIP run did not")
            except (RuntimeError, AssertionError) as e_gpu:
                if type(e_gpu) is not type(e_ref) or (isinstance(e_ref, RuntimeError) and str(e_gpu) != str(e_ref)):
                    bad += 1
                    print("MISMATCH", tag, "different errors", repr(e_ref)[:80], repr(e_gpu)[:80])
            continue
        hmm = est.fit()
        ok = (len(est.likelihoods) == len(ref.likelihoods) and np.allclose(est.likelihoods, ref.likelihoods, rtol=1e-9)
              and np.allclose(hmm.transition_matrix, href.transition_matrix, rtol=1e-6, atol=1e-9))
        if kind == "gaussian":
            ok = ok and np.allclose(hmm.output_model.means, href.output_model.means, rtol=1e-7, atol=1e-9) \
                and np.allclose(hmm.output_model.sigmas, href.output_model.sigmas, rtol=1e-6)
        else:
            ok = ok and np.allclose(hmm.output_model.output_probabilities, href.output_model.output_probabilities, rtol=1e-6, atol=1e-9)
        vit = all(np.array_equal(a, b) for a, b in zip(hmm.hidden_state_trajectories, href.hidden_state_trajectories))
        if not ok or not vit:
            bad += 1
            print("MISMATCH", tag, "iterations", len(est.likelihoods), len(ref.likelihoods), "viterbi equal", vit,
                  "last likelihoods", est.likelihoods[-1:], ref.likelihoods[-1:])
    except Exception as e:  # noqa
        bad += 1
        print("EXCEPTION", tag, repr(e)[:300])
        if os.environ.get("VERBOSE"):
            import traceback
            traceback.print_exc()
print("stress_em: %d cases, %d failures" % (ncase, bad))
sys.exit(1 if bad else 0)
