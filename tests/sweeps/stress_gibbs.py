"""Test infrastructure (uses the oracle).  Random sweep of Gibbs chains: BayesianHMMSampler on the HIP engine against the same chain on the
oracle-backed CPU engine double (same numpy seed for the parameter draws, same counter-based uniforms
for the hidden paths): the sampled models must coincide.  usage: python tests/sweeps/stress_gibbs.py [seed [cases]]"""
import os, sys, warnings
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import bhmm_amd
from oracle_engine import OracleEngine
warnings.simplefilter("ignore")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for case in range(ncase):
    n = int(rng.integers(2, 6))
    kind = "gaussian" if rng.random() < 0.6 else "discrete"
    K = int(rng.integers(1, 5))
    lens = [int(x) for x in rng.integers(60, int(rng.choice([300, 2500])), K)]
    rev = bool(rng.random() < 0.5)
    A = rng.random((n, n)) + 4 * np.eye(n); A /= A.sum(axis=1, keepdims=True)
    if kind == "gaussian":
        mu, sig = np.sort(rng.normal(0, 4, n)), rng.uniform(0.4, 1.2, n)
    else:
        M = int(rng.choice([3, 10, 40])); B = rng.dirichlet(np.ones(M) * 0.3, size=n)
    obs = []
    for T in lens:
        s = np.empty(T, dtype=int); s[0] = rng.integers(0, n)
        for t in range(1, T):
            s[t] = rng.choice(n, p=A[s[t - 1]])
        obs.append(rng.normal(mu[s], sig[s]) if kind == "gaussian" else np.array([rng.choice(M, p=B[x]) for x in s]))
    tag = "case %d: %s n=%d K=%d lens=%s reversible=%s" % (case, kind, n, K, lens, rev)
    try:
        pi0 = np.full(n, 1.0 / n)
        init = bhmm_amd.gaussian_hmm(pi0, A, mu, sig) if kind == "gaussian" else bhmm_amd.discrete_hmm(pi0, A, 0.9 * B + 0.1 / M)
        chains = []
        for factory in (OracleEngine, None):
            np.random.seed(1234 + case)
            smp = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=init, reversible=rev, engine_factory=factory,
                                              transition_matrix_sampling_steps=50)
            chains.append(smp.sample(4, save_hidden_state_trajectory=True))
        ok = True
        for a, b in zip(*chains):
            ok = ok and np.allclose(a.transition_matrix, b.transition_matrix, rtol=1e-9, atol=1e-12)
            if kind == "gaussian":
                ok = ok and np.allclose(a.output_model.means, b.output_model.means, rtol=1e-9, atol=1e-12)
                ok = ok and np.allclose(a.output_model.sigmas, b.output_model.sigmas, rtol=1e-9)
            else:
                ok = ok and np.allclose(a.output_model.output_probabilities, b.output_model.output_probabilities, rtol=1e-9, atol=1e-12)
            ok = ok and all(np.array_equal(x, y) for x, y in zip(a.hidden_state_trajectories, b.hidden_state_trajectories))
        if not ok:
            bad += 1
            print("MISMATCH", tag)
    except Exception as e:  # noqa
        bad += 1
        import traceback
        print("EXCEPTION", tag, repr(e)[:300]); traceback.print_exc()
print("stress_gibbs: %d cases, %d failures" % (ncase, bad))
sys.exit(1 if bad else 0)
