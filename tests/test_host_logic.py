"""Host-side logic on CPU: M-step estimators, partitioning, EM-loop semantics of
MaximumLikelihoodEstimator / BayesianHMMSampler (driven by the oracle-backed test double),
closed-form cases of bhmm/tests/test_mlhmm_patho.py."""
import numpy as np
import pytest

import bhmm_amd
from bhmm_amd.estimators import _tmatrix
from bhmm_amd.sharding import lpt_partition
from oracle_engine import OracleEngine


def test_lpt_partition_balances_and_is_deterministic():
    lengths = [100, 90, 10, 10, 50, 50, 30, 1]
    parts = lpt_partition(lengths, 3)
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(lengths[k] for k in p) for p in parts]
    assert max(loads) - min(loads) <= 20
    assert parts == lpt_partition(lengths, 3)
    assert lpt_partition([5, 5], 4)[2] == [] and lpt_partition([5, 5], 1) == [[0, 1]]


def test_nonreversible_mstep_is_row_normalisation():
    C = np.array([[4.0, 1.0, 0.0], [2.0, 2.0, 0.0], [0.0, 0.0, 0.0]])
    P = _tmatrix.estimate_P(C, reversible=False)
    assert np.allclose(P, [[0.8, 0.2, 0], [0.5, 0.5, 0], [0, 0, 1]])  # empty row -> C_ii = 1


def test_reversible_mle_properties():
    rng = np.random.default_rng(0)
    C = rng.integers(1, 50, (5, 5)).astype(float)
    P = _tmatrix.estimate_P(C, reversible=True, maxerr=1e-13)
    assert _tmatrix.is_transition_matrix(P)
    pi = _tmatrix.stationary_vector(P)
    X = pi[:, None] * P
    assert np.allclose(X, X.T, atol=1e-10)                  # detailed balance
    assert np.allclose(pi @ P, pi, atol=1e-10)
    # likelihood optimality among reversible matrices: symmetric counts -> closed form
    Cs = C + C.T
    assert np.allclose(_tmatrix.estimate_P(Cs, reversible=True, maxerr=1e-13),
                       Cs / Cs.sum(axis=1)[:, None], atol=1e-9)
    ll = lambda Q: np.sum(C * np.log(Q))
    for _ in range(20):                                      # random reversible competitors
        S = rng.random((5, 5)); S = S + S.T
        assert ll(P) >= ll(S / S.sum(axis=1)[:, None]) - 1e-9
    # fixed stationary distribution
    pi_fix = np.array([0.1, 0.2, 0.3, 0.25, 0.15])
    Pf = _tmatrix.estimate_P(C, reversible=True, fixed_statdist=pi_fix, maxerr=1e-13)
    assert np.allclose(pi_fix @ Pf, pi_fix, atol=1e-8)
    assert np.allclose(pi_fix[:, None] * Pf, (pi_fix[:, None] * Pf).T, atol=1e-8)


def test_partial_reversible_closed_form():
    # test_mlhmm_patho.py:41-55: A = [[0.8, 0.2], [0, 1]] from C = [[4, 1], [0, 3]]
    C = np.array([[4.0, 1.0], [0.0, 3.0]])
    P = _tmatrix.estimate_P(C, reversible=True, maxerr=1e-12, mincount_connectivity=1e-16)
    assert np.allclose(P, [[0.8, 0.2], [0.0, 1.0]], atol=1e-8)
    pi = _tmatrix.stationary_distribution(P, C=C)
    assert np.allclose(pi.sum(), 1.0)


def _discrete_init(B=None):
    A = np.array([[0.7, 0.3], [0.2, 0.8]])
    B = np.array([[0.8, 0.2], [0.3, 0.7]]) if B is None else B
    return bhmm_amd.discrete_hmm([0.5, 0.5], A, B)


def test_mlhmm_patho_2state_step():
    """bhmm/tests/test_mlhmm_patho.py:41-55 (closed-form EM answer, up to permutation)."""
    obs = np.array([0, 0, 0, 0, 0, 1, 1, 1, 1], dtype=int)
    hmm = bhmm_amd.estimate_hmm([obs], nstates=2, lag=1, accuracy=1e-6,
                                initial_model=_discrete_init(), engine_factory=OracleEngine)
    assert np.allclose(hmm.initial_distribution, [1, 0], atol=1e-5)
    assert np.allclose(hmm.transition_matrix, [[0.8, 0.2], [0.0, 1.0]], atol=1e-5)
    assert np.allclose(hmm.output_model.output_probabilities, np.eye(2), atol=1e-5)
    assert np.array_equal(hmm.hidden_state_trajectories[0], [0, 0, 0, 0, 0, 1, 1, 1, 1])
    assert hmm.hidden_state_trajectories[0].dtype == np.int32


def test_mlhmm_patho_2state_2step():
    """bhmm/tests/test_mlhmm_patho.py:57-71."""
    obs = np.array([0, 1, 0], dtype=int)
    init = bhmm_amd.discrete_hmm([0.6, 0.4], [[0.3, 0.7], [0.6, 0.4]], [[0.8, 0.2], [0.3, 0.7]])
    hmm = bhmm_amd.estimate_hmm([obs], nstates=2, lag=1, accuracy=1e-6, initial_model=init,
                                engine_factory=OracleEngine)
    assert np.allclose(hmm.initial_distribution, [1, 0], atol=1e-5)
    assert np.allclose(hmm.transition_matrix, [[0, 1], [1, 0]], atol=1e-5)
    assert np.allclose(hmm.output_model.output_probabilities, np.eye(2), atol=1e-5)


def test_mlhmm_patho_1state():
    """bhmm/tests/test_mlhmm_patho.py:27-35."""
    obs = np.array([0, 0, 0, 0, 0], dtype=int)
    init = bhmm_amd.discrete_hmm([1.0], [[1.0]], [[1.0]])
    hmm = bhmm_amd.estimate_hmm([obs], nstates=1, lag=1, accuracy=1e-6, initial_model=init,
                                engine_factory=OracleEngine)
    assert np.allclose(hmm.transition_matrix, [[1.0]])
    assert np.allclose(hmm.output_model.output_probabilities, [[1.0]])


def _gauss_problem(seed=0, K=5, T=400):
    rng = np.random.default_rng(seed)
    A = np.array([[0.95, 0.05, 0.0], [0.03, 0.9, 0.07], [0.0, 0.1, 0.9]])
    mu, sig = np.array([-2.0, 0.5, 3.0]), np.array([0.6, 0.5, 0.9])
    obs = []
    for k in range(K):
        s = np.zeros(T + 37 * k, dtype=int)
        for t in range(1, len(s)):
            s[t] = rng.choice(3, p=A[s[t - 1]])
        obs.append(rng.normal(mu[s], sig[s]))
    init = bhmm_amd.gaussian_hmm([0.4, 0.3, 0.3], 0.8 * A + 0.2 / 3, mu + 0.4, sig * 1.3)
    return obs, init


def test_em_loop_semantics_and_monotone_likelihood():
    obs, init = _gauss_problem()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=False,
                                              accuracy=1e-4, maxit=60, store_gamma=True,
                                              engine_factory=OracleEngine)
    hmm = est.fit()
    L = est.likelihoods
    assert len(L) >= 3 and len(L) <= 60
    assert np.all(np.diff(L) > -1e-8)                       # EM never decreases the likelihood
    assert (L[-1] - L[-2]) < 1e-4                           # stopped by the signed criterion
    assert hmm.likelihood == L[-1] == est.likelihood        # value before the last M-step
    assert np.allclose(est.count_matrix.sum(), sum(len(o) - 1 for o in obs))
    assert np.allclose(est.initial_count.sum(), len(obs))
    assert np.allclose(hmm.transition_matrix.sum(axis=1), 1.0)
    assert len(hmm.hidden_state_trajectories) == len(obs)
    g = est.hidden_state_probabilities
    assert g[2].shape == (len(obs[2]), 3) and np.allclose(g[2].sum(axis=1), 1.0)
    assert np.allclose(sorted(hmm.output_model.means), [-2.0, 0.5, 3.0], atol=0.25)
    # maxit stops the loop without convergence
    est2 = bhmm_amd.MLHMM(obs, 3, initial_model=init, reversible=False, accuracy=1e-12, maxit=3,
                          engine_factory=OracleEngine)
    est2.fit()
    assert len(est2.likelihoods) == 3
    assert np.allclose(est2.likelihoods, L[:3], rtol=1e-12)


def test_estimator_argument_handling():
    obs, init = _gauss_problem(K=2, T=50)
    # no initial model: gaussian data get the heuristic start, discrete data must bring one
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, engine_factory=OracleEngine)
    assert est.hmm.nstates == 3 and est.hmm.output_model.model_type == 'gaussian'
    with pytest.raises(NotImplementedError):
        bhmm_amd.MaximumLikelihoodEstimator([np.array([0, 1, 1, 0, 2])], 2, output='discrete',
                                            engine_factory=OracleEngine)
    with pytest.raises(ValueError):
        bhmm_amd.MaximumLikelihoodEstimator(obs, 2, initial_model=init, engine_factory=OracleEngine)
    # fixed initial distribution (p with stationary=False), maximum_likelihood.py:118-126
    p = np.array([0.2, 0.3, 0.5])
    est = bhmm_amd.MLHMM(obs, 3, initial_model=init, reversible=False, p=p, maxit=2,
                         engine_factory=OracleEngine)
    assert np.allclose(est.fit().initial_distribution, p)
    # stationary=True: pi is the stationary vector of T
    est = bhmm_amd.MLHMM(obs, 3, initial_model=init, reversible=True, stationary=True, maxit=2,
                         engine_factory=OracleEngine)
    h = est.fit()
    assert np.allclose(h.initial_distribution @ h.transition_matrix, h.initial_distribution,
                       atol=1e-8)
    assert bhmm_amd.MaximumLikelihoodHMM is bhmm_amd.MLHMM and bhmm_amd.BayesianHMM is bhmm_amd.BHMM
    lagged = bhmm_amd.lag_observations([np.arange(10)], 3)
    assert [list(x) for x in lagged] == [[0, 3, 6, 9], [1, 4, 7], [2, 5, 8]]


def test_bayesian_sampler_sweeps():
    obs, init = _gauss_problem(K=3, T=300)
    mle = bhmm_amd.estimate_hmm(obs, 3, initial_model=init, reversible=False, maxit=20,
                                engine_factory=OracleEngine)
    np.random.seed(0)
    models = bhmm_amd.bayesian_hmm(obs, mle, nsample=12, reversible=False, store_hidden=True,
                                   engine_factory=OracleEngine)
    assert len(models) == 12
    means = np.array([m.output_model.means for m in models])
    assert means.std(axis=0).min() > 0                       # parameters move
    assert np.allclose(means.mean(axis=0), mle.output_model.means, atol=0.3)
    for m in models:
        assert np.allclose(m.transition_matrix.sum(axis=1), 1.0)
        assert len(m.hidden_state_trajectories) == 3
        assert m.hidden_state_trajectories[0].shape == (300,)
    # reversible sampling keeps detailed balance; disconnected + reversible is refused
    models = bhmm_amd.bayesian_hmm(obs, mle, nsample=2, reversible=True, engine_factory=OracleEngine)
    for m in models:
        assert _tmatrix.is_reversible(m.transition_matrix)
    dis = bhmm_amd.gaussian_hmm([0.5, 0.5], np.eye(2), [0.0, 1.0], [1.0, 1.0])
    with pytest.raises(NotImplementedError):                 # bayesian_sampling.py:187-191
        bhmm_amd.BayesianHMMSampler(obs, 2, initial_model=dis, reversible=True,
                                    transition_matrix_prior=None, engine_factory=OracleEngine)


def test_gaussian_initial_model_heuristic():
    """bhmm_amd.init (the role of bhmm/init/gaussian.py:26-92): mixture fit from a deterministic
    start, fractional transition counts as one matrix product (checked against the reference's
    per-step outer-product loop), a proper HMM out."""
    from bhmm_amd.init.gaussian import fit_gmm1d, fractional_counts, init_model_gaussian1d
    rng = np.random.default_rng(5)
    mu, sg = np.array([-3.0, 0.0, 4.0]), np.array([0.5, 1.0, 0.7])
    P = np.array([[0.9, 0.1, 0.0], [0.05, 0.9, 0.05], [0.0, 0.1, 0.9]])
    obs = []
    for T in (4000, 2500):
        s = np.zeros(T, dtype=int)
        for t in range(1, T):
            s[t] = rng.choice(3, p=P[s[t - 1]])
        obs.append(mu[s] + sg[s] * rng.standard_normal(T))
    w, m, s_ = fit_gmm1d(np.concatenate(obs), 3)
    np.testing.assert_allclose(m, mu, atol=0.15)
    np.testing.assert_allclose(s_, sg, atol=0.15)
    assert abs(w.sum() - 1) < 1e-12 and np.all(np.diff(m) > 0)
    assert np.array_equal(fit_gmm1d(np.concatenate(obs), 3)[1], m)       # reproducible
    # fractional counts: the reference's loop, init/gaussian.py:66-78
    N = fractional_counts(obs, m, s_)
    Nref = np.zeros((3, 3))
    for o in obs:
        p = np.exp(-0.5 * ((o[:, None] - m[None, :]) / s_[None, :]) ** 2) / (np.sqrt(2 * np.pi) * s_)
        p /= p.sum(axis=1)[:, None]
        for t in range(len(o) - 1):
            Nref += np.outer(p[t], p[t + 1])
    np.testing.assert_allclose(N, Nref, rtol=1e-9)
    assert abs(N.sum() - sum(len(o) - 1 for o in obs)) < 1e-6
    for rev in (True, False):
        hmm = init_model_gaussian1d(obs, 3, reversible=rev)
        T = hmm.transition_matrix
        np.testing.assert_allclose(T.sum(axis=1), 1.0, atol=1e-12)
        np.testing.assert_allclose(hmm.initial_distribution @ T, hmm.initial_distribution, atol=1e-8)
        np.testing.assert_allclose(np.diag(T), np.diag(P), atol=0.12)   # metastable as generated (a start, not a fit)
        assert hmm.output_model.model_type == 'gaussian'
