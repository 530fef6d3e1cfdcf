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
    # no initial model: both kinds of data get the heuristic start of bhmm.init_hmm
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, engine_factory=OracleEngine)
    assert est.hmm.nstates == 3 and est.hmm.output_model.model_type == 'gaussian'
    est = bhmm_amd.MaximumLikelihoodEstimator([np.array([0, 1, 1, 0, 2, 2, 1, 0])], 2,
                                              output='discrete', engine_factory=OracleEngine)
    assert est.hmm.nstates == 2 and est.hmm.output_model.model_type == 'discrete'
    assert est.hmm.output_model.output_probabilities.shape == (2, 3)
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
    # no initial model: a maximum-likelihood fit from the heuristic start (bayesian_sampling.py:375-385)
    smp = bhmm_amd.BayesianHMMSampler(obs, 3, reversible=False, engine_factory=OracleEngine)
    assert smp.model.nstates == 3 and np.all(np.diff(smp.model.output_model.means) > 0)
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


def _markov_chain(P, T, seed):
    rng = np.random.RandomState(seed)
    cs = np.cumsum(P, axis=1)
    u = rng.random_sample(T)
    s = np.zeros(T, dtype=np.int32)
    for t in range(1, T):
        s[t] = min(np.searchsorted(cs[s[t - 1]], u[t]), P.shape[0] - 1)
    return s


_SPLIT = np.array([0, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,
                   0, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 0, 1, 2, 2, 2, 2, 2, 2])


def test_discrete_initial_model_reference_values():
    """The known answers of bhmm/tests/test_init_discrete.py:182-213 (state splitting with and
    without empty labels): they pin the count matrix, the neighbour prior, the reversible
    estimator, the coarse-graining and the regularisation."""
    piref = np.array([0.35801876, 0.55535398, 0.08662726])
    Aref = np.array([[0.76462978, 0.10261978, 0.13275044],
                     [0.06615566, 0.89464821, 0.03919614],
                     [0.54863966, 0.25128039, 0.20007995]])
    h = bhmm_amd.init_discrete_hmm([_SPLIT], 3, separate=[0])
    np.testing.assert_allclose(h.initial_distribution, piref, atol=1e-6)
    np.testing.assert_allclose(h.transition_matrix, Aref, atol=1e-6)
    Bref = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]])
    assert np.max(np.abs(h.output_model.output_probabilities - Bref)) < 0.01
    h = bhmm_amd.init_discrete_hmm([_SPLIT + 2], 3, separate=[1, 2])
    np.testing.assert_allclose(h.initial_distribution, piref, atol=1e-6)
    np.testing.assert_allclose(h.transition_matrix, Aref, atol=1e-6)
    Bref = np.array([[0, 0, 0, 1, 0], [0, 0, 0, 0, 1], [0, 0, 1, 0, 0]])
    assert np.max(np.abs(h.output_model.output_probabilities - Bref)) < 0.01
    with pytest.raises(ValueError):                          # test_init_discrete.py:215-218
        bhmm_amd.init_discrete_hmm([np.array([0, 0, 1, 1])], 2, separate=[0, 2])


def test_discrete_initial_model_metastable_chains():
    """test_init_discrete.py:33-109: metastable chains coarse-grain to their blocks (PCCA+)."""
    from bhmm_amd.init.discrete import pcca_memberships, count_matrix
    P2 = np.array([[0.99, 0.01], [0.01, 0.99]])
    h = bhmm_amd.init_discrete_hmm([_markov_chain(P2, 10000, 1)], 2)
    A, B = h.transition_matrix, h.output_model.output_probabilities
    if B[0, 0] < B[1, 0]:
        B = B[::-1]
    assert np.max(A - P2) < 0.01 and np.max(B - np.eye(2)) < 0.01
    P4 = np.array([[0.90, 0.10, 0.00, 0.00], [0.10, 0.89, 0.01, 0.00],
                   [0.00, 0.01, 0.89, 0.10], [0.00, 0.00, 0.10, 0.90]])
    h = bhmm_amd.init_discrete_hmm([_markov_chain(P4, 10000, 2)], 2)
    A, B = h.transition_matrix, h.output_model.output_probabilities
    Bref = np.array([[0.5, 0.5, 0.0, 0.0], [0.0, 0.0, 0.5, 0.5]])
    assert np.max(A - np.array([[0.99, 0.01], [0.01, 0.99]])) < 0.01
    assert np.max(B - Bref) < 0.05 or np.max(B[::-1] - Bref) < 0.05
    P6 = np.array([[0.90, 0.10, 0.00, 0.00, 0.00, 0.00], [0.20, 0.79, 0.01, 0.00, 0.00, 0.00],
                   [0.00, 0.01, 0.84, 0.15, 0.00, 0.00], [0.00, 0.00, 0.05, 0.94, 0.01, 0.00],
                   [0.00, 0.00, 0.00, 0.02, 0.78, 0.20], [0.00, 0.00, 0.00, 0.00, 0.10, 0.90]])
    d6 = _markov_chain(P6, 10000, 3)
    h = bhmm_amd.init_discrete_hmm([d6], 3)
    assert _tmatrix.is_transition_matrix(h.transition_matrix) and h.is_reversible
    np.testing.assert_allclose(h.output_model.output_probabilities.sum(axis=1), 1.0)
    # memberships: a partition of unity whose crisp assignment is the three blocks
    chi = pcca_memberships(_tmatrix.mle_reversible(count_matrix([d6], 1)), 3)
    np.testing.assert_allclose(chi.sum(axis=1), 1.0, atol=1e-12)
    assert chi.min() >= 0 and chi.max(axis=1).min() > 0.9
    crisp = chi.argmax(axis=1)
    assert crisp[0] == crisp[1] and crisp[2] == crisp[3] and crisp[4] == crisp[5]
    assert len(set(crisp)) == 3
    # disconnected matrix: one metastable state per closed set, transient states by absorption
    Pd = np.array([[0.9, 0.1, 0.0, 0.0, 0.0], [0.1, 0.9, 0.0, 0.0, 0.0],
                   [0.0, 0.0, 0.8, 0.2, 0.0], [0.0, 0.0, 0.2, 0.8, 0.0],
                   [0.0, 0.3, 0.3, 0.0, 0.4]])
    chi = pcca_memberships(Pd, 2)
    np.testing.assert_allclose(chi[:4], [[1, 0], [1, 0], [0, 1], [0, 1]])
    np.testing.assert_allclose(chi[4], [0.5, 0.5])
    with pytest.raises(ValueError):
        pcca_memberships(Pd, 1)


def test_discrete_initial_model_pathological():
    """test_init_discrete.py:115-180."""
    from bhmm_amd.init.discrete import init_discrete_hmm_spectral, count_matrix
    for rev in (True, False):
        h = bhmm_amd.init_discrete_hmm([np.array([0, 0, 0, 0, 0])], 1, reversible=rev)
        assert np.allclose(h.transition_matrix, [[1.0]])
        assert np.allclose(h.output_model.output_probabilities, [[1.0]])
        h = bhmm_amd.init_discrete_hmm([np.array([0, 0, 0, 0, 1])], 1, reversible=rev)
        B = h.output_model.output_probabilities
        assert np.allclose(h.transition_matrix, [[1.0]]) and B.shape == (1, 2) and np.all(B > 0)
        C = count_matrix([np.array([0, 0, 1, 1, 0])], 1)
        p0, A0, B0 = init_discrete_hmm_spectral(C, 1, reversible=rev,
                                                P=_tmatrix.estimate_P(C, reversible=rev))
        assert np.allclose(A0, [[1.0]]) and B0.shape == (1, 2) and np.all(B0 > 0)
        h = bhmm_amd.init_discrete_hmm([np.array([0, 0, 0, 0, 1])], 2, reversible=rev,
                                       method='spectral', regularize=False)
        assert np.allclose(h.transition_matrix, [[0.75, 0.25], [0, 1]])
        assert np.allclose(h.output_model.output_probabilities, np.eye(2))
        h = bhmm_amd.init_discrete_hmm([np.array([0, 1, 2, 0, 3, 4])], 3, reversible=rev)
        assert _tmatrix.is_transition_matrix(h.transition_matrix)
        assert (not rev) or h.is_reversible
        assert np.allclose(h.output_model.output_probabilities.sum(axis=1), 1)
    with pytest.raises(NotImplementedError):
        bhmm_amd.init_discrete_hmm([np.array([0, 1, 0, 0, 1, 1])], 3, reversible=False)
    h = bhmm_amd.init_hmm([_SPLIT], 2, lag=2)
    assert h.lag == 2 and h.output_model.model_type == 'discrete'
    # two disconnected symbol sets: ValueError (test_mlhmm.py:127-140)
    rng = np.random.RandomState(1)
    with pytest.raises(ValueError):
        bhmm_amd.init_discrete_hmm([rng.randint(0, 5, 100), rng.randint(6, 11, 100)], 2, lag=5)


def test_spectral_properties_and_sampled_statistics():
    """generic_hmm.py:203-296 and generic_sampled_hmm.py / util/statistics.py."""
    from bhmm_amd.util.statistics import confidence_interval, confidence_interval_arr
    P = np.array([[0.9, 0.1, 0.0], [0.1, 0.8, 0.1], [0.0, 0.1, 0.9]])
    h = bhmm_amd.gaussian_hmm([1 / 3.] * 3, P, [-1.0, 0.0, 1.0], [1.0, 1.0, 1.0])
    h._lag = 2
    lam = np.sort(np.linalg.eigvals(P).real)[::-1]
    np.testing.assert_allclose(h.eigenvalues, lam, atol=1e-12)
    R, L = h.eigenvectors_right, h.eigenvectors_left
    np.testing.assert_allclose(L @ R, np.eye(3), atol=1e-12)
    np.testing.assert_allclose(R @ np.diag(h.eigenvalues) @ L, P, atol=1e-12)
    np.testing.assert_allclose(L[0], h.stationary_distribution, atol=1e-12)
    np.testing.assert_allclose(R[:, 0], 1.0, atol=1e-12)
    np.testing.assert_allclose(h.timescales, -2.0 / np.log(lam[1:]))
    np.testing.assert_allclose(h.lifetimes, -2.0 / np.log(np.diag(P)))
    assert h._spectral_decomp_available and hasattr(h, '_ensure_spectral_decomposition')
    sub = h.sub_hmm([0, 1])
    np.testing.assert_allclose(sub.transition_matrix, [[0.9, 0.1], [1 / 9., 8 / 9.]])
    assert np.array_equal(sub.output_model.means, [-1.0, 0.0])
    # non-reversible model: complex decomposition, still L R = 1
    Pn = np.array([[0.8, 0.2, 0.0], [0.0, 0.7, 0.3], [0.4, 0.0, 0.6]])
    hn = bhmm_amd.discrete_hmm([1, 0, 0], Pn, np.eye(3))
    assert not hn.is_reversible
    np.testing.assert_allclose(hn.eigenvectors_left @ hn.eigenvectors_right, np.eye(3), atol=1e-12)
    assert abs(hn.eigenvalues[0] - 1) < 1e-12
    # confidence interval: the reference's interpolation on a known sample
    data = np.arange(101, dtype=float)
    m, lo, hi = confidence_interval(data, 0.9)
    assert m == 50.0 and abs(lo - 5.0) < 1e-12 and abs(hi - 95.9) < 1e-12
    with pytest.raises(ValueError):
        confidence_interval(data, 1.5)
    rng = np.random.RandomState(0)
    arr = rng.normal(size=(2000, 2, 3))
    lo, hi = confidence_interval_arr(arr, conf=0.95)
    assert lo.shape == (2, 3) and np.all(np.abs(lo + 1.96) < 0.2) and np.all(np.abs(hi - 1.96) < 0.2)
    # SampledHMM over perturbed copies
    samples = []
    for k in range(50):
        Pk = P + 0.01 * rng.random_sample((3, 3))
        Pk /= Pk.sum(axis=1)[:, None]
        samples.append(bhmm_amd.gaussian_hmm([1 / 3.] * 3, Pk, np.array([-1.0, 0.0, 1.0]) + 0.1 * rng.normal(size=3),
                                             [1.0, 1.0, 1.0]))
    sh = bhmm_amd.SampledHMM(h, samples, conf=0.9)
    assert sh.nsamples == 50 and len(sh) == 50 and sh[3] is samples[3]
    assert sh.transition_matrix_samples.shape == (50, 3, 3)
    np.testing.assert_allclose(sh.transition_matrix_mean,
                               np.mean([x.transition_matrix for x in samples], axis=0))
    np.testing.assert_allclose(sh.transition_matrix_std,
                               np.std([x.transition_matrix for x in samples], axis=0))
    lo, hi = sh.transition_matrix_conf
    assert np.all(lo <= sh.transition_matrix_mean + 1e-15) and np.all(hi >= sh.transition_matrix_mean - 1e-15)
    assert sh.eigenvalues_mean.shape == (3,) and sh.timescales_samples.shape == (50, 2)
    assert sh.lifetimes_std.shape == (3,) and sh.eigenvectors_left_mean.shape == (3, 3)
    assert sh.stationary_distribution_mean.shape == (3,)
    assert sh.means_samples.shape == (50, 3) and sh.sigmas_std.shape == (3,)
    lo, hi = sh.means_conf
    assert np.all(lo < hi)
    with pytest.raises(AttributeError):
        sh.output_probabilities_mean
    np.testing.assert_allclose(sh.transition_matrix, P)       # the estimated model itself
    # typed views (gaussian_hmm.py / discrete_hmm.py)
    gh = bhmm_amd.GaussianHMM(h)
    assert np.array_equal(gh.means, h.output_model.means) and gh.nstates == 3 and gh.lag == 2
    with pytest.raises(TypeError):
        bhmm_amd.DiscreteHMM(h)
    dh = bhmm_amd.DiscreteHMM(hn)
    assert dh.nsymbols == 3 and np.array_equal(dh.output_probabilities, np.eye(3))
    sg = bhmm_amd.SampledGaussianHMM(h, samples)
    assert sg.means_mean.shape == (3,)
    with pytest.raises(TypeError):
        bhmm_amd.SampledDiscreteHMM(h, samples)


def test_output_model_sampling_with_reference_signatures():
    """gaussian.py:274-382, discrete.py:217-318: sample(observations per state) and the per-state
    generators; sample() must equal sample_from_statistics on the same statistics and stream."""
    om = bhmm_amd.GaussianOutputModel(3, means=[-1.0, 0.0, 1.0], sigmas=[0.5, 1.0, 2.0])
    rs = np.random.RandomState(0)
    obs = [om.generate_observations_from_state(i, 5000, rng=rs) for i in range(3)]
    assert all(o.shape == (5000,) for o in obs)
    om2 = bhmm_amd.GaussianOutputModel(3, means=[-1.0, 0.0, 1.0], sigmas=[0.5, 1.0, 2.0])
    om.sample(obs, rng=np.random.RandomState(7))
    d = [o - m for o, m in zip(obs, [-1.0, 0.0, 1.0])]
    om2.sample_from_statistics([5000] * 3, [x.sum() for x in d], [np.dot(x, x) for x in d],
                               rng=np.random.RandomState(7))
    np.testing.assert_allclose(om.means, om2.means, rtol=1e-12)
    np.testing.assert_allclose(om.sigmas, om2.sigmas, rtol=1e-12)
    np.testing.assert_allclose(om.means, [-1.0, 0.0, 1.0], atol=0.1)
    np.testing.assert_allclose(om.sigmas, [0.5, 1.0, 2.0], rtol=0.05)
    assert np.isfinite(om.generate_observation_from_state(2, rng=rs))
    dm = bhmm_amd.DiscreteOutputModel(np.array([[0.5, 0.5, 0.0], [0.1, 0.2, 0.7]]))
    o = [dm.generate_observations_from_state(i, 4000, rng=rs) for i in range(2)]
    assert o[0].max() <= 1 and o[1].dtype == np.int32
    dm.sample(o, rng=rs)
    B = dm.output_probabilities
    np.testing.assert_allclose(B.sum(axis=1), 1.0)
    assert B[0, 2] == 0.0 and abs(B[1, 2] - 0.7) < 0.05      # unseen symbols keep probability 0
    assert dm.generate_observation_from_state(0, rng=rs) in (0, 1)


def test_testsystems_and_synthetic_trajectories():
    """testsystems.py:26-250, generic_hmm.py:433-589."""
    rs = np.random.RandomState(3)
    T = bhmm_amd.testsystems.generate_transition_matrix(5, rng=rs)
    assert _tmatrix.is_transition_matrix(T) and _tmatrix.is_reversible(T)
    lt = np.linspace(np.log(10), np.log(100), 5)
    np.testing.assert_allclose(np.diag(T), 1 - np.exp(-lt), rtol=1e-12)     # lifetimes 10 .. 100
    model, O, S = bhmm_amd.testsystems.generate_synthetic_observations(
        nstates=3, ntrajectories=4, length=3000, rng=rs)
    assert model.is_stationary and len(O) == 4 and O[0].shape == (3000,) and S[0].dtype == np.int32
    assert bhmm_amd.testsystems.total_state_visits(3, S).sum() == 12000
    emp = np.array([O[0][S[0] == i].mean() for i in range(3) if np.any(S[0] == i)])
    assert np.all(np.diff(emp) > 0)                           # means -5, 0, 5 in state order
    md, Od, Sd = bhmm_amd.testsystems.generate_synthetic_observations(
        nstates=3, ntrajectories=2, length=500, output='discrete', rng=rs)
    assert md.output_model.model_type == 'discrete' and Od[0].max() <= 2
    s = model.generate_synthetic_state_trajectory(50, start=1, stop=2, rng=rs)
    assert s[0] == 1 and (s[-1] == 2 or len(s) == 50) and 2 not in s[:-1]
    with pytest.raises(ValueError):
        model.generate_synthetic_state_trajectory(5, initial_Pi=[1, 0, 0], start=0)


def test_gaussian_mixture_start_finds_rare_distant_state():
    """A state visited 6 % of the time, far from the bulk, is missed by a quantile start (EM then
    converges to a poor optimum and so does the whole estimation); the multi-start fit finds it."""
    from bhmm_amd.init.gaussian import fit_gmm1d
    rs = np.random.RandomState(0)
    model, O, S = bhmm_amd.testsystems.generate_synthetic_observations(
        nstates=3, ntrajectories=8, length=20000, rng=rs)
    w, m, sg = fit_gmm1d(np.concatenate(O), 3)
    np.testing.assert_allclose(m, [-5.0, 0.0, 5.0], atol=0.15)
    np.testing.assert_allclose(sg, [0.5, 1.25, 2.0], rtol=0.1)
    frac = np.bincount(np.concatenate(S), minlength=3) / (8 * 20000.0)
    np.testing.assert_allclose(w, frac, atol=0.02)
    w2, m2, sg2 = fit_gmm1d(np.concatenate(O), 3)
    assert np.array_equal(m, m2) and np.array_equal(sg, sg2)          # reproducible


def test_multi_start_recovers_overlapping_states():
    """estimate_hmm from raw data, 4 overlapping Gaussian states (the reference's dalton model):
    the mixture start alone ends in a poor optimum (the marginal distribution hardly identifies the
    states), the kinetic start (bins -> Markov model -> PCCA+) finds the generating model; the
    estimator tries both for a few iterations and continues with the better one."""
    from bhmm_amd.init.gaussian import init_model_gaussian1d_kinetic
    rs = np.random.RandomState(2)
    model, O, S = bhmm_amd.testsystems.generate_synthetic_observations(
        nstates=4, ntrajectories=4, length=20000, rng=rs)
    alt = init_model_gaussian1d_kinetic(O, 4)
    assert alt is not None and alt.nstates == 4 and _tmatrix.is_reversible(alt.transition_matrix)
    assert np.all(np.diff(alt.output_model.means) > 0)
    assert init_model_gaussian1d_kinetic([np.zeros(50)], 2) is None          # does not apply
    est = bhmm_amd.MaximumLikelihoodEstimator(O, 4, engine_factory=OracleEngine, maxit=100)
    h = est.fit()
    np.testing.assert_allclose(h.output_model.means, model.output_model.means, atol=0.1)
    np.testing.assert_allclose(h.output_model.sigmas, model.output_model.sigmas, atol=0.1)
    np.testing.assert_allclose(h.transition_matrix, model.transition_matrix, atol=0.03)


def test_native_reversible_mle_equals_numpy_restatement():
    """bhmm_mle_reversible (host code of the library, no GPU) runs the same fixed-point iteration
    as the numpy restatement _tmatrix.mle_reversible: same matrices to rounding, detailed balance,
    structural zeros kept, and the two-state closed form."""
    from bhmm_amd.estimators import _tmatrix
    rng = np.random.default_rng(4)
    for n in (2, 3, 8, 20, 64):
        C = rng.random((n, n)) * rng.integers(1, 1000, (n, n))
        C[rng.random((n, n)) < 0.2] = 0.0
        C += np.diag(rng.random(n) + 0.1)
        C[0, 1] += 1.0                                   # keep it connected
        C[np.arange(1, n), np.arange(n - 1)] += 0.5
        C[np.arange(n - 1), np.arange(1, n)] += 0.5
        P1 = _tmatrix.mle_reversible(C, maxiter=200000, maxerr=1e-13)
        P0 = _tmatrix.mle_reversible(C, maxiter=200000, maxerr=1e-13, native=False)
        np.testing.assert_allclose(P1, P0, rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(P1.sum(axis=1), 1.0, rtol=1e-13)
        pi = _tmatrix.stationary_vector(P1)
        F = pi[:, None] * P1
        np.testing.assert_allclose(F, F.T, atol=1e-10)   # detailed balance
        assert np.all(P1[(C + C.T) == 0] == 0.0)
    # every two-state chain is reversible: the reversible MLE is the plain row-normalised one
    C = np.array([[4.0, 1.0], [2.0, 4.0]])
    P = _tmatrix.mle_reversible(C, maxerr=1e-15)
    np.testing.assert_allclose(P, [[0.8, 0.2], [1.0 / 3.0, 2.0 / 3.0]], atol=1e-9)


def test_lagged_list_that_was_edited_is_not_taken_for_its_recipe():
    """A LaggedObservations list is mutable; the device-side fast path (upload the originals, cut
    the views on the GPU) may only be taken while every entry still IS the recorded view
    (ADVICE round 2)."""
    from bhmm_amd.estimators.maximum_likelihood import _views_intact, _load_observations
    base = [np.arange(20.0), np.arange(7.0)]
    lagged = bhmm_amd.lag_observations(base, 3)
    assert _views_intact(lagged)
    edited = bhmm_amd.lag_observations(base, 3)
    edited[1] = edited[1].copy()                    # same values, other memory
    assert not _views_intact(edited)
    swapped = bhmm_amd.lag_observations(base, 3)
    swapped[0], swapped[1] = swapped[1], swapped[0]
    assert not _views_intact(swapped)
    cut = bhmm_amd.lag_observations(base, 3)
    cut[2] = cut[2][:-1]
    assert not _views_intact(cut)

    class Rec(object):
        def __init__(self):
            self.calls = []

        def set_observations_lagged(self, *a, **k):
            self.calls.append('lagged')

        def set_observations(self, *a, **k):
            self.calls.append('plain')

    class NoComm(object):
        active = False

    for given, want in ((lagged, 'lagged'), (edited, 'plain'), (swapped, 'plain')):
        rec = Rec()
        _load_observations(rec, 'gaussian', given, list(given), range(len(given)), NoComm(), 2, 0)
        assert rec.calls == [want]
