"""End-to-end estimators on the HIP engine: the same EM run as the oracle-backed CPU double
must give the same likelihood history, parameters and Viterbi paths."""
import numpy as np
import pytest

import bhmm_amd
from oracle_engine import OracleEngine
from test_host_logic import _gauss_problem

pytestmark = pytest.mark.gpu


def test_mle_fit_matches_oracle_driven_fit():
    obs, init = _gauss_problem(seed=1, K=6, T=700)
    ref = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=True,
                                              accuracy=1e-5, maxit=25, engine_factory=OracleEngine)
    href = ref.fit()
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, 3, initial_model=init, reversible=True,
                                              accuracy=1e-5, maxit=25, store_gamma=True)
    hmm = est.fit()
    assert len(est.likelihoods) == len(ref.likelihoods)
    np.testing.assert_allclose(est.likelihoods, ref.likelihoods, rtol=1e-10)
    np.testing.assert_allclose(hmm.transition_matrix, href.transition_matrix, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(hmm.output_model.means, href.output_model.means, rtol=1e-8)
    np.testing.assert_allclose(hmm.output_model.sigmas, href.output_model.sigmas, rtol=1e-7)
    for a, b in zip(hmm.hidden_state_trajectories, href.hidden_state_trajectories):
        assert np.array_equal(a, b)
    g = est.hidden_state_probabilities
    assert np.allclose(g[0].sum(axis=1), 1.0) and g[0].shape == (len(obs[0]), 3)


def test_discrete_patho_on_gpu():
    obs = np.array([0, 0, 0, 0, 0, 1, 1, 1, 1], dtype=int)
    init = bhmm_amd.discrete_hmm([0.5, 0.5], [[0.7, 0.3], [0.2, 0.8]], [[0.8, 0.2], [0.3, 0.7]])
    hmm = bhmm_amd.estimate_hmm([obs], nstates=2, accuracy=1e-6, initial_model=init)
    assert np.allclose(hmm.transition_matrix, [[0.8, 0.2], [0.0, 1.0]], atol=1e-5)
    assert np.allclose(hmm.output_model.output_probabilities, np.eye(2), atol=1e-5)


def test_bayesian_sampler_on_gpu():
    obs, init = _gauss_problem(seed=2, K=4, T=500)
    mle = bhmm_amd.estimate_hmm(obs, 3, initial_model=init, reversible=False, maxit=15)
    np.random.seed(1)
    models = bhmm_amd.bayesian_hmm(obs, mle, nsample=10, reversible=False, store_hidden=True)
    means = np.array([m.output_model.means for m in models])
    assert means.std(axis=0).min() > 0
    assert np.allclose(means.mean(axis=0), mle.output_model.means, atol=0.3)
    assert all(p.shape == (len(o),) for p, o in zip(models[-1].hidden_state_trajectories, obs))


def test_output_models_on_gpu(golden):
    g = golden("pobs_gauss3")
    om = bhmm_amd.GaussianOutputModel(3, means=g["mu"], sigmas=g["sigma"])
    np.testing.assert_allclose(om.p_obs(g["obs"]), g["pobs"], rtol=1e-13)
    out = np.zeros((len(g["obs"]) + 3, 3))
    om.p_obs(g["obs"], out=out)
    np.testing.assert_allclose(out[:-3], g["pobs"], rtol=1e-13)
    assert np.all(out[-3:] == 1.0)      # stale zero rows are "outliers" too (gaussian.py:194-195)
    gd = golden("d8_ragged")
    from conftest import split
    obs = split(gd["obs"].astype(np.int32), gd["lengths"])
    dm = bhmm_amd.DiscreteOutputModel(gd["B"])
    from oracle import oracle as orc
    r = orc.estep("discrete", obs, gd["A"], gd["pi"], gd["B"], want_gamma=True)
    dm.estimate(obs, r["gammas"])
    np.testing.assert_allclose(dm.output_probabilities, gd["B_new"], rtol=1e-10, atol=1e-15)


def test_estimate_hmm_from_raw_gaussian_data():
    """estimate_hmm(observations, nstates) without an initial model (bhmm/api.py:309-372 with the
    initialiser of bhmm/init/gaussian.py): the heuristic start + Baum-Welch on the GPU recover
    the generating model."""
    import bhmm_amd
    rng = np.random.default_rng(9)
    mu, sg = np.array([-2.0, 1.0, 5.0]), np.array([0.6, 0.8, 0.7])
    P = np.array([[0.95, 0.05, 0.0], [0.03, 0.94, 0.03], [0.0, 0.06, 0.94]])
    obs = []
    for T in (6000, 5000, 3000):
        s = np.zeros(T, dtype=int)
        for t in range(1, T):
            s[t] = rng.choice(3, p=P[s[t - 1]])
        obs.append(mu[s] + sg[s] * rng.standard_normal(T))
    hmm = bhmm_amd.estimate_hmm(obs, 3, reversible=False, accuracy=1e-4, maxit=200)
    order = np.argsort(hmm.output_model.means)
    np.testing.assert_allclose(hmm.output_model.means[order], mu, atol=0.1)
    np.testing.assert_allclose(hmm.output_model.sigmas[order], sg, atol=0.1)
    np.testing.assert_allclose(hmm.transition_matrix[np.ix_(order, order)], P, atol=0.03)


def test_estimate_hmm_from_raw_discrete_data():
    """estimate_hmm(dtrajs, nstates) without an initial model: count matrix + PCCA+ start
    (bhmm/init/discrete.py) and Baum-Welch on the GPU recover a hidden 2-state chain seen
    through 6 symbols; then bayesian_hmm gives a SampledHMM whose statistics bracket it."""
    from test_host_logic import _markov_chain
    P = np.array([[0.97, 0.03], [0.05, 0.95]])
    B = np.array([[0.5, 0.3, 0.15, 0.05, 0.0, 0.0], [0.0, 0.0, 0.05, 0.15, 0.3, 0.5]])
    rng = np.random.RandomState(4)
    obs = []
    for k, T in enumerate((20000, 15000)):
        s = _markov_chain(P, T, 10 + k)
        cs = np.cumsum(B, axis=1)
        obs.append(np.minimum((rng.random_sample(T)[:, None] > cs[s]).sum(axis=1), 5).astype(np.int32))
    hmm = bhmm_amd.estimate_hmm(obs, 2, accuracy=1e-4, maxit=200)
    A, Bh = hmm.transition_matrix, hmm.output_model.output_probabilities
    if Bh[0, 0] < Bh[1, 0]:
        A, Bh = A[::-1, ::-1], Bh[::-1]
    np.testing.assert_allclose(A, P, atol=0.01)
    np.testing.assert_allclose(Bh, B, atol=0.02)
    assert hmm.is_reversible
    np.random.seed(3)
    sampled = bhmm_amd.bayesian_hmm(obs, hmm, nsample=20, reversible=True)
    assert isinstance(sampled, bhmm_amd.SampledHMM) and sampled.nsamples == 20
    lo, hi = sampled.transition_matrix_conf
    assert np.all(lo <= hmm.transition_matrix + 0.02) and np.all(hi >= hmm.transition_matrix - 0.02)
    assert sampled.output_probabilities_mean.shape == (2, 6)
    assert np.all(sampled.timescales_mean > 5)


def test_estimate_hmm_multi_start_five_overlapping_states():
    """estimate_hmm(observations, 5) on the reference's dalton test system (means -5 .. 5, sigmas
    0.5 .. 2, lifetimes 10 .. 100): with the opt-in multi_start=True the estimator tries the mixture
    start and the kinetic start for a few GPU iterations each and recovers the generating model
    (the default is the reference's single start, maximum_likelihood.py:111-116)."""
    rs = np.random.RandomState(3)
    model, O, S = bhmm_amd.testsystems.generate_synthetic_observations(
        nstates=5, ntrajectories=10, length=30000, rng=rs)
    hmm = bhmm_amd.estimate_hmm(O, 5, multi_start=True)
    np.testing.assert_allclose(hmm.output_model.means, model.output_model.means, atol=0.05)
    np.testing.assert_allclose(hmm.output_model.sigmas, model.output_model.sigmas, atol=0.05)
    np.testing.assert_allclose(hmm.transition_matrix, model.transition_matrix, atol=0.02)
    paths = hmm.hidden_state_trajectories
    assert np.mean(np.concatenate(paths) == np.concatenate(S)) > 0.9
