"""Synthetic test systems with the recipes of bhmm/util/testsystems.py:26-250 (the models the
reference's tests, examples and the benchmark of this repository are built on)."""
import math

import numpy as np


def generate_transition_matrix(nstates=3, lifetime_max=100, lifetime_min=10, reversible=True,
                               rng=np.random):
    """testsystems.py:26-65: random metastable transition matrix, lifetimes log-spaced."""
    lt = np.linspace(math.log(lifetime_min), math.log(lifetime_max), num=nstates)
    diag = 1.0 - 1.0 / np.exp(lt)
    X = rng.random_sample((nstates, nstates))
    if reversible:
        X = X + X.T
    T = X / np.sum(X, axis=1)[:, None]
    for i in range(nstates):
        T[i, i] = 0
        T[i, :] *= (1.0 - diag[i]) / np.sum(T[i, :])
        T[i, i] = 1.0 - np.sum(T[i, :])
    return T


def force_spectroscopy_model():
    """testsystems.py:68-102: the fixed three-state model of the reference's examples (a force
    spectroscopy experiment; the numbers are the model's definition)."""
    from ..estimators import _tmatrix
    from ..hmm import HMM
    from ..output_models import GaussianOutputModel
    output_model = GaussianOutputModel(3, means=[3.0, 4.7, 5.6], sigmas=[1.0, 0.3, 0.2])
    Tij = np.array([[0.98, 0.01540412, 0.00459588],
                    [0.06331175, 0.9, 0.03668825],
                    [0.00339873, 0.00660127, 0.99]])
    return HMM(_tmatrix.stationary_vector(Tij), Tij, output_model)


def dalton_model(nstates=3, omin=-5, omax=5, sigma_min=0.5, sigma_max=2.0, lifetime_max=100,
                 lifetime_min=10, reversible=True, output='gaussian', rng=np.random):
    """testsystems.py:105-188."""
    from ..estimators import _tmatrix
    from ..hmm import HMM
    from ..output_models import DiscreteOutputModel, GaussianOutputModel
    means = np.linspace(omin, omax, num=nstates)
    sigmas = np.linspace(sigma_min, sigma_max, num=nstates)
    if output == 'gaussian':
        output_model = GaussianOutputModel(nstates, means=means, sigmas=sigmas)
    elif output == 'discrete':
        B = np.exp(-0.5 * (means[:, None] - means[None, :]) / (sigmas[:, None] * sigmas[None, :]))
        output_model = DiscreteOutputModel(B / B.sum(axis=1)[:, None])
    else:
        raise Exception("output_model_type = '%s' unknown, must be one of ['gaussian', 'discrete']"
                        % output)
    Tij = generate_transition_matrix(nstates, lifetime_max=lifetime_max, lifetime_min=lifetime_min,
                                     reversible=reversible, rng=rng)
    return HMM(_tmatrix.stationary_vector(Tij), Tij, output_model)


def generate_synthetic_observations(nstates=3, ntrajectories=10, length=10000, omin=-5, omax=5,
                                    sigma_min=0.5, sigma_max=2.0, lifetime_max=100, lifetime_min=10,
                                    reversible=True, output='gaussian', rng=np.random):
    """testsystems.py:191-250: [model, observations, hidden states]."""
    model = dalton_model(nstates, omin=omin, omax=omax, sigma_min=sigma_min, sigma_max=sigma_max,
                         lifetime_max=lifetime_max, lifetime_min=lifetime_min, reversible=reversible,
                         output=output, rng=rng)
    O, S = model.generate_synthetic_observation_trajectories(ntrajectories=ntrajectories,
                                                             length=length, rng=rng)
    return [model, O, S]


def generate_random_bhmm(nstates=3, ntrajectories=10, length=10000, omin=-5, omax=5, sigma_min=0.5,
                         sigma_max=2.0, lifetime_max=100, lifetime_min=10, reversible=True,
                         output='gaussian', rng=np.random, **sampler_kwargs):
    """testsystems.py:253-316: (model, observations, hidden states, BayesianHMMSampler)."""
    from ..estimators.bayesian_sampling import BayesianHMMSampler
    model, O, S = generate_synthetic_observations(
        nstates=nstates, ntrajectories=ntrajectories, length=length, omin=omin, omax=omax,
        sigma_min=sigma_min, sigma_max=sigma_max, lifetime_max=lifetime_max,
        lifetime_min=lifetime_min, reversible=reversible, output=output, rng=rng)
    return model, O, S, BayesianHMMSampler(O, nstates, output=output, **sampler_kwargs)


def total_state_visits(nstates, S):
    """testsystems.py:319-330."""
    return np.sum([np.bincount(s, minlength=nstates) for s in S], axis=0)
