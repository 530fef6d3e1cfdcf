"""Functional wrappers with the signatures of bhmm/api.py:309-372 (estimate_hmm) and
:375-470 (bayesian_hmm), plus lag_observations (:70-94) and the model factories (:97-158)."""
import numpy as np

from .estimators.bayesian_sampling import BayesianHMMSampler
from .estimators.maximum_likelihood import MaximumLikelihoodEstimator
from .hmm import HMM, SampledHMM
from .output_models import DiscreteOutputModel, GaussianOutputModel


def _guess_output_type(observations):
    """bhmm/api.py:36-66."""
    o1 = np.asarray(observations[0])
    if np.issubdtype(o1.dtype, np.integer) and o1.ndim == 1:
        return 'discrete'
    if np.issubdtype(o1.dtype, np.floating) and o1.ndim == 1:
        return 'gaussian'
    raise TypeError('Observations is neither sequences of integers nor 1D-sequences of floats.')


class LaggedObservations(list):
    """What lag_observations returns: a plain list of sub-sampled trajectories (numpy views, so
    existing code keeps working) that also remembers how it was made -- the original
    trajectories, the lag, and (trajectory, shift) of every entry -- so that the estimators can
    upload the ORIGINAL data once and cut the views on the GPU
    (bhmm_ctx_set_observations_lagged) instead of uploading `lag` host-side copies."""

    def __init__(self, base, lag, stride):
        list.__init__(self)
        self.base = [np.asarray(o) for o in base]
        self.lag = int(lag)
        self.stride = int(stride)
        self.views = []          # (trajectory index, shift) of every entry


def lag_observations(observations, lag, stride=1):
    """Sub-sampled, shifted trajectories (bhmm/api.py:70-94): trajectory k contributes
    obs_k[shift::lag] for shift = 0, stride, 2 stride, ... < lag; pieces with fewer than two
    observations are left out (they carry no transition)."""
    out = LaggedObservations(observations, lag, stride)
    for k, traj in enumerate(out.base):
        for shift in range(0, out.lag, out.stride):
            piece = traj[shift::out.lag]
            if piece.shape[0] > 1:
                out.append(piece)
                out.views.append((k, shift))
    return out


def gaussian_hmm(pi, P, means, sigmas):
    """bhmm/api.py:97-126."""
    nstates = len(means)
    return HMM(pi, P, GaussianOutputModel(nstates, means, sigmas))


def discrete_hmm(pi, P, pout):
    """bhmm/api.py:129-158."""
    return HMM(pi, P, DiscreteOutputModel(pout))


def init_gaussian_hmm(observations, nstates, lag=1, reversible=True):
    """bhmm/api.py:202-228."""
    from .init.gaussian import init_model_gaussian1d
    if lag > 1:
        observations = lag_observations(observations, lag)
    hmm0 = init_model_gaussian1d(observations, nstates, reversible=reversible)
    hmm0._lag = lag
    return hmm0


def init_discrete_hmm(observations, nstates, lag=1, reversible=True, stationary=True,
                      regularize=True, method='connect-spectral', separate=None):
    """bhmm/api.py:231-306."""
    from .init import discrete as _init
    p0, P, B = _init.init_discrete_hmm(observations, nstates, lag=lag, reversible=reversible,
                                       stationary=stationary, regularize=regularize,
                                       method=method, separate=separate)
    hmm0 = discrete_hmm(p0, P, B)
    hmm0._lag = lag
    return hmm0


def init_hmm(observations, nstates, lag=1, output=None, reversible=True):
    """bhmm/api.py:161-199."""
    if output is None:
        output = _guess_output_type(observations)
    if output == 'discrete':
        return init_discrete_hmm(observations, nstates, lag=lag, reversible=reversible)
    if output == 'gaussian':
        return init_gaussian_hmm(observations, nstates, lag=lag, reversible=reversible)
    raise NotImplementedError('output model type ' + str(output) + ' not yet implemented.')


def estimate_hmm(observations, nstates, lag=1, initial_model=None, output=None, reversible=True,
                 stationary=False, p=None, accuracy=1e-3, maxit=1000, maxit_P=100000,
                 mincount_connectivity=1e-2, **engine_kwargs):
    """bhmm/api.py:309-372."""
    if output is None:
        output = _guess_output_type(observations)
    if lag > 1:
        observations = lag_observations(observations, lag)
    est = MaximumLikelihoodEstimator(observations, nstates, initial_model=initial_model,
                                     output=output, reversible=reversible, stationary=stationary,
                                     p=p, accuracy=accuracy, maxit=maxit, maxit_P=maxit_P,
                                     **engine_kwargs)
    est.fit()
    est.hmm._lag = lag
    return est.hmm


def bayesian_hmm(observations, estimated_hmm, nsample=100, reversible=True, stationary=False,
                 p0_prior='mixed', transition_matrix_prior='mixed', store_hidden=False,
                 call_back=None, **engine_kwargs):
    """bhmm/api.py:375-470.  Returns a SampledHMM (which also iterates over / indexes the
    sampled models)."""
    sampler = BayesianHMMSampler(observations, estimated_hmm.nstates, initial_model=estimated_hmm,
                                 reversible=reversible, stationary=stationary,
                                 transition_matrix_sampling_steps=1000, p0_prior=p0_prior,
                                 transition_matrix_prior=transition_matrix_prior,
                                 output=estimated_hmm.output_model.model_type, **engine_kwargs)
    sampled = sampler.sample(nsamples=nsample, save_hidden_state_trajectory=store_hidden,
                             call_back=call_back)
    return SampledHMM(estimated_hmm, sampled)
