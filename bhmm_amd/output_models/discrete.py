"""Discrete emission model -- hot-path subset of bhmm/output_models/discrete.py."""
import numpy as np

from .. import _lib
from .outputmodel import OutputModel


class DiscreteOutputModel(OutputModel):
    def __init__(self, B, prior=None, ignore_outliers=False):
        self._output_probabilities = np.array(B, dtype=np.float64)
        nstates, self._nsymbols = self._output_probabilities.shape[0], self._output_probabilities.shape[1]
        if not np.allclose(np.sum(self._output_probabilities, axis=1), 1):
            raise ValueError('B is no stochastic matrix')          # discrete.py:73-75
        if prior is None:
            prior = np.zeros((nstates, self._nsymbols))
        self.prior = np.broadcast_to(np.asarray(prior, dtype=np.float64),
                                     (nstates, self._nsymbols)).copy()
        OutputModel.__init__(self, nstates, ignore_outliers=ignore_outliers)

    def __repr__(self):
        return 'DiscreteOutputModel(%r)' % (self._output_probabilities,)

    @property
    def model_type(self):
        return 'discrete'

    @property
    def output_probabilities(self):
        return self._output_probabilities

    @property
    def nsymbols(self):
        return self._nsymbols

    def parameters(self):
        return self._output_probabilities, None

    def set_parameters(self, B, _unused=None):
        """Adopt parameters drawn elsewhere (Gibbs: rank 0 draws, the others receive)."""
        self._output_probabilities = np.array(B, dtype=np.float64).reshape(
            self._output_probabilities.shape)

    def sub_output_model(self, states):
        return DiscreteOutputModel(self._output_probabilities[states])

    def p_obs(self, obs, out=None):
        """discrete.py:130-157: column gather  pobs[t,:] = B[:, obs[t]]  (numpy in the
        reference as well; inside the batched E-step the gather is fused into the kernels)."""
        obs = np.asarray(obs)
        if out is None:
            out = self._output_probabilities[:, obs].T
            return self._handle_outliers(out)
        if obs.shape[0] == out.shape[0]:
            np.copyto(out, self._output_probabilities[:, obs].T)
        elif obs.shape[0] < out.shape[0]:
            out[:obs.shape[0], :] = self._output_probabilities[:, obs].T
        else:
            raise ValueError('output array out is too small: ' + str(out.shape[0]) + ' < '
                             + str(obs.shape[0]))
        return self._handle_outliers(out)

    def estimate_from_statistics(self, symbol_counts):
        """discrete.py:202-215: row-normalise the weighted symbol counts."""
        B = np.array(symbol_counts, dtype=np.float64)
        self._output_probabilities = B / np.sum(B, axis=1)[:, None]

    def estimate(self, observations, weights):
        """discrete.py:159-215; the scatter-add is bhmm_update_pout (_discrete.c:1-32)."""
        L = _lib.load()
        _lib.require_device()
        N, M = self._output_probabilities.shape
        B = np.zeros((N, M))
        for o, w in zip(observations, weights):
            o = np.ascontiguousarray(o, dtype=np.int32)
            w = _lib.f64(w)
            _lib.check(L.bhmm_update_pout(_lib.dp(B), _lib.ip(o), _lib.dp(w), o.shape[0], N, M))
        self._output_probabilities = B / np.sum(B, axis=1)[:, None]

    def sample_from_statistics(self, symbol_counts, rng=np.random):
        """Gibbs update of discrete.py:217-251 from per-state symbol counts."""
        for i in range(self.nstates):
            count = np.asarray(symbol_counts[i], dtype=np.float64) + self.prior[i]
            positive = count > 0
            if np.any(positive):
                self._output_probabilities[i, positive] = rng.dirichlet(count[positive])

    def sample(self, observations_by_state, rng=np.random):
        """discrete.py:217-251 with the reference's signature: observations_by_state[k] are the
        symbols observed while the hidden path was in state k."""
        counts = [np.bincount(np.asarray(o, dtype=np.int64), minlength=self._nsymbols)[:self._nsymbols]
                  for o in observations_by_state]
        self.sample_from_statistics(counts, rng=rng)

    def generate_observation_from_state(self, state_index, rng=np.random):
        """discrete.py:253-282."""
        return int(rng.choice(self._nsymbols, p=self._output_probabilities[state_index]))

    def generate_observations_from_state(self, state_index, nobs, rng=np.random):
        """discrete.py:284-318."""
        return rng.choice(self._nsymbols, size=nobs,
                          p=self._output_probabilities[state_index]).astype(np.int32)

    def generate_observation_trajectory(self, s_t, rng=np.random):
        s_t = np.asarray(s_t)
        cdf = np.cumsum(self._output_probabilities, axis=1)
        u = rng.random_sample(s_t.shape[0])
        o = (u[:, None] > cdf[s_t]).sum(axis=1)
        return np.minimum(o, self._nsymbols - 1).astype(np.int32)
