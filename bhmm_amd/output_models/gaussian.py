"""1-D Gaussian emission model -- hot-path subset of bhmm/output_models/gaussian.py.

`p_obs` runs the HIP kernel behind bhmm_pobs_gaussian (replacing gaussian.pyx:87-105 /
_gaussian.c:45-70).  Inside the batched E-step the pdf is fused into the recursions and this
method is not called at all; `estimate_from_statistics` is the M-step on the sufficient
statistics the E-step returns, `estimate` keeps the reference's (observations, weights)
signature.
"""
import numpy as np

from .. import _lib
from .outputmodel import OutputModel


class GaussianOutputModel(OutputModel):
    def __init__(self, nstates, means=None, sigmas=None, ignore_outliers=True):
        OutputModel.__init__(self, nstates, ignore_outliers=ignore_outliers)
        self._means = (np.array(means, dtype=np.float64) if means is not None
                       else np.zeros(nstates))
        self._sigmas = (np.array(sigmas, dtype=np.float64) if sigmas is not None
                        else np.zeros(nstates))
        if self._means.shape != (self.nstates,) or self._sigmas.shape != (self.nstates,):
            raise ValueError('means / sigmas must have shape (nstates,)')

    def __repr__(self):
        return 'GaussianOutputModel(%d, means=%r, sigmas=%r)' % (
            self.nstates, self._means, self._sigmas)

    @property
    def model_type(self):
        return 'gaussian'

    @property
    def dimension(self):
        return 1

    @property
    def means(self):
        return self._means

    @property
    def sigmas(self):
        return self._sigmas

    def parameters(self):
        """(par0, par1) as the C ABI wants them for BHMM_EMIT_GAUSSIAN."""
        return self._means, self._sigmas

    def set_parameters(self, means, sigmas):
        """Adopt parameters drawn elsewhere (Gibbs: rank 0 draws, the others receive)."""
        self._means = np.array(means, dtype=np.float64).reshape(self.nstates)
        self._sigmas = np.array(sigmas, dtype=np.float64).reshape(self.nstates)

    def sub_output_model(self, states):
        return GaussianOutputModel(len(states), self._means[states], self._sigmas[states])

    def p_obs(self, obs, out=None):
        """gaussian.py:170-212: (T, N) matrix of emission densities (+ outlier rule)."""
        L = _lib.load()
        _lib.require_device()
        obs = _lib.f64(obs)
        T, N = obs.shape[0], self.nstates
        if out is None:
            res = np.zeros((T, N))
        else:
            res = out
        direct = res.flags.c_contiguous and res.dtype == np.float64 and res.shape[0] == T
        buf = res if direct else np.empty((T, N))
        _lib.check(L.bhmm_pobs_gaussian(_lib.dp(buf), _lib.dp(obs), _lib.dp(_lib.f64(self._means)),
                                        _lib.dp(_lib.f64(self._sigmas)), N, T, 0))
        if not direct:
            res[:T] = buf
        # the reference scans the whole buffer, stale rows included (gaussian.py:194-195)
        return self._handle_outliers(res)

    def estimate_from_statistics(self, state_counts, sum_gd, sum_gdd):
        """M-step of gaussian.py:214-272 from  sum_t gamma,  sum_t gamma (o - mu_old),
        sum_t gamma (o - mu_old)^2.  Equals the reference's two-pass result: the new mean is
        mu_old + <d>, and the variance around the NEW mean is <d^2> - <d>^2."""
        w = np.asarray(state_counts, dtype=np.float64)
        m1 = np.asarray(sum_gd) / w
        m2 = np.asarray(sum_gdd) / w
        self._means = self._means + m1
        self._sigmas = np.sqrt(np.maximum(m2 - m1 * m1, 0.0))
        if np.any(self._sigmas < np.finfo(self._sigmas.dtype).eps):
            raise RuntimeError('at least one sigma is too small to continue.')

    def estimate(self, observations, weights):
        """gaussian.py:214-272 with the reference signature (host arrays)."""
        N = self.nstates
        w_sum = np.zeros(N)
        num = np.zeros(N)
        for o, w in zip(observations, weights):
            num += np.dot(np.asarray(w).T, np.asarray(o, dtype=np.float64))
            w_sum += np.sum(w, axis=0)
        means = num / w_sum
        var = np.zeros(N)
        for o, w in zip(observations, weights):
            d = np.asarray(o, dtype=np.float64)[:, None] - means[None, :]
            var += np.sum(np.asarray(w) * d * d, axis=0)
        self._means = means
        self._sigmas = np.sqrt(var / w_sum)
        if np.any(self._sigmas < np.finfo(self._sigmas.dtype).eps):
            raise RuntimeError('at least one sigma is too small to continue.')

    def sample_from_statistics(self, n_i, sum_d, sum_dd, rng=np.random):
        """Gibbs update of gaussian.py:274-320 from per-state hidden-path statistics
        (count, sum (o - mu_old), sum (o - mu_old)^2)."""
        for i in range(self.nstates):
            n = int(round(n_i[i]))
            if n > 0:
                mean_obs = self._means[i] + sum_d[i] / n
                old_mu = self._means[i]
                self._means[i] = rng.randn() * self._sigmas[i] / np.sqrt(n) + mean_obs
                if n > 1:
                    chi2 = rng.chisquare(n - 1)
                    shift = self._means[i] - old_mu
                    # mean((o - mu_new)^2) from the shifted moments
                    sigmahat2 = (sum_dd[i] - 2.0 * shift * sum_d[i]) / n + shift * shift
                    self._sigmas[i] = np.sqrt(max(sigmahat2, 0.0)) / np.sqrt(chi2 / n)

    def sample(self, observations, prior=None, rng=np.random):
        """gaussian.py:274-320 with the reference's signature: observations[k] are the
        observations assigned to state k.  (The Gibbs sampler of this package feeds
        sample_from_statistics with sums the GPU path pass already produced.)"""
        n_i = np.zeros(self.nstates)
        sum_d = np.zeros(self.nstates)
        sum_dd = np.zeros(self.nstates)
        for i in range(self.nstates):
            o = np.asarray(observations[i], dtype=np.float64)
            d = o - self._means[i]
            n_i[i], sum_d[i], sum_dd[i] = o.size, d.sum(), np.dot(d, d)
        self.sample_from_statistics(n_i, sum_d, sum_dd, rng=rng)

    def generate_observation_from_state(self, state_index, rng=np.random):
        """gaussian.py:322-350."""
        return self._sigmas[state_index] * rng.standard_normal() + self._means[state_index]

    def generate_observations_from_state(self, state_index, nobs, rng=np.random):
        """gaussian.py:352-382."""
        return self._sigmas[state_index] * rng.standard_normal(nobs) + self._means[state_index]

    def generate_observation_trajectory(self, s_t, rng=np.random):
        s_t = np.asarray(s_t)
        return self._means[s_t] + self._sigmas[s_t] * rng.standard_normal(s_t.shape[0])
