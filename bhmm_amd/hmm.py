"""Model container: the fields of bhmm/hmm/generic_hmm.py:25-93 the hot path reads and writes
(initial distribution, transition matrix, output model, hidden paths, likelihood), the
spectral properties SampledHMM reports (:124-274) and sub-models (:276-296).  Synthetic data
generation of the reference class is out of scope."""
import numpy as np

from .estimators import _tmatrix
from .util.statistics import confidence_interval_arr


class HMM(object):
    def __init__(self, Pi, Tij, output_model, lag=1):
        self._nstates = np.shape(Tij)[0]
        self._lag = lag
        self.output_model = output_model
        self.hidden_state_trajectories = None
        self.likelihood = None
        self._spectral = None
        self.update(Pi, Tij)

    def update(self, Pi, Tij):
        """generic_hmm.py:79-93 (same assertions)."""
        self._Tij = np.array(Tij, dtype=np.float64)
        assert _tmatrix.is_transition_matrix(self._Tij), \
            'Given transition matrix is not a stochastic matrix'
        assert self._Tij.shape[0] == self._nstates, \
            'Given transition matrix has unexpected number of states '
        Pi = np.asarray(Pi, dtype=np.float64)
        assert np.all(Pi >= 0), 'Given initial distribution contains negative elements.'
        assert np.any(Pi > 0), 'Given initial distribution is zero'
        self._Pi = np.array(Pi) / np.sum(Pi)
        self._spectral = None

    def __repr__(self):
        return 'HMM(%r, %r, %r)' % (self._Pi, self._Tij, self.output_model)

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) -- what the Gibbs sampler stores per posterior sample
        (bayesian_sampling.py:259) -- without the generic object walk (150 us per sample): arrays
        are copied, hidden paths too, cached spectral data is dropped."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                new.__dict__[k] = v.copy()
            elif k == '_spectral':
                new.__dict__[k] = None
            elif k == 'hidden_state_trajectories' and v is not None:
                new.__dict__[k] = [None if p is None else np.array(p) for p in v]
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    @property
    def lag(self):
        return self._lag

    @property
    def nstates(self):
        return self._nstates

    @property
    def initial_distribution(self):
        return self._Pi

    @property
    def Pi(self):
        return self._Pi

    @property
    def transition_matrix(self):
        return self._Tij

    @property
    def Tij(self):
        return self._Tij

    @property
    def is_strongly_connected(self):
        return _tmatrix.is_connected(self._Tij, strong=True)

    @property
    def is_reversible(self):
        return _tmatrix.is_reversible(self._Tij)

    @property
    def is_stationary(self):
        """generic_hmm.py: initial distribution equals a stationary vector of Tij."""
        return np.allclose(np.dot(self._Pi, self._Tij), self._Pi)

    @property
    def stationary_distribution(self):
        assert self.is_strongly_connected, 'No unique stationary distribution: not connected.'
        return _tmatrix.stationary_vector(self._Tij)

    def _rdl(self):
        self._ensure_spectral_decomposition()
        return self._spectral

    # generic_hmm.py:127-136 (names kept: the reference's tests look for them)
    @property
    def _spectral_decomp_available(self):
        return self._spectral is not None

    def _ensure_spectral_decomposition(self):
        if self._spectral is None:
            R, D, L = _tmatrix.rdl_decomposition(self._Tij, reversible=self.is_reversible)
            self._spectral = (R, np.diag(D), L)

    @property
    def eigenvalues(self):
        """generic_hmm.py:203-213: sorted by descending norm (within a connected set)."""
        return self._rdl()[1]

    @property
    def eigenvectors_left(self):
        """generic_hmm.py:215-226: row matrix."""
        return self._rdl()[2]

    @property
    def eigenvectors_right(self):
        """generic_hmm.py:228-239: column matrix."""
        return self._rdl()[0]

    @property
    def timescales(self):
        """generic_hmm.py:241-257: -lag / ln|lambda_i|, i >= 2 (infinite for |lambda| = 1)."""
        lam = np.abs(self.eigenvalues[1:])
        with np.errstate(divide='ignore'):
            return np.where(lam >= 1.0, np.inf, -self._lag / np.log(lam))

    @property
    def lifetimes(self):
        """generic_hmm.py:259-274: -lag / ln p_ii."""
        return -self._lag / np.log(np.diag(self._Tij))

    def sub_hmm(self, states):
        """generic_hmm.py:276-296."""
        states = np.asarray(states)
        pi_sub = self._Pi[states] / self._Pi[states].sum()
        P_sub = self._Tij[states, :][:, states]
        assert np.all(P_sub.sum(axis=1) > 0), \
            'Illegal sub_hmm request: transition matrix cannot be normalized on ' + str(states)
        P_sub = P_sub / P_sub.sum(axis=1)[:, None]
        return HMM(pi_sub, P_sub, self.output_model.sub_output_model(states), lag=self.lag)

    def count_matrix(self):
        """generic_hmm.py:297-319: lag-1 transition counts of the hidden paths."""
        if self.hidden_state_trajectories is None:
            raise RuntimeError('HMM model does not have a hidden state trajectory.')
        C = np.zeros((self._nstates, self._nstates))
        for s in self.hidden_state_trajectories:
            s = np.asarray(s)
            np.add.at(C, (s[:-1], s[1:]), 1.0)
        return C

    def count_init(self):
        """generic_hmm.py:321-334."""
        if self.hidden_state_trajectories is None:
            raise RuntimeError('HMM model does not have a hidden state trajectory.')
        n = [traj[0] for traj in self.hidden_state_trajectories]
        return np.bincount(n, minlength=self.nstates)

    def collect_observations_in_state(self, observations, state_index):
        """generic_hmm.py:398-431."""
        if not self.hidden_state_trajectories:
            raise RuntimeError('HMM model does not have a hidden state trajectory.')
        parts = [np.asarray(o)[np.asarray(s) == state_index]
                 for s, o in zip(self.hidden_state_trajectories, observations)]
        return np.concatenate(parts) if parts else np.array([])

    # ---- synthetic data (generic_hmm.py:433-589) ------------------------------------------------
    def generate_synthetic_state_trajectory(self, nsteps, initial_Pi=None, start=None, stop=None,
                                            dtype=np.int32, rng=np.random):
        """generic_hmm.py:433-479: a Markov chain of the hidden transition matrix (inverse-CDF
        draws; the reference delegates to msmtools.generation.generate_traj)."""
        if initial_Pi is not None and start is not None:
            raise ValueError('Arguments initial_Pi and start are exclusive. Only set one of them.')
        if start is None:
            p0 = self._Pi if initial_Pi is None else np.asarray(initial_Pi, dtype=np.float64)
            start = int(rng.choice(self._nstates, p=p0 / p0.sum()))
        cdf = np.cumsum(self._Tij, axis=1)
        u = rng.random_sample(int(nsteps))
        traj = np.empty(int(nsteps), dtype=dtype)
        s = int(start)
        for t in range(int(nsteps)):
            if t > 0:
                s = min(int(np.searchsorted(cdf[s], u[t], side='right')), self._nstates - 1)
            traj[t] = s
            if stop is not None and s == stop:
                return traj[:t + 1]
        return traj

    def generate_synthetic_observation(self, state, rng=np.random):
        """generic_hmm.py:481-504."""
        return self.output_model.generate_observation_from_state(state, rng=rng)

    def generate_synthetic_observation_trajectory(self, length, initial_Pi=None, rng=np.random):
        """generic_hmm.py:506-545: (observations, hidden states)."""
        s_t = self.generate_synthetic_state_trajectory(length, initial_Pi=initial_Pi, rng=rng)
        return [self.output_model.generate_observation_trajectory(s_t, rng=rng), s_t]

    def generate_synthetic_observation_trajectories(self, ntrajectories, length, initial_Pi=None,
                                                    rng=np.random):
        """generic_hmm.py:547-589: ([observations], [hidden states])."""
        O, S = [], []
        for _ in range(ntrajectories):
            o_t, s_t = self.generate_synthetic_observation_trajectory(length, initial_Pi=initial_Pi,
                                                                      rng=rng)
            O.append(o_t)
            S.append(s_t)
        return [O, S]


def _stat_property(name, doc):
    def samples(self):
        return np.array([np.asarray(getattr(h, name)) for h in self._sampled_hmms])

    def mean(self):
        return np.mean(samples(self), axis=0)

    def std(self):
        return np.std(samples(self), axis=0)

    def conf(self):
        return confidence_interval_arr(samples(self), conf=self._conf)

    return (property(samples, doc=doc + ' of every sample'), property(mean), property(std),
            property(conf))


class SampledHMM(HMM):
    """bhmm/hmm/generic_sampled_hmm.py:27-251: the estimated model plus statistics (mean, std,
    confidence interval) of a list of sampled models.  The Gaussian / discrete output
    statistics of SampledGaussianHMM / SampledDiscreteHMM (gaussian_hmm.py:66-112,
    discrete_hmm.py:62-83) are available on the same object when the output model has them.
    Iterating or indexing gives the sampled models."""

    def __init__(self, estimated_hmm, sampled_hmms, conf=0.95):
        HMM.__init__(self, estimated_hmm.initial_distribution, estimated_hmm.transition_matrix,
                     estimated_hmm.output_model, lag=estimated_hmm.lag)
        self.hidden_state_trajectories = estimated_hmm.hidden_state_trajectories
        self.likelihood = estimated_hmm.likelihood
        self._sampled_hmms = list(sampled_hmms)
        self._nsamples = len(self._sampled_hmms)
        self.set_confidence(conf)

    def set_confidence(self, conf):
        self._conf = conf

    @property
    def nsamples(self):
        return self._nsamples

    @property
    def sampled_hmms(self):
        return self._sampled_hmms

    @property
    def confidence_interval(self):
        return self._conf

    def __len__(self):
        return self._nsamples

    def __iter__(self):
        return iter(self._sampled_hmms)

    def __getitem__(self, i):
        return self._sampled_hmms[i]

    (initial_distribution_samples, initial_distribution_mean, initial_distribution_std,
     initial_distribution_conf) = _stat_property('initial_distribution', 'initial distribution')
    (transition_matrix_samples, transition_matrix_mean, transition_matrix_std,
     transition_matrix_conf) = _stat_property('transition_matrix', 'transition matrix')
    (eigenvalues_samples, eigenvalues_mean, eigenvalues_std,
     eigenvalues_conf) = _stat_property('eigenvalues', 'eigenvalues')
    (eigenvectors_left_samples, eigenvectors_left_mean, eigenvectors_left_std,
     eigenvectors_left_conf) = _stat_property('eigenvectors_left', 'left eigenvectors')
    (eigenvectors_right_samples, eigenvectors_right_mean, eigenvectors_right_std,
     eigenvectors_right_conf) = _stat_property('eigenvectors_right', 'right eigenvectors')
    (timescales_samples, timescales_mean, timescales_std,
     timescales_conf) = _stat_property('timescales', 'relaxation timescales')
    (lifetimes_samples, lifetimes_mean, lifetimes_std,
     lifetimes_conf) = _stat_property('lifetimes', 'state lifetimes')

    @property
    def stationary_distribution_samples(self):
        """generic_sampled_hmm.py:92-98."""
        if self.is_stationary:
            return self.initial_distribution_samples
        return np.array([h.stationary_distribution for h in self._sampled_hmms])

    @property
    def stationary_distribution_mean(self):
        return np.mean(self.stationary_distribution_samples, axis=0)

    @property
    def stationary_distribution_std(self):
        return np.std(self.stationary_distribution_samples, axis=0)

    @property
    def stationary_distribution_conf(self):
        return confidence_interval_arr(self.stationary_distribution_samples, conf=self._conf)

    def _output_samples(self, name):
        return np.array([np.asarray(getattr(h.output_model, name)) for h in self._sampled_hmms])

    def __getattr__(self, attr):
        # means_* / sigmas_* (Gaussian) and output_probabilities_* (discrete)
        if attr.startswith('_') or attr == 'output_model':
            raise AttributeError(attr)
        for base in ('means', 'sigmas', 'output_probabilities'):
            if attr.startswith(base + '_') and hasattr(self.output_model, base):
                x = self._output_samples(base)
                kind = attr[len(base) + 1:]
                if kind == 'samples':
                    return x
                if kind == 'mean':
                    return np.mean(x, axis=0)
                if kind == 'std':
                    return np.std(x, axis=0)
                if kind == 'conf':
                    return confidence_interval_arr(x, conf=self._conf)
        raise AttributeError(attr)


# ---- typed convenience views (bhmm/hmm/gaussian_hmm.py, discrete_hmm.py) ---------------------
# The reference builds these by multiple inheritance from the output-model classes; here they are
# HMM objects that forward the output model's attributes, which is all their users read.
class _TypedHMM(HMM):
    _kind = None

    def __init__(self, hmm):
        if hmm.output_model.model_type != self._kind:
            raise TypeError('Given hmm is not a %s HMM, but has an output model of type: %s'
                            % (self._kind, type(hmm.output_model)))
        HMM.__init__(self, hmm.initial_distribution, hmm.transition_matrix, hmm.output_model,
                     lag=hmm.lag)
        self.hidden_state_trajectories = hmm.hidden_state_trajectories
        self.likelihood = hmm.likelihood

    def __getattr__(self, attr):
        if attr.startswith('_') or attr == 'output_model':
            raise AttributeError(attr)
        return getattr(self.output_model, attr)


class GaussianHMM(_TypedHMM):
    """gaussian_hmm.py:29-40: `means`, `sigmas`, ... of the output model on the HMM itself."""
    _kind = 'gaussian'


class DiscreteHMM(_TypedHMM):
    """discrete_hmm.py:29-40: `output_probabilities`, `nsymbols`, ... on the HMM itself."""
    _kind = 'discrete'


class SampledGaussianHMM(SampledHMM):
    """gaussian_hmm.py:43-112: means_* / sigmas_* statistics over the samples."""

    def __init__(self, estimated_hmm, sampled_hmms, conf=0.95):
        GaussianHMM(estimated_hmm)      # type check
        SampledHMM.__init__(self, estimated_hmm, sampled_hmms, conf=conf)


class SampledDiscreteHMM(SampledHMM):
    """discrete_hmm.py:43-83: output_probabilities_* statistics over the samples."""

    def __init__(self, estimated_hmm, sampled_hmms, conf=0.95):
        DiscreteHMM(estimated_hmm)      # type check
        SampledHMM.__init__(self, estimated_hmm, sampled_hmms, conf=conf)
