"""Plain model container: the fields of bhmm/hmm/generic_hmm.py:25-93 the hot path reads and
writes (initial distribution, transition matrix, output model, hidden paths, likelihood).
Spectral analysis, sub-models and synthetic data generation of the reference class are out
of scope."""
import numpy as np

from .estimators import _tmatrix


class HMM(object):
    def __init__(self, Pi, Tij, output_model, lag=1):
        self._nstates = np.shape(Tij)[0]
        self._lag = lag
        self.output_model = output_model
        self.hidden_state_trajectories = None
        self.likelihood = None
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

    def __repr__(self):
        return 'HMM(%r, %r, %r)' % (self._Pi, self._Tij, self.output_model)

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
