"""Emission-model base class: the part of bhmm/output_models/outputmodel.py:24-131 that sits
on the hot path (state count, implementation switch, outlier rule)."""
import warnings

import numpy as np


class OutputModel(object):
    def __init__(self, nstates, ignore_outliers=True):
        self._nstates = int(nstates)
        self.ignore_outliers = ignore_outliers
        self.found_outliers = False

    @property
    def nstates(self):
        return self._nstates

    def __deepcopy__(self, memo):
        """Array attributes copied, the rest by reference semantics of copy.copy (scalars, flags)."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = v.copy() if isinstance(v, np.ndarray) else v
        return new

    def set_implementation(self, impl):
        """outputmodel.py:69-86.  Only 'hip' exists here; other names warn and keep it."""
        if impl.lower() != 'hip':
            warnings.warn('Implementation ' + impl + ' is not available in bhmm_amd. Using the '
                          'hip implementation.')

    def _handle_outliers(self, p_o):
        """outputmodel.py:119-131: rows summing to exactly zero become uniform ones."""
        if self.ignore_outliers:
            outliers = np.where(p_o.sum(axis=1) == 0)[0]
            if outliers.size > 0:
                p_o[outliers, :] = 1.0
                self.found_outliers = True
        return p_o

    def log_p_obs(self, obs, out=None):
        p = self.p_obs(obs, out=out)
        return np.log(p, out=p)
