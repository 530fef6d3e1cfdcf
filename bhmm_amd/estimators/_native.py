"""Host-side model updates as ONE native call each (include/bhmm_amd.h, section 3): the whole
M-step of an EM iteration (maximum_likelihood.py:284-330) and the parameter draws of a Gibbs
sweep (bayesian_sampling.py:333-373).  Host code of the shared library, no device work; the
numpy functions in _tmatrix.py are the restatements these are tested against.
"""
import ctypes

import numpy as np

from .. import _lib

_KIND = {'gaussian': _lib.EMIT_GAUSSIAN, 'discrete': _lib.EMIT_DISCRETE, 'explicit': _lib.EMIT_EXPLICIT}


def _addr(a):
    """Address of a C-contiguous float64 array as a ctypes pointer, without numpy's `.ctypes`
    helper object (5 us per use; these calls run once per EM iteration / Gibbs sweep)."""
    if a is None:
        return None
    return ctypes.cast(a.__array_interface__['data'][0], _lib.c_double_p)


def _c64(a):
    if isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a, dtype=np.float64)


class MStep(object):
    """Persistent argument block of bhmm_mstep for one estimator (buffers and ctypes pointers are
    made once: the call itself must stay in the tens of microseconds)."""

    def __init__(self, kind, n, M=0):
        self._L = _lib.load()
        self._fn = self._L.bhmm_mstep
        self.kind, self.n, self.M = kind, int(n), int(M)
        self._code = _KIND[kind]
        self.T = np.empty((n, n))
        self.pi = np.empty(n)
        if kind == 'gaussian':
            self.par0, self.par1 = np.empty(n), np.empty(n)
        elif kind == 'discrete':
            self.par0, self.par1 = np.empty((n, M)), None
        else:
            self.par0 = self.par1 = None
        self.info = np.zeros(2, dtype=np.int32)
        self.warm = np.zeros(1 + n)       # state of the reversible fixed point, carried along
        self._out = (_addr(self.T), _addr(self.pi), _addr(self.par0), _addr(self.par1),
                     _lib.ip(self.info), _addr(self.warm))

    def __call__(self, packed, T_old, par0_old, par1_old, reversible, stationary, fixed_pi,
                 maxiter, maxerr, mincount):
        """reversible: True / False / None (= 'iff T_old is reversible', what the reference's
        self._hmm.is_reversible evaluates to).  Returns fresh copies (T, pi, par0, par1)."""
        packed = _c64(packed)
        rev = -1 if reversible is None else int(bool(reversible))
        fp = _c64(fixed_pi) if fixed_pi is not None else None
        T_old = _c64(T_old)
        p0o = _c64(par0_old) if par0_old is not None else None
        p1o = _c64(par1_old) if par1_old is not None else None
        _lib.check(self._fn(self._code, self.n, self.M, _addr(packed), _addr(T_old), _addr(p0o),
                            _addr(p1o), rev, int(bool(stationary)), _addr(fp), int(maxiter),
                            float(maxerr), float(mincount), *self._out))
        return (self.T.copy(), self.pi.copy(),
                None if self.par0 is None else self.par0.copy(),
                None if self.par1 is None else self.par1.copy())


class GibbsParameters(object):
    """Persistent argument block of bhmm_gibbs_parameters for one sampler."""

    def __init__(self, kind, n, M=0, prior_C=None, prior_n0=None, prior_B=None, reversible=True,
                 stationary=False, nsteps=1000):
        self._L = _lib.load()
        self._fn = self._L.bhmm_gibbs_parameters
        self.kind, self.n, self.M = kind, int(n), int(M)
        self._code = _KIND[kind]
        self.prior_C = _lib.f64(prior_C) if prior_C is not None else None
        self.prior_n0 = _lib.f64(prior_n0) if prior_n0 is not None else None
        self.prior_B = _lib.f64(prior_B) if prior_B is not None else None
        self.reversible, self.stationary, self.nsteps = bool(reversible), bool(stationary), int(nsteps)
        self.T = np.empty((n, n))
        self.p0 = np.empty(n)
        self.info = np.zeros(1, dtype=np.int32)
        self._priors = (_addr(self.prior_C), _addr(self.prior_n0), _addr(self.prior_B))
        self._out = (_addr(self.T), _addr(self.p0))
        self._info = _lib.ip(self.info)

    def __call__(self, packed_path_stats, par0, par1, seed, sweep):
        """par0 / par1: current emission parameters (copied, then updated in the copies).
        Returns (T, p0, par0_new, par1_new)."""
        packed = _c64(packed_path_stats)
        p0n = np.array(par0, dtype=np.float64) if par0 is not None else None
        p1n = np.array(par1, dtype=np.float64) if par1 is not None else None
        _lib.check(self._fn(self._code, self.n, self.M, _addr(packed), *self._priors,
                            int(self.reversible), int(self.stationary), self.nsteps,
                            int(seed) & 0xFFFFFFFFFFFFFFFF, int(sweep), *self._out, _addr(p0n),
                            _addr(p1n), self._info))
        return self.T.copy(), self.p0.copy(), p0n, p1n
