"""Hidden-variable kernels with the signatures of bhmm/hidden/api.py, executed on MI355X.

Every function mirrors its reference twin (cited per function): same argument meaning, same
return values, same treatment of T / *_out buffers (buffers may have more rows than T; rows
>= T are left untouched, maximum_likelihood.py:128-130,253-265), same exceptions.  The
arithmetic runs in the HIP library through the C ABI (include/bhmm_amd.h); there is no
Python/CPU implementation here and none is selected silently.
"""
import warnings

import numpy as np

from .. import _lib
from ..util import config

__all__ = ['set_implementation', 'forward', 'backward', 'state_probabilities', 'state_counts',
           'transition_counts', 'viterbi', 'sample_path']

__IMPL_HIP__ = 2
__impl__ = __IMPL_HIP__


def set_implementation(impl):
    """bhmm/hidden/api.py:44-62.  Only 'hip' exists in this package; any other name warns
    (like the reference does for unknown names) and keeps 'hip'."""
    global __impl__
    if impl.lower() != 'hip':
        warnings.warn('Implementation ' + impl + ' is not available in bhmm_amd. Using the hip '
                      'implementation.')
    __impl__ = __IMPL_HIP__


def _check_dtype():
    if config.dtype != np.float64:
        raise TypeError('bhmm_amd kernels are float64 only (hidden.pyx:59-68)')


def _rows(T, pobs, what='pobs'):
    if T is None:
        return pobs.shape[0]
    if T > pobs.shape[0]:
        raise ValueError('T must be at most the length of ' + what + '.')
    return int(T)


def forward(A, pobs, pi, T=None, alpha_out=None):
    """bhmm/hidden/api.py:65-96 -> _forward (_hidden.c:16-66).  Returns (logprob, alpha)."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    T = _rows(T, pobs)
    N = A.shape[0]
    if alpha_out is None:
        alpha_out = np.zeros((T, N), dtype=np.float64, order='C')
    elif T > alpha_out.shape[0]:
        raise ValueError('alpha_out must at least have length T in order to fit trajectory.')
    A_, p_, pi_ = _lib.f64(A), _lib.f64(pobs[:T]), _lib.f64(pi)
    direct = alpha_out.flags.c_contiguous and alpha_out.dtype == np.float64
    buf = alpha_out if direct else np.empty((T, N))
    import ctypes
    ll = ctypes.c_double(0.0)
    _lib.check(L.bhmm_forward(_lib.dp(buf), ctypes.byref(ll), _lib.dp(A_), _lib.dp(p_),
                              _lib.dp(pi_), N, T))
    if not direct:
        alpha_out[:T] = buf
    return ll.value, alpha_out


def backward(A, pobs, T=None, beta_out=None):
    """bhmm/hidden/api.py:99-125 -> _backward (_hidden.c:69-110).  Returns beta."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    T = _rows(T, pobs)
    N = A.shape[0]
    if beta_out is None:
        beta_out = np.zeros((T, N), dtype=np.float64, order='C')
    elif T > beta_out.shape[0]:
        raise ValueError('beta_out must at least have length T in order to fit trajectory.')
    A_, p_ = _lib.f64(A), _lib.f64(pobs[:T])
    direct = beta_out.flags.c_contiguous and beta_out.dtype == np.float64
    buf = beta_out if direct else np.empty((T, N))
    _lib.check(L.bhmm_backward(_lib.dp(buf), _lib.dp(A_), _lib.dp(p_), N, T))
    if not direct:
        beta_out[:T] = buf
    return beta_out


def state_probabilities(alpha, beta, T=None, gamma_out=None):
    """bhmm/hidden/api.py:133-188 (numpy there; _computeGamma _hidden.c:113-131 here)."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    if alpha.shape[0] != beta.shape[0]:
        raise ValueError('Inconsistent sizes of alpha and beta.')
    if T is None:
        T = alpha.shape[0] if gamma_out is None else gamma_out.shape[0]
    T = int(T)
    N = alpha.shape[1]
    if gamma_out is None:
        gamma_out = np.empty((T, N), dtype=np.float64)
        rows = T
    else:
        # reference: writes T rows if gamma_out is shorter than alpha, else all rows of alpha
        rows = T if gamma_out.shape[0] < alpha.shape[0] else alpha.shape[0]
    a_, b_ = _lib.f64(alpha[:rows]), _lib.f64(beta[:rows])
    direct = gamma_out.flags.c_contiguous and gamma_out.dtype == np.float64
    buf = gamma_out if direct else np.empty((rows, N))
    _lib.check(L.bhmm_state_probabilities(_lib.dp(buf), _lib.dp(a_), _lib.dp(b_), N, rows))
    if not direct:
        gamma_out[:rows] = buf
    return gamma_out


def state_counts(gamma, T, out=None):
    """bhmm/hidden/api.py:191-211 (a numpy column sum in the reference as well)."""
    return np.sum(gamma[0:T], axis=0, out=out)


def transition_counts(alpha, beta, A, pobs, T=None, out=None):
    """bhmm/hidden/api.py:214-248 -> _compute_transition_counts (_hidden.c:148-183).
    `out` is overwritten, not accumulated (_hidden.c:160-162)."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    T = _rows(T, pobs)
    N = len(A)
    if out is None:
        out = np.zeros((N, N), dtype=np.float64, order='C')
    A_, p_ = _lib.f64(A), _lib.f64(pobs[:T])
    a_, b_ = _lib.f64(alpha[:T]), _lib.f64(beta[:T])
    direct = out.flags.c_contiguous and out.dtype == np.float64
    buf = out if direct else np.empty((N, N))
    _lib.check(L.bhmm_transition_counts(_lib.dp(buf), _lib.dp(A_), _lib.dp(p_), _lib.dp(a_),
                                        _lib.dp(b_), N, T))
    if not direct:
        out[:] = buf
    return out


def viterbi(A, pobs, pi):
    """bhmm/hidden/api.py:251-274 -> _compute_viterbi (_hidden.c:203-281).  int32 path,
    bit-identical to the reference C for the same pobs."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    T, N = pobs.shape[0], A.shape[0]
    path = np.zeros(T, dtype=np.int32)
    A_, p_, pi_ = _lib.f64(A), _lib.f64(pobs), _lib.f64(pi)
    _lib.check(L.bhmm_viterbi(_lib.ip(path), _lib.dp(A_), _lib.dp(p_), _lib.dp(pi_), N, T))
    return path


def sample_path(alpha, A, pobs, T=None, seed=None, u=None):
    """bhmm/hidden/api.py:277-304 -> set_seed/_sample_path (_hidden.c:321-378).

    The reference draws from the C library generator (seeded only when `seed` is given);
    so does this function, so the path equals the reference's for the same seed.  `u`
    (extension) supplies the uniforms directly: u[t] is used for step t."""
    _check_dtype()
    L = _lib.load()
    _lib.require_device()
    N = pobs.shape[1]
    if T is None:
        T = pobs.shape[0]
    elif T > pobs.shape[0] or T > alpha.shape[0]:
        raise ValueError('T must be at most the length of pobs and alpha.')
    T = int(T)
    if u is None:
        u = np.empty(T)
        _lib.check(L.bhmm_libc_uniforms(_lib.dp(u), T, -1 if seed is None else int(seed)))
    u_ = _lib.f64(u)
    path = np.zeros(T, dtype=np.int32)
    a_, A_ = _lib.f64(alpha[:T]), _lib.f64(A)
    _lib.check(L.bhmm_sample_path(_lib.ip(path), _lib.dp(a_), _lib.dp(A_), _lib.dp(u_), N, T))
    return path
