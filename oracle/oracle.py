"""ctypes front-end of the parity checker (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package bhmm_amd never does.

Two libraries live here:

* ``liboracle.so``          -- oracle/bhmm_oracle.c, this repo's CPU restatement of the
                               reference kernels (each function cites the reference lines).
* ``_ref/libbhmm_ref.so``   -- the reference's own C sources (bhmm/hidden/impl_c/_hidden.c,
                               bhmm/output_models/impl_c/_gaussian.c, _discrete.c) compiled
                               in place by oracle/Makefile.  Optional: present only after a
                               build in a container that has /root/reference; the prebuilt
                               file travels to the GPU box.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_i64_p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bhmm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/bhmm") and (
            force or not os.path.exists(os.path.join(_HERE, "_ref", "libbhmm_ref.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _dp(a):
    return a.ctypes.data_as(_c_double_p)


def _ip(a):
    return a.ctypes.data_as(_c_int_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_forward.restype = ctypes.c_double
        L.orc_pobs_gaussian.restype = ctypes.c_long
        _lib = L
    return _lib


# ---------------------------------------------------------------------------------------
# restatement (liboracle.so)
# ---------------------------------------------------------------------------------------

def pobs_gaussian(obs, mu, sigma, ignore_outliers=True):
    obs, mu, sigma = _f64(obs), _f64(mu), _f64(sigma)
    T, N = obs.shape[0], mu.shape[0]
    out = np.empty((T, N))
    lib().orc_pobs_gaussian(_dp(obs), ctypes.c_long(T), _dp(mu), _dp(sigma), N,
                            int(bool(ignore_outliers)), _dp(out))
    return out


def pobs_discrete(obs, B):
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    B = _f64(B)
    N, M = B.shape
    out = np.empty((obs.shape[0], N))
    lib().orc_pobs_discrete(_ip(obs), ctypes.c_long(obs.shape[0]), _dp(B), N, M, _dp(out))
    return out


def update_pout(obs, w, pout):
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    w = _f64(w)
    N, M = pout.shape
    assert pout.flags.c_contiguous and pout.dtype == np.float64
    lib().orc_update_pout(_ip(obs), _dp(w), ctypes.c_long(obs.shape[0]), N, M, _dp(pout))
    return pout


def forward(A, pobs, pi):
    A, pobs, pi = _f64(A), _f64(pobs), _f64(pi)
    T, N = pobs.shape
    alpha = np.empty((T, N))
    logL = lib().orc_forward(_dp(alpha), _dp(A), _dp(pobs), _dp(pi), N, ctypes.c_long(T))
    return logL, alpha


def backward(A, pobs):
    A, pobs = _f64(A), _f64(pobs)
    T, N = pobs.shape
    beta = np.empty((T, N))
    lib().orc_backward(_dp(beta), _dp(A), _dp(pobs), N, ctypes.c_long(T))
    return beta


def gamma(alpha, beta):
    alpha, beta = _f64(alpha), _f64(beta)
    T, N = alpha.shape
    g = np.empty((T, N))
    lib().orc_gamma(_dp(g), _dp(alpha), _dp(beta), N, ctypes.c_long(T))
    return g


def state_counts(gam):
    gam = _f64(gam)
    T, N = gam.shape
    c = np.empty(N)
    lib().orc_state_counts(_dp(c), _dp(gam), N, ctypes.c_long(T))
    return c


def transition_counts(alpha, beta, A, pobs):
    alpha, beta, A, pobs = _f64(alpha), _f64(beta), _f64(A), _f64(pobs)
    T, N = pobs.shape
    C = np.empty((N, N))
    rc = lib().orc_transition_counts(_dp(C), _dp(A), _dp(pobs), _dp(alpha), _dp(beta), N,
                                     ctypes.c_long(T))
    if rc:
        raise MemoryError()
    return C


def viterbi(A, pobs, pi):
    A, pobs, pi = _f64(A), _f64(pobs), _f64(pi)
    T, N = pobs.shape
    path = np.empty(T, dtype=np.int32)
    rc = lib().orc_viterbi(_ip(path), _dp(A), _dp(pobs), _dp(pi), N, ctypes.c_long(T))
    if rc:
        raise MemoryError()
    return path


def sample_path(alpha, A, u=None, seed=None):
    """Backward sampling.  u: uniforms per step (u[t] used at step t); None -> libc rand()."""
    alpha, A = _f64(alpha), _f64(A)
    T, N = alpha.shape
    path = np.empty(T, dtype=np.int32)
    if seed is not None:
        lib().orc_set_seed(int(seed))
    up = _dp(_f64(u)) if u is not None else None
    if u is not None:
        u = _f64(u)
        up = _dp(u)
    rc = lib().orc_sample_path(_ip(path), _dp(alpha), _dp(A), N, ctypes.c_long(T), up)
    if rc:
        raise RuntimeError("sample_path failed: %d" % rc)
    return path


def libc_uniforms(T, seed):
    """The uniforms the reference's _sample_path would draw after set_seed(seed)."""
    u = np.empty(T)
    lib().orc_set_seed(int(seed))
    lib().orc_fill_uniforms_libc(_dp(u), ctypes.c_long(T))
    return u


def path_counts(paths, N):
    C = np.zeros((N, N), dtype=np.int64)
    n0 = np.zeros(N, dtype=np.int64)
    for p in paths:
        p = np.ascontiguousarray(p, dtype=np.int32)
        lib().orc_path_counts(_ip(p), ctypes.c_long(p.shape[0]), N,
                              C.ctypes.data_as(_c_i64_p), n0.ctypes.data_as(_c_i64_p))
    return C, n0


def estep(kind, observations, A, pi, par0, par1=None, want_gamma=False):
    """E-step over a list of trajectories, trajectory-ordered sums like
    maximum_likelihood.py:271-282,383-385.  kind: 'gaussian' (par0=means, par1=sigmas) or
    'discrete' (par0=B).  Returns dict(logL[K], gamma0_sum, C, gammas (opt), state_counts)."""
    A, pi, par0 = _f64(A), _f64(pi), _f64(par0)
    N = A.shape[0]
    K = len(observations)
    M = par0.shape[1] if kind == 'discrete' else 0
    par1 = _f64(par1) if par1 is not None else par0
    logL = np.zeros(K)
    g0 = np.zeros(N)
    Csum = np.zeros((N, N))
    sc = np.zeros(N)
    gammas = []
    for k, o in enumerate(observations):
        T = len(o)
        o = _f64(o) if kind == 'gaussian' else np.ascontiguousarray(o, dtype=np.int32)
        work = np.empty(3 * T * N)
        g = np.empty((T, N))
        C = np.empty((N, N))
        ll = ctypes.c_double(0.0)
        rc = lib().orc_estep_one(0 if kind == 'gaussian' else 1,
                                 o.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(T), N, M,
                                 _dp(A), _dp(pi), _dp(par0), _dp(par1), _dp(work), _dp(g), _dp(C),
                                 ctypes.byref(ll))
        if rc:
            raise MemoryError()
        logL[k] = ll.value
        g0 += g[0]
        Csum += C
        sc += g.sum(axis=0)
        gammas.append(g)
    out = dict(logL=logL, gamma0_sum=g0, C=Csum, state_counts=sc)
    if want_gamma:
        out['gammas'] = gammas
    return out


def estimate_gaussian(observations, gammas):
    N = gammas[0].shape[1]
    off = np.zeros(len(observations) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(o) for o in observations])
    obs = _f64(np.concatenate(observations))
    gam = _f64(np.concatenate(gammas, axis=0))
    mu, sig = np.empty(N), np.empty(N)
    lib().orc_estimate_gaussian(_dp(obs), _dp(gam), off.ctypes.data_as(_c_i64_p),
                                len(observations), N, _dp(mu), _dp(sig))
    return mu, sig


def estimate_discrete(observations, gammas, M):
    """discrete.py:159-215: zero, scatter-add per trajectory, row-normalise."""
    N = gammas[0].shape[1]
    B = np.zeros((N, M))
    for o, g in zip(observations, gammas):
        update_pout(o, g, B)
    B /= B.sum(axis=1)[:, None]
    return B


# ---------------------------------------------------------------------------------------
# the reference's own compiled C (oracle/_ref), when present
# ---------------------------------------------------------------------------------------

_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libbhmm_ref.so"))


def ref():
    global _ref
    if _ref is None:
        R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libbhmm_ref.so"))
        R._forward.restype = ctypes.c_double
        R._backward.restype = None
        R._p_obs.restype = None
        R._update_pout.restype = None
        _ref = R
    return _ref


_omp = {}


def estep_batch_omp(kind, obs, A, pi, par0, par1=None, threads=0):
    """The per-trajectory E-step sequence of maximum_likelihood.py:249-265 over a (K, T) batch, one
    OpenMP task per trajectory (oracle/omp_driver.c): with the reference's own C kernels when
    oracle/_ref/libbhmm_ref_omp.so exists, else with this repo's restatement.  Returns
    (per-trajectory log-likelihoods, threads used, 'reference' | 'port')."""
    which = "reference" if os.path.exists(os.path.join(_HERE, "_ref", "libbhmm_ref_omp.so")) else "port"
    if which not in _omp:
        path = os.path.join(_HERE, "_ref", "libbhmm_ref_omp.so") if which == "reference" else \
            os.path.join(_HERE, "liboracle_omp.so")
        if which == "port" and not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_omp.so"])
        L = ctypes.CDLL(path)
        L.orc_estep_batch_omp.restype = ctypes.c_int
        _omp[which] = L
    A, pi, par0 = _f64(A), _f64(pi), _f64(par0)
    par1 = _f64(par1) if par1 is not None else par0
    obs = np.ascontiguousarray(obs, dtype=np.float64 if kind == 'gaussian' else np.int32)
    K, T = obs.shape
    N = A.shape[0]
    M = par0.shape[1] if kind == 'discrete' else 0
    logL = np.zeros(K)
    used = _omp[which].orc_estep_batch_omp(0 if kind == 'gaussian' else 1, obs.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_int(K), ctypes.c_long(T), ctypes.c_int(N), ctypes.c_int(M),
                                           _dp(A), _dp(pi), _dp(par0), _dp(par1), ctypes.c_int(threads), _dp(logL))
    if used < 0:
        raise MemoryError()
    return logL, used, which


def ref_pobs_gaussian(obs, mu, sigma, out=None):
    obs, mu, sigma = _f64(obs), _f64(mu), _f64(sigma)
    T, N = obs.shape[0], mu.shape[0]
    if out is None:
        out = np.empty((T, N))
    ref()._p_obs(_dp(obs), _dp(mu), _dp(sigma), N, T, _dp(out))
    return out


def ref_forward(A, pobs, pi, alpha=None):
    T, N = pobs.shape
    if alpha is None:
        alpha = np.empty((T, N))
    ll = ref()._forward(_dp(alpha), _dp(A), _dp(pobs), _dp(pi), N, T)
    return ll, alpha


def ref_backward(A, pobs, beta=None):
    T, N = pobs.shape
    if beta is None:
        beta = np.empty((T, N))
    ref()._backward(_dp(beta), _dp(A), _dp(pobs), N, T)
    return beta


def ref_transition_counts(alpha, beta, A, pobs, C=None):
    T, N = pobs.shape
    if C is None:
        C = np.empty((N, N))
    rc = ref()._compute_transition_counts(_dp(C), _dp(A), _dp(pobs), _dp(alpha), _dp(beta), N, T)
    if rc:
        raise MemoryError()
    return C


def ref_viterbi(A, pobs, pi):
    T, N = pobs.shape
    path = np.empty(T, dtype=np.int32)
    rc = ref()._compute_viterbi(_ip(path), _dp(A), _dp(pobs), _dp(pi), N, T)
    if rc:
        raise MemoryError()
    return path


def ref_sample_path(alpha, A, pobs, seed=None):
    T, N = alpha.shape
    path = np.empty(T, dtype=np.int32)
    if seed is not None:
        ref().set_seed(int(seed))
    rc = ref()._sample_path(_ip(path), _dp(alpha), _dp(A), _dp(pobs), N, T)
    if rc:
        raise MemoryError()
    return path


def ref_update_pout(obs, w, pout):
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    N, M = pout.shape
    ref()._update_pout(_ip(obs), _dp(w), obs.shape[0], N, M, _dp(pout))
    return pout


def ref_gamma(alpha, beta, out=None):
    """numpy lines of bhmm/hidden/api.py:176-186 (multiply, dot-with-ones, divide)."""
    ones = np.ones(alpha.shape[1])[:, None]
    if out is None:
        out = alpha * beta
    else:
        np.multiply(alpha, beta, out)
    np.divide(out, np.dot(out, ones), out=out)
    return out
