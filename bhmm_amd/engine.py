"""Device-resident batch engine: the host-side handle of one `bhmm_ctx` (include/bhmm_amd.h).

One Engine holds a set of observation trajectories on one GPU (uploaded once -- they are
constant across EM iterations / Gibbs sweeps, maximum_likelihood.py:101) and runs the whole
per-iteration loop over trajectories of the reference
(maximum_likelihood.py:383-385, bayesian_sampling.py:288-290) as one call.
"""
import ctypes

import numpy as np

from . import _lib


class EStepResult(object):
    """Reduced sufficient statistics of one E-step (what maximum_likelihood.py:271-282 and the
    emission `estimate` methods consume)."""

    def __init__(self, kind, n, M, packed, logL_k):
        self.kind = kind
        self.packed = packed
        self.logL_k = logL_k
        o = 0
        self.loglik = float(packed[o]); o += 1
        self.gamma0_sum = packed[o:o + n].copy(); o += n
        self.C = packed[o:o + n * n].reshape(n, n).copy(); o += n * n
        self.state_counts = packed[o:o + n].copy(); o += n
        self.sum_gd = self.sum_gdd = self.symbol_counts = None
        if kind == 'gaussian':
            self.sum_gd = packed[o:o + n].copy(); o += n
            self.sum_gdd = packed[o:o + n].copy(); o += n
        elif kind == 'discrete':
            self.symbol_counts = packed[o:o + n * M].reshape(n, M).copy(); o += n * M


_KINDS = {'gaussian': _lib.EMIT_GAUSSIAN, 'discrete': _lib.EMIT_DISCRETE,
          'explicit': _lib.EMIT_EXPLICIT}


class Engine(object):
    def __init__(self, device=0, stream=None):
        self._L = _lib.load()
        _lib.require_device()
        h = ctypes.c_void_p()
        _lib.check(self._L.bhmm_ctx_create(ctypes.byref(h), int(device),
                                           ctypes.c_void_p(stream) if stream else None))
        self._h = h
        self.kind = None
        self.nstates = 0
        self.nsymbols = 0
        self.lengths = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.bhmm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ---------------------------------------------------------------------------
    def set_observations(self, kind, observations, nstates, nsymbols=0, chunk=0):
        """observations: list of 1-d arrays (gaussian: float, discrete: int) or, for kind
        'explicit', list of (T_k, nstates) pobs matrices."""
        code = _KINDS[kind]
        lengths = np.array([len(o) for o in observations], dtype=np.int64)
        off = np.zeros(len(observations) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lengths)
        if kind == 'gaussian':
            flat = np.ascontiguousarray(np.concatenate([np.asarray(o, dtype=np.float64)
                                                        for o in observations]))
        elif kind == 'discrete':
            flat = np.ascontiguousarray(np.concatenate([np.asarray(o).astype(np.int32)
                                                        for o in observations]))
            if flat.size and (flat.min() < 0 or flat.max() >= nsymbols):
                raise ValueError("discrete observation outside [0, nsymbols)")
        else:
            flat = np.ascontiguousarray(np.concatenate(
                [np.asarray(o, dtype=np.float64).reshape(-1, nstates) for o in observations]))
        _lib.check(self._L.bhmm_ctx_set_observations(
            self._h, code, flat.ctypes.data_as(ctypes.c_void_p), _lib.lp(off), len(observations),
            int(nstates), int(nsymbols), int(chunk), 0))
        self._adopt(kind, nstates, nsymbols, lengths)

    def set_observations_device(self, kind, dev_ptr, offsets, nstates, nsymbols=0, chunk=0):
        """Same, for a trajectory-concatenated buffer already resident on this GPU
        (dev_ptr: integer device address, e.g. torch.Tensor.data_ptr())."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        _lib.check(self._L.bhmm_ctx_set_observations(
            self._h, _KINDS[kind], ctypes.c_void_p(int(dev_ptr)), _lib.lp(off), len(off) - 1,
            int(nstates), int(nsymbols), int(chunk), 1))
        self._adopt(kind, nstates, nsymbols, np.diff(off))

    def _adopt(self, kind, nstates, nsymbols, lengths):
        self.kind = kind
        self.nstates = int(nstates)
        self.nsymbols = int(nsymbols)
        self.lengths = lengths
        self.offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)

    @property
    def stats_size(self):
        return self._L.bhmm_ctx_stats_size(self._h)

    @property
    def total_steps(self):
        return self._L.bhmm_ctx_total_steps(self._h)

    @property
    def num_chunks(self):
        return self._L.bhmm_ctx_num_chunks(self._h)

    @property
    def chunk_len(self):
        return self._L.bhmm_ctx_chunk_len(self._h)

    @property
    def stream(self):
        return self._L.bhmm_ctx_stream(self._h)

    def set_option(self, name, value):
        _lib.check(self._L.bhmm_ctx_set_option(self._h, name.encode(), float(value)))

    def get_option(self, name):
        v = ctypes.c_double(0.0)
        _lib.check(self._L.bhmm_ctx_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def sync(self):
        _lib.check(self._L.bhmm_ctx_sync(self._h))

    def kernel_ms(self, which):
        return self._L.bhmm_ctx_last_kernel_ms(self._h, which)

    # -- E-step --------------------------------------------------------------------------
    def _model_ptrs(self, A, pi, par0, par1):
        A = _lib.f64(A)
        pi = _lib.f64(pi)
        p0 = _lib.f64(par0) if par0 is not None else None
        p1 = _lib.f64(par1) if par1 is not None else None
        n = self.nstates
        if A.shape != (n, n) or pi.shape != (n,):
            raise ValueError("model shape does not match nstates=%d" % n)
        return A, pi, p0, p1

    def estep_launch(self, A, pi, par0=None, par1=None, stats_dev=None, store_gamma=False):
        """Enqueue one E-step.  stats_dev: optional device address receiving the packed
        statistics (e.g. a torch tensor that is all-reduced across ranks afterwards)."""
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        flags = _lib.FLAG_STORE_GAMMA if store_gamma else 0
        _lib.check(self._L.bhmm_estep(self._h, _lib.dp(A), _lib.dp(pi), _lib.dp(p0), _lib.dp(p1),
                                      ctypes.c_void_p(int(stats_dev)) if stats_dev else None,
                                      flags))

    def estep_fetch(self):
        S = self.stats_size
        packed = np.empty(S)
        logL_k = np.empty(len(self.lengths))
        _lib.check(self._L.bhmm_estep_fetch(self._h, _lib.dp(packed), _lib.dp(logL_k)))
        return EStepResult(self.kind, self.nstates, self.nsymbols, packed, logL_k)

    def estep(self, A, pi, par0=None, par1=None, store_gamma=False):
        self.estep_launch(A, pi, par0, par1, store_gamma=store_gamma)
        return self.estep_fetch()

    def unpack(self, packed, logL_k=None):
        return EStepResult(self.kind, self.nstates, self.nsymbols, np.asarray(packed), logL_k)

    def gamma(self, k):
        T = int(self.lengths[k])
        g = np.empty((T, self.nstates))
        _lib.check(self._L.bhmm_get_gamma(self._h, int(k), _lib.dp(g)))
        return g

    # -- paths ---------------------------------------------------------------------------
    def viterbi(self, A, pi, par0=None, par1=None):
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        paths = np.empty(int(self.offsets[-1]), dtype=np.int32)
        _lib.check(self._L.bhmm_viterbi_batch(self._h, _lib.dp(A), _lib.dp(pi), _lib.dp(p0),
                                              _lib.dp(p1), _lib.ip(paths)))
        return [paths[self.offsets[k]:self.offsets[k + 1]] for k in range(len(self.lengths))]

    def sample_paths(self, A, pi, par0=None, par1=None, u=None, seed=0, want_paths=True):
        """Gibbs hidden-path step.  Returns (paths or None, C int64 (n,n), n0 int64 (n,), emis)."""
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        n = self.nstates
        total = int(self.offsets[-1])
        paths = np.empty(total, dtype=np.int32) if want_paths else None
        C = np.zeros((n, n), dtype=np.int64)
        n0 = np.zeros(n, dtype=np.int64)
        if self.kind == 'gaussian':
            emis = np.zeros((3, n))
        elif self.kind == 'discrete':
            emis = np.zeros((n, self.nsymbols))
        else:
            emis = None
        uu = _lib.f64(np.concatenate(u)) if u is not None else None
        _lib.check(self._L.bhmm_sample_paths(self._h, _lib.dp(A), _lib.dp(pi), _lib.dp(p0),
                                             _lib.dp(p1), _lib.dp(uu), ctypes.c_uint64(int(seed)),
                                             _lib.ip(paths), _lib.lp(C), _lib.lp(n0),
                                             _lib.dp(emis)))
        plist = None
        if want_paths:
            plist = [paths[self.offsets[k]:self.offsets[k + 1]] for k in range(len(self.lengths))]
        return plist, C, n0, emis
