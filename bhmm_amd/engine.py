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
        self.device = int(device)
        self.kind = None
        self._stage = None
        self._keep = {}
        self._fetch = None
        self._kms = None
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

    def set_observations_lagged(self, kind, observations, lag, views, nstates, nsymbols=0, chunk=0):
        """Lagged views (bhmm/api.py:70-94) cut on the device: `observations` are the ORIGINAL
        trajectories (uploaded once), `views` a list of (trajectory index, shift); context
        trajectory v becomes observations[k][shift::lag]."""
        code = _KINDS[kind]
        lens = np.array([len(o) for o in observations], dtype=np.int64)
        off = np.zeros(len(observations) + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        if kind == 'gaussian':
            flat = np.ascontiguousarray(np.concatenate([np.asarray(o, dtype=np.float64)
                                                        for o in observations]))
        elif kind == 'discrete':
            flat = np.ascontiguousarray(np.concatenate([np.asarray(o).astype(np.int32)
                                                        for o in observations]))
            # only what the views touch has to be a symbol (stride > 1 skips shifts)
            for k, s in views:
                piece = np.asarray(observations[k])[int(s)::int(lag)]
                if piece.size and (piece.min() < 0 or piece.max() >= nsymbols):
                    raise ValueError("discrete observation outside [0, nsymbols)")
        else:
            flat = np.ascontiguousarray(np.concatenate(
                [np.asarray(o, dtype=np.float64).reshape(-1, nstates) for o in observations]))
        vt = np.ascontiguousarray([k for k, _ in views], dtype=np.int32)
        vs = np.ascontiguousarray([s for _, s in views], dtype=np.int32)
        _lib.check(self._L.bhmm_ctx_set_observations_lagged(
            self._h, code, flat.ctypes.data_as(ctypes.c_void_p), _lib.lp(off), len(observations),
            int(lag), _lib.ip(vt), _lib.ip(vs), len(views), int(nstates), int(nsymbols), int(chunk), 0))
        vlen = np.array([max(0, -(-(int(lens[k]) - int(s)) // int(lag))) for k, s in views],
                        dtype=np.int64)
        self._adopt(kind, nstates, nsymbols, vlen)

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
        self._stage = None
        self._fetch = None

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

    def kernel_ms_all(self):
        """The five HIP-event intervals of the last E-step (prescan, stitch, sweep, finalise, whole)
        in an array owned by the engine (overwritten by the next call)."""
        k = self._kms
        if k is None:
            buf = np.zeros(5)
            k = self._kms = (buf, _lib.dp(buf))
        _lib.check(self._L.bhmm_ctx_last_kernel_ms_all(self._h, k[1]))
        return k[0]

    # -- E-step --------------------------------------------------------------------------
    _STAGE_LIMIT = 1 << 16   # elements: larger emission tables are passed as they are

    def _model_ptrs(self, A, pi, par0, par1):
        """ctypes pointers (A, pi, par0, par1) for one call.  The model is copied into arrays owned
        by the engine whose pointers are made once per set of observations: building four ctypes
        pointers from numpy arrays costs ~13 us per call, a sixtieth of an E-step of configs[1]."""
        n = self.nstates
        if np.shape(A) != (n, n) or np.shape(pi) != (n,):
            raise ValueError("model shape does not match nstates=%d" % n)
        st = self._stage
        if st is None:
            st = self._stage = {}
        out = []
        for key, a in (("A", A), ("pi", pi), ("p0", par0), ("p1", par1)):
            if a is None:
                out.append(None)
                continue
            shp = np.shape(a)
            ent = st.get(key)
            if ent is None or ent[0].shape != shp:
                if int(np.prod(shp, dtype=np.int64)) > self._STAGE_LIMIT:
                    a = _lib.f64(a)
                    self._keep[key] = a             # alive until the call returns
                    out.append(_lib.dp(a))
                    continue
                buf = np.empty(shp, dtype=np.float64)
                ent = st[key] = (buf, _lib.dp(buf))
            np.copyto(ent[0], a)
            out.append(ent[1])
        return out

    def estep_launch(self, A, pi, par0=None, par1=None, stats_dev=None, store_gamma=False):
        """Enqueue one E-step.  stats_dev: optional device address receiving the packed
        statistics (e.g. a torch tensor that is all-reduced across ranks afterwards)."""
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        flags = _lib.FLAG_STORE_GAMMA if store_gamma else 0
        _lib.check(self._L.bhmm_estep(self._h, A, pi, p0, p1,
                                      ctypes.c_void_p(int(stats_dev)) if stats_dev else None,
                                      flags))

    def estep_fetch(self):
        S = self.stats_size
        packed = np.empty(S)
        logL_k = np.empty(len(self.lengths))
        _lib.check(self._L.bhmm_estep_fetch(self._h, _lib.dp(packed), _lib.dp(logL_k)))
        return EStepResult(self.kind, self.nstates, self.nsymbols, packed, logL_k)

    def estep_fetch_packed(self, out=None):
        """Wait for the last E-step and return only its packed statistics vector (layout:
        include/bhmm_amd.h, bhmm_ctx_stats_size) -- no per-field copies, no logL_k."""
        if out is None:
            out = np.empty(self.stats_size)
        f = self._fetch
        if f is None or f[0] is not out:            # the pointer of a re-used `out` is made once
            f = self._fetch = (out, _lib.dp(out))
        _lib.check(self._L.bhmm_estep_fetch(self._h, f[1], None))
        return out

    def estep_fetch_logL(self):
        """Per-trajectory log-likelihoods of the last E-step only (waits for it).  Used when the
        packed statistics stay on the device for an all-reduce."""
        logL_k = np.empty(len(self.lengths))
        _lib.check(self._L.bhmm_estep_fetch(self._h, None, _lib.dp(logL_k)))
        return logL_k

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
        _lib.check(self._L.bhmm_viterbi_batch(self._h, A, pi, p0, p1, _lib.ip(paths)))
        return [paths[self.offsets[k]:self.offsets[k + 1]] for k in range(len(self.lengths))]

    def viterbi_u8(self, A, pi, par0=None, par1=None, out=None):
        """Viterbi paths as one byte per step.  out: None (a numpy uint8 array is allocated), a
        numpy uint8 array, or any object with data_ptr()/is_cuda (a torch uint8 tensor: on this
        engine's GPU the kernels write it directly, pinned host memory is copied at link rate).
        Returns `out` (concatenated over trajectories like the observations)."""
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        total = int(self.offsets[-1])
        on_dev = 0
        if out is None:
            out = np.empty(total, dtype=np.uint8)
        if isinstance(out, np.ndarray):
            if out.dtype != np.uint8 or out.size < total or not out.flags.c_contiguous:
                raise ValueError("out must be a contiguous uint8 array of sum(T_k) elements")
            ptr = out.ctypes.data
        else:
            if out.numel() < total or out.element_size() != 1 or not out.is_contiguous():
                raise ValueError("out must be a contiguous uint8 tensor of sum(T_k) elements")
            on_dev = 1 if out.is_cuda else 0
            if on_dev and out.device.index != self.device:
                raise ValueError("out lives on another GPU than this engine")
            ptr = out.data_ptr()
        _lib.check(self._L.bhmm_viterbi_batch_u8(self._h, A, pi, p0, p1, ctypes.c_void_p(int(ptr)), on_dev))
        return out

    def set_stream_offsets(self, soff):
        """Position of each loaded trajectory in the device random stream (include/bhmm_amd.h):
        a sharded caller passes the offsets in the unsharded concatenation, None resets."""
        if soff is None:
            _lib.check(self._L.bhmm_ctx_set_stream_offsets(self._h, None))
            return
        soff = np.ascontiguousarray(soff, dtype=np.int64)
        if soff.shape != (len(self.lengths),):
            raise ValueError("one stream offset per loaded trajectory")
        _lib.check(self._L.bhmm_ctx_set_stream_offsets(self._h, _lib.lp(soff)))

    @property
    def path_stats_size(self):
        return self._L.bhmm_ctx_path_stats_size(self._h)

    def sample_paths_dev(self, A, pi, par0, par1, stats_dev, u=None, seed=0, want_paths=False):
        """Gibbs hidden-path step with the packed path statistics left in the device buffer
        `stats_dev` (integer address, path_stats_size doubles).  Returns paths or None."""
        A, pi, p0, p1 = self._model_ptrs(A, pi, par0, par1)
        total = int(self.offsets[-1])
        paths = np.empty(total, dtype=np.int32) if want_paths else None
        uu = _lib.f64(np.concatenate(u)) if u is not None else None
        _lib.check(self._L.bhmm_sample_paths_dev(self._h, A, pi, p0, p1, _lib.dp(uu), ctypes.c_uint64(int(seed)),
                                                 _lib.ip(paths), ctypes.c_void_p(int(stats_dev))))
        if not want_paths:
            return None
        return [paths[self.offsets[k]:self.offsets[k + 1]] for k in range(len(self.lengths))]

    def unpack_path_stats(self, packed):
        """[counts n*n | n0 n | emission block] -> (C int64 (n,n), n0 int64 (n,), emis)."""
        n, M = self.nstates, self.nsymbols
        packed = np.asarray(packed, dtype=np.float64)
        C = np.rint(packed[:n * n]).astype(np.int64).reshape(n, n)
        n0 = np.rint(packed[n * n:n * n + n]).astype(np.int64)
        rest = packed[n * n + n:]
        if self.kind == 'gaussian':
            emis = rest[:3 * n].reshape(3, n).copy()
        elif self.kind == 'discrete':
            emis = rest[:n * M].reshape(n, M).copy()
        else:
            emis = None
        return C, n0, emis

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
        _lib.check(self._L.bhmm_sample_paths(self._h, A, pi, p0, p1, _lib.dp(uu), ctypes.c_uint64(int(seed)),
                                             _lib.ip(paths), _lib.lp(C), _lib.lp(n0),
                                             _lib.dp(emis)))
        plist = None
        if want_paths:
            plist = [paths[self.offsets[k]:self.offsets[k + 1]] for k in range(len(self.lengths))]
        return plist, C, n0, emis


class NativeComm(object):
    """The library's own communicator (include/bhmm_amd.h, section 2b): RCCL behind the C ABI, for
    callers that do not bring torch.distributed -- one all-reduce of the packed statistics per EM
    iteration / Gibbs sweep (maximum_likelihood.py:271-282 in distributed form).  Rank 0 calls
    NativeComm.unique_id() and hands the 128 bytes to the other ranks; every rank then constructs
    NativeComm(device, nranks, rank, uid) (a collective)."""

    def __init__(self, device, nranks=1, rank=0, uid=None):
        self._L = _lib.load()
        if uid is None:
            if nranks != 1:
                raise ValueError("more than one rank: pass the unique id of rank 0")
            uid = NativeComm.unique_id()
        self._uid = ctypes.create_string_buffer(bytes(uid), 128)
        self._h = ctypes.c_void_p()
        _lib.check(self._L.bhmm_comm_init_rank(ctypes.byref(self._h), int(device), int(nranks), int(rank),
                                               ctypes.cast(self._uid, ctypes.c_void_p)))
        self.nranks, self.rank, self.device = int(nranks), int(rank), int(device)

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().bhmm_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return buf.raw

    def allreduce_stats(self, engine, dev_ptr, count):
        """In-place sum over the ranks of `count` doubles at device address dev_ptr, enqueued on the
        engine's stream (no host synchronisation)."""
        _lib.check(self._L.bhmm_ctx_allreduce_stats(engine._h, self._h, ctypes.c_void_p(int(dev_ptr)),
                                                    ctypes.c_int64(int(count))))

    def close(self):
        if self._h:
            self._L.bhmm_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_observations(kind, obs_dev, A, pi, par0, par1, K, T, seed, device=0, stream=None,
                       states_dev=None, first_traj=0):
    """Draw K synthetic trajectories of T steps on the GPU into the device buffer at address
    `obs_dev` (K*T doubles for 'gaussian', int32 for 'discrete'); see bhmm_synth_observations in
    include/bhmm_amd.h.  states_dev: optional device address of K*T bytes for the hidden paths.
    first_traj: index of this call's first trajectory in a larger (sharded) set -- the slice is then
    exactly what one call for the whole set would have drawn for these trajectories."""
    L = _lib.load()
    _lib.require_device()
    A = _lib.f64(A)
    n = A.shape[0]
    p0 = _lib.f64(par0)
    p1 = _lib.f64(par1) if par1 is not None else None
    M = p0.shape[1] if kind == 'discrete' else 0
    _lib.check(L.bhmm_synth_observations_at(
        ctypes.c_void_p(int(obs_dev)), ctypes.c_void_p(int(states_dev)) if states_dev else None,
        int(device), ctypes.c_void_p(stream) if stream else None, _KINDS[kind], _lib.dp(A),
        _lib.dp(_lib.f64(pi)), _lib.dp(p0), _lib.dp(p1), int(n), int(M), int(K), int(T),
        ctypes.c_uint64(int(seed)), int(first_traj)))
