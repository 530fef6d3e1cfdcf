"""Test double for bhmm_amd.engine.Engine backed by the CPU oracle (TEST INFRASTRUCTURE).

Lets the host-side logic of the estimators (EM loop semantics, M-step, sharding and
all-reduce) run in the GPU-less CI container.  It is injected through the estimators'
`engine_factory` argument and is never importable from the product package.
"""
import numpy as np

from bhmm_amd.engine import EStepResult
from oracle import oracle as orc


def device_uniforms(seed, start, count):
    """numpy restatement of the engine's counter-based stream (path_kernels.hpp uniform01):
    u[t] = splitmix64-finaliser(seed + golden * (start + t + 1)) >> 11, scaled to [0, 1)."""
    with np.errstate(over='ignore'):
        x = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


class OracleEngine(object):
    def __init__(self, device=0):
        self.device = device
        self.soff = None

    def set_stream_offsets(self, soff):
        self.soff = None if soff is None else np.asarray(soff, dtype=np.int64)

    def set_observations(self, kind, observations, nstates, nsymbols=0, chunk=0):
        self.kind, self.obs, self.n, self.M = kind, [np.asarray(o) for o in observations], nstates, nsymbols
        self.lengths = np.array([len(o) for o in observations], dtype=np.int64)
        self.soff = None
        self._gammas = None

    def _pobs(self, o, par0, par1):
        if self.kind == 'gaussian':
            return orc.pobs_gaussian(o, par0, par1)
        return orc.pobs_discrete(o, par0)

    def estep(self, A, pi, par0=None, par1=None, store_gamma=False):
        n = self.n
        if len(self.obs) == 0:
            S = 1 + n + n * n + n + (2 * n if self.kind == 'gaussian' else n * self.M)
            return EStepResult(self.kind, n, self.M, np.zeros(S), np.zeros(0))
        r = orc.estep(self.kind, self.obs, A, pi, par0, par1, want_gamma=True)
        self._gammas = r['gammas']
        parts = [[r['logL'].sum()], r['gamma0_sum'], r['C'].ravel(), r['state_counts']]
        if self.kind == 'gaussian':
            mu = np.asarray(par0)
            sd = sum((g * (o[:, None] - mu[None, :])).sum(axis=0) for o, g in zip(self.obs, r['gammas']))
            sdd = sum((g * (o[:, None] - mu[None, :]) ** 2).sum(axis=0)
                      for o, g in zip(self.obs, r['gammas']))
            parts += [sd, sdd]
        else:
            cnt = np.zeros((n, self.M))
            for o, g in zip(self.obs, r['gammas']):
                orc.update_pout(o, g, cnt)
            parts.append(cnt.ravel())
        return EStepResult(self.kind, n, self.M, np.concatenate([np.ravel(p) for p in parts]),
                           r['logL'])

    def unpack(self, packed, logL_k=None):
        return EStepResult(self.kind, self.n, self.M, np.asarray(packed), logL_k)

    def gamma(self, k):
        return self._gammas[k]

    def viterbi(self, A, pi, par0=None, par1=None):
        return [orc.viterbi(A, self._pobs(o, par0, par1), pi) for o in self.obs]

    def sample_paths(self, A, pi, par0=None, par1=None, u=None, seed=0, want_paths=True):
        n = self.n
        paths = []
        soff = self.soff if self.soff is not None else np.concatenate([[0], np.cumsum(self.lengths)])
        for k, o in enumerate(self.obs):
            _, alpha = orc.forward(A, self._pobs(o, par0, par1), pi)
            uu = u[k] if u is not None else device_uniforms(seed, int(soff[k]), len(o))
            paths.append(orc.sample_path(alpha, A, u=uu))
        C, n0 = orc.path_counts(paths, n) if paths else (np.zeros((n, n), np.int64), np.zeros(n, np.int64))
        if self.kind == 'gaussian':
            emis = np.zeros((3, n))
            for p, o in zip(paths, self.obs):
                for i in range(n):
                    d = o[p == i] - par0[i]
                    emis[0, i] += d.size
                    emis[1, i] += d.sum()
                    emis[2, i] += (d * d).sum()
        else:
            emis = np.zeros((n, self.M))
            for p, o in zip(paths, self.obs):
                np.add.at(emis, (p, o), 1.0)
        return (paths if want_paths else None), C, n0, emis
