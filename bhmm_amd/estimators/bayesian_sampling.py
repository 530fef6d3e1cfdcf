"""Gibbs sampler driver with the interface of bhmm/estimators/bayesian_sampling.py:31-373.

The hot step of every sweep -- sampling all hidden paths (forward pass + backward sampling per
trajectory, :283-331) and collecting their statistics (generic_hmm.py:297-334,398-431) -- is
one call into the device engine, which returns the integer transition / start counts and the
per-state emission statistics.  The parameter updates that follow (emission parameters, p0,
transition matrix; :333-373) are ONE native host call (bhmm_gibbs_parameters, host code of the
shared library) driven by a counter-based generator: given the all-reduced statistics every rank
draws the SAME parameters, so a sharded chain needs no broadcast.  The reversible
transition-matrix sampler lives in msmtools in the reference; it is restated from the
publication (Trendelkamp-Schroer et al. 2015) and is "parity unpinned" against msmtools itself.
`native_parameters=False` selects the numpy draws (np.random stream, rank 0 draws + broadcast).
"""
import copy

import numpy as np

from .. import hidden
from ..sharding import Comm, lpt_partition
from ..util import config
from . import _tmatrix


def _default_engine_factory(device):
    from ..engine import Engine
    return Engine(device)


def _unpack_path_stats(packed, kind, n, M):
    """[counts n*n | n0 n | emission block] (include/bhmm_amd.h, bhmm_sample_paths_dev)."""
    packed = np.asarray(packed, dtype=np.float64)
    C = np.rint(packed[:n * n]).astype(np.int64).reshape(n, n)
    n0 = np.rint(packed[n * n:n * n + n]).astype(np.int64)
    rest = packed[n * n + n:]
    if kind == 'gaussian':
        return C, n0, rest[:3 * n].reshape(3, n).copy()
    if kind == 'discrete':
        return C, n0, rest[:n * M].reshape(n, M).copy()
    return C, n0, None


class BayesianHMMSampler(object):
    def __init__(self, observations, nstates, initial_model=None, reversible=True,
                 stationary=False, transition_matrix_sampling_steps=1000, p0_prior='mixed',
                 transition_matrix_prior='mixed', output='gaussian', device=None,
                 process_group=None, engine_factory=None, native_parameters=True):
        if len(observations) == 0:
            raise Exception("No observations were provided.")
        self.reversible = reversible
        self.stationary = stationary
        self.nstates = nstates
        self.observations = copy.deepcopy(observations)
        self.nobs = len(observations)
        self.Ts = [len(o) for o in observations]
        self.maxT = np.max(self.Ts)
        if not initial_model:
            initial_model = self._generateInitialModel(output, device, process_group, engine_factory)
        self.model = copy.deepcopy(initial_model)
        self._output = self.model.output_model.model_type

        # priors, bayesian_sampling.py:158-184
        if p0_prior is None or (isinstance(p0_prior, str) and p0_prior == 'sparse'):
            self.prior_n0 = np.zeros(self.nstates)
        elif isinstance(p0_prior, np.ndarray):
            if p0_prior.ndim == 1 and p0_prior.shape[0] == self.nstates:
                self.prior_n0 = np.array(p0_prior)
            else:
                raise ValueError('initial distribution prior must have dimension ' + str(nstates))
        elif p0_prior == 'mixed':
            self.prior_n0 = np.array(self.model.initial_distribution)
        elif p0_prior == 'uniform':
            self.prior_n0 = np.ones(nstates)
        else:
            raise ValueError('initial distribution prior mode undefined: ' + str(p0_prior))
        if transition_matrix_prior is None or (isinstance(p0_prior, str) and p0_prior == 'sparse'):
            self.prior_C = np.zeros((self.nstates, self.nstates))
        elif isinstance(transition_matrix_prior, np.ndarray):
            if np.array_equal(transition_matrix_prior.shape, (self.nstates, self.nstates)):
                self.prior_C = np.array(transition_matrix_prior)
            else:
                raise ValueError('transition matrix prior must have shape (n, n)')
        elif transition_matrix_prior == 'mixed':
            self.prior_C = np.array(self.model.transition_matrix)
        elif transition_matrix_prior == 'uniform':
            self.prior_C = np.ones((nstates, nstates))
        else:
            raise ValueError('transition matrix prior mode undefined: '
                             + str(transition_matrix_prior))
        if reversible:
            if not _tmatrix.is_connected(self.model.transition_matrix + self.prior_C, strong=True):
                raise NotImplementedError('Trying to sample disconnected HMM with option '
                                          'reversible:\n ' + str(self.model.transition_matrix)
                                          + '\nUse prior to connect, select connected subset, '
                                          'or use reversible=False.')
        self.transition_matrix_sampling_steps = transition_matrix_sampling_steps
        hidden.set_implementation(config.kernel)
        self.model.output_model.set_implementation(config.kernel)

        self._comm = Comm(process_group)
        self._parts = lpt_partition(self.Ts, self._comm.world)
        self._mine = self._parts[self._comm.rank]
        if device is None:
            from .maximum_likelihood import default_device
            device = default_device(self._comm)
        self._comm.bind_device(device)       # collectives run on the engine's GPU
        factory = engine_factory or _default_engine_factory
        self._engine = factory(device)
        M = self.model.output_model.nsymbols if self._output == 'discrete' else 0
        self._nsymbols = M
        if self._mine:
            from .maximum_likelihood import _load_observations
            _load_observations(self._engine, self._output, observations, self.observations,
                               self._mine, self._comm, nstates, M)
            if self._comm.active:
                # uniforms are addressed by the position in the UNSHARDED concatenation of all
                # trajectories: the sampled paths do not depend on the partition over ranks
                goff = np.concatenate([[0], np.cumsum(self.Ts)]).astype(np.int64)
                self._engine.set_stream_offsets(goff[self._mine])
        self._sweep = 0
        self._rng = np.random
        self._native = bool(native_parameters)
        self._param_seed = None
        self._draw = None

    def _generateInitialModel(self, output_model_type, device=None, process_group=None,
                              engine_factory=None):
        """bayesian_sampling.py:375-385: a maximum-likelihood fit from the heuristic start."""
        from .maximum_likelihood import MaximumLikelihoodEstimator
        mlhmm = MaximumLikelihoodEstimator(self.observations, self.nstates,
                                           reversible=self.reversible, output=output_model_type,
                                           device=device, process_group=process_group,
                                           engine_factory=engine_factory)
        return mlhmm.fit()

    def sample(self, nsamples, nburn=0, nthin=1, save_hidden_state_trajectory=False,
               call_back=None, seed=None):
        """bayesian_sampling.py:206-267."""
        first = True
        for _ in range(nburn):
            self._update(seed=seed if first else None, keep_paths=False)
            first = False
        models = list()
        for _ in range(nsamples):
            for _thin in range(nthin):
                self._update(seed=seed if first else None,
                             keep_paths=save_hidden_state_trajectory)
                first = False
            model_copy = copy.deepcopy(self.model)
            if not save_hidden_state_trajectory:
                model_copy.hidden_state_trajectories = None
            models.append(model_copy)
            if call_back is not None:
                call_back()
        return models

    def _update(self, seed=None, keep_paths=False):
        """One Gibbs sweep, bayesian_sampling.py:269-281.  With several ranks the hidden-path
        step runs sharded and its statistics are all-reduced; the parameter draws are then a
        deterministic function of (statistics, seed, sweep) that every rank evaluates itself
        (native path), or are drawn on rank 0 and broadcast (numpy path)."""
        packed = self._updateHiddenStateTrajectories(seed=seed, keep_paths=keep_paths)
        if self._native:
            self._update_parameters_native(packed, seed)
            return
        C, n0, emis = _unpack_path_stats(packed, self._output, self.nstates, self._nsymbols)
        if not self._comm.active:
            self._updateEmissionProbabilities(emis)
            self._updateTransitionMatrix(C, n0)
            return
        err = None
        if self._comm.rank == 0:
            try:
                self._updateEmissionProbabilities(emis)
                self._updateTransitionMatrix(C, n0)
            except Exception as e:      # the other ranks wait in the broadcast: tell them
                err = e
        self._broadcast_parameters(failed=err is not None)
        if err is not None:
            raise err

    def _update_parameters_native(self, packed, seed):
        """bayesian_sampling.py:333-373 as one call of bhmm_gibbs_parameters."""
        if self._param_seed is None or seed is not None:
            if seed is not None:
                base = int(seed)
            else:
                # no seed given: one draw from numpy's global generator (np.random.seed makes the
                # whole chain reproducible), identical on every rank
                base = int(self._rng.randint(0, 2 ** 31 - 1)) * 2 ** 31 + int(self._rng.randint(0, 2 ** 31 - 1))
                base = int(self._comm.broadcast_numpy(np.array([base], dtype=np.int64), src=0)[0])
            self._param_seed = (base * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFFFFFFFFFF
        if self._draw is None:
            from ._native import GibbsParameters
            om = self.model.output_model
            self._draw = GibbsParameters(
                self._output, self.nstates, self._nsymbols, prior_C=self.prior_C,
                prior_n0=self.prior_n0, prior_B=getattr(om, 'prior', None),
                reversible=self.reversible, stationary=self.stationary,
                nsteps=self.transition_matrix_sampling_steps)
        om = self.model.output_model
        par0, par1 = om.parameters()
        Tij, p0, new0, new1 = self._draw(packed, par0, par1, self._param_seed, self._sweep)
        om.set_parameters(new0, new1)
        self.model.update(p0, Tij)

    def _broadcast_parameters(self, failed=False):
        """[ok | T (n*n) | p0 (n) | emission parameters] from rank 0 to every rank."""
        n = self.nstates
        om = self.model.output_model
        par0, par1 = om.parameters()
        parts = [np.array([0.0 if failed else 1.0]), np.ravel(self.model.transition_matrix),
                 np.ravel(self.model.initial_distribution), np.ravel(par0)]
        if par1 is not None:
            parts.append(np.ravel(par1))
        vec = self._comm.broadcast_numpy(np.concatenate(parts).astype(np.float64), src=0)
        if vec[0] != 1.0:
            if self._comm.rank != 0:
                raise RuntimeError('parameter update failed on rank 0 (see its exception)')
            return
        if self._comm.rank != 0:
            o = 1
            Tij = vec[o:o + n * n].reshape(n, n); o += n * n
            p0 = vec[o:o + n]; o += n
            k0 = np.size(par0)
            new0 = vec[o:o + k0].reshape(np.shape(par0)); o += k0
            new1 = vec[o:o + np.size(par1)].reshape(np.shape(par1)) if par1 is not None else None
            om.set_parameters(new0, new1)
            self.model.update(p0, Tij)

    def _updateHiddenStateTrajectories(self, seed=None, keep_paths=False):
        """bayesian_sampling.py:283-331 for all trajectories at once.  Returns the hidden-path
        statistics the two parameter updates need as ONE packed fp64 vector
        [C n*n | n0 n | emission block] (summed over ranks)."""
        if seed is not None:
            self._seed_base = int(seed)
        base = getattr(self, '_seed_base', 0x5EED)
        sweep_seed = (base * 1000003 + self._sweep * 7919) & 0xFFFFFFFFFFFF   # same on every rank
        self._sweep += 1
        om = self.model.output_model
        par0, par1 = om.parameters()
        eng, comm = self._engine, self._comm
        A, pi = self.model.transition_matrix, self.model.initial_distribution
        n, M = self.nstates, self._nsymbols
        esz = 3 * n if self._output == 'gaussian' else (n * M if self._output == 'discrete' else 0)
        paths = []

        def pack(C, n0, emis):
            return np.concatenate([np.ravel(C).astype(np.float64), np.ravel(n0).astype(np.float64)]
                                  + ([np.ravel(emis)] if esz else []))

        if not comm.active:
            paths, C, n0, emis = eng.sample_paths(A, pi, par0, par1, seed=sweep_seed,
                                                  want_paths=keep_paths)
            packed = pack(C, n0, emis)
        elif hasattr(eng, 'sample_paths_dev'):
            # ONE packed fp64 vector [C | n0 | emission block] stays on the engine's GPU, ONE
            # all-reduce, ONE copy to the host; the integer counts are exact in fp64 (< 2^53)
            buf = comm.stats_buffer(n * n + n + esz)
            if self._mine:
                paths = eng.sample_paths_dev(A, pi, par0, par1, buf.data_ptr(), seed=sweep_seed,
                                             want_paths=keep_paths)
            else:
                buf.zero_()
            packed = comm.allreduce_stats(buf)
        else:
            # host-side engine (CPU test double): same packed vector, host all-reduce
            packed = np.zeros(n * n + n + esz)
            if self._mine:
                paths, C, n0, emis = eng.sample_paths(A, pi, par0, par1, seed=sweep_seed,
                                                      want_paths=keep_paths)
                packed = pack(C, n0, emis)
            packed = comm.allreduce_sum_numpy(packed)
        if keep_paths:
            full = [None] * self.nobs
            if self._comm.active:
                for part, plist in zip(self._parts, self._comm.gather_objects(paths)):
                    for k, pth in zip(part, plist):
                        full[k] = pth
            else:
                for k, pth in zip(self._mine, paths):
                    full[k] = pth
            self.model.hidden_state_trajectories = full
        else:
            self.model.hidden_state_trajectories = None
        return np.asarray(packed, dtype=np.float64)

    def _updateEmissionProbabilities(self, emis):
        """bayesian_sampling.py:333-339 from per-state statistics instead of gathered arrays."""
        om = self.model.output_model
        if self._output == 'gaussian':
            om.sample_from_statistics(emis[0], emis[1], emis[2], rng=self._rng)
        else:
            om.sample_from_statistics(emis, rng=self._rng)

    def _updateTransitionMatrix(self, Cint, n0int):
        """bayesian_sampling.py:341-373."""
        C = Cint.astype(np.float64) + self.prior_C
        if self.reversible and not _tmatrix.is_connected(C, strong=True):
            raise NotImplementedError('Encountered disconnected count matrix with sampling '
                                      'option reversible:\n ' + str(C) + '\nUse prior to ensure '
                                      'connectivity or use reversible=False.')
        if self.reversible:
            P0 = _tmatrix.mle_reversible(C, maxiter=10000)
            C = C.copy()
            C[np.where(P0 + P0.T == 0)] = 0
            Tij = _tmatrix.sample_reversible(C, nsteps=self.transition_matrix_sampling_steps,
                                             P0=P0, rng=self._rng)
        else:
            Tij = _tmatrix.sample_nonreversible(C, rng=self._rng)
        if self.stationary:
            p0 = _tmatrix.stationary_distribution(Tij, C=C)
        else:
            n0 = n0int.astype(float)
            w = n0 + self.prior_n0
            positive = w > 0
            p0 = np.zeros_like(n0)
            p0[positive] = self._rng.dirichlet(w[positive])
        self.model.update(p0, Tij)
