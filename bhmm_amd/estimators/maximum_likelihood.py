"""Baum-Welch EM driver with the interface of bhmm/estimators/maximum_likelihood.py:36-446.

Same constructor, `fit()`, properties, convergence rule and final Viterbi pass as the
reference; what changes is where the loop over trajectories runs.  The reference walks the
trajectories in Python and calls five kernels per trajectory (:221-269, :383-385); here the
whole E-step is one call into the device-resident engine (bhmm_amd/engine.py), which returns
only the reduced sufficient statistics.  With torch.distributed initialised, trajectories are
sharded over ranks and the statistics are all-reduced (bhmm_amd/sharding.py); every rank then
performs the identical O(N^2) M-step.
"""
import copy
import time

import numpy as np

from .. import hidden
from ..engine import EStepResult
from ..sharding import Comm, lpt_partition
from ..util import config
from . import _tmatrix


def _default_engine_factory(device):
    from ..engine import Engine
    return Engine(device)


def default_device(comm):
    """One process per GPU: rank r drives GPU r (modulo the GPUs visible to this process).
    Counting devices does not initialise the GPU."""
    if not comm.active:
        return 0
    try:
        import torch
        ndev = torch.cuda.device_count()
    except ImportError:
        ndev = 0
    return comm.rank % ndev if ndev > 0 else comm.rank


def _views_intact(lagged):
    """True while every entry of a LaggedObservations list still IS the recorded view
    base[k][shift::lag] (same memory, length, stride, dtype).  A list is mutable: after
    `obs[i] = ...`, a sort or a remove/append the recipe no longer describes the data, and the
    estimators must upload the entries themselves."""
    try:
        for entry, (k, shift) in zip(lagged, lagged.views):
            want = lagged.base[k][shift::lagged.lag]
            if not isinstance(entry, np.ndarray) or entry.shape != want.shape \
                    or entry.dtype != want.dtype or entry.strides != want.strides \
                    or entry.__array_interface__['data'][0] != want.__array_interface__['data'][0]:
                return False
    except (AttributeError, IndexError, TypeError):
        return False
    return True


def _load_observations(engine, kind, given, copied, mine, comm, nstates, nsymbols):
    """Hand this rank's trajectories to the engine.  Lagged views made by lag_observations
    (bhmm/api.py:70-94) are not uploaded piece by piece: the original trajectories go up once and
    the views are cut on the GPU (bhmm_ctx_set_observations_lagged) -- single process only, a
    sharded run uploads each rank's own pieces."""
    views = getattr(given, 'views', None)
    if (views is not None and not comm.active and hasattr(engine, 'set_observations_lagged')
            and len(views) == len(copied) and _views_intact(given)):
        engine.set_observations_lagged(kind, given.base, given.lag, views, nstates,
                                       nsymbols=nsymbols)
    else:
        engine.set_observations(kind, [copied[k] for k in mine], nstates, nsymbols=nsymbols)


def packed_stats_size(kind, n, M):
    """Length of the packed E-step statistics vector (include/bhmm_amd.h, bhmm_ctx_stats_size)."""
    return 1 + n + n * n + n + (2 * n if kind == 'gaussian' else (n * M if kind == 'discrete' else 0))


class MaximumLikelihoodEstimator(object):
    def __init__(self, observations, nstates, initial_model=None, output='gaussian',
                 reversible=True, stationary=False, p=None, accuracy=1e-3, maxit=1000,
                 maxit_P=100000, device=None, process_group=None, store_gamma=False,
                 engine_factory=None, multi_start=False):
        # maximum_likelihood.py:101-104
        self._observations = copy.deepcopy(observations)
        self._nobs = len(observations)
        self._Ts = [len(o) for o in observations]
        self._maxT = np.max(self._Ts)
        self._nstates = nstates
        self._reversible = reversible
        self._stationary = stationary
        self._alternative_starts = []
        if initial_model is None:
            # maximum_likelihood.py:112-118 -> bhmm.init_hmm.  Gaussian: mixture fit +
            # fractional counts (bhmm_amd/init/gaussian.py); discrete: count matrix + PCCA+
            # (bhmm_amd/init/discrete.py).
            from .. import api as _api
            initial_model = _api.init_hmm(observations, nstates, output=output,
                                          reversible=reversible)
            if multi_start and output == 'gaussian' and nstates > 1:
                # opt-in extension (the reference starts from init_hmm alone, :111-116): EM only
                # finds the optimum next to its start and an E-step costs milliseconds here, so a
                # second, kinetic start is tried for a few iterations in fit() and the better one
                # continues (init/gaussian.py: init_model_gaussian1d_kinetic)
                from ..init.gaussian import init_model_gaussian1d_kinetic
                alt = init_model_gaussian1d_kinetic(observations, nstates, reversible=reversible)
                if alt is not None:
                    self._alternative_starts.append(alt)
        self._hmm = copy.deepcopy(initial_model)
        if self._hmm.nstates != nstates:
            raise ValueError('initial_model has %d states, nstates=%d' % (self._hmm.nstates, nstates))
        self._output = self._hmm.output_model.model_type
        self._fixed_stationary_distribution = None
        self._fixed_initial_distribution = None
        if p is not None:
            if stationary:
                self._fixed_stationary_distribution = np.array(p)
            else:
                self._fixed_initial_distribution = np.array(p)
        self._accuracy = accuracy
        self._maxit = maxit
        self._maxit_P = maxit_P
        self._likelihoods = None
        self._store_gamma = store_gamma
        self._gammas = None
        self._last = None
        self._mstep = None
        hidden.set_implementation(config.kernel)
        self._hmm.output_model.set_implementation(config.kernel)

        # ---- device-resident batch (replaces the shared alpha/beta/pobs buffers, :128-133) ----
        self._comm = Comm(process_group)
        self._parts = lpt_partition(self._Ts, self._comm.world)
        self._mine = self._parts[self._comm.rank]
        if device is None:
            device = default_device(self._comm)
        self._device = device
        self._comm.bind_device(device)       # collectives run on the engine's GPU, not on "current"
        factory = engine_factory or _default_engine_factory
        self._engine = factory(device)
        M = self._hmm.output_model.nsymbols if self._output == 'discrete' else 0
        self._nsymbols = M
        if self._mine:
            _load_observations(self._engine, self._output, observations, self._observations,
                               self._mine, self._comm, nstates, M)

    # ---- properties (maximum_likelihood.py:145-219) -------------------------------------
    @property
    def observations(self):
        return self._observations

    @property
    def nobservations(self):
        return self._nobs

    @property
    def observation_lengths(self):
        return self._Ts

    @property
    def is_reversible(self):
        return self._reversible

    @property
    def nstates(self):
        return self._nstates

    @property
    def accuracy(self):
        return self._accuracy

    @property
    def maxit(self):
        return self._maxit

    @property
    def likelihood(self):
        return self._likelihoods[-1]

    @property
    def likelihoods(self):
        return self._likelihoods

    @property
    def hidden_state_probabilities(self):
        """gamma per trajectory (T_k, N), fetched from the device on demand (construct the
        estimator with store_gamma=True).  With several ranks only the local trajectories are
        filled, the others are None."""
        if not self._store_gamma:
            raise RuntimeError('construct the estimator with store_gamma=True to keep gamma')
        out = [None] * self._nobs
        for j, k in enumerate(self._mine):
            out[k] = self._engine.gamma(j)
        return out

    @property
    def local_trajectories(self):
        """Indices of the trajectories this rank holds (all of them in a single process)."""
        return list(self._mine)

    @property
    def hmm(self):
        return self._hmm

    @property
    def output_model(self):
        return self._hmm.output_model

    @property
    def transition_matrix(self):
        return self._hmm.transition_matrix

    @property
    def initial_probability(self):
        return self._hmm.initial_distribution

    @property
    def stationary_probability(self):
        assert self._stationary, 'Estimator is not stationary'
        return self._hmm.initial_distribution

    # ---- E-step --------------------------------------------------------------------------
    def _estep(self):
        """All trajectories of this rank on the GPU, then the cross-rank reduction
        (maximum_likelihood.py:221-282, 383-385).  Returns an EStepResult."""
        om = self._hmm.output_model
        par0, par1 = om.parameters()
        eng, comm = self._engine, self._comm
        A, pi = self._hmm.transition_matrix, self._hmm.initial_distribution
        if not comm.active and hasattr(eng, 'estep_fetch_packed'):
            # (the per-trajectory log-likelihoods stay on the device: the EM loop needs their sum,
            # which is packed[0]; with 1e6 short trajectories they would be 8 MB per iteration)
            eng.estep_launch(A, pi, par0, par1, store_gamma=self._store_gamma)
            res = EStepResult(self._output, self._nstates, self._nsymbols, eng.estep_fetch_packed(), None)
        elif not comm.active:
            res = eng.estep(A, pi, par0, par1, store_gamma=self._store_gamma)
        elif hasattr(eng, 'estep_launch'):
            # the E-step leaves its packed statistics in a device buffer on the engine's GPU; ONE
            # in-place all-reduce (RCCL over xGMI) and ONE device-to-host copy follow -- the
            # distributed form of the sums at maximum_likelihood.py:271-282
            S = packed_stats_size(self._output, self._nstates, self._nsymbols)
            buf = comm.stats_buffer(S)
            logL_k = np.zeros(0)
            failed = None
            if self._mine:
                eng.estep_launch(A, pi, par0, par1, stats_dev=buf.data_ptr(),
                                 store_gamma=self._store_gamma)
                try:
                    # waits for the E-step of this shard (and repairs what the library can repair,
                    # bhmm_estep_fetch: non-finite counts -> one chunk per trajectory, repeated)
                    logL_k = eng.estep_fetch_logL()
                except (AssertionError, RuntimeError) as e:
                    # A shard that cannot be evaluated must not leave the other ranks alone in the
                    # collective: its buffer holds the non-finite statistics the error is about, the
                    # sum is non-finite on EVERY rank, and every rank fails the checks below together.
                    failed = e
                    buf[0] = float('nan')
            else:
                buf.zero_()
            packed = comm.allreduce_stats(buf)
            if failed is not None:
                raise failed
            res = EStepResult(self._output, self._nstates, self._nsymbols, packed, logL_k)
        else:
            # host-side engine (the CPU test double of tests/): host all-reduce
            if self._mine:
                res = eng.estep(A, pi, par0, par1, store_gamma=self._store_gamma)
                packed, logL_k = res.packed, res.logL_k
            else:
                packed = np.zeros(packed_stats_size(self._output, self._nstates, self._nsymbols))
                logL_k = np.zeros(0)
            packed = comm.allreduce_sum_numpy(packed)
            res = EStepResult(self._output, self._nstates, self._nsymbols, packed, logL_k)
        assert np.isfinite(res.loglik)       # maximum_likelihood.py:385
        nn = self._nstates
        assert np.all(np.isfinite(res.packed[:1 + 2 * nn + nn * nn])), \
            'counts of the E-step are not finite'
        return res

    # ---- M-step --------------------------------------------------------------------------
    def _update_model(self, res, maxiter=10000000):
        """maximum_likelihood.py:284-330 on the reduced statistics, as ONE native host call
        (bhmm_mstep: transition matrix incl. connected sets / reversible fixed points, initial or
        stationary distribution, emission parameters).  `reversible` is decided from the current
        transition matrix, like the reference's `self._hmm.is_reversible`."""
        om = self._hmm.output_model
        par0, par1 = om.parameters()
        if self._mstep is None:
            from ._native import MStep
            self._mstep = MStep(self._output, self._nstates, self._nsymbols)
        fixed = (self._fixed_stationary_distribution if self._stationary
                 else self._fixed_initial_distribution)
        T, pi, new0, new1 = self._mstep(res.packed, self._hmm.transition_matrix, par0, par1, None,
                                        self._stationary, fixed, maxiter, 1e-12, 1e-16)
        self._hmm.update(pi, T)
        om.set_parameters(new0, new1)

    def _update_model_numpy(self, res, maxiter=10000000):
        """The same M-step in numpy (_tmatrix.py): the restatement bhmm_mstep is tested against."""
        gamma0_sum, C = res.gamma0_sum, res.C
        T = _tmatrix.estimate_P(C, reversible=self._hmm.is_reversible,
                                fixed_statdist=self._fixed_stationary_distribution,
                                maxiter=maxiter, maxerr=1e-12, mincount_connectivity=1e-16)
        if self._stationary:
            if self._fixed_stationary_distribution is None:
                pi = _tmatrix.stationary_distribution(T, C=C, mincount_connectivity=1e-16)
            else:
                pi = self._fixed_stationary_distribution
        else:
            if self._fixed_initial_distribution is None:
                pi = gamma0_sum / np.sum(gamma0_sum)
            else:
                pi = self._fixed_initial_distribution
        self._hmm.update(pi, T)
        om = self._hmm.output_model
        if self._output == 'gaussian':
            om.estimate_from_statistics(res.state_counts, res.sum_gd, res.sum_gdd)
        else:
            om.estimate_from_statistics(res.symbol_counts)

    def em_step(self):
        """ONE whole EM iteration -- E-step over all trajectories (every rank's shard on its GPU,
        statistics all-reduced), then the M-step -- as `fit` performs it (maximum_likelihood.py:383-399
        without the convergence bookkeeping).  Returns the log-likelihood of the model the E-step
        was evaluated with."""
        res = self._estep()
        self._update_model(res, maxiter=self._maxit_P)
        self._last = res
        return res.loglik

    def compute_viterbi_paths(self):
        """maximum_likelihood.py:332-352.  Paths of the local trajectories; with several
        ranks they are gathered so that every rank returns the full list."""
        om = self._hmm.output_model
        par0, par1 = om.parameters()
        local = []
        if self._mine:
            local = self._engine.viterbi(self._hmm.transition_matrix,
                                         self._hmm.initial_distribution, par0, par1)
        paths = np.empty(self._nobs, dtype=object)
        if self._comm.active:
            for part, plist in zip(self._parts, self._comm.gather_objects(local)):
                for k, pth in zip(part, plist):
                    paths[k] = pth
        else:
            for k, pth in zip(self._mine, local):
                paths[k] = pth
        return paths

    def _select_start(self, ntrial=20):
        """Multi-start: run `ntrial` EM iterations from every candidate initial model (only when
        the caller gave none) and keep the model with the highest likelihood.  The iterations of
        the winner are not wasted -- fit() continues from where they ended."""
        if not self._alternative_starts:
            return
        candidates = [self._hmm] + [copy.deepcopy(h) for h in self._alternative_starts]
        self._alternative_starts = []
        best, best_ll = None, -np.inf
        for cand in candidates:
            self._hmm = cand
            self._reset_mstep_state()      # (the warm start of the fixed point belongs to one model sequence)
            cand.output_model.set_implementation(config.kernel)
            ll = -np.inf
            try:
                for _ in range(ntrial):
                    res = self._estep()
                    self._update_model(res, maxiter=self._maxit_P)
                ll = self._estep().loglik
            except (AssertionError, RuntimeError, ValueError, FloatingPointError):
                ll = -np.inf        # a start that degenerates is simply not chosen
            if ll > best_ll:
                best, best_ll = cand, ll
        if best is not None:
            self._hmm = best

    def _reset_mstep_state(self):
        """The native M-step warm-starts its reversible fixed point from the previous call's solution
        (within maxerr of the cold result); a new model sequence starts cold, so that a fit
        does not depend on what the estimator object did before."""
        if self._mstep is not None:
            self._mstep.warm[:] = 0.0

    def fit(self):
        """maximum_likelihood.py:354-446."""
        self._select_start()
        self._reset_mstep_state()
        it = 0
        self._likelihoods = np.zeros(self.maxit)
        loglik = 0.0
        tmatrix_nonzeros = self.hmm.transition_matrix.nonzero()
        converged = False
        res = None
        while not converged and it < self.maxit:
            res = self._estep()
            loglik = res.loglik
            if it > 0:
                dL = loglik - self._likelihoods[it - 1]
                if dL < self._accuracy:          # signed, as in the reference (:389-394)
                    converged = True
            self._update_model(res, maxiter=self._maxit_P)
            tmatrix_nonzeros_new = self.hmm.transition_matrix.nonzero()
            if not np.array_equal(tmatrix_nonzeros, tmatrix_nonzeros_new):
                converged = False                # likelihood is discontinuous here (:401-404)
                tmatrix_nonzeros = tmatrix_nonzeros_new
            self._likelihoods[it] = loglik
            it += 1
        self._likelihoods = self._likelihoods[:it]
        self._hmm.likelihood = loglik            # of the model before the last M-step (:423)
        if res is not None:
            self.count_matrix = res.C.copy()
            self.initial_count = res.gamma0_sum.copy()
        self._last = res
        self._hmm.hidden_state_trajectories = self.compute_viterbi_paths()
        return self._hmm
