"""Host-side model updates: the native one-call M-step and Gibbs
parameter step of the C ABI against (a) fixtures written from the reference's own pure-numpy
code (tests/golden/gen_golden_host.py: _tmatrix_disconnected.py:126-190, gaussian.py:274-320,
discrete.py:217-251, util/statistics.py:34-151), (b) the numpy restatements in
bhmm_amd/estimators/_tmatrix.py, (c) a pure-Python restatement of the counter-based generator, and
(d) brute-force posteriors.  No GPU: this is host code of the shared library."""
import ctypes
import os

import numpy as np
import pytest
from scipy import stats as sps

import bhmm_amd
from bhmm_amd import _lib
from bhmm_amd.estimators import _tmatrix
from bhmm_amd.estimators._native import GibbsParameters, MStep
from bhmm_amd.output_models import DiscreteOutputModel, GaussianOutputModel
from bhmm_amd.util.statistics import confidence_interval, confidence_interval_arr

from host_rng_ref import Rng as PyRng

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_refs.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def L():
    return _lib.load()


def native_estimate(C, reversible, fixed=None, maxerr=1e-12, mincount=1e-16):
    C = _lib.f64(C)
    n = C.shape[0]
    P = np.empty((n, n))
    it = ctypes.c_int64(0)
    fp = _lib.f64(fixed) if fixed is not None else None
    _lib.check(L().bhmm_host_estimate_tmatrix(_lib.dp(P), _lib.dp(C), n, int(reversible), _lib.dp(fp),
                                              1000000, maxerr, mincount, ctypes.byref(it)))
    return P


# ---- (a) pinned by the reference's own numpy code -----------------------------------------------
def test_partially_reversible_estimator_equals_the_reference(gold):
    """_tmatrix_disconnected.py:126-190 -- numpy restatement and native code, both against outputs
    of the reference function itself."""
    for c in range(int(gold["prev_cases"])):
        C, S, want = gold["prev%d_C" % c], gold["prev%d_S" % c], gold["prev%d_P" % c]
        n = C.shape[0]
        P = np.eye(n)
        _tmatrix.transition_matrix_partial_rev(C, P, S, maxiter=1000000, maxerr=1e-12)
        np.testing.assert_allclose(P, want, rtol=1e-12, atol=1e-15)
        Pn = np.eye(n)
        _lib.check(L().bhmm_host_partial_rev(_lib.dp(Pn), _lib.dp(_lib.f64(C)), n,
                                             _lib.ip(np.ascontiguousarray(S, dtype=np.int32)), 1000000, 1e-12))
        np.testing.assert_allclose(Pn, want, rtol=1e-12, atol=1e-15)
        # rows outside S untouched, rows in S stochastic, detailed balance inside S
        assert np.array_equal(Pn[~S], np.eye(n)[~S])
        np.testing.assert_allclose(Pn[S].sum(axis=1), 1.0, rtol=1e-14)
    assert np.array_equal(_tmatrix.nonempty_set(gold["nonempty_C"]), gold["nonempty_0"])
    assert np.array_equal(_tmatrix.nonempty_set(gold["nonempty_C"], 1.0), gold["nonempty_1"])


def test_gaussian_gibbs_draw_equals_the_reference_under_the_same_numpy_stream(gold):
    """gaussian.py:274-320: `sample` (arrays per state) and `sample_from_statistics` (the sums the
    GPU path pass returns) consume np.random exactly like the reference (randn, then chisquare, per
    state) and give its means / sigmas."""
    for c in range(int(gold["gs_cases"])):
        mu, sig, sizes = gold["gs%d_mu" % c], gold["gs%d_sigma" % c], gold["gs%d_sizes" % c]
        flat = gold["gs%d_obs" % c]
        obs = np.split(flat, np.cumsum(sizes)[:-1])
        seed = int(gold["gs%d_seed" % c])
        gm = GaussianOutputModel(len(mu), means=mu.copy(), sigmas=sig.copy())
        np.random.seed(seed)
        gm.sample([o.copy() for o in obs])
        np.testing.assert_allclose(gm.means, gold["gs%d_mu_new" % c], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(gm.sigmas, gold["gs%d_sigma_new" % c], rtol=1e-10)
        gm2 = GaussianOutputModel(len(mu), means=mu.copy(), sigmas=sig.copy())
        n_i = np.array([len(o) for o in obs], dtype=float)
        sd = np.array([np.sum(o - m) for o, m in zip(obs, mu)])
        sdd = np.array([np.sum((o - m) ** 2) for o, m in zip(obs, mu)])
        np.random.seed(seed)
        gm2.sample_from_statistics(n_i, sd, sdd)
        np.testing.assert_allclose(gm2.means, gold["gs%d_mu_new" % c], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(gm2.sigmas, gold["gs%d_sigma_new" % c], rtol=1e-9)


def test_discrete_gibbs_draw_equals_the_reference_under_the_same_numpy_stream(gold):
    """discrete.py:217-251."""
    for c in range(int(gold["ds_cases"])):
        B, sizes, flat = gold["ds%d_B" % c], gold["ds%d_sizes" % c], gold["ds%d_obs" % c]
        obs = np.split(flat, np.cumsum(sizes)[:-1])
        dm = DiscreteOutputModel(B.copy())
        np.testing.assert_array_equal(dm.prior, gold["ds%d_prior" % c])
        np.random.seed(int(gold["ds%d_seed" % c]))
        dm.sample([o.copy() for o in obs])
        np.testing.assert_allclose(dm.output_probabilities, gold["ds%d_B_new" % c], rtol=1e-13, atol=0)
        dm2 = DiscreteOutputModel(B.copy())
        counts = np.array([np.bincount(o, minlength=B.shape[1]) for o in obs], dtype=float)
        np.random.seed(int(gold["ds%d_seed" % c]))
        dm2.sample_from_statistics(counts)
        np.testing.assert_allclose(dm2.output_probabilities, gold["ds%d_B_new" % c], rtol=1e-13, atol=0)


def test_sample_statistics_equal_the_reference(gold):
    """util/statistics.py:34-151 (what SampledHMM reports)."""
    for i in range(int(gold["ci_n"])):
        a = gold["ci%d_in" % i]
        alpha, size, d = a[0], int(a[1]), a[2:]
        assert len(d) == size
        np.testing.assert_allclose(confidence_interval(d, alpha), gold["ci%d_out" % i], rtol=1e-14)
    lo, up = confidence_interval_arr(gold["cia2"], conf=0.9)
    np.testing.assert_allclose(lo, gold["cia2_lo"], rtol=1e-14)
    np.testing.assert_allclose(up, gold["cia2_up"], rtol=1e-14)
    lo, up = confidence_interval_arr(gold["cia3"])
    np.testing.assert_allclose(lo, gold["cia3_lo"], rtol=1e-14)
    np.testing.assert_allclose(up, gold["cia3_up"], rtol=1e-14)
    lo, up = confidence_interval_arr([gold["cia2"][i] for i in range(30)], conf=0.5)
    np.testing.assert_allclose(lo, gold["cial_lo"], rtol=1e-14)
    np.testing.assert_allclose(up, gold["cial_up"], rtol=1e-14)
    with pytest.raises(ValueError):
        confidence_interval(np.arange(3.0), 1.5)


# ---- (b) native code against the numpy restatements -----------------------------------------------
def test_native_transition_matrix_estimators_equal_numpy(monkeypatch):
    """estimate_P (_tmatrix_disconnected.py:68-123) on random count matrices of every connectivity
    structure: disconnected, closed / open sets, empty rows, singletons."""
    monkeypatch.setattr(_tmatrix, "NATIVE", False)      # the numpy side is numpy all the way down
    rng = np.random.default_rng(0)
    for trial in range(400):
        n = int(rng.integers(1, 10))
        C = rng.random((n, n)) * rng.integers(1, 100)
        C[rng.random((n, n)) < rng.choice([0, 0.3, 0.6, 0.85])] = 0
        for rev in (True, False):
            want = _tmatrix.estimate_P(C.copy(), reversible=rev, maxerr=1e-12, mincount_connectivity=1e-16)
            np.testing.assert_allclose(native_estimate(C, rev), want, rtol=0, atol=1e-13, err_msg=str(C))
    for trial in range(60):
        n = int(rng.integers(2, 8))
        C = rng.random((n, n)) * 50 + 0.1
        pi = rng.dirichlet(np.ones(n))
        want = _tmatrix.estimate_P(C, reversible=True, fixed_statdist=pi, maxerr=1e-12)
        got = native_estimate(C, True, pi)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(pi @ got, pi, atol=1e-9)               # pi is stationary
        np.testing.assert_allclose(pi[:, None] * got, (pi[:, None] * got).T, atol=1e-10)


def test_native_connected_sets_stationary_vectors_reversibility():
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(1)
    for trial in range(200):
        n = int(rng.integers(1, 90))
        C = rng.random((n, n))
        C[rng.random((n, n)) < rng.choice([0.5, 0.9, 0.97, 0.99])] = 0
        for strong in (True, False):
            lab = np.empty(n, dtype=np.int32)
            _lib.check(L().bhmm_host_connected_sets(_lib.ip(lab), _lib.dp(_lib.f64(C)), n, 0.0, int(strong)))
            k, ref = connected_components(csr_matrix(C), directed=True,
                                          connection='strong' if strong else 'weak')
            assert len(np.unique(lab)) == k
            # same partition, and numbered by decreasing size
            assert len(set(zip(lab.tolist(), ref.tolist()))) == k
            sizes = np.bincount(lab)
            assert np.all(np.diff(sizes) <= 0)
            ours = _tmatrix.connected_sets(C, strong=strong)
            assert [sorted(s.tolist()) for s in ours] == [sorted(np.where(lab == i)[0].tolist()) for i in range(k)]
    for trial in range(100):
        n = int(rng.integers(1, 40))
        P = rng.random((n, n)) ** 3 + 1e-9
        P /= P.sum(axis=1)[:, None]
        pi = np.empty(n)
        _lib.check(L().bhmm_host_stationary_vector(_lib.dp(pi), _lib.dp(P), n))
        np.testing.assert_allclose(pi @ P, pi, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(pi, _tmatrix.stationary_vector(P), rtol=1e-9, atol=1e-13)
        assert L().bhmm_host_is_reversible(_lib.dp(P), n) == int(_tmatrix.is_reversible(P))
        X = rng.random((n, n))
        Prev = (X + X.T) / (X + X.T).sum(axis=1)[:, None]
        assert L().bhmm_host_is_reversible(_lib.dp(_lib.f64(Prev)), n) == 1
    # nearly uncoupled chain: GTH keeps full relative accuracy where a linear solve loses it
    P = np.array([[1 - 1e-13, 1e-13], [3e-13, 1 - 3e-13]])
    pi = np.empty(2)
    _lib.check(L().bhmm_host_stationary_vector(_lib.dp(pi), _lib.dp(P), 2))
    np.testing.assert_allclose(pi, [0.75, 0.25], rtol=1e-3)   # (entries carry 1e-16 / 1e-13 rounding)
    # reducible block: the closed class carries the mass
    P = np.array([[0.5, 0.5, 0.0], [0.0, 0.7, 0.3], [0.0, 0.4, 0.6]])
    pi = np.empty(3)
    _lib.check(L().bhmm_host_stationary_vector(_lib.dp(pi), _lib.dp(P), 3))
    np.testing.assert_allclose(pi, [0.0, 4 / 7., 3 / 7.], atol=1e-10)


@pytest.mark.parametrize("kind", ["gaussian", "discrete"])
def test_one_call_mstep_equals_the_numpy_mstep(kind, monkeypatch):
    """bhmm_mstep == maximum_likelihood.py:284-330 evaluated in numpy (_update_model_numpy), for
    reversible / non-reversible models, stationary and fixed distributions."""
    from bhmm_amd.engine import EStepResult
    from bhmm_amd.estimators.maximum_likelihood import MaximumLikelihoodEstimator, packed_stats_size
    from oracle_engine import OracleEngine
    monkeypatch.setattr(_tmatrix, "NATIVE", False)
    rng = np.random.default_rng(5)
    n, M = 4, 6
    for trial in range(40):
        X = rng.random((n, n)) + np.eye(n)
        if trial % 2:
            X = X + X.T
        A = X / X.sum(axis=1)[:, None]
        pi0 = rng.dirichlet(np.ones(n))
        if kind == "gaussian":
            init = bhmm_amd.gaussian_hmm(pi0, A, np.sort(rng.normal(0, 2, n)), rng.random(n) + 0.3)
            obs = [rng.normal(0, 2, 50)]
        else:
            init = bhmm_amd.discrete_hmm(pi0, A, rng.dirichlet(np.ones(M), size=n))
            obs = [rng.integers(0, M, 50).astype(np.int32)]
        stationary = trial % 3 == 1
        p = rng.dirichlet(np.ones(n)) if trial % 5 == 2 else None
        ests = [MaximumLikelihoodEstimator(obs, n, initial_model=init, stationary=stationary, p=p,
                                           engine_factory=OracleEngine) for _ in range(2)]
        S = packed_stats_size(kind, n, M if kind == "discrete" else 0)
        packed = np.zeros(S)
        C = rng.random((n, n)) * 100
        if trial % 4 == 3:
            C[rng.random((n, n)) < 0.5] = 0
            C += np.diag(rng.random(n))
        w = rng.random(n) * 1000 + 1
        packed[0] = -123.0
        packed[1:1 + n] = rng.dirichlet(np.ones(n))
        packed[1 + n:1 + n + n * n] = C.ravel()
        packed[1 + n + n * n:1 + 2 * n + n * n] = w
        if kind == "gaussian":
            m1 = rng.normal(0, 0.1, n)
            packed[1 + 2 * n + n * n:1 + 3 * n + n * n] = m1 * w
            packed[1 + 3 * n + n * n:] = (m1 ** 2 + rng.random(n) + 0.1) * w
        else:
            packed[1 + 2 * n + n * n:] = (rng.random((n, M)) * 10).ravel()
        res = EStepResult(kind, n, M if kind == "discrete" else 0, packed, np.zeros(1))
        ests[0]._update_model(res, maxiter=100000)
        ests[1]._update_model_numpy(res, maxiter=100000)
        a, b = ests[0].hmm, ests[1].hmm
        np.testing.assert_allclose(a.transition_matrix, b.transition_matrix, rtol=0, atol=1e-12)
        np.testing.assert_allclose(a.initial_distribution, b.initial_distribution, rtol=0, atol=1e-11)
        for x, y in zip(a.output_model.parameters(), b.output_model.parameters()):
            if x is not None:
                np.testing.assert_allclose(x, y, rtol=1e-13)
    # a vanishing variance is the reference's RuntimeError (gaussian.py:271-272)
    ms = MStep("gaussian", 2)
    packed = np.array([0.0, 0.5, 0.5, 5, 1, 1, 5, 6.0, 6.0, 0.0, 0.0, 0.0, 0.0])
    with pytest.raises(RuntimeError, match="sigma is too small"):
        ms(packed, np.eye(2), np.zeros(2), np.ones(2), False, False, None, 1000, 1e-12, 1e-16)


# ---- (c) the counter-based generator and the algebra around it --------------------------------------
def draws(count, what, param=0.0, seed=1, stream=0):
    out = np.empty(count)
    _lib.check(L().bhmm_host_rng_draws(_lib.dp(out), count, what, param, seed, stream))
    return out


def test_generator_is_reproduced_bit_for_bit_by_the_python_restatement():
    for what, fn in ((0, lambda r: r.u01()), (1, lambda r: r.normal()), (3, lambda r: r.u01_open())):
        r = PyRng(77, 3)
        want = np.array([fn(r) for _ in range(3000)])
        assert np.array_equal(draws(3000, what, seed=77, stream=3), want)
    for k in (0.3, 1.0, 7.5, 1e5):
        r = PyRng(5, 0)
        want = np.array([r.gamma(k) for _ in range(500)])
        np.testing.assert_allclose(draws(500, 2, k, seed=5), want, rtol=1e-15)


def test_generator_distributions():
    assert sps.kstest(draws(200000, 0), 'uniform').pvalue > 1e-3
    z = draws(1000000, 1, seed=3)
    assert sps.kstest(z, 'norm').pvalue > 1e-3
    assert abs((np.abs(z) > 3.5).mean() / (2 * sps.norm.sf(3.5)) - 1) < 0.15     # ziggurat tail
    for k in (0.05, 0.5, 1.0, 2.5, 100.0, 1e6):
        assert sps.kstest(draws(200000, 2, k, seed=11), 'gamma', args=(k,)).pvalue > 1e-3, k


def _gibbs(kind, n, M, packed, seed, sweep, par0, par1, **kw):
    g = GibbsParameters(kind, n, M, **kw)
    return g(packed, par0, par1, seed, sweep)


def test_gibbs_parameter_call_equals_the_python_restatement_draw_for_draw():
    """Non-reversible sweep: emission draws (gaussian.py:303-318 | discrete.py:243-251), Dirichlet
    rows, Dirichlet p0 -- recomputed in Python from the same generator, same order."""
    rng = np.random.default_rng(9)
    n = 3
    C = rng.integers(0, 50, (n, n)).astype(float)
    C[2, 0] = 0
    n0 = np.array([3.0, 0.0, 1.0])
    mu, sig = np.array([-1.0, 0.5, 2.0]), np.array([0.5, 1.0, 0.8])
    cnt = np.array([120.0, 1.0, 0.0])
    sd = np.array([3.0, -0.2, 0.0])
    sdd = np.array([40.0, 0.04, 0.0])
    prior_C = np.full((n, n), 0.25)
    prior_n0 = np.array([0.5, 0.0, 0.5])
    packed = np.concatenate([C.ravel(), n0, cnt, sd, sdd])
    T, p0, mu_new, sig_new = _gibbs("gaussian", n, 0, packed, 42, 7, mu, sig, prior_C=prior_C,
                                    prior_n0=prior_n0, reversible=False, nsteps=1000)
    r = PyRng(42, 7)
    emu, esig = mu.copy(), sig.copy()
    for i in range(n):
        ni = round(cnt[i])
        if ni > 0:
            mean_obs = emu[i] + sd[i] / ni
            old = emu[i]
            emu[i] = r.normal() * esig[i] / np.sqrt(ni) + mean_obs
            if ni > 1:
                chi2 = r.chisquare(ni - 1.0)
                sh = emu[i] - old
                s2 = (sdd[i] - 2.0 * sh * sd[i]) / ni + sh * sh
                esig[i] = np.sqrt(max(s2, 0.0)) / np.sqrt(chi2 / ni)
    np.testing.assert_allclose(mu_new, emu, rtol=1e-15)
    np.testing.assert_allclose(sig_new, esig, rtol=1e-14)
    eT = np.zeros((n, n))
    for i in range(n):
        r.dirichlet(list(C[i] + prior_C[i]), eT[i])
    np.testing.assert_allclose(T, eT, rtol=1e-14)
    ep0 = np.zeros(n)
    r.dirichlet(list(n0 + prior_n0), ep0)
    np.testing.assert_allclose(p0, ep0, rtol=1e-14)
    assert p0[1] == 0.0 and abs(p0.sum() - 1) < 1e-15
    # discrete emissions: rows with the model's prior, zero-count symbols keep their entry
    M = 4
    B = rng.dirichlet(np.ones(M), size=n)
    cnts = rng.integers(0, 9, (n, M)).astype(float)
    cnts[:, 2] = 0
    packed = np.concatenate([C.ravel(), n0, cnts.ravel()])
    T, p0, B_new, _ = _gibbs("discrete", n, M, packed, 1, 0, B, None, prior_C=prior_C, prior_n0=prior_n0,
                             reversible=False)
    r = PyRng(1, 0)
    eB = B.copy()
    for i in range(n):
        r.dirichlet(list(cnts[i]), eB[i])
    np.testing.assert_allclose(B_new, eB, rtol=1e-14)
    assert np.array_equal(B_new[:, 2], B[:, 2])


def test_gaussian_gibbs_draws_follow_the_conditional_posterior():
    """mu | sigma ~ N(mean, sigma^2 / n);  n s^2(mu) / sigma_new^2 ~ chi2(n - 1)."""
    n, cnt = 1, 40.0
    mu, sig = np.array([0.3]), np.array([1.7])
    rng = np.random.default_rng(2)
    o = rng.normal(0.5, 1.5, int(cnt))
    sd, sdd = np.sum(o - mu[0]), np.sum((o - mu[0]) ** 2)
    packed = np.array([1.0, 1.0, cnt, sd, sdd])
    g = GibbsParameters("gaussian", 1, 0, reversible=False)
    res = np.array([[x[0] for x in g(packed, mu, sig, 99, s)[2:]] for s in range(20000)])
    z = (res[:, 0] - o.mean()) / (sig[0] / np.sqrt(cnt))
    assert sps.kstest(z, 'norm').pvalue > 1e-3
    s2 = np.array([np.mean((o - m) ** 2) for m in res[:, 0]])
    q = cnt * s2 / res[:, 1] ** 2
    assert sps.kstest(q, 'chi2', args=(cnt - 1,)).pvalue > 1e-3


# ---- (d) the reversible sampler against brute-force posteriors ---------------------------------------
def _rev_chain(C, nsamples, nsteps, seed=123):
    n = C.shape[0]
    packed = np.concatenate([C.ravel(), np.ones(n)])
    g = GibbsParameters("explicit", n, 0, reversible=True, nsteps=nsteps)
    return np.array([g(packed, None, None, seed, s)[0] for s in range(nsamples)])


def test_reversible_sampler_two_states_against_a_grid_posterior():
    """p(X | C) ~ prod_ij (x_ij / x_i)^c_ij  prod_{i<=j} x_ij^-1 on x11 + 2 x12 + x22 = 1
    (Trendelkamp-Schroer et al. 2015; the prior msmtools' reversible sampler uses)."""
    C = np.array([[5.0, 3.0], [2.0, 7.0]])
    S = _rev_chain(C, 30000, 20)
    g = 2000
    a = (np.arange(g) + 0.5) / g
    x11, x12 = np.meshgrid(a, a / 2, indexing='ij')
    x22 = 1 - x11 - 2 * x12
    ok = x22 > 0
    x22s = np.where(ok, x22, 1.0)
    x1, x2 = x11 + x12, x12 + x22s
    logd = (C[0, 0] * np.log(x11 / x1) + C[0, 1] * np.log(x12 / x1) + C[1, 0] * np.log(x12 / x2)
            + C[1, 1] * np.log(x22s / x2) - np.log(x11) - np.log(x12) - np.log(x22s))
    w = np.where(ok, np.exp(logd - logd[ok].max()), 0.0)
    for q, samp in ((x12 / x1, S[:, 0, 1]), (x12 / x2, S[:, 1, 0])):
        m = (w * q).sum() / w.sum()
        sd = np.sqrt((w * q * q).sum() / w.sum() - m * m)
        assert abs(samp.mean() - m) < 4 * sd / np.sqrt(len(samp) / 4.0), (samp.mean(), m)
        assert abs(samp.std() / sd - 1) < 0.03
    for T in S[:50]:
        np.testing.assert_allclose(T.sum(axis=1), 1.0, rtol=1e-14)


def test_reversible_sampler_three_states_against_importance_sampling():
    C = np.array([[8.0, 2.0, 1.0], [3.0, 9.0, 2.0], [0.0, 3.0, 6.0]])
    S = _rev_chain(C, 20000, 10, seed=5)
    # importance sampling: X ~ symmetric matrix from Dirichlet over the 6 free elements
    rng = np.random.default_rng(0)
    iu = np.triu_indices(3)
    c0 = (C + C.T)[iu] - np.diag(np.diag(C))[iu]
    alpha = c0 * 0.7 + 0.2
    Y = rng.dirichlet(alpha, size=1200000)            # y = elements of the upper triangle, sum 1
    X = np.zeros((len(Y), 3, 3))
    X[:, iu[0], iu[1]] = Y
    X = X + np.transpose(X, (0, 2, 1)) - np.einsum('kij,ij->kij', X, np.eye(3))
    # off-diagonal elements count twice in sum X: measure on {sum_{i<=j} y = 1} differs from
    # {sum X = 1} only by a per-sample rescaling, and P = X / rowsum is scale invariant
    P = X / X.sum(axis=2)[:, :, None]
    logp = np.sum(np.where(C > 0, C * np.log(np.maximum(P, 1e-300)), 0.0), axis=(1, 2)) - np.log(Y).sum(axis=1)
    logq = ((alpha - 1) * np.log(Y)).sum(axis=1)
    lw = logp - logq
    w = np.exp(lw - lw.max())
    ess = w.sum() ** 2 / (w ** 2).sum()
    assert ess > 2000
    for (i, j) in ((0, 1), (1, 2), (2, 1), (0, 0)):
        m = (w * P[:, i, j]).sum() / w.sum()
        sd = np.sqrt((w * P[:, i, j] ** 2).sum() / w.sum() - m * m)
        assert abs(S[:, i, j].mean() - m) < 5 * sd / np.sqrt(min(ess, len(S) / 3.0)), (i, j, S[:, i, j].mean(), m)
        assert abs(S[:, i, j].std() / sd - 1) < 0.1


def test_reversible_sampler_properties_and_errors():
    rng = np.random.default_rng(4)
    n = 6
    C = rng.integers(0, 40, (n, n)).astype(float) + np.diag(rng.integers(50, 90, n))
    C[0, 3] = C[3, 0] = 0                     # structural zero in both directions
    packed = np.concatenate([C.ravel(), np.ones(n)])
    g = GibbsParameters("explicit", n, 0, reversible=True, nsteps=25, stationary=True)
    T, p0, _, _ = g(packed, None, None, 8, 0)
    assert T[0, 3] == 0 and T[3, 0] == 0
    np.testing.assert_allclose(T.sum(axis=1), 1.0, rtol=1e-14)
    np.testing.assert_allclose(p0 @ T, p0, atol=1e-12)                       # stationary = True
    np.testing.assert_allclose(p0[:, None] * T, (p0[:, None] * T).T, atol=1e-14)   # detailed balance
    T2 = g(packed, None, None, 8, 0)[0]
    assert np.array_equal(T, T2)                                              # function of (seed, sweep)
    assert not np.array_equal(T, g(packed, None, None, 8, 1)[0])
    Cd = np.array([[5.0, 0.0], [0.0, 5.0]])
    with pytest.raises(NotImplementedError, match="disconnected"):
        GibbsParameters("explicit", 2, 0, reversible=True)(np.concatenate([Cd.ravel(), [1, 1]]), None, None, 1, 0)
    # posterior concentrates on the reversible MLE for large counts
    big = C * 1e5
    Tb = GibbsParameters("explicit", n, 0, reversible=True, nsteps=50)(np.concatenate([big.ravel(), np.ones(n)]),
                                                                        None, None, 3, 0)[0]
    np.testing.assert_allclose(Tb, _tmatrix.mle_reversible(big, maxerr=1e-13), atol=2e-3)


def test_reversible_sampler_is_the_same_chain_at_every_vector_width():
    """The element updates of a round run in SIMD lanes (csrc/host_rev_sampler.hpp: four with AVX2 + FMA, one
    without); every update draws from a stream of its own, so the chain must not depend on the width -- to the
    bit, on dense, sparse, huge and fractional count matrices (Gamma shapes below one, degenerate rows)."""
    import ctypes
    rng = np.random.default_rng(0)

    def run(C, nsweeps, base, lanes):
        n = C.shape[0]
        X = _lib.f64((C + C.T) / (2 * C.sum()))
        r = L().bhmm_host_sample_reversible(_lib.dp(X), _lib.dp(_lib.f64(C)), n, nsweeps, ctypes.c_uint64(base), lanes)
        return X, r

    widest = None
    for n in (2, 3, 5, 8, 13):
        for trial in range(12):
            C = rng.integers(0, 50, (n, n)).astype(float) + np.diag(rng.integers(10, 200, n))
            if trial % 3 == 0:
                C *= 1e4
            if trial % 4 == 1 and n > 2:
                C[0, 1] = C[1, 0] = 0
            if trial % 5 == 2:
                C = C * rng.random((n, n)) * 0.02
            X1, r1 = run(C, 40, 77 + trial, 1)
            Xw, rw = run(C, 40, 77 + trial, 0)
            assert r1 == 1 and rw in (1, 4)
            widest = rw
            assert np.array_equal(X1, Xw), (n, trial)
            assert np.allclose(X1, X1.T, rtol=0, atol=0) and abs(X1.sum() - 1) < 1e-12 and np.all(X1 >= 0)
            assert np.array_equal(X1 == 0, (C + C.T) == 0)              # structural zeros stay
            assert not np.array_equal(X1, run(C, 40, 78 + trial, 1)[0])  # a function of the base
    assert widest in (1, 4)
