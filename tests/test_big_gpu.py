"""GPU parity of the row-batched matrix-core E-step for MORE THAN 128 states (csrc/big_kernels.hpp: 129 .. 512
states, the transition matrix streamed from L2 in matrix-operand order, vectors normalised at every step,
xi counts by the time-parallel GEMM k_big_xi_gemm), against the CPU oracle of
bhmm/hidden/impl_c/_hidden.c:16-183 and hidden/api.py:176-186."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _model(n, rng, kind, M=0, dense=True):
    A = rng.random((n, n)) + 0.02
    if not dense:
        A[rng.random((n, n)) < 0.4] = 0.0
    A += np.eye(n) * (6.0 if dense else 0.5)
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-8, 8, n), rng.uniform(0.3, 1.2, n)
    if kind == "discrete":
        return A, pi, rng.dirichlet(np.ones(M), n), None
    return A, pi, None, None


def _observations(kind, rng, lengths, n, M):
    if kind == "gaussian":
        return [rng.normal(0, 5, T) for T in lengths]
    if kind == "discrete":
        return [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    return [rng.random((T, n)) * rng.random((T, 1)) + 1e-3 for T in lengths]


def _reference(kind, obs, A, pi, p0, p1):
    if kind != "explicit":
        return orc.estep(kind, obs, A, pi, p0, p1, want_gamma=True)
    n = A.shape[0]
    lls, Cs, g0, sc, gam = [], np.zeros((n, n)), np.zeros(n), np.zeros(n), []
    for o in obs:
        ll, al = orc.forward(A, o, pi)
        be = orc.backward(A, o)
        gm = orc.gamma(al, be)
        lls.append(ll)
        gam.append(gm)
        g0 += gm[0]
        sc += gm.sum(axis=0)
        if len(o) > 1:
            Cs += orc.transition_counts(al, be, A, o)
    return dict(logL=np.array(lls), C=Cs, gamma0_sum=g0, state_counts=sc, gammas=gam)


# (column tiles per wavefront: 3 -- A stays in registers --, 4, 6, 8; state counts on and off the 64-grid)
@pytest.mark.parametrize("n,kind", [(129, "gaussian"), (192, "discrete"), (193, "explicit"), (256, "gaussian"),
                                    (257, "discrete"), (300, "gaussian"), (384, "explicit"), (400, "discrete"),
                                    (512, "gaussian")])
def test_big_estep_matches_the_oracle(n, kind):
    """Ragged batch cut into time segments (warm-up boundaries verified on the device): log-likelihoods,
    xi counts, gamma sums, stored gamma rows and the emission statistics against the oracle; the big
    kernels did the work, repeated calls are bit-identical."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(2000 + n)
    M = 23
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (1203, 1, 700, 2, 333, 3) if n <= 300 else (601, 1, 250, 2)
    obs = _observations(kind, rng, lengths, n, M)
    ref = _reference(kind, obs, A, pi, p0, p1)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 200)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    res = eng.estep(A, pi, p0, p1, store_gamma=True)
    assert eng.get_option("tile") == 1 and eng.get_option("wide_trouble") == 0, eng.get_option("tile_reason")
    assert eng.get_option("wide_segments") > len(lengths) and eng.get_option("spec_fail") == 0
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res.C.sum(), sum(max(T - 1, 0) for T in lengths), rtol=1e-11)
    for k in range(len(lengths)):
        np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
    if kind == "gaussian":
        sd = sum((g * (o[:, None] - p0[None, :])).sum(axis=0) for o, g in zip(obs, ref["gammas"]))
        sdd = sum((g * (o[:, None] - p0[None, :]) ** 2).sum(axis=0) for o, g in zip(obs, ref["gammas"]))
        np.testing.assert_allclose(res.sum_gd, sd, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(res.sum_gdd, sdd, rtol=1e-8, atol=1e-9)
    elif kind == "discrete":
        cnt = np.zeros((n, M))
        for o, g in zip(obs, ref["gammas"]):
            orc.update_pout(o, g, cnt)
        np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
    r2 = eng.estep(A, pi, p0, p1)               # statistics only: the same numbers, run to run
    np.testing.assert_allclose(r2.packed, res.packed, rtol=1e-12, atol=1e-12)
    r3 = eng.estep(A, pi, p0, p1)
    assert np.array_equal(r2.packed, r3.packed)
    eng.close()


@pytest.mark.parametrize("n", [200, 320])
def test_big_estep_sparse_model_and_zero_start_probabilities(n):
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(77 + n)
    A, pi, mu, sig = _model(n, rng, "gaussian", dense=False)
    pi[::3] = 0.0
    pi /= pi.sum()
    obs = [rng.normal(0, 5, T) for T in (901, 640, 300)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 150)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    assert eng.get_option("tile") == 1, eng.get_option("tile_reason")
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9, atol=1e-11)
    eng.close()


def test_big_kernels_hand_outlier_rows_to_the_order_faithful_family():
    """An observation 60 sigma from every state: all densities underflow to exactly 0, the step's normaliser
    is 0 -- outputmodel.py:126-130 turns such a row into ones.  The big kernels only report it (flags ->
    wide_trouble); the E-step is repeated by the order-faithful any-N kernels, which implement the rule, and
    the context stays there for these observations."""
    from bhmm_amd.engine import Engine
    n = 160
    rng = np.random.default_rng(5)
    A, pi, mu, sig = _model(n, rng, "gaussian")
    obs = [rng.normal(0, 5, T) for T in (700, 300)]
    obs[0][350] = 500.0
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 150)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    assert eng.get_option("tile") == 0 and eng.get_option("wide_trouble") != 0
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-11)
    r2 = eng.estep(A, pi, mu, sig)
    assert eng.get_option("tile") == 0
    np.testing.assert_allclose(r2.packed, res.packed, rtol=1e-12)
    eng.close()


def test_big_forward_pass_feeds_the_backward_draw():
    """The Gibbs hidden-path step above 128 states takes its alpha rows from the segmented matrix-core forward
    pass (any positive factor per row cancels in the draw): the oracle's paths for the caller's uniforms."""
    from bhmm_amd.engine import Engine
    n = 200
    rng = np.random.default_rng(9)
    A, pi, mu, sig = _model(n, rng, "gaussian")
    obs = [rng.normal(0, 5, T) for T in (900, 1, 400)]
    u = [rng.random(len(o)) for o in obs]
    eng = Engine(0)
    eng.set_option("wide_segment_len", 150)
    eng.set_observations("gaussian", obs, n)
    eng.estep(A, pi, mu, sig)
    paths = eng.sample_paths(A, pi, mu, sig, u=u)[0]
    assert eng.get_option("sample_forward_segmented") == 1
    for p, o, uu in zip(paths, obs, u):
        po = orc.pobs_gaussian(o, mu, sig)
        assert np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu))
    eng.close()


def test_em_and_gibbs_run_on_the_big_kernels():
    """MaximumLikelihoodEstimator / BayesianHMMSampler with 160 Gaussian states: every E-step of the EM run
    stays on the matrix-core kernels of big_kernels.hpp (time-segmented), the likelihood rises, and the
    log-likelihood of every iteration equals the oracle's for the model that iteration was evaluated with."""
    import bhmm_amd
    from bhmm_amd.engine import Engine
    n = 160
    rng = np.random.default_rng(21)
    A, pi, mu, sig = _model(n, rng, "gaussian")
    obs = [rng.normal(0, 4.5, T) for T in (1500, 700, 300)]
    init = bhmm_amd.gaussian_hmm(pi, A, mu, sig)
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, n, initial_model=init, reversible=False, maxit=5, accuracy=-1.0)
    est._engine.set_option("wide_segment_len", 150)
    est._engine.set_observations("gaussian", obs, n)      # (re-plan with the shorter segments)
    models, lls = [], []
    for _ in range(4):
        h = est.hmm
        models.append((h.transition_matrix.copy(), h.initial_distribution.copy(),
                       h.output_model.means.copy(), h.output_model.sigmas.copy()))
        lls.append(est.em_step())
        assert est._engine.get_option("tile") == 1 and est._engine.get_option("wide_trouble") == 0
        assert est._engine.get_option("wide_segments") > len(obs)
    assert np.all(np.diff(lls) > 0)
    for (Am, pim, mum, sgm), ll in zip(models, lls):
        ref = orc.estep("gaussian", obs, Am, pim, mum, sgm)
        np.testing.assert_allclose(ll, ref["logL"].sum(), rtol=1e-11)
    smp = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=est.hmm, reversible=False)
    ms = smp.sample(2, seed=5)
    assert len(ms) == 2
    np.testing.assert_allclose(ms[-1].transition_matrix.sum(axis=1), 1.0, rtol=1e-12)
