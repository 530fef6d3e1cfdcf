"""More than 64 hidden states (bhmm/hidden/impl_c/_hidden.c:16-378 has no limit on N): the any-N
kernel family (bhmm_amd/csrc/gen_kernels.hpp) against the oracle at N = 65 ... 300 -- single-trajectory
hidden API, batched E-step (xi counts as an MFMA GEMM), Viterbi (bit-exact) and path sampling given
the uniforms (bit-exact)."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle_engine import OracleEngine, device_uniforms

pytestmark = pytest.mark.gpu


def _model(n, rng, kind, M=0, sparse=0.3):
    A = rng.random((n, n)) ** 3 + 1e-3
    A[rng.random((n, n)) < sparse] = 0.0
    A += np.eye(n) * rng.uniform(1.0, 5.0)
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-0.1 * n, 0.1 * n, n), rng.uniform(0.5, 1.5, n)
    if kind == "discrete":
        return A, pi, rng.dirichlet(np.ones(M), n), None
    return A, pi, None, None


@pytest.mark.parametrize("n", [65, 100, 200, 257])
def test_hidden_api_any_number_of_states(n):
    """forward / backward rows, gamma, xi counts, Viterbi and sampled paths of the reference-shaped
    single-trajectory entry points: rows and paths BIT-identical to the oracle (order-faithful
    kernels), log-likelihood and counts to 1e-12 / 1e-9."""
    from bhmm_amd import hidden
    rng = np.random.default_rng(n)
    T = 150 if n > 200 else 400
    A, pi, _, _ = _model(n, rng, "explicit")
    pobs = rng.random((T, n)) ** 4 + 1e-6
    pobs[5] = 0.0
    pobs[5, 3] = 0.7                                         # a row with a single possible state
    ll_ref, a_ref = orc.forward(A, pobs, pi)
    b_ref = orc.backward(A, pobs)
    ll, alpha = hidden.forward(A, pobs, pi)
    beta = hidden.backward(A, pobs)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-12)
    assert np.array_equal(alpha, a_ref) and np.array_equal(beta, b_ref)
    g = hidden.state_probabilities(alpha, beta)
    np.testing.assert_allclose(g, orc.gamma(a_ref, b_ref), rtol=1e-12, atol=1e-300)
    C = hidden.transition_counts(alpha, beta, A, pobs)
    np.testing.assert_allclose(C, orc.transition_counts(a_ref, b_ref, A, pobs), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(C.sum(), T - 1, rtol=1e-11)
    assert np.array_equal(hidden.viterbi(A, pobs, pi), orc.viterbi(A, pobs, pi))
    u = orc.libc_uniforms(T, 5)
    L = __import__("bhmm_amd")._lib.load()
    from bhmm_amd import _lib
    path = np.empty(T, dtype=np.int32)
    _lib.check(L.bhmm_sample_path(_lib.ip(path), _lib.dp(_lib.f64(a_ref)), _lib.dp(_lib.f64(A)),
                                  _lib.dp(_lib.f64(u)), n, T))
    assert np.array_equal(path, orc.sample_path(a_ref, A, u=u))


@pytest.mark.parametrize("n,kind", [(100, "gaussian"), (200, "discrete"), (70, "explicit"), (300, "gaussian")])
def test_batched_engine_any_number_of_states(n, kind):
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(1000 + n)
    M = 37
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (300, 1, 2, 77, 129) if n < 300 else (120, 1, 40)
    if kind == "gaussian":
        obs = [rng.normal(0, 0.06 * n, T) for T in lengths]
    elif kind == "discrete":
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    else:
        obs = [rng.random((T, n)) ** 3 + 1e-5 for T in lengths]
    eng = Engine(0)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    res = eng.estep(A, pi, p0, p1, store_gamma=True)
    if kind == "explicit":
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
        ref = dict(logL=np.array(lls), C=Cs, gamma0_sum=g0, state_counts=sc, gammas=gam)
    else:
        ref = orc.estep(kind, obs, A, pi, p0, p1, want_gamma=True)
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.C.sum(), sum(max(T - 1, 0) for T in lengths), rtol=1e-11)
    for k in (0, 1, len(lengths) - 1):
        np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-9, atol=1e-14)
    if kind == "gaussian":
        mu_new, sig_new = orc.estimate_gaussian(obs, ref["gammas"])
        w = res.state_counts
        m1 = res.sum_gd / w
        np.testing.assert_allclose(p0 + m1, mu_new, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(np.sqrt(res.sum_gdd / w - m1 * m1), sig_new, rtol=1e-7)
    if kind == "discrete":
        Bn = orc.estimate_discrete(obs, ref["gammas"], M)
        np.testing.assert_allclose(res.symbol_counts / res.symbol_counts.sum(axis=1)[:, None], Bn,
                                   rtol=1e-9, atol=1e-13)
    # Viterbi: bit-exact
    oe = OracleEngine()
    oe.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    vp = eng.viterbi(A, pi, p0, p1)
    vref = [orc.viterbi(A, o, pi) for o in obs] if kind == "explicit" else oe.viterbi(A, pi, p0, p1)
    for a, b in zip(vp, vref):
        assert np.array_equal(a, b)
    if n <= 256:
        v8 = eng.viterbi_u8(A, pi, p0, p1)
        assert np.array_equal(v8, np.concatenate(vp).astype(np.uint8))
    else:
        with pytest.raises(ValueError):
            eng.viterbi_u8(A, pi, p0, p1)
    # Gibbs hidden-path step with the device stream == oracle given the same uniforms
    if kind != "explicit":
        paths, C, n0, emis = eng.sample_paths(A, pi, p0, p1, seed=77)
        rp, rC, rn0, remis = oe.sample_paths(A, pi, p0, p1, seed=77)
        for a, b in zip(paths, rp):
            assert np.array_equal(a, b)
        assert np.array_equal(C, rC) and np.array_equal(n0, rn0)
        np.testing.assert_allclose(emis, remis, rtol=1e-10, atol=1e-9)
        import torch
        buf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
        eng.sample_paths_dev(A, pi, p0, p1, buf.data_ptr(), seed=77)
        C2, n02, emis2 = eng.unpack_path_stats(buf.cpu().numpy())
        assert np.array_equal(C2, rC) and np.array_equal(n02, rn0)
    eng.close()


def test_estimators_run_with_more_than_64_states():
    """estimate_hmm / bayesian sampling on a 72-state discrete problem: the likelihood rises, the
    sampler keeps a valid model (the host-side M-step / parameter draws have no state limit either)."""
    import bhmm_amd
    rng = np.random.default_rng(3)
    n, M = 72, 90
    A, pi, B, _ = _model(n, rng, "discrete", M, sparse=0.0)
    obs = [rng.integers(0, M, 600).astype(np.int32) for _ in range(3)]
    init = bhmm_amd.discrete_hmm(pi, A, B)
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, n, initial_model=init, reversible=False, maxit=4,
                                              accuracy=-1.0)
    hmm = est.fit()
    assert len(est.likelihoods) == 4 and np.all(np.diff(est.likelihoods) > 0)
    assert len(hmm.hidden_state_trajectories) == 3 and hmm.hidden_state_trajectories[0].max() < n
    smp = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=hmm, reversible=False)
    models = smp.sample(2, seed=4)
    assert len(models) == 2
    np.testing.assert_allclose(models[-1].transition_matrix.sum(axis=1), 1.0, rtol=1e-12)


@pytest.mark.parametrize("n", [139, 140, 141])
def test_transition_matrix_in_lds_up_to_the_size_that_fits(n):
    """Up to 140 states the any-N kernels keep their own copy of A in LDS (n^2 doubles + the vectors within the
    160 KB of a CU), from 141 on they read it from global memory: both sides of the boundary launch and give
    the oracle's log-likelihoods bit for bit (order-faithful rows), its counts, Viterbi and sampled paths."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(1000 + n)
    A, pi, mu, sig = _model(n, rng, "gaussian")
    obs = [rng.normal(0, 0.05 * n, T) for T in (40, 7, 1)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    pobs = [orc.pobs_gaussian(o, mu, sig) for o in obs]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-13)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
    for p, po in zip(eng.viterbi(A, pi, mu, sig), pobs):
        assert np.array_equal(p, orc.viterbi(A, po, pi))
    u = [rng.random(len(o)) for o in obs]
    paths = eng.sample_paths(A, pi, mu, sig, u=u)[0]
    for p, po, uu in zip(paths, pobs, u):
        assert np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, uu))
    eng.close()


@pytest.mark.parametrize("n,kind", [(65, "gaussian"), (100, "discrete"), (128, "gaussian"), (200, "gaussian"), (300, "discrete")])
def test_backward_draw_over_time_segments_is_the_serial_draw(n, kind):
    """More than 64 states (k_gen_sample_seg, SPL = 2 / 4 / 8 states per lane): the draw over time segments,
    coupled through the per-step uniforms -- the oracle's paths (_hidden.c:330-378) state for state with the
    caller's uniforms, the serial kernel's with the device stream; up to 128 states the alpha rows come
    from the tile forward pass."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(4000 + n)
    M = 25
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (20011, 1, 7000, 2, 300) if n <= 128 else (6000, 1, 2500)
    if kind == "gaussian":
        obs = [rng.normal(0, 0.12 * n, T) for T in lengths]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    u = [rng.random(T) for T in lengths]
    eng = Engine(0)
    eng.set_option("sample_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    for rep in range(2):
        paths, C, n0, emis = eng.sample_paths(A, pi, p0, p1, u=u)
        assert eng.get_option("sample_segmented") == 1 and eng.get_option("sample_segments") > 10
        if n in (65, 100):                # (the tile forward pass verified on these models; it need not)
            assert eng.get_option("sample_forward_segmented") == 1
        ref = [orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu) for po, uu in zip(pobs, u)]
        assert sum(int((p != r).sum()) for p, r in zip(paths, ref)) == 0
        Cr, n0r = orc.path_counts(ref, n)
        assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    seeded = eng.sample_paths(A, pi, p0, p1, seed=11)[0]
    eng.set_option("spec_enabled", 0)
    serial = eng.sample_paths(A, pi, p0, p1, seed=11)[0]
    assert eng.get_option("sample_segmented") == 0
    assert all(np.array_equal(a, b) for a, b in zip(seeded, serial))
    eng.close()


@pytest.mark.parametrize("n,kind", [(65, "gaussian"), (100, "discrete"), (128, "gaussian"), (97, "gaussian")])
def test_viterbi_over_time_segments_65_to_128_states(n, kind):
    """65..128 states, two target states per lane (k_gen_viterbi_seg): segments whose start vector is not
    the predecessor's to the bit are repeated from it until none is left; the paths are the oracle's
    (_hidden.c:186-276) byte for byte, ragged lengths and single steps included."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(5000 + n)
    M = 25
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (20011, 1, 7000, 2, 300)
    if kind == "gaussian":
        obs = [rng.normal(0, 0.12 * n, T) for T in lengths]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    for rep in range(2):
        paths = eng.viterbi(A, pi, p0, p1)
        assert eng.get_option("viterbi_chunked") == 1 and eng.get_option("viterbi_segments") > 10
        for p, po in zip(paths, pobs):
            assert np.array_equal(p, orc.viterbi(A, po, pi)), rep
    p8 = eng.viterbi_u8(A, pi, p0, p1)
    assert np.array_equal(p8, np.concatenate(paths).astype(np.uint8))
    eng.close()
