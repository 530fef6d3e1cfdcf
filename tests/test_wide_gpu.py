"""GPU parity of the 9..64-state kernel family (BASELINE configs[3] shape: N = 64)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _random_model(n, rng, kind, M=0):
    A = rng.random((n, n)) + 0.02
    A[rng.random((n, n)) < 0.2] = 0.0
    A += np.eye(n) * 0.5
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-6, 6, n), rng.uniform(0.3, 1.2, n)
    return A, pi, rng.dirichlet(np.ones(M), n), None


def _check(res, ref):
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=RTOL)
    np.testing.assert_allclose(res.C, ref["C"], rtol=RTOL, atol=1e-11)
    np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=RTOL, atol=1e-11)


def test_g64_golden(golden):
    from bhmm_amd.engine import Engine
    g = golden("g64")
    eng = Engine(0)
    eng.set_observations("gaussian", [g["obs"]], 64)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"], store_gamma=True)
    np.testing.assert_allclose(res.loglik, float(g["logL"]), rtol=1e-12)
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(res.state_counts, g["state_counts"], rtol=RTOL)
    np.testing.assert_allclose(res.gamma0_sum, g["gamma0"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(eng.gamma(0)[g["rows"]], g["gamma_rows"], rtol=1e-8, atol=1e-13)
    eng.close()


@pytest.mark.parametrize("n,kind", [(9, "gaussian"), (16, "gaussian"), (17, "discrete"),
                                    (32, "gaussian"), (40, "discrete"), (64, "gaussian")])
def test_random_wide_models(n, kind):
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(n)
    M = 30
    A, pi, p0, p1 = _random_model(n, rng, kind, M)
    lengths = (700, 1, 2, 333, 64)
    if kind == "gaussian":
        obs = [rng.normal(0, 4, T) for T in lengths]
    else:
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    ref = orc.estep(kind, obs, A, pi, p0, p1, want_gamma=True)
    eng = Engine(0)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    res = eng.estep(A, pi, p0, p1, store_gamma=True)
    _check(res, ref)
    for k in (0, 3):
        np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
    if kind == "gaussian":
        sd = sum((g * (o[:, None] - p0[None, :])).sum(axis=0) for o, g in zip(obs, ref["gammas"]))
        sdd = sum((g * (o[:, None] - p0[None, :]) ** 2).sum(axis=0)
                  for o, g in zip(obs, ref["gammas"]))
        np.testing.assert_allclose(res.sum_gd, sd, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(res.sum_gdd, sdd, rtol=1e-8, atol=1e-9)
    else:
        cnt = np.zeros((n, M))
        for o, g in zip(obs, ref["gammas"]):
            orc.update_pout(o, g, cnt)
        np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
    r2 = eng.estep(A, pi, p0, p1)
    assert np.array_equal(res.packed, r2.packed)
    eng.close()


@pytest.mark.parametrize("n", [12, 33, 64])
def test_hidden_api_wide(n):
    import bhmm_amd.hidden as hidden
    rng = np.random.default_rng(100 + n)
    A, pi, mu, sig = _random_model(n, rng, "gaussian")
    pobs = orc.pobs_gaussian(rng.normal(0, 4, 900), mu, sig)
    ll_ref, a_ref = orc.forward(A, pobs, pi)
    b_ref = orc.backward(A, pobs)
    ll, alpha = hidden.forward(A, pobs, pi)
    beta = hidden.backward(A, pobs)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-12)
    np.testing.assert_allclose(alpha, a_ref, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(beta, b_ref, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(hidden.state_probabilities(alpha, beta), orc.gamma(a_ref, b_ref),
                               rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(hidden.transition_counts(alpha, beta, A, pobs),
                               orc.transition_counts(a_ref, b_ref, A, pobs), rtol=1e-9, atol=1e-12)
