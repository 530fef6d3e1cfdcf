"""GPU parity: batched HIP E-step (through the C ABI) vs the CPU oracle and the golden
fixtures.  Tolerance: 1e-6 relative in fp64 (BASELINE.json north_star); in practice the
agreement is ~1e-12, and the tests assert 1e-9 so that regressions are visible.
"""
import numpy as np
import pytest

from conftest import split
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _engine():
    from bhmm_amd.engine import Engine
    return Engine(0)


def _cmp(res, ref, n):
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=RTOL)
    np.testing.assert_allclose(res.loglik, ref["logL"].sum(), rtol=RTOL)
    np.testing.assert_allclose(res.C, ref["C"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=RTOL, atol=1e-12)


def _gauss_stats(obs, gammas, mu):
    sd = sum((g * (o[:, None] - mu[None, :])).sum(axis=0) for o, g in zip(obs, gammas))
    sdd = sum((g * (o[:, None] - mu[None, :]) ** 2).sum(axis=0) for o, g in zip(obs, gammas))
    return sd, sdd


@pytest.mark.parametrize("chunk", [0, 1, 3, 7, 64, 100000])
def test_g8_ragged_golden(golden, chunk):
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    ref = orc.estep("gaussian", obs, g["A"], g["pi"], g["mu"], g["sigma"], want_gamma=True)
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"], store_gamma=True)
    # against the reference's own numbers (fixture) ...
    np.testing.assert_allclose(res.logL_k, g["logL"], rtol=RTOL)
    np.testing.assert_allclose(res.C, g["C"].sum(axis=0), rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(res.gamma0_sum, g["gamma0"].sum(axis=0), rtol=RTOL)
    np.testing.assert_allclose(res.state_counts, g["state_counts"].sum(axis=0), rtol=RTOL)
    # ... and against the oracle run here
    _cmp(res, ref, 8)
    sd, sdd = _gauss_stats(obs, ref["gammas"], g["mu"])
    np.testing.assert_allclose(res.sum_gd, sd, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(res.sum_gdd, sdd, rtol=1e-8)
    # gamma itself (hidden_state_probabilities, maximum_likelihood.py:190-193)
    for k in range(len(obs)):
        np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(eng.gamma(4), g["gamma4"], rtol=1e-8, atol=1e-13)
    # emission M-step from the sufficient statistics == two-pass reference (gaussian.py:214-272)
    mu_new = g["mu"] + res.sum_gd / res.state_counts
    var = res.sum_gdd / res.state_counts - (res.sum_gd / res.state_counts) ** 2
    np.testing.assert_allclose(mu_new, g["mu_new"], rtol=1e-9)
    np.testing.assert_allclose(np.sqrt(var), g["sigma_new"], rtol=1e-8)
    eng.close()


@pytest.mark.parametrize("chunk", [0, 2, 5, 33])
def test_d8_ragged_golden(golden, chunk):
    g = golden("d8_ragged")
    obs = split(g["obs"].astype(np.int32), g["lengths"])
    M = g["B"].shape[1]
    ref = orc.estep("discrete", obs, g["A"], g["pi"], g["B"], want_gamma=True)
    eng = _engine()
    eng.set_observations("discrete", obs, 8, nsymbols=M, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["B"], store_gamma=True)
    np.testing.assert_allclose(res.logL_k, g["logL"], rtol=RTOL)
    _cmp(res, ref, 8)
    Bn = res.symbol_counts / res.symbol_counts.sum(axis=1)[:, None]
    np.testing.assert_allclose(Bn, g["B_new"], rtol=1e-8, atol=1e-14)
    np.testing.assert_allclose(eng.gamma(1), g["gamma1"], rtol=1e-8, atol=1e-13)
    eng.close()


@pytest.mark.parametrize("chunk", [0, 4])
def test_outliers_and_zeros(golden, chunk):
    g = golden("g8_outliers")
    eng = _engine()
    eng.set_observations("gaussian", [g["obs"]], 8, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"], store_gamma=True)
    np.testing.assert_allclose(res.loglik, float(g["logL"]), rtol=RTOL)
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(eng.gamma(0), g["gamma"], rtol=1e-8, atol=1e-13)
    eng.close()
    # structural zeros in A, B and pi (3 states -> padded to 4)
    g = golden("d3_zeros")
    eng = _engine()
    eng.set_observations("discrete", [g["obs"]], 3, nsymbols=4, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["B"], store_gamma=True)
    np.testing.assert_allclose(res.loglik, float(g["logL"]), rtol=RTOL)
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(eng.gamma(0), g["gamma"], rtol=1e-8, atol=1e-13)
    eng.close()


def test_kat_fixtures(golden):
    g = golden("kat1_toy")   # explicit pobs, 2 states
    eng = _engine()
    eng.set_observations("explicit", [g["pobs"]], 2, chunk=3)
    res = eng.estep(g["A"], g["pi"], store_gamma=True)
    assert abs(res.loglik - (-4.6323247916806176)) < 1e-12
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL)
    np.testing.assert_allclose(eng.gamma(0), g["gamma"], rtol=RTOL)
    eng.close()
    g = golden("kat2_gauss3")  # 3 states -> padded to 4
    eng = _engine()
    eng.set_observations("gaussian", [g["obs"].astype(np.float64)], 3, chunk=37)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"], store_gamma=True)
    np.testing.assert_allclose(res.loglik, -15289.770127434271, rtol=1e-12)
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL)
    np.testing.assert_allclose(res.state_counts, g["state_counts"], rtol=RTOL)
    np.testing.assert_allclose(eng.gamma(0)[g["rows"]], g["gamma_rows"], rtol=1e-8, atol=1e-14)
    eng.close()


def test_doublewell_reference_trajectory(golden):
    g = golden("d2_doublewell")
    obs = g["obs"].astype(np.int32)
    eng = _engine()
    eng.set_observations("discrete", [obs], 2, nsymbols=g["B"].shape[1])
    res = eng.estep(g["A"], g["pi"], g["B"], store_gamma=True)
    np.testing.assert_allclose(res.loglik, float(g["logL"]), rtol=1e-11)
    np.testing.assert_allclose(res.C, g["C"], rtol=RTOL)
    np.testing.assert_allclose(res.state_counts, g["state_counts"], rtol=RTOL)
    np.testing.assert_allclose(eng.gamma(0)[g["rows"]], g["gamma_rows"], rtol=1e-8, atol=1e-14)
    eng.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
def test_random_models_all_state_counts(n):
    rng = np.random.default_rng(100 + n)
    A = rng.random((n, n)) + 0.05
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    mu = np.linspace(-2, 2, n) if n > 1 else np.array([0.3])
    sig = rng.uniform(0.4, 1.2, n)
    obs = [rng.normal(0, 2, T) for T in (513, 64, 1, 1000)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = _engine()
    eng.set_observations("gaussian", obs, n, chunk=50)
    res = eng.estep(A, pi, mu, sig)
    _cmp(res, ref, n)
    eng.close()


def test_repeated_estep_is_deterministic_and_reusable(golden):
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=16)
    r1 = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    r2 = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    assert np.array_equal(r1.packed, r2.packed)          # fixed-order reductions
    A2 = 0.5 * g["A"] + 0.5 / 8
    r3 = eng.estep(A2, g["pi"], g["mu"] - 0.2, g["sigma"] * 1.1)
    ref = orc.estep("gaussian", obs, A2, g["pi"], g["mu"] - 0.2, g["sigma"] * 1.1)
    _cmp(r3, ref, 8)
    eng.close()


def test_nonfinite_likelihood_is_reported():
    # a symbol no state can emit: c_t == 0 -> logL = -inf in the reference (_hidden.c:57-62),
    # and MaximumLikelihoodEstimator asserts (maximum_likelihood.py:385)
    A = np.array([[0.9, 0.1], [0.2, 0.8]])
    B = np.array([[0.5, 0.5, 0.0], [0.3, 0.7, 0.0]])
    obs = [np.array([0, 1, 2, 1, 0], dtype=np.int32)]
    eng = _engine()
    eng.set_observations("discrete", obs, 2, nsymbols=3)
    with pytest.raises(AssertionError):
        eng.estep(A, np.array([0.5, 0.5]), B)
    eng.close()


def test_full_size_properties():
    """BASELINE configs[1] shape (8 states, 256 x 1e5 Gaussian): size-independent checks.
    (1) sum_t gamma == T per trajectory, sum C == T-1;  (2) a sub-batch run separately gives
    the same per-trajectory log-likelihoods (chunking differs: auto chunk depends on total);
    (3) one trajectory against the oracle."""
    import torch
    from bench import make_c2_model, synth_gaussian_device
    K, T = 256, 100000
    model = make_c2_model()
    obs_dev = synth_gaussian_device(model, K, T, seed=5, device="cuda:0")
    off = np.arange(K + 1, dtype=np.int64) * T
    eng = _engine()
    eng.set_observations_device("gaussian", obs_dev.data_ptr(), off, 8)
    res = eng.estep(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
    assert np.all(np.isfinite(res.logL_k))
    np.testing.assert_allclose(res.state_counts.sum(), K * T, rtol=1e-10)
    np.testing.assert_allclose(res.C.sum(), K * (T - 1), rtol=1e-10)
    np.testing.assert_allclose(res.gamma0_sum.sum(), K, rtol=1e-10)
    sub = obs_dev[: 3 * T].contiguous()
    eng2 = _engine()
    eng2.set_observations_device("gaussian", sub.data_ptr(), off[:4], 8, chunk=777)
    res2 = eng2.estep(model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
    np.testing.assert_allclose(res2.logL_k, res.logL_k[:3], rtol=1e-11)
    o0 = obs_dev[:T].cpu().numpy()
    ref = orc.estep("gaussian", [o0], model["A_eval"], model["pi"], model["mu_eval"],
                    model["sigma"])
    np.testing.assert_allclose(res.logL_k[0], ref["logL"][0], rtol=1e-11)
    eng.close()
    eng2.close()
    del obs_dev, sub
    torch.cuda.empty_cache()


def test_speculative_boundaries_verify_or_fall_back(golden):
    """The E-step first tries chunk boundaries obtained by warm-up (no prescan / stitch) and
    verifies them; a warm-up that is too short must be detected and the exact pipeline used."""
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    ref = orc.estep("gaussian", obs, g["A"], g["pi"], g["mu"], g["sigma"])
    # (1) far too short a warm-up: every boundary is wrong -> detected, exact fallback
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=16)
    eng.set_option("spec_W", 2)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    assert eng.get_option("spec_fail") == 1 and eng.get_option("spec_ok") == 0
    assert eng.get_option("spec_last_dev") > 1e-6
    _cmp(res, ref, 8)
    assert eng.get_option("spec_W") > 2                  # lengthened for the next call
    for _ in range(4):
        res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
        _cmp(res, ref, 8)
    # the warm-up grows until it verifies (or speculation is switched off); either way exact
    assert eng.get_option("spec_ok") >= 1 or eng.get_option("spec_enabled") == 0
    eng.close()
    # (2) normal case: verified at once, and identical to the exact pipeline within 1e-10
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=500)
    r_spec = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    assert eng.get_option("spec_ok") == 1 and eng.get_option("spec_last_dev") <= 1e-11
    eng.set_option("spec_enabled", 0)
    r_exact = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    np.testing.assert_allclose(r_spec.packed, r_exact.packed, rtol=1e-10, atol=1e-12)
    _cmp(r_spec, ref, 8)
    eng.close()
    # (3) a chain that does not forget (near-identity A, uninformative emissions): speculation
    # cannot verify, is abandoned, results stay exact
    A = np.full((4, 4), 1e-7)
    np.fill_diagonal(A, 1.0 - 3e-7)
    pi = np.array([0.4, 0.3, 0.2, 0.1])
    mu, sig = np.array([0.0, 0.01, 0.02, 0.03]), np.ones(4)
    rng = np.random.default_rng(0)
    o = [rng.normal(0, 1, 6000)]
    ref = orc.estep("gaussian", o, A, pi, mu, sig)
    eng = _engine()
    eng.set_observations("gaussian", o, 4, chunk=50)
    for _ in range(6):
        res = eng.estep(A, pi, mu, sig)
        _cmp(res, ref, 4)
    assert eng.get_option("spec_fail") >= 1
    eng.close()


def test_exp_of_gaussian_density_is_ulp_accurate():
    """The E-step evaluates exp(-z^2/2) (_gaussian.c:18) with its own branch-free kernel for
    non-positive arguments: bound its error against libm over the whole range, including the
    gradual underflow and the clamp."""
    import ctypes
    from bhmm_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([
        -rng.uniform(0.0, 50.0, 200000), -rng.uniform(0.0, 1e-3, 20000),
        -rng.uniform(50.0, 745.0, 100000), -np.exp(rng.uniform(-40, 0, 20000)),
        np.array([0.0, -0.0, -1e-300, -708.3, -745.13, -745.2, -750.0, -1e4, -1e300, -np.inf]),
        -np.log(2.0) * (np.arange(2000) + 0.5)])  # reduction boundaries
    y = np.empty_like(x)
    _lib.check(L.bhmm_diag_exp_nonpos(_lib.dp(y), _lib.dp(x), x.size))
    ref = np.exp(x)
    normal = ref > 2.3e-308
    ulp = np.abs(y[normal] - ref[normal]) / np.spacing(ref[normal])
    assert ulp.max() <= 1.0, ulp.max()
    # gradual underflow: absolute error of one denormal spacing, exact zero below the range
    assert np.all(np.abs(y[~normal] - ref[~normal]) <= 4.95e-324 * 1.5)
    assert np.all(y[x < -746.0] == 0.0)
    assert y[0 + np.flatnonzero(x == 0.0)[0]] == 1.0


@pytest.mark.parametrize("mu,sigma", [(0.3, 1.7), (-5.0, 1e-3), (100.0, 25.0), (1.0, 1e3), (0.0, 1.0)])
def test_gaussian_density_of_the_sweep(mu, sigma):
    """gauss_pdf (estep_sweep.hpp): the density of _gaussian.c:18-20 with constant, exponent and
    range reduction fused.  Against 80-bit arithmetic: a few 1e-16 plus the rounding of the
    exponent's argument (which libm's exp(-z*z/2) carries as well); exact zeros far out and for
    infinite observations; NaN is kept apart from a hit by the variant of the checked kernels."""
    from bhmm_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(11)
    z = np.concatenate([rng.uniform(-8, 8, 200000), rng.uniform(-38.6, 38.6, 100000),
                        rng.normal(0, 1e-6, 1000), rng.uniform(-47, 47, 20000),
                        np.array([0.0, 1e-160, 39.0, 45.0, 60.0, 91.0, 1e5, 1e9, 1e150, 1e200])])
    o = mu + sigma * z
    for nansafe in (0, 1):
        y = np.empty_like(o)
        _lib.check(L.bhmm_diag_gauss_pdf(_lib.dp(y), _lib.dp(o), o.size, mu, sigma, nansafe))
        ol, ml, sl = o.astype(np.longdouble), np.longdouble(mu), np.longdouble(sigma)
        x = ((ol - ml) / sl) ** 2 / 2
        ref = np.exp(-x) / (np.sqrt(2 * np.pi * np.longdouble(1)) * sl)
        normal = ref > 2.3e-308
        rel = np.abs((y[normal] - ref[normal]) / ref[normal]).astype(float)
        bound = 6e-16 + 4e-16 * x[normal].astype(float)
        assert np.all(rel <= bound), (rel / bound).max()
        sub = ref[~normal].astype(float)
        assert np.all(np.abs(y[~normal] - sub) <= 4.95e-324 * 2 + 3e-13 * sub)
        assert np.all(y[np.abs(z) > 50] == 0.0)
        special = np.array([np.inf, -np.inf, np.nan])
        ys = np.empty(3)
        _lib.check(L.bhmm_diag_gauss_pdf(_lib.dp(ys), _lib.dp(special), 3, mu, sigma, nansafe))
        assert ys[0] == 0.0 and ys[1] == 0.0
        if nansafe:
            assert ys[2] == 0.0  # -> all-zero row -> fix_outlier restores the NaN
    # an invalid sigma poisons every density
    for bad in (0.0, -1.0, np.nan, np.inf):
        yb = np.empty(4)
        _lib.check(L.bhmm_diag_gauss_pdf(_lib.dp(yb), _lib.dp(np.array([0.0, 1.0, -3.0, 1e9])), 4, mu, bad, 0))
        assert np.all(np.isnan(yb))


@pytest.mark.parametrize("want_gamma", [False, True])
def test_infinite_observations_are_outliers(want_gamma):
    """+-inf observations have density 0 for every state: the outlier rule (outputmodel.py:126-130)
    makes the row all ones, on the padded 3-state model (4 lanes) as on the 8-state one, in the
    branch-free and in the checked kernels."""
    rng = np.random.default_rng(21)
    for n in (3, 8):
        A = rng.random((n, n)) + 2 * np.eye(n)
        A /= A.sum(axis=1, keepdims=True)
        pi = rng.dirichlet(np.ones(n))
        mu, sig = np.linspace(-2, 2, n), rng.uniform(0.5, 1.5, n)
        obs = [rng.normal(0, 2, T) for T in (300, 171, 64)]
        obs[0][17] = np.inf
        obs[1][5] = -np.inf
        obs[1][170] = np.inf
        ref = orc.estep("gaussian", obs, A, pi, mu, sig, want_gamma=want_gamma)
        eng = _engine()
        eng.set_observations("gaussian", obs, n, chunk=32)
        for _ in range(2):
            res = eng.estep(A, pi, mu, sig, store_gamma=want_gamma)
            np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
            np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
        eng.close()


@pytest.mark.parametrize("n", [1, 2, 5, 12, 64])
def test_densities_in_the_denormal_range(n):
    """Observations 37-39 sigma from every state: densities of 1e-310 .. 1e-324.  The reference divides
    by the (denormal) row sums; a reciprocal of such a sum is infinite, and a density with constant
    and exponential fused rounds differently from exp-then-scale down there.  The checked kernels
    re-evaluate such rows in the reference's operation order and carry them times 2^900: results
    stay finite, every step still contributes unit mass to the counts, and with one state (where the
    reference's own arithmetic is exact) the log-likelihood is the reference's."""
    rng = np.random.default_rng(5)
    mu = np.linspace(-3.2, -2.2, n) if n > 1 else np.array([-2.9472305])
    sig = np.full(n, 0.3) if n > 1 else np.array([0.32470804])
    A = rng.random((n, n)) + np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = np.full(n, 1.0 / n)
    obs = [rng.normal(0, 3, T) for T in (26419, 15186, 3754)]
    with np.errstate(all="ignore"):
        ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    if n <= 12:
        from ld_reference import estep_longdouble
        ld_logL, ld_C = estep_longdouble(A, pi, [orc.pobs_gaussian(o, mu, sig) for o in obs])
    for chunk in (7, 0):
        eng = _engine()
        eng.set_observations("gaussian", obs, n, chunk=chunk)
        for _ in range(2):
            res = eng.estep(A, pi, mu, sig)
            assert np.all(np.isfinite(res.logL_k)) and np.all(np.isfinite(res.C))
            np.testing.assert_allclose(res.C.sum(), sum(len(o) - 1 for o in obs), rtol=1e-10)
            np.testing.assert_allclose(res.state_counts.sum(), sum(len(o) for o in obs), rtol=1e-10)
            # the reference's own sums are coarsely rounded denormals there: 1e-6 is its noise ...
            if np.all(np.isfinite(ref["logL"])):
                np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-6)
            # ... so the comparison that is tight is the one with the reference's recursions carried
            # out in 80-bit arithmetic on the reference's own double-precision emission rows
            if n <= 12:
                np.testing.assert_allclose(res.logL_k, ld_logL, rtol=1e-11)
                np.testing.assert_allclose(res.C, ld_C, rtol=1e-8, atol=1e-9)
            if n == 1:
                np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-13)
                assert res.C[0, 0] == sum(len(o) - 1 for o in obs)
        eng.close()


@pytest.mark.parametrize("case", ["stress_case_53_204", "stress_case_51_278", "stress_case_55_240",
                                  "stress_case_72_35"])
def test_cases_found_by_the_random_sweep(case):
    """Inputs on which tests/sweeps/stress_small.py (random parity sweep against the oracle) found defects:
    sparse transition matrices with very narrow states -- p o beta in the denormal range although
    neither factor is (NaN counts; 6 and 12 states) -- and an absorbing state among far outliers
    (zero rows of the chunk transfer matrices took part in the exponent alignment of the exact
    boundary pass: non-finite likelihood whenever the speculation gave up); alpha concentrated on a
    state whose A (p o beta) is denormal while p o beta as a whole is not (24 states: NaN counts)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "stress", case + ".npz"))
    A, pi, mu, sig = d["A"], d["pi"], d["par0"], d["par1"]
    obs = np.split(d["obs"], np.cumsum(d["lens"])[:-1])
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    for chunk in (int(d["chunk"]), 0):
        for sg in (False, True):
            eng = _engine()
            eng.set_observations("gaussian", obs, len(mu), chunk=chunk)
            for _ in range(2):
                res = eng.estep(A, pi, mu, sig, store_gamma=sg)
                np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
                np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
                np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10)
            eng.close()


def test_sampled_paths_with_a_sparse_transition_matrix():
    """The map kernels of the Gibbs step also draw for next states the path cannot take; with zeros
    in A such a draw has no possible predecessor.  It must not raise (only a walk that really uses
    such an entry fails): same paths as the reference for the same uniforms."""
    rng = np.random.default_rng(8)
    n = 6
    for trial in range(6):
        A = rng.random((n, n)) * (rng.random((n, n)) < 0.4)
        A[np.arange(n), np.arange(n)] = rng.random(n) + 0.5
        A /= A.sum(axis=1, keepdims=True)
        pi = rng.dirichlet(np.ones(n))
        mu, sig = np.linspace(-4, 4, n), rng.uniform(0.5, 1.5, n)
        obs = [rng.normal(0, 3, T) for T in (700, 33, 1500)]
        u = [rng.random(len(o)) for o in obs]
        ref = [orc.sample_path(orc.forward(A, orc.pobs_gaussian(o, mu, sig), pi)[1], A, u=uu)
               for o, uu in zip(obs, u)]
        eng = _engine()
        eng.set_observations("gaussian", obs, n, chunk=int(rng.choice([0, 16, 100])))
        paths, C, n0, _ = eng.sample_paths(A, pi, mu, sig, u=u)
        assert all(np.array_equal(a, b) for a, b in zip(paths, ref))
        Cr, n0r = orc.path_counts(ref, n)
        assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
        eng.close()


def test_nan_observation_is_not_a_hit(golden):
    """A NaN observation must poison its trajectory (the reference's pobs row is NaN) on every
    path that evaluates the density with the clamp modifier: the upload finds it and keeps the
    context on the checked kernels."""
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    obs = [o.copy() for o in obs]
    obs[1][len(obs[1]) // 2] = np.nan
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=16)
    assert eng.get_option("careful") == 1.0
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    with pytest.raises(AssertionError):
        eng.estep(*margs)
    ok = [o for i, o in enumerate(obs) if i != 1]
    eng.set_observations("gaussian", ok, 8, chunk=16)
    assert eng.get_option("careful") == 0.0
    res = eng.estep(*margs)
    ref = orc.estep("gaussian", ok, *margs)
    np.testing.assert_allclose(res.loglik, ref["logL"].sum(), rtol=1e-11)
    eng.close()


# ---------------------------------------------------------------------------------------------
# The statistics-only E-step (no gamma rows) runs the branch-free instantiation of k_estep: every
# second alpha row in HBM, rescaling every 4th step, zero / tiny vectors only reported.  The tests
# above ask for gamma and therefore run the careful instantiation; these cover the other one over
# every alignment of chunk length, unroll group and trajectory end.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec", [1, 0])
@pytest.mark.parametrize("chunk", [0, 1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 17, 23, 33, 64])
def test_statistics_only_estep_all_alignments(golden, chunk, spec):
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    ref = orc.estep("gaussian", obs, g["A"], g["pi"], g["mu"], g["sigma"], want_gamma=True)
    eng = _engine()
    eng.set_option("spec_enabled", spec)
    eng.set_observations("gaussian", obs, 8, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
    assert eng.get_option("careful") == 0.0
    np.testing.assert_allclose(res.logL_k, g["logL"], rtol=RTOL)
    _cmp(res, ref, 8)
    sd, sdd = _gauss_stats(obs, ref["gammas"], g["mu"])
    np.testing.assert_allclose(res.sum_gd, sd, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(res.sum_gdd, sdd, rtol=1e-8)
    # same numbers as the careful instantiation up to rounding
    res2 = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"], store_gamma=True)
    np.testing.assert_allclose(res.packed, res2.packed, rtol=1e-11, atol=1e-13)
    eng.close()


@pytest.mark.parametrize("chunk", [0, 2, 5, 8, 13, 33])
def test_statistics_only_estep_discrete_and_explicit(golden, chunk):
    g = golden("d8_ragged")
    obs = split(g["obs"], g["lengths"])
    M = g["B"].shape[1]
    ref = orc.estep("discrete", obs, g["A"], g["pi"], g["B"], want_gamma=True)
    eng = _engine()
    eng.set_observations("discrete", obs, 8, nsymbols=M, chunk=chunk)
    res = eng.estep(g["A"], g["pi"], g["B"])
    np.testing.assert_allclose(res.logL_k, g["logL"], rtol=RTOL)
    _cmp(res, ref, 8)
    cnt = np.zeros_like(g["B"])
    for o, gm in zip(obs, ref["gammas"]):
        for s in range(M):
            cnt[:, s] += gm[o == s].sum(axis=0)
    np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
    eng.close()
    # caller-supplied pobs rows (hidden/api.py signatures)
    pobs = [np.ascontiguousarray(g["B"][:, o].T) for o in obs]
    eng = _engine()
    eng.set_observations("explicit", pobs, 8, chunk=chunk)
    res = eng.estep(g["A"], g["pi"])
    np.testing.assert_allclose(res.logL_k, g["logL"], rtol=RTOL)
    _cmp(res, ref, 8)
    eng.close()


@pytest.mark.parametrize("chunk", [0, 4, 9])
def test_outlier_rows_switch_to_the_careful_kernel(golden, chunk):
    """An all-zero emission row (outputmodel.py:126-130) is only *reported* by the branch-free
    kernel; the library must repeat the E-step with the kernel that applies the rule, give the
    reference's numbers, and stay on that kernel for the data set."""
    g = golden("g8_outliers")
    eng = _engine()
    eng.set_observations("gaussian", [g["obs"]], 8, chunk=chunk)
    assert eng.get_option("careful") == 0.0
    for _ in range(2):
        res = eng.estep(g["A"], g["pi"], g["mu"], g["sigma"])
        assert eng.get_option("careful") == 1.0
        np.testing.assert_allclose(res.loglik, float(g["logL"]), rtol=RTOL)
        np.testing.assert_allclose(res.C, g["C"], rtol=RTOL, atol=1e-12)
    # new observations start on the fast kernel again
    g2 = golden("g8_ragged")
    eng.set_observations("gaussian", split(g2["obs"], g2["lengths"]), 8, chunk=chunk)
    eng.estep(g2["A"], g2["pi"], g2["mu"], g2["sigma"])
    assert eng.get_option("careful") == 0.0
    eng.close()


def test_tiny_emissions_are_rescaled_in_time():
    """Emission densities around 1e-150 per step: four unscaled steps would underflow.  The
    branch-free kernel has to notice (its running maximum falls below 2^-800) and hand over to the
    kernel that rescales every step; the result must match the oracle."""
    rng = np.random.default_rng(11)
    n, T = 4, 600
    A = rng.random((n, n)) + np.eye(n) * 3
    A /= A.sum(axis=1, keepdims=True)
    pi = np.full(n, 1.0 / n)
    mu = np.array([-1.0, 0.0, 1.0, 2.0])
    sigma = np.full(n, 0.02)          # observations up to 25 sigma from the nearest mean:
    obs = [rng.uniform(-1.5, 2.5, T) for _ in range(3)]  # densities down to 1e-136, no denormals
    ref = orc.estep("gaussian", obs, A, pi, mu, sigma)
    eng = _engine()
    eng.set_observations("gaussian", obs, n, chunk=64)
    res = eng.estep(A, pi, mu, sigma)
    assert eng.get_option("careful") == 1.0
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=RTOL)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-12)
    eng.close()


def test_warmup_length_is_measured_not_guessed():
    """The first E-step on new observations measures the forgetting curve of the model at hand
    and takes the warm-up length from it; the boundary check must then pass at once, with a
    deviation below the tolerance but not absurdly far below it (i.e. the warm-up is not
    wastefully long).  A caller-supplied length is left alone."""
    rng = np.random.default_rng(21)
    n, K, T = 6, 4, 30000
    mu, sig = np.linspace(-4, 4, n), np.full(n, 1.0)
    pi = np.full(n, 1.0 / n)
    res = {}
    for stay in (0.7, 0.97):           # fast / slowly mixing chain
        A = np.full((n, n), (1 - stay) / (n - 1))
        np.fill_diagonal(A, stay)
        s = np.zeros((K, T), dtype=int)
        u = rng.random((K, T))
        for t in range(1, T):
            s[:, t] = np.where(u[:, t] < stay, s[:, t - 1], rng.integers(0, n, K))
        obs = [mu[s[k]] + sig[s[k]] * rng.standard_normal(T) for k in range(K)]
        eng = _engine()
        eng.set_observations("gaussian", obs, n, chunk=2000)
        r = eng.estep(A, pi, mu, sig)
        assert eng.get_option("spec_ok") == 1 and eng.get_option("spec_fail") == 0
        assert 1e-16 < eng.get_option("spec_last_dev") <= 1e-11
        res[stay] = eng.get_option("spec_W")
        ref = orc.estep("gaussian", obs, A, pi, mu, sig)
        np.testing.assert_allclose(r.logL_k, ref["logL"], rtol=RTOL)
        # a fixed length is respected
        eng.set_option("spec_W", 400)
        eng.set_observations("gaussian", obs, n, chunk=2000)
        eng.estep(A, pi, mu, sig)
        assert eng.get_option("spec_W") == 400
        eng.close()
    assert res[0.97] > res[0.7]        # the slower chain needs the longer warm-up


@pytest.mark.parametrize("seed", range(12))
def test_randomised_models_and_shapes(seed):
    """Random state counts (incl. the padded ones), trajectory counts, ragged lengths, chunk
    lengths and emission kinds, default settings (probe, speculative boundaries, branch-free
    kernel): statistics against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([2, 3, 4, 5, 7, 8]))
    K = int(rng.integers(1, 9))
    lengths = rng.integers(1, 2500, K)
    if seed % 4 == 0:
        lengths[0] = 1                                  # T = 1 trajectory
    A = rng.random((n, n)) + np.eye(n) * rng.uniform(0, 6)
    if seed % 3 == 0:
        A[rng.integers(0, n), rng.integers(0, n)] = 0.0  # a structural zero
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    chunk = int(rng.choice([0, 0, 3, 10, 37, 128, 1000]))
    kind = ["gaussian", "discrete", "explicit"][seed % 3]
    eng = _engine()
    if kind == "gaussian":
        mu, sig = np.sort(rng.normal(0, 3, n)), rng.uniform(0.3, 2.0, n)
        obs = [rng.normal(0, 4, T) for T in lengths]
        ref = orc.estep("gaussian", obs, A, pi, mu, sig)
        eng.set_observations("gaussian", obs, n, chunk=chunk)
        res = eng.estep(A, pi, mu, sig)
    else:
        M = int(rng.integers(2, 12))
        B = rng.dirichlet(np.ones(M), size=n)
        sym = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        ref = orc.estep("discrete", sym, A, pi, B)
        if kind == "discrete":
            eng.set_observations("discrete", sym, n, nsymbols=M, chunk=chunk)
            res = eng.estep(A, pi, B)
        else:
            eng.set_observations("explicit", [np.ascontiguousarray(B[:, o].T) for o in sym], n,
                                 chunk=chunk)
            res = eng.estep(A, pi)
    _cmp(res, ref, n)
    # and the same numbers when called again (run-to-run identical)
    again = eng.estep(*((A, pi, mu, sig) if kind == "gaussian" else
                        ((A, pi, B) if kind == "discrete" else (A, pi))))
    assert np.array_equal(res.packed, again.packed)
    eng.close()


def test_carried_boundary_vectors_discrete_kind():
    """The same for discrete emissions (the backward sweep runs in groups of eight steps there)."""
    rng = np.random.default_rng(78)
    n, M, K, T = 8, 40, 16, 60000
    A0 = rng.random((n, n)) + 5 * np.eye(n)
    A0 /= A0.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    B0 = rng.dirichlet(np.ones(M) * 0.5, size=n)
    obs = [rng.integers(0, M, T).astype(np.int32) for _ in range(K)]
    a, b = _engine(), _engine()
    a.set_observations("discrete", obs, n, nsymbols=M, chunk=2048)
    b.set_observations("discrete", obs, n, nsymbols=M, chunk=2048)
    b.set_option("carry", 0)
    used = []
    for it in range(12):
        drift = 2e-3 * 0.6 ** it
        A = A0 * (1 + drift * rng.normal(size=(n, n)))
        A /= A.sum(axis=1, keepdims=True)
        B = B0 * (1 + drift * rng.normal(size=(n, M)))
        B /= B.sum(axis=1, keepdims=True)
        ra, rb = a.estep(A, pi, B), b.estep(A, pi, B)
        used.append(int(a.get_option("carry_W")))
        np.testing.assert_allclose(ra.packed, rb.packed, rtol=1e-9, atol=1e-9)
    assert a.get_option("carry_ok") >= 5 and a.get_option("carry_fail") == 0, used
    ref = orc.estep("discrete", obs[:2], A, pi, B)
    np.testing.assert_allclose(ra.logL_k[:2], ref["logL"], rtol=1e-10)
    a.close()
    b.close()


def test_em_sequence_on_carried_boundary_vectors():
    """Round 3: in a sequence of E-steps on slowly changing models the warm-ups start from the
    PREVIOUS E-step's boundary vectors, a shorter distance out (estep_sweep.hpp: Carry).  Same
    statistics as with full warm-ups (both verified to 1e-11) and as the oracle; never used when the
    model did not change; a deliberately wrong sensitivity bound makes the check fail, the E-step is
    repeated with full warm-ups and still returns the right statistics."""
    rng = np.random.default_rng(77)
    n, K, T = 8, 24, 40000
    A0 = rng.random((n, n)) + 6 * np.eye(n)
    A0 /= A0.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    mu0, sig0 = np.linspace(-4, 4, n), np.linspace(0.6, 1.4, n)
    obs = [rng.normal(mu0[rng.integers(0, n, T)], 1.0) for _ in range(K)]
    a, b = _engine(), _engine()
    a.set_observations("gaussian", obs, n, chunk=2048)      # (chunks long against the warm-up)
    b.set_observations("gaussian", obs, n, chunk=2048)
    b.set_option("carry", 0)
    used = []
    for it in range(14):
        drift = 3e-3 * 0.6 ** it
        A = A0 * (1 + drift * rng.normal(size=(n, n)))
        A /= A.sum(axis=1, keepdims=True)
        mu = mu0 + drift * rng.normal(size=n)
        sig = sig0 * (1 + drift * rng.normal(size=n))
        ra, rb = a.estep(A, pi, mu, sig), b.estep(A, pi, mu, sig)
        used.append(int(a.get_option("carry_W")))
        np.testing.assert_allclose(ra.packed, rb.packed, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ra.logL_k, rb.logL_k, rtol=1e-12)
    W = a.get_option("spec_W")
    assert a.get_option("carry_ok") >= 6 and a.get_option("carry_fail") == 0, used
    assert 0 < min(u for u in used if u > 0) < 0.8 * W, (used, W)
    assert b.get_option("carry_ok") == 0 and b.get_option("carry_W") == 0
    ref = orc.estep("gaussian", obs[:3], A, pi, mu, sig)
    np.testing.assert_allclose(ra.logL_k[:3], ref["logL"], rtol=1e-10)
    # the same model again: nothing to gain from carried vectors that are exact -- full warm-ups
    a.estep(A, pi, mu, sig)
    assert a.get_option("carry_W") == 0
    # a sensitivity bound that is far too optimistic: the short warm-up fails the check, the
    # E-step is repeated with full warm-ups
    a.set_option("carry_kappa", 1e-12)
    a.estep(A, pi, mu + 1e-7, sig)        # (full warm-ups; captures for a warm-up of the minimum length)
    assert a.get_option("carry_W") == 0
    fails = a.get_option("carry_fail")
    A2 = A * (1 + 2e-2 * rng.random((n, n)))
    A2 /= A2.sum(axis=1, keepdims=True)
    r2 = a.estep(A2, pi, mu + 0.02, sig)
    assert a.get_option("carry_fail") == fails + 1
    ref = orc.estep("gaussian", obs[:3], A2, pi, mu + 0.02, sig)
    np.testing.assert_allclose(r2.logL_k[:3], ref["logL"], rtol=1e-10)
    np.testing.assert_allclose(r2.state_counts.sum(), K * T, rtol=1e-10)
    a.close()
    b.close()


def test_explicit_rows_of_1e_minus_222_do_not_underflow_the_rebuilt_alpha_row():
    """tests/sweeps/stress_small.py seed 7 case 921 (saved under tests/golden/cases): explicit emission rows
    like [0, 1e-222] followed by [9e-126, 1e-198].  The per-step-checked backward sweep rebuilds every
    second alpha row from its predecessor; left at the magnitude of its emission row, its product with
    A (p o beta) underflowed to zero and gamma / the counts were NaN where the reference is finite."""
    import os
    from bhmm_amd.engine import Engine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cases", "explicit2_nan_counts_7_921.npz"))
    A, pi, lens = d["A"], d["pi"], d["lens"]
    pobs = np.split(d["pobs"], np.cumsum(lens)[:-1])
    n = A.shape[0]
    Cref = np.zeros((n, n))
    lls = []
    for p in pobs:
        ll, al = orc.forward(A, p, pi)
        be = orc.backward(A, p)
        Cref += orc.transition_counts(al, be, A, p)
        lls.append(ll)
    assert np.all(np.isfinite(Cref))
    for chunk in (int(d["chunk"]), 0):
        eng = Engine(0)
        eng.set_observations("explicit", pobs, n, chunk=chunk)
        res = eng.estep(A, pi, None, None, store_gamma=True)
        np.testing.assert_allclose(res.logL_k, lls, rtol=1e-11)
        np.testing.assert_allclose(res.C, Cref, rtol=1e-8, atol=1e-10)
        g = eng.gamma(0)
        np.testing.assert_allclose(g, orc.gamma(orc.forward(A, pobs[0], pi)[1], orc.backward(A, pobs[0])),
                                   rtol=1e-8, atol=1e-12)
        eng.close()


@pytest.mark.parametrize("spec,store_gamma", [(1, False), (1, True), (0, False)])
def test_discrete_emission_probabilities_spread_over_hundreds_of_decades(spec, store_gamma):
    """tests/sweeps/stress_small.py seed 8001 case 1411 (saved): B with entries down to 2e-294.  The discrete
    kind keeps every fourth alpha row and rebuilds three from each; unscaled, three steps of probabilities
    of 1e-150 took the rebuilt rows to zero (S = 0: NaN counts) in the branch-free AND the per-step-checked
    kernel.  The checked kernel now rescales every rebuilt row; the branch-free one reports and is repeated."""
    import os
    from bhmm_amd.engine import Engine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cases", "disc5_tiny_B_8001_1411.npz"),
                allow_pickle=True)
    A, pi, B, lens = d["A"], d["pi"], d["par0"], d["lens"]
    obs = [o.astype(np.int32) for o in np.split(d["obs"], np.cumsum(lens)[:-1])]
    ref = orc.estep("discrete", obs, A, pi, B, None, want_gamma=True)
    assert np.all(np.isfinite(ref["C"]))
    eng = Engine(0)
    eng.set_option("spec_enabled", spec)
    eng.set_observations("discrete", obs, A.shape[0], nsymbols=B.shape[1], chunk=int(d["chunk"]))
    for _ in range(2):
        res = eng.estep(A, pi, B, store_gamma=store_gamma)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10)
        Bn = orc.estimate_discrete(obs, ref["gammas"], B.shape[1])       # discrete.py:202-215 on the reference's gammas
        np.testing.assert_allclose(res.symbol_counts / res.symbol_counts.sum(axis=1)[:, None], Bn, rtol=1e-8, atol=1e-12)
    if store_gamma:
        np.testing.assert_allclose(eng.gamma(0), ref["gammas"][0], rtol=1e-8, atol=1e-12)
    eng.close()


def test_reducible_model_with_wide_emissions_is_repeated_on_one_chunk_per_trajectory():
    """tests/sweeps/stress_small.py seed 9201 case 840 (saved): A = I with emission probabilities hundreds
    of decades apart.  The reference's gamma depends on the order in which its sequential recursions lose
    a block; the chunked evaluation came out 0/0.  The library notices the
    non-finite counts, re-plans with one chunk per trajectory -- the sequential recursions -- and repeats
    the E-step: finite, and equal to the reference's."""
    import os
    from bhmm_amd.engine import Engine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cases", "disc2_tiny_B_9201_840.npz"),
                allow_pickle=True)
    A, pi, B, lens = d["A"], d["pi"], d["par0"], d["lens"]
    obs = [o.astype(np.int32) for o in np.split(d["obs"], np.cumsum(lens)[:-1])]
    ref = orc.estep("discrete", obs, A, pi, B, None)
    assert np.all(np.isfinite(ref["C"]))
    eng = Engine(0)
    eng.set_observations("discrete", obs, 2, nsymbols=B.shape[1], chunk=int(d["chunk"]))
    assert eng.num_chunks > len(obs)
    for _ in range(2):
        res = eng.estep(A, pi, B)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10)
    assert eng.num_chunks == len(obs)
    eng.close()
    # the sharded estimator's call sequence (maximum_likelihood.py:271-282 in distributed form): the
    # statistics go to the CALLER's device buffer, bhmm_estep_fetch(logL_k only) waits for them -- and must
    # repair them there, before the all-reduce that follows (a rank that raised instead would leave the
    # others alone in the collective)
    import torch
    eng = Engine(0)
    eng.set_observations("discrete", obs, 2, nsymbols=B.shape[1], chunk=int(d["chunk"]))
    buf = torch.full((eng.stats_size,), -1.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    eng.estep_launch(A, pi, B, stats_dev=buf.data_ptr())
    logL_k = eng.estep_fetch_logL()
    assert eng.num_chunks == len(obs)
    res = eng.unpack(buf.cpu().numpy(), logL_k)
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-8, atol=1e-10)
    eng.close()


@pytest.mark.parametrize("case,kind", [("gauss8_denormal_entries_8101_681", "gaussian"),
                                       ("disc4_M1150_9501_20", "discrete"),
                                       ("gauss7_denormal_entries_16001_2087", "gaussian")])
@pytest.mark.parametrize("spec,store_gamma", [(1, False), (1, True), (0, False)])
def test_weight_on_entries_in_the_denormal_range(case, kind, spec, store_gamma):
    """Two saved cases of tests/sweeps/stress_small.py in which a sparse transition matrix puts the whole
    weight of a step on a state whose emission probability (seed 8101 case 681: single entries of a Gaussian
    row below 2^-1022 beside representable ones) or whose p o beta entry (seed 9501 case 20: beta spread over
    more than 300 decades) is in the denormal range.  The reference multiplies alpha A p beta left to right and
    divides by the sum of the same products, so the lost bits cancel and it stays within 2e-15 of the 80-bit
    recursion; products formed in another order, or rounded once more in A (p o beta), left counts off by 6e-4
    and 2e-6.  The per-step-checked kernel forms such rows times 2^900; the branch-free one notices weights
    alpha_i / S of 2^850 and reports."""
    import os
    from bhmm_amd.engine import Engine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cases", case + ".npz"), allow_pickle=True)
    A, pi, lens = d["A"], d["pi"], d["lens"]
    par0, par1 = d["par0"], (d["par1"] if d["par1"].size else None)
    obs = np.split(d["obs"], np.cumsum(lens)[:-1])
    if kind == "discrete":
        obs = [o.astype(np.int32) for o in obs]
    ref = orc.estep(kind, obs, A, pi, par0, par1, want_gamma=True)
    assert np.all(np.isfinite(ref["C"]))
    eng = Engine(0)
    eng.set_option("spec_enabled", spec)
    eng.set_observations(kind, obs, A.shape[0], nsymbols=par0.shape[1] if kind == "discrete" else 0,
                         chunk=int(d["chunk"]))
    # the log-likelihood of the Gaussian case: the reference's own row sums hold denormal terms there and it is
    # 2.3e-5 off the 80-bit recursion on the same (double) emission rows (the kernels, whose density rounds
    # differently below 2^-1022, 2e-6): no further from the 80-bit value than twice the reference's distance
    from ld_reference import estep_longdouble
    pobs = [orc.pobs_gaussian(o, par0, par1) if kind == "gaussian" else orc.pobs_discrete(o, par0) for o in obs]
    with np.errstate(all="ignore"):
        ld_logL, ld_C = estep_longdouble(A, pi, pobs)
    np.testing.assert_allclose(ref["C"], ld_C, rtol=1e-9, atol=1e-11)
    for _ in range(2):
        res = eng.estep(A, pi, par0, par1, store_gamma=store_gamma)
        ld = np.asarray(ld_logL, dtype=np.float64)
        assert np.all(np.abs(res.logL_k - ld) <= np.maximum(2.0 * np.abs(ref["logL"] - ld), 1e-10 * np.abs(ld)))
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9, atol=1e-11)
    if store_gamma:
        np.testing.assert_allclose(eng.gamma(0), ref["gammas"][0], rtol=1e-8, atol=1e-12)
    eng.close()


def test_boundary_tolerance_option_shortens_the_warm_up():
    """spec_tol (N <= 8): the boundary check's tolerance, the warm-up calibrated a hundred times inside it.
    1e-9 gives a shorter warm-up than the default 1e-11, boundaries that verify, and statistics equal to
    the default's far inside the 1e-6 contract; values outside [1e-13, 1e-7] are refused."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(77)
    n = 8
    A = rng.random((n, n)) + np.eye(n) * 6.0
    A /= A.sum(axis=1)[:, None]
    pi = np.full(n, 1.0 / n)
    mu, sig = np.linspace(-4, 4, n), np.full(n, 1.3)
    obs = [rng.normal(0, 3, T) for T in (300000, 200000)]
    out = {}
    for tol in (1e-11, 1e-9):
        eng = Engine(0)
        eng.set_option("spec_tol", tol)
        assert eng.get_option("spec_tol") == tol
        eng.set_observations("gaussian", obs, n)
        r = eng.estep(A, pi, mu, sig)
        r = eng.estep(A, pi, mu, sig)
        assert eng.get_option("spec_fail") == 0 and eng.get_option("spec_last_dev") <= tol
        out[tol] = (r, eng.get_option("spec_W"))
        eng.close()
    assert out[1e-9][1] < out[1e-11][1]
    np.testing.assert_allclose(out[1e-9][0].logL_k, out[1e-11][0].logL_k, rtol=1e-9)
    np.testing.assert_allclose(out[1e-9][0].C, out[1e-11][0].C, rtol=1e-7, atol=1e-7)
    eng = Engine(0)
    with pytest.raises(Exception):
        eng.set_option("spec_tol", 1e-3)
    eng.close()
