"""Pins the CPU oracle (oracle/bhmm_oracle.c) to the reference.

Two anchors: (1) the golden fixtures captured from the reference's own C and Python
kernels (tests/golden/gen_golden.py); (2) oracle/_ref -- the reference C sources compiled
in place -- on fresh random inputs, where the restatement must be BIT-identical
(same operation order, no contraction).  Mirrors bhmm/tests/test_hidden.py:265-335.
"""
import hashlib

import numpy as np
import pytest

from conftest import split
from oracle import oracle as orc

RTOL = 1e-13


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_kat1_toy(golden):
    g = golden("kat1_toy")
    A, pi, pobs = g["A"], g["pi"], g["pobs"]
    logL, alpha = orc.forward(A, pobs, pi)
    beta = orc.backward(A, pobs)
    gam = orc.gamma(alpha, beta)
    assert logL == float(g["logL"]) == -4.6323247916806176
    assert np.array_equal(alpha, g["alpha"])
    assert np.array_equal(beta, g["beta"])
    np.testing.assert_allclose(gam, g["gamma"], rtol=RTOL)
    assert np.array_equal(orc.transition_counts(alpha, beta, A, pobs), g["C"])
    assert np.array_equal(orc.viterbi(A, pobs, pi), g["viterbi"])
    # path sampling: libc stream after srand(42), and the same draw from explicit uniforms
    assert np.array_equal(orc.sample_path(alpha, A, seed=42), g["sample_path_seed42"])
    assert np.array_equal(orc.sample_path(alpha, A, u=g["sample_u"]), g["sample_path_seed42"])
    np.testing.assert_allclose(orc.state_counts(gam), g["gamma"].sum(axis=0), rtol=RTOL)


def test_kat2_gauss3(golden):
    g = golden("kat2_gauss3")
    obs = g["obs"].astype(np.float64)
    pobs = orc.pobs_gaussian(obs, g["mu"], g["sigma"])
    logL, alpha = orc.forward(g["A"], pobs, g["pi"])
    beta = orc.backward(g["A"], pobs)
    gam = orc.gamma(alpha, beta)
    rows = g["rows"]
    assert logL == float(g["logL"])
    assert abs(logL - (-15289.770127434271)) < 1e-9
    assert np.array_equal(alpha[rows], g["alpha_rows"])
    assert np.array_equal(beta[rows], g["beta_rows"])
    np.testing.assert_allclose(gam[rows], g["gamma_rows"], rtol=RTOL)
    np.testing.assert_allclose(gam.sum(axis=0), g["state_counts"], rtol=1e-12)
    assert np.array_equal(orc.transition_counts(alpha, beta, g["A"], pobs), g["C"])
    v = orc.viterbi(g["A"], pobs, g["pi"])
    assert np.array_equal(v, g["viterbi"])
    assert sha1(v.astype(np.int32)) == str(g["viterbi_sha1"]) == \
        "a15f23bffe22a93b68d1750509c01ebdd839bac9"


@pytest.mark.parametrize("name,kind", [("g8_ragged", "gaussian"), ("d8_ragged", "discrete")])
def test_ragged_batches(golden, name, kind):
    g = golden(name)
    lengths = g["lengths"]
    obs = split(g["obs"], lengths)
    if kind == "gaussian":
        r = orc.estep(kind, obs, g["A"], g["pi"], g["mu"], g["sigma"], want_gamma=True)
    else:
        r = orc.estep(kind, obs, g["A"], g["pi"], g["B"], want_gamma=True)
    assert np.array_equal(r["logL"], g["logL"])
    np.testing.assert_allclose(r["C"], g["C"].sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(r["gamma0_sum"], g["gamma0"].sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(r["state_counts"], g["state_counts"].sum(axis=0), rtol=1e-12)
    vit = split(g["viterbi"], lengths)
    for k, o in enumerate(obs):
        if kind == "gaussian":
            pobs = orc.pobs_gaussian(o, g["mu"], g["sigma"])
        else:
            pobs = orc.pobs_discrete(o, g["B"])
        assert np.array_equal(orc.viterbi(g["A"], pobs, g["pi"]), vit[k])
    # emission M-step (gaussian.py:214-272 / discrete.py:159-215)
    if kind == "gaussian":
        mu, sig = orc.estimate_gaussian(obs, r["gammas"])
        np.testing.assert_allclose(mu, g["mu_new"], rtol=1e-12)
        np.testing.assert_allclose(sig, g["sigma_new"], rtol=1e-12)
        np.testing.assert_allclose(r["gammas"][4], g["gamma4"], rtol=RTOL)
        pobs0 = orc.pobs_gaussian(obs[0], g["mu"], g["sigma"])
        _, a0 = orc.forward(g["A"], pobs0, g["pi"])
        assert np.array_equal(orc.sample_path(a0, g["A"], u=g["sample_u0"]),
                              g["sample_path0_seed7"])
        assert np.array_equal(orc.libc_uniforms(len(obs[0]), 7), g["sample_u0"])
    else:
        Bn = orc.estimate_discrete(obs, r["gammas"], g["B"].shape[1])
        np.testing.assert_allclose(Bn, g["B_new"], rtol=1e-12)
        np.testing.assert_allclose(r["gammas"][1], g["gamma1"], rtol=RTOL)


def test_outlier_rows(golden):
    g = golden("g8_outliers")
    pobs = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"])
    assert np.all(pobs[[0, 100, 256]] == 1.0)   # outputmodel.py:126-130
    logL, alpha = orc.forward(g["A"], pobs, g["pi"])
    beta = orc.backward(g["A"], pobs)
    assert logL == float(g["logL"])
    np.testing.assert_allclose(orc.gamma(alpha, beta), g["gamma"], rtol=RTOL)
    assert np.array_equal(orc.transition_counts(alpha, beta, g["A"], pobs), g["C"])
    assert np.array_equal(orc.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])
    raw = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"], ignore_outliers=False)
    assert np.all(raw[[0, 100, 256]] == 0.0)


def test_structural_zeros(golden):
    g = golden("d3_zeros")
    pobs = orc.pobs_discrete(g["obs"], g["B"])
    assert np.array_equal(pobs, g["pobs"]) if "pobs" in g.files else True
    logL, alpha = orc.forward(g["A"], pobs, g["pi"])
    beta = orc.backward(g["A"], pobs)
    assert logL == float(g["logL"])
    assert np.array_equal(alpha, g["alpha"]) and np.array_equal(beta, g["beta"])
    assert np.array_equal(orc.transition_counts(alpha, beta, g["A"], pobs), g["C"])
    assert np.array_equal(orc.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])


def test_doublewell_reference_data(golden):
    """Kernel-level check on the reference's own 100k-step test trajectory
    (bhmm/tests/data/2well_traj_100K.dat, used by bhmm/tests/test_mlhmm.py:32-45)."""
    g = golden("d2_doublewell")
    obs = g["obs"].astype(np.int32)
    assert obs.shape[0] == 99990 and obs.min() == 18 and obs.max() == 84
    pobs = orc.pobs_discrete(obs, g["B"])
    logL, alpha = orc.forward(g["A"], pobs, g["pi"])
    beta = orc.backward(g["A"], pobs)
    gam = orc.gamma(alpha, beta)
    rows = g["rows"]
    assert logL == float(g["logL"])
    assert np.array_equal(alpha[rows], g["alpha_rows"])
    assert np.array_equal(beta[rows], g["beta_rows"])
    np.testing.assert_allclose(gam[rows], g["gamma_rows"], rtol=RTOL)
    assert np.array_equal(orc.transition_counts(alpha, beta, g["A"], pobs), g["C"])
    v = orc.viterbi(g["A"], pobs, g["pi"])
    assert sha1(v.astype(np.int32)) == str(g["viterbi_sha1"])
    assert np.array_equal(np.packbits(v.astype(np.uint8)), g["viterbi_bits"])


def test_n64(golden):
    g = golden("g64")
    pobs = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"])
    logL, alpha = orc.forward(g["A"], pobs, g["pi"])
    beta = orc.backward(g["A"], pobs)
    assert logL == float(g["logL"])
    assert np.array_equal(alpha[g["rows"]], g["alpha_rows"])
    assert np.array_equal(orc.transition_counts(alpha, beta, g["A"], pobs), g["C"])
    assert np.array_equal(orc.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])


def test_pobs_gaussian_anchor(golden):
    """bhmm/tests/test_output_gaussian.py:43-59 (C vs python p_obs, allclose)."""
    g = golden("pobs_gauss3")
    p = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"])
    np.testing.assert_allclose(p, g["pobs"], rtol=1e-15)


def test_single_step_trajectory():
    A = np.array([[0.7, 0.3], [0.4, 0.6]])
    pi = np.array([0.25, 0.75])
    pobs = np.array([[0.2, 0.5]])
    logL, alpha = orc.forward(A, pobs, pi)
    assert np.isclose(logL, np.log(0.25 * 0.2 + 0.75 * 0.5))
    beta = orc.backward(A, pobs)
    assert np.array_equal(beta, [[0.5, 0.5]])
    assert np.array_equal(orc.transition_counts(alpha, beta, A, pobs), np.zeros((2, 2)))
    assert np.array_equal(orc.viterbi(A, pobs, pi), [1])


def test_path_counts():
    C, n0 = orc.path_counts([np.array([0, 0, 1, 2, 2, 2, 0]), np.array([1]), np.array([2, 1])], 3)
    assert np.array_equal(C, [[1, 1, 0], [0, 0, 1], [1, 1, 2]])
    assert np.array_equal(n0, [1, 1, 1])


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("N,T,seed", [(2, 50, 1), (3, 1000, 2), (8, 5000, 3), (17, 300, 4)])
def test_bit_identical_to_compiled_reference(N, T, seed):
    rng = np.random.default_rng(seed)
    A = rng.random((N, N))
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(N))
    mu = np.linspace(-3, 3, N)
    sig = rng.uniform(0.3, 1.5, N)
    obs = rng.normal(0, 3, T)
    pobs = orc.pobs_gaussian(obs, mu, sig, ignore_outliers=False)
    assert np.array_equal(pobs, orc.ref_pobs_gaussian(obs, mu, sig))
    logL, alpha = orc.forward(A, pobs, pi)
    rl, ra = orc.ref_forward(A, pobs, pi)
    assert logL == rl and np.array_equal(alpha, ra)
    beta = orc.backward(A, pobs)
    assert np.array_equal(beta, orc.ref_backward(A, pobs))
    assert np.array_equal(orc.transition_counts(alpha, beta, A, pobs),
                          orc.ref_transition_counts(alpha, beta, A, pobs))
    assert np.array_equal(orc.viterbi(A, pobs, pi), orc.ref_viterbi(A, pobs, pi))
    assert np.array_equal(orc.sample_path(alpha, A, seed=11),
                          orc.ref_sample_path(alpha, A, pobs, seed=11))
    np.testing.assert_allclose(orc.gamma(alpha, beta), orc.ref_gamma(alpha, beta), rtol=1e-14)
    w = orc.gamma(alpha, beta)
    o_int = rng.integers(0, 5, T).astype(np.int32)
    assert np.array_equal(orc.update_pout(o_int, w, np.zeros((N, 5))),
                          orc.ref_update_pout(o_int, w, np.zeros((N, 5))))


def test_longdouble_recursions_agree_with_the_oracle_in_the_normal_range():
    """tests/ld_reference.py (the reference's recursions in 80-bit arithmetic: the arbiter where the
    double-precision reference loses precision in the denormal range) is itself pinned here: on
    ordinary inputs it reproduces the oracle to rounding."""
    from ld_reference import estep_longdouble, hidden_longdouble
    rng = np.random.default_rng(3)
    for n in (1, 3, 8, 13):
        A = rng.random((n, n)) + np.eye(n)
        A /= A.sum(axis=1, keepdims=True)
        pi = rng.dirichlet(np.ones(n))
        mu, sig = np.linspace(-2, 2, n), rng.uniform(0.5, 1.5, n)
        obs = [rng.normal(0, 2, T) for T in (120, 33)]
        ref = orc.estep("gaussian", obs, A, pi, mu, sig)
        pobs = [orc.pobs_gaussian(o, mu, sig) for o in obs]
        logL, C = estep_longdouble(A, pi, pobs)
        np.testing.assert_allclose(logL, ref["logL"], rtol=1e-13)
        np.testing.assert_allclose(C, ref["C"], rtol=1e-11, atol=1e-13)
        l, a, b, g, C1 = hidden_longdouble(A, pobs[0], pi)
        lr, ar = orc.forward(A, pobs[0], pi)
        np.testing.assert_allclose(l, lr, rtol=1e-13)
        np.testing.assert_allclose(a, ar, rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(b, orc.backward(A, pobs[0]), rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(g, orc.gamma(ar, orc.backward(A, pobs[0])), rtol=1e-11, atol=1e-300)
