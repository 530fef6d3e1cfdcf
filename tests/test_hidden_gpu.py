"""GPU parity of the bhmm.hidden-compatible API (bhmm_amd/hidden/api.py) -- the counterpart of
bhmm/tests/test_hidden.py:265-335 (each kernel with and without preallocated *_out buffers,
results compared with the reference implementation), plus the bit-exact contracts:
Viterbi paths and sampled paths identical to the reference C given the same inputs.
"""
import hashlib

import numpy as np
import pytest

from conftest import split
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def hidden():
    import bhmm_amd.hidden as h
    h.set_implementation('hip')
    return h


def examples(golden):
    g1 = golden("kat1_toy")
    ex = [(g1["A"], g1["pi"], g1["pobs"])]
    g2 = golden("kat2_gauss3")
    pobs = orc.pobs_gaussian(g2["obs"].astype(np.float64), g2["mu"], g2["sigma"])
    ex.append((g2["A"], g2["pi"], pobs))
    g3 = golden("g8_ragged")
    o = split(g3["obs"], g3["lengths"])[3]
    ex.append((g3["A"], g3["pi"], orc.pobs_gaussian(o, g3["mu"], g3["sigma"])))
    return ex


def test_forward_backward_gamma_counts(hidden, golden):
    for A, pi, pobs in examples(golden):
        T, N = pobs.shape
        ll_ref, a_ref = orc.forward(A, pobs, pi)
        b_ref = orc.backward(A, pobs)
        g_ref = orc.gamma(a_ref, b_ref)
        C_ref = orc.transition_counts(a_ref, b_ref, A, pobs)
        # fresh allocation
        ll, alpha = hidden.forward(A, pobs, pi)
        beta = hidden.backward(A, pobs)
        gam = hidden.state_probabilities(alpha, beta)
        C = hidden.transition_counts(alpha, beta, A, pobs)
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-12)
        np.testing.assert_allclose(alpha, a_ref, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(beta, b_ref, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(gam, g_ref, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(C, C_ref, rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(hidden.state_counts(gam, T), g_ref.sum(axis=0), rtol=1e-10)
        # preallocated buffers longer than T (maximum_likelihood.py:128-130): rows >= T untouched
        big_a = np.full((T + 5, N), -7.0)
        big_b = np.full((T + 5, N), -7.0)
        ll2, a2 = hidden.forward(A, pobs, pi, T=T, alpha_out=big_a)
        b2 = hidden.backward(A, pobs, T=T, beta_out=big_b)
        assert a2 is big_a and b2 is big_b
        assert np.array_equal(big_a[:T], alpha) and np.all(big_a[T:] == -7.0)
        assert np.array_equal(big_b[:T], beta) and np.all(big_b[T:] == -7.0)
        g_out = np.zeros((T, N))
        g2 = hidden.state_probabilities(big_a, big_b, T=T, gamma_out=g_out)
        assert g2 is g_out and np.array_equal(g_out, gam)
        C_out = np.full((N, N), 99.0)                      # overwritten, not accumulated
        C2 = hidden.transition_counts(big_a, big_b, A, np.vstack([pobs, np.ones((5, N))]), T=T,
                                      out=C_out)
        assert C2 is C_out and np.array_equal(C_out, C)
        # gamma of the truncated problem T' < T (hidden/api.py:181-184)
        if T > 4:
            g_short = hidden.state_probabilities(alpha, beta, T=4)
            np.testing.assert_allclose(g_short, g_ref[:4], rtol=1e-9)


def test_errors_mirror_reference(hidden, golden):
    g1 = golden("kat1_toy")
    A, pi, pobs = g1["A"], g1["pi"], g1["pobs"]
    with pytest.raises(ValueError):
        hidden.forward(A, pobs, pi, T=11)                  # impl_python/hidden.py:64-65
    with pytest.raises(ValueError):
        hidden.backward(A, pobs, T=11)
    with pytest.raises(ValueError):
        hidden.forward(A, pobs, pi, alpha_out=np.zeros((3, 2)))
    with pytest.raises(ValueError):
        hidden.state_probabilities(np.zeros((5, 2)), np.zeros((4, 2)))  # hidden/api.py:167-168
    with pytest.warns(UserWarning):
        hidden.set_implementation('fortran')               # hidden/api.py:59-62


def test_viterbi_bit_exact(hidden, golden):
    g1 = golden("kat1_toy")
    assert np.array_equal(hidden.viterbi(g1["A"], g1["pobs"], g1["pi"]), g1["viterbi"])
    assert hidden.viterbi(g1["A"], g1["pobs"], g1["pi"]).dtype == np.int32   # hidden.pyx:161-162
    g2 = golden("kat2_gauss3")
    pobs = orc.pobs_gaussian(g2["obs"].astype(np.float64), g2["mu"], g2["sigma"])
    v = hidden.viterbi(g2["A"], pobs, g2["pi"])
    assert np.array_equal(v, g2["viterbi"])
    assert sha1(v) == "a15f23bffe22a93b68d1750509c01ebdd839bac9"
    g = golden("d2_doublewell")
    pobs = orc.pobs_discrete(g["obs"].astype(np.int32), g["B"])
    v = hidden.viterbi(g["A"], pobs, g["pi"])
    assert sha1(v) == str(g["viterbi_sha1"])
    g = golden("d3_zeros")
    pobs = orc.pobs_discrete(g["obs"], g["B"])
    assert np.array_equal(hidden.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])
    g = golden("g8_outliers")
    pobs = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"])
    assert np.array_equal(hidden.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])


def test_viterbi_ties_first_maximum_wins(hidden):
    # all-ties input: every comparison is an exact tie
    for N in (2, 3, 4, 7, 8):
        A = np.full((N, N), 1.0 / N)
        pi = np.full(N, 1.0 / N)
        pobs = np.full((300, N), 0.25)
        assert np.array_equal(hidden.viterbi(A, pobs, pi), orc.viterbi(A, pobs, pi))
    rng = np.random.default_rng(5)
    for N in (1, 2, 3, 5, 8):
        A = rng.integers(1, 4, (N, N)).astype(float)
        A /= A.sum(axis=1)[:, None]
        pi = np.full(N, 1.0 / N)
        pobs = rng.integers(1, 3, (2000, N)) / 4.0         # few distinct values -> many ties
        assert np.array_equal(hidden.viterbi(A, pobs, pi), orc.viterbi(A, pobs, pi))
    for T in (1, 2, 7, 8, 9, 16, 17):                      # back-pointer word boundaries
        pobs = rng.random((T, 3))
        A = rng.dirichlet(np.ones(3), 3)
        assert np.array_equal(hidden.viterbi(A, pobs, np.array([.2, .3, .5])),
                              orc.viterbi(A, pobs, np.array([.2, .3, .5])))


def test_sample_path_reproduces_reference_stream(hidden, golden):
    g1 = golden("kat1_toy")
    s = hidden.sample_path(g1["alpha"], g1["A"], g1["pobs"], seed=42)
    assert np.array_equal(s, g1["sample_path_seed42"]) and s.dtype == np.int32
    s = hidden.sample_path(g1["alpha"], g1["A"], g1["pobs"], u=g1["sample_u"])
    assert np.array_equal(s, g1["sample_path_seed42"])
    g = golden("g8_ragged")
    o = split(g["obs"], g["lengths"])[0]
    pobs = orc.pobs_gaussian(o, g["mu"], g["sigma"])
    _, alpha = orc.forward(g["A"], pobs, g["pi"])
    s = hidden.sample_path(alpha, g["A"], pobs, seed=7)
    assert np.array_equal(s, g["sample_path0_seed7"])
    rng = np.random.default_rng(9)
    for N in (1, 2, 3, 6, 8):
        A = rng.dirichlet(np.ones(N), N)
        alpha = rng.dirichlet(np.ones(N), 1500)
        u = rng.random(1500)
        assert np.array_equal(hidden.sample_path(alpha, A, np.ones((1500, N)), u=u),
                              orc.sample_path(alpha, A, u=u))
        s_short = hidden.sample_path(alpha, A, np.ones((1500, N)), T=100, u=u[:100])
        assert np.array_equal(s_short, orc.sample_path(alpha[:100], A, u=u[:100]))


def test_batched_viterbi_matches_per_trajectory(golden):
    from bhmm_amd.engine import Engine
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    eng = Engine(0)
    eng.set_observations("gaussian", obs, 8)
    paths = eng.viterbi(g["A"], g["pi"], g["mu"], g["sigma"])
    ref = split(g["viterbi"], g["lengths"])
    for p, r in zip(paths, ref):
        assert np.array_equal(p, r)
    eng.close()
    g = golden("d8_ragged")
    obs = split(g["obs"].astype(np.int32), g["lengths"])
    eng = Engine(0)
    eng.set_observations("discrete", obs, 8, nsymbols=g["B"].shape[1])
    paths = eng.viterbi(g["A"], g["pi"], g["B"])
    for p, r in zip(paths, split(g["viterbi"], g["lengths"])):
        assert np.array_equal(p, r)
    eng.close()


def test_batched_path_sampling_and_statistics(golden):
    from bhmm_amd.engine import Engine
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    rng = np.random.default_rng(3)
    u = [rng.random(len(o)) for o in obs]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, 8, chunk=50)
    paths, C, n0, emis = eng.sample_paths(g["A"], g["pi"], g["mu"], g["sigma"], u=u)
    ref_paths = []
    for o, uu in zip(obs, u):
        pobs = orc.pobs_gaussian(o, g["mu"], g["sigma"])
        _, alpha = orc.forward(g["A"], pobs, g["pi"])
        ref_paths.append(orc.sample_path(alpha, g["A"], u=uu))
    mism = sum(int((p != r).sum()) for p, r in zip(paths, ref_paths))
    # alpha comes from the chunk-parallel forward pass (1e-13 relative to the serial one), so a
    # draw can differ only if a uniform falls within ~1e-13 of a CDF step: not in this data
    assert mism == 0
    Cr, n0r = orc.path_counts(ref_paths, 8)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)      # integer statistics: exact
    allp = np.concatenate(ref_paths)
    allo = np.concatenate(obs)
    for i in range(8):
        sel = allo[allp == i]
        assert emis[0, i] == len(sel)
        np.testing.assert_allclose(emis[1, i], (sel - g["mu"][i]).sum(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(emis[2, i], ((sel - g["mu"][i]) ** 2).sum(), rtol=1e-9)
    # device-generated uniforms: statistics only, same seed -> same draw
    p1, C1, _, _ = eng.sample_paths(g["A"], g["pi"], g["mu"], g["sigma"], seed=11)
    p2, C2, _, _ = eng.sample_paths(g["A"], g["pi"], g["mu"], g["sigma"], seed=11)
    p3, C3, _, _ = eng.sample_paths(g["A"], g["pi"], g["mu"], g["sigma"], seed=12)
    assert np.array_equal(C1, C2) and not np.array_equal(C1, C3)
    assert C1.sum() == sum(len(o) - 1 for o in obs)
    eng.close()
    # discrete: symbol counts per hidden state
    g = golden("d8_ragged")
    obs = split(g["obs"].astype(np.int32), g["lengths"])
    M = g["B"].shape[1]
    u = [rng.random(len(o)) for o in obs]
    eng = Engine(0)
    eng.set_observations("discrete", obs, 8, nsymbols=M, chunk=40)
    paths, C, n0, emis = eng.sample_paths(g["A"], g["pi"], g["B"], u=u)
    ref_paths = []
    for o, uu in zip(obs, u):
        _, alpha = orc.forward(g["A"], orc.pobs_discrete(o, g["B"]), g["pi"])
        ref_paths.append(orc.sample_path(alpha, g["A"], u=uu))
    assert all(np.array_equal(p, r) for p, r in zip(paths, ref_paths))
    Cr, n0r = orc.path_counts(ref_paths, 8)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    cnt = np.zeros((8, M))
    np.add.at(cnt, (np.concatenate(ref_paths), np.concatenate(obs)), 1.0)
    assert np.array_equal(emis, cnt)
    eng.close()


def test_output_model_kernels(golden):
    from bhmm_amd import _lib
    L = _lib.load()
    g = golden("pobs_gauss3")                       # bhmm/tests/test_output_gaussian.py:43-59
    T, N = g["pobs"].shape
    out = np.empty((T, N))
    _lib.check(L.bhmm_pobs_gaussian(_lib.dp(out), _lib.dp(_lib.f64(g["obs"])),
                                    _lib.dp(_lib.f64(g["mu"])), _lib.dp(_lib.f64(g["sigma"])),
                                    N, T, 1))
    np.testing.assert_allclose(out, g["pobs"], rtol=1e-13)
    g = golden("d8_ragged")
    obs = g["obs"].astype(np.int32)[:5000]
    w = np.random.default_rng(1).dirichlet(np.ones(8), 5000)
    pout = np.zeros((8, 64))
    _lib.check(L.bhmm_update_pout(_lib.dp(pout), _lib.ip(obs), _lib.dp(w), 5000, 8, 64))
    np.testing.assert_allclose(pout, orc.update_pout(obs, w, np.zeros((8, 64))), rtol=1e-11)


def test_batched_sampling_exact_on_ties():
    """Uniforms that hit CDF steps exactly (margin 0): the chunk-parallel sampler must take the
    reference's normalise-then-cumsum fallback and reproduce its choice."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(12)
    for n in (2, 4, 8):
        A = np.full((n, n), 1.0 / n)                       # dyadic everywhere
        B = np.full((n, 4), 0.25)
        pi = np.full(n, 1.0 / n)
        obs = [rng.integers(0, 4, T).astype(np.int32) for T in (257, 31, 1)]
        u = [rng.integers(0, 2 * n, len(o)) / (2.0 * n) for o in obs]   # exact multiples of 1/(2n)
        eng = Engine(0)
        eng.set_observations("discrete", obs, n, nsymbols=4, chunk=16)
        paths, C, n0, _ = eng.sample_paths(A, pi, B, u=u)
        for p, o, uu in zip(paths, obs, u):
            _, alpha = orc.forward(A, orc.pobs_discrete(o, B), pi)
            assert np.array_equal(p, orc.sample_path(alpha, A, u=uu))
        eng.close()
    # impossible draw (u >= 1 can never be reached): reported, not fatal (_hidden.c:299-304)
    eng = Engine(0)
    eng.set_observations("discrete", [obs[0]], 8, nsymbols=4)
    from bhmm_amd import _lib
    with pytest.raises(_lib.BhmmAmdError):
        eng.sample_paths(np.full((8, 8), 0.125), np.full(8, 0.125), np.full((8, 4), 0.25),
                         u=[np.full(len(obs[0]), 1.5)])
    eng.close()


@pytest.mark.parametrize("chunk", [0, 5, 16, 40, 200])
def test_chunk_parallel_viterbi_is_bit_exact_or_falls_back(golden, chunk):
    """bhmm_viterbi_batch first runs parallel over time chunks (warm-up boundaries, verified;
    decisions must not be close) and otherwise serially; either way the path is the reference's
    (_hidden.c:203-281), bit for bit."""
    from bhmm_amd.engine import Engine
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    ref = split(g["viterbi"], g["lengths"])
    eng = Engine(0)
    eng.set_observations("gaussian", obs, 8, chunk=chunk)
    for W in (288, 24, 2):
        eng.set_option("spec_W", W)
        paths = eng.viterbi(g["A"], g["pi"], g["mu"], g["sigma"])
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r)
        ran_chunked = eng.get_option("viterbi_chunked")
        if W == 2 and 0 < chunk < 100:
            assert ran_chunked == 0.0        # two warm-up steps cannot verify -> serial run
    eng.close()
    # a longer problem where the chunked run must be the one that is used
    rng = np.random.default_rng(4)
    n, T, K = 5, 20000, 6
    A = rng.random((n, n)) + 4 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = np.full(n, 1.0 / n)
    mu, sig = np.linspace(-3, 3, n), np.full(n, 0.8)
    s = np.zeros((K, T), dtype=int)
    for t in range(1, T):
        s[:, t] = [rng.choice(n, p=A[i]) for i in s[:, t - 1]]
    obs = [mu[s[k]] + sig[s[k]] * rng.standard_normal(T) for k in range(K)]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, n, chunk=500)
    paths = eng.viterbi(A, pi, mu, sig)
    assert eng.get_option("viterbi_chunked") == 1.0
    for k in range(K):
        pobs = orc.pobs_gaussian(obs[k], mu, sig)
        assert np.array_equal(paths[k], orc.viterbi(A, pobs, pi))
    # exact ties everywhere (uniform model): every decision is "close", but the boundary vectors
    # are bit-identical, so the chunked run is the serial run and the first maximum wins as in the
    # reference
    Au = np.full((n, n), 1.0 / n)
    eng.set_observations("explicit", [np.full((3000, n), 0.3)], n, chunk=100)
    p = eng.viterbi(Au, pi)
    assert eng.get_option("viterbi_close") > 0
    assert np.array_equal(p[0], orc.viterbi(Au, np.full((3000, n), 0.3), pi))
    eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_randomised_viterbi_and_sampling(seed):
    """Random shapes / chunk lengths / kinds: batched Viterbi (chunk-parallel or serial, whichever
    the library picks) and batched path sampling against the oracle, bit for bit."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.choice([2, 3, 5, 8]))
    K = int(rng.integers(1, 7))
    lengths = rng.integers(1, 3000, K)
    A = rng.random((n, n)) + np.eye(n) * rng.uniform(0, 5)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    chunk = int(rng.choice([0, 7, 64, 300, 1000]))
    eng = Engine(0)
    if seed % 2 == 0:
        mu, sig = np.sort(rng.normal(0, 3, n)), rng.uniform(0.4, 1.5, n)
        obs = [rng.normal(0, 3, T) for T in lengths]
        pobs = [orc.pobs_gaussian(o, mu, sig) for o in obs]
        eng.set_observations("gaussian", obs, n, chunk=chunk)
        args = (A, pi, mu, sig)
    else:
        M = int(rng.integers(2, 9))
        B = rng.dirichlet(np.ones(M), size=n)
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, B) for o in obs]
        eng.set_observations("discrete", obs, n, nsymbols=M, chunk=chunk)
        args = (A, pi, B)
    paths = eng.viterbi(*args)
    for p, pb in zip(paths, pobs):
        assert np.array_equal(p, orc.viterbi(A, pb, pi))
    u = [rng.random(T) for T in lengths]
    sp, C, n0, _ = eng.sample_paths(*args, u=u)
    ref = [orc.sample_path(orc.forward(A, pb, pi)[1], A, u=uu) for pb, uu in zip(pobs, u)]
    assert sum(int((p != r).sum()) for p, r in zip(sp, ref)) == 0
    Cr, n0r = orc.path_counts(ref, n)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    eng.close()


def test_forward_rows_whose_product_with_alpha_is_denormal():
    """An emission row that is not tiny as a whole but whose product with the current alpha is: A = I,
    alpha = [1, 0], pobs = [1e-312, 1e-80] (found by tests/sweeps/stress_hidden.py seed 2002 case 1789:
    the reciprocal of the denormal row sum was infinite, the rows NaN from there on).  The reference
    divides by the denormal sum and stays finite; so must the kernels, with the same log-likelihood."""
    from bhmm_amd import hidden
    rng = np.random.default_rng(1789)
    for n in (2, 3, 8):
        A = np.eye(n)
        pi = np.full(n, 1.0 / n)
        pi[0] += 0.5
        pi /= pi.sum()
        T = 300
        pobs = np.exp(-60.0 * rng.random((T, n)))
        pobs[:20, 1:] *= 1e-30                         # state 0 wins: alpha -> [1, 0, ...] EXACTLY
        for t in (40, 41, 150, 299):
            pobs[t, 0] = 1.01e-312                     # a denormal for the only state that is alive
        ll_ref, a_ref = orc.forward(A, pobs, pi)
        b_ref = orc.backward(A, pobs)
        assert np.isfinite(ll_ref) and a_ref[45, 0] == 1.0
        ll, alpha = hidden.forward(A, pobs, pi)
        beta = hidden.backward(A, pobs)
        assert np.all(np.isfinite(alpha)) and np.all(np.isfinite(beta))
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-12)
        np.testing.assert_allclose(alpha, a_ref, rtol=1e-9, atol=1e-250)
        # (beta: weights below 1e-308 of a row are lost for good by the double-precision reference and kept by
        # the kernels' separate exponents -- tests/sweeps/stress_hidden.py; compared where both are representable)
        np.testing.assert_allclose(beta.sum(axis=1), 1.0, rtol=1e-12)
        big = b_ref > 1e-150
        np.testing.assert_allclose(beta[big], b_ref[big], rtol=1e-9)
