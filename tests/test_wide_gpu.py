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


def test_wide_viterbi_bit_exact(golden):
    import bhmm_amd.hidden as hidden
    from bhmm_amd.engine import Engine
    g = golden("g64")
    pobs = orc.pobs_gaussian(g["obs"], g["mu"], g["sigma"])
    assert np.array_equal(hidden.viterbi(g["A"], pobs, g["pi"]), g["viterbi"])
    rng = np.random.default_rng(77)
    for n in (9, 16, 20, 33, 64):
        A, pi, B, _ = _random_model(n, rng, "discrete", 25)
        for T in (1, 2, 63, 64, 65, 129, 1000):
            o = rng.integers(0, 25, T).astype(np.int32)
            pobs = orc.pobs_discrete(o, B)
            assert np.array_equal(hidden.viterbi(A, pobs, pi), orc.viterbi(A, pobs, pi)), (n, T)
    # ties: first maximum wins
    A = np.full((12, 12), 1.0 / 12)
    assert np.array_equal(hidden.viterbi(A, np.full((200, 12), 0.5), np.full(12, 1.0 / 12)),
                          orc.viterbi(A, np.full((200, 12), 0.5), np.full(12, 1.0 / 12)))
    # batched, fused emissions
    n, M = 24, 25
    A, pi, B, _ = _random_model(n, rng, "discrete", M)
    obs = [rng.integers(0, M, T).astype(np.int32) for T in (500, 1, 130)]
    eng = Engine(0)
    eng.set_observations("discrete", obs, n, nsymbols=M)
    for p, o in zip(eng.viterbi(A, pi, B), obs):
        assert np.array_equal(p, orc.viterbi(A, orc.pobs_discrete(o, B), pi))
    eng.close()
    A, pi, mu, sig = _random_model(64, rng, "gaussian")
    obs = [rng.normal(0, 4, T) for T in (800, 70)]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, 64)
    for p, o in zip(eng.viterbi(A, pi, mu, sig), obs):
        assert np.array_equal(p, orc.viterbi(A, orc.pobs_gaussian(o, mu, sig), pi))
    eng.close()


@pytest.mark.parametrize("n,kind", [(64, "gaussian"), (33, "gaussian"), (16, "discrete"), (9, "gaussian"),
                                    (40, "discrete")])
def test_wide_viterbi_over_time_segments_is_the_serial_run(n, kind):
    """9..64 states, trajectories cut into time segments (k_wide_viterbi_seg): segments whose start vector
    is not the predecessor's to the bit are repeated from it until none is left, and then the paths are
    the oracle's (_hidden.c:186-276) byte for byte; ragged lengths, a trajectory shorter than a warm-up
    and one of a single step included."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(900 + n)
    M = 30
    A, pi, p0, p1 = _random_model(n, rng, kind, M)
    lengths = (30011, 1, 9000, 257, 12345)
    if kind == "gaussian":
        obs = [rng.normal(0, 3, T) for T in lengths]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    for rep in range(3):                              # (later calls search a shorter warm-up)
        paths = eng.viterbi(A, pi, p0, p1)
        assert eng.get_option("viterbi_chunked") == 1 and eng.get_option("viterbi_segments") > 40
        for p, po in zip(paths, pobs):
            assert np.array_equal(p, orc.viterbi(A, po, pi)), rep
    eng.close()


def test_wide_viterbi_segments_that_do_not_coalesce_go_to_the_serial_kernel():
    """A model that never forgets (the identity plus a whisper) and flat emissions: warm-ups from the
    uniform vector do not reproduce the serial run's vectors, and the repeated segments do not fall
    back onto the first pass's either; after a bounded number of rounds the serial kernel decides, and
    keeps these observations -- the paths are still the oracle's."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(12)
    n = 16
    A = np.eye(n) * (1 - 1e-9) + 1e-9 / n
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    mu, sig = np.zeros(n) + 1e-3 * np.arange(n), np.ones(n)
    obs = [rng.normal(0, 1, 200000)]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, n)
    for _ in range(2):
        p = eng.viterbi(A, pi, mu, sig)[0]
        assert eng.get_option("viterbi_chunked") == 0
        assert np.array_equal(p, orc.viterbi(A, orc.pobs_gaussian(obs[0], mu, sig), pi))
    eng.close()


def test_wide_path_sampling():
    import bhmm_amd.hidden as hidden
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(5)
    for n in (10, 32, 50):
        A = rng.dirichlet(np.ones(n), n)
        alpha = rng.dirichlet(np.ones(n), 700)
        u = rng.random(700)
        assert np.array_equal(hidden.sample_path(alpha, A, np.ones((700, n)), u=u),
                              orc.sample_path(alpha, A, u=u))
    n, M = 20, 15
    A, pi, B, _ = _random_model(n, rng, "discrete", M)
    obs = [rng.integers(0, M, T).astype(np.int32) for T in (600, 1, 77)]
    u = [rng.random(len(o)) for o in obs]
    eng = Engine(0)
    eng.set_observations("discrete", obs, n, nsymbols=M)
    paths, C, n0, emis = eng.sample_paths(A, pi, B, u=u)
    ref = []
    for o, uu in zip(obs, u):
        _, al = orc.forward(A, orc.pobs_discrete(o, B), pi)
        ref.append(orc.sample_path(al, A, u=uu))
    assert sum(int((p != r).sum()) for p, r in zip(paths, ref)) == 0
    Cr, n0r = orc.path_counts(ref, n)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    cnt = np.zeros((n, M))
    np.add.at(cnt, (np.concatenate(ref), np.concatenate(obs)), 1.0)
    assert np.array_equal(emis, cnt)
    eng.close()
    A, pi, mu, sig = _random_model(12, rng, "gaussian")
    obs = [rng.normal(0, 4, T) for T in (400, 90)]
    eng = Engine(0)
    eng.set_observations("gaussian", obs, 12)
    paths, C, n0, emis = eng.sample_paths(A, pi, mu, sig, seed=3)
    allp, allo = np.concatenate(paths), np.concatenate(obs)
    Cr, n0r = orc.path_counts(paths, 12)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    for i in range(12):
        sel = allo[allp == i]
        assert emis[0, i] == len(sel)
        np.testing.assert_allclose(emis[1, i], (sel - mu[i]).sum(), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(emis[2, i], ((sel - mu[i]) ** 2).sum(), rtol=1e-9, atol=1e-9)
    eng.close()


@pytest.mark.parametrize("n,kind", [(64, "gaussian"), (20, "discrete"), (33, "gaussian")])
def test_wide_path_sampling_over_time_segments_is_the_serial_draw(n, kind):
    """9..64 states: the backward draw (_hidden.c:331-380) over time segments coupled through the
    per-step uniforms -- with the caller's uniforms the paths are the oracle's, state for state; with the
    device's counter-based stream they are the serial kernel's (spec_enabled = 0)."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(300 + n)
    M = 18
    A, pi, p0, p1 = _random_model(n, rng, kind, M)
    lengths = (40011, 1, 9000, 2, 257, 20000)
    if kind == "gaussian":
        obs = [rng.normal(0, 3, T) for T in lengths]
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
        assert eng.get_option("sample_segmented") == 1 and eng.get_option("sample_segments") > 40
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


def test_wide_time_segments_verified_or_fallback():
    """9..64 states: trajectories cut into time segments with warm-up boundaries; the result
    must equal the serial recursion, and a too-short warm-up must be caught."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(21)
    n = 20
    A, pi, mu, sig = _random_model(n, rng, "gaussian")
    obs = [rng.normal(0, 4, T) for T in (5000, 700, 1, 2600)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig, want_gamma=True)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 400)
    eng.set_option("spec_W", 150)
    eng.set_observations("gaussian", obs, n)
    assert eng.get_option("wide_segments") > len(obs)
    res = eng.estep(A, pi, mu, sig, store_gamma=True)
    assert eng.get_option("spec_ok") == 1 and eng.get_option("spec_fail") == 0
    _check(res, ref)
    np.testing.assert_allclose(eng.gamma(0), ref["gammas"][0], rtol=1e-8, atol=1e-13)
    np.testing.assert_allclose(eng.gamma(3), ref["gammas"][3], rtol=1e-8, atol=1e-13)
    eng.close()
    eng = Engine(0)
    eng.set_option("wide_segment_len", 400)
    eng.set_option("spec_W", 1)                     # hopeless warm-up: detected, serial fallback
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    assert eng.get_option("spec_fail") == 1
    _check(res, ref)
    res = eng.estep(A, pi, mu, sig)
    _check(res, ref)
    eng.close()
    # discrete, 40 states
    n, M = 40, 30
    A, pi, B, _ = _random_model(n, rng, "discrete", M)
    obs = [rng.integers(0, M, T).astype(np.int32) for T in (3000, 1200)]
    ref = orc.estep("discrete", obs, A, pi, B)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 500)
    eng.set_observations("discrete", obs, n, nsymbols=M)
    res = eng.estep(A, pi, B)
    _check(res, ref)
    eng.close()


def test_wide64_segments_lazy_scaling_and_mfma_counts():
    """64 states, time segments: the kernels that run here carry alpha / beta lazily scaled,
    multiply on DPP row broadcasts and accumulate the transition counts on the matrix cores.
    Result = the serial reference recursion; a stretch of observations ~35 sigma away from every
    state drives the lazily scaled vectors out of range, which must be noticed and repeated with
    the per-step normalisation."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(77)
    n = 64
    A, pi, mu, sig = _random_model(n, rng, "gaussian")
    obs = [rng.normal(0, 4, T) for T in (6001, 500, 3, 2500, 1)]
    for far, careful in ((False, 0.0), (True, 1.0)):
        if far:
            obs[0][3000:3007] = 47.0 + rng.normal(0, 0.1, 7)
            obs[3][100:104] = -46.0
        ref = orc.estep("gaussian", obs, A, pi, mu, sig, want_gamma=True)
        eng = Engine(0)
        eng.set_option("wide_segment_len", 500)
        eng.set_option("spec_W", 200)
        eng.set_observations("gaussian", obs, n)
        assert eng.get_option("wide_segments") > len(obs)
        res = eng.estep(A, pi, mu, sig, store_gamma=True)
        assert eng.get_option("spec_ok") >= 1 and eng.get_option("spec_fail") == 0
        assert eng.get_option("careful") == careful
        _check(res, ref)
        for k in (0, 3):
            np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
        sd = sum((g * (o[:, None] - mu[None, :])).sum(axis=0) for o, g in zip(obs, ref["gammas"]))
        np.testing.assert_allclose(res.sum_gd, sd, rtol=1e-8, atol=1e-9)
        r2 = eng.estep(A, pi, mu, sig)
        assert np.array_equal(res.packed, r2.packed)
        eng.close()


def test_configs3_full_size_properties():
    """BASELINE configs[3] shape (64 states, 128 x 1e5 Gaussian) on the row-batched matrix-core
    kernels (k_tile_fwd / k_tile_bwd, asserted):  (1) size-independent: sum gamma = K T,
    sum C = K (T - 1), sum gamma_0 = K;  (2) the time-segmented run equals the serial plan (one
    segment per trajectory, per-step normalisation, the 64-lane kernels);  (3) AGAINST THE ORACLE
    at the benchmark's own segment geometry: a two-trajectory sub-batch of the same data, T = 1e5,
    cut with the segment length and the warm-up the full batch settled on (3 125-step segments,
    W ~ 900): log-likelihoods, C, sum gamma, sum gamma_0, emission sums and stored gamma rows
    against orc.estep (_hidden.c:16-183);  (4) the full batch's own log-likelihoods of those two
    trajectories against the same oracle run."""
    import torch
    from bench import metastable_matrix, stationary
    from bhmm_amd.engine import Engine
    n, K, T = 64, 128, 100000
    rng = np.random.default_rng(64)
    A = metastable_matrix(n, rng)
    pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    dwell = 50
    s = torch.randint(0, n, (K, T // dwell), device="cuda:0", generator=g).repeat_interleave(dwell, dim=1)
    obs = (torch.tensor(mu, device="cuda:0")[s] + torch.tensor(sig, device="cuda:0")[s]
           * torch.randn((K, T), device="cuda:0", dtype=torch.float64, generator=g)).reshape(-1)
    off = np.arange(K + 1, dtype=np.int64) * T
    args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), off, n)
    assert eng.get_option("wide_segments") >= 1024
    res = eng.estep(*args)                  # the probe sets a warm-up that verifies at once (the tile
    #                                         kernels then refine it with up to four more verified runs)
    assert 1 <= eng.get_option("spec_ok") <= 5 and eng.get_option("spec_fail") == 0
    assert eng.get_option("spec_last_dev") < 1e-11 and eng.get_option("spec_W") > 288  # measured
    assert eng.get_option("careful") == 0.0
    assert np.all(np.isfinite(res.logL_k))
    np.testing.assert_allclose(res.state_counts.sum(), K * T, rtol=1e-10)
    np.testing.assert_allclose(res.C.sum(), K * (T - 1), rtol=1e-10)
    np.testing.assert_allclose(res.gamma0_sum.sum(), K, rtol=1e-10)
    ser = Engine(0)
    ser.set_option("wide_segments", 0)
    ser.set_observations_device("gaussian", obs.data_ptr(), off, n)
    assert ser.get_option("wide_segments") == 0
    rs = ser.estep(*args)
    np.testing.assert_allclose(res.logL_k, rs.logL_k, rtol=1e-11)
    np.testing.assert_allclose(res.C, rs.C, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(res.state_counts, rs.state_counts, rtol=1e-9)
    np.testing.assert_allclose(res.sum_gd, rs.sum_gd, rtol=1e-7, atol=1e-6)
    np.testing.assert_allclose(res.sum_gdd, rs.sum_gdd, rtol=1e-8)
    assert eng.get_option("tile") == 1, eng.get_option("tile_reason")
    assert eng.get_option("wide_trouble") == 0
    seg_len, W = int(eng.get_option("wide_segment_len")), int(eng.get_option("spec_W"))
    assert 2000 <= seg_len <= 4000 and W < seg_len, (seg_len, W)      # 4096 segments of ~3 125 steps
    sub = [obs[k * T:(k + 1) * T].cpu().numpy() for k in (0, 77)]
    ref = orc.estep("gaussian", sub, *args, want_gamma=True)
    np.testing.assert_allclose(res.logL_k[[0, 77]], ref["logL"], rtol=1e-11)
    e2 = Engine(0)
    e2.set_option("wide_segment_len", seg_len)
    e2.set_option("spec_W", W)
    e2.set_observations("gaussian", sub, n)
    r2 = e2.estep(*args, store_gamma=True)
    assert e2.get_option("tile") == 1 and e2.get_option("careful") == 0
    assert e2.get_option("spec_fail") == 0 and e2.get_option("wide_trouble") == 0
    assert e2.get_option("wide_segments") >= 2 * (T // seg_len) and e2.get_option("spec_W") == W
    np.testing.assert_allclose(r2.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(r2.C, ref["C"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r2.state_counts, ref["state_counts"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(r2.gamma0_sum, ref["gamma0_sum"], rtol=1e-9, atol=1e-13)
    mu_e = args[2]
    sd = sum((g * (o[:, None] - mu_e[None, :])).sum(axis=0) for o, g in zip(sub, ref["gammas"]))
    sdd = sum((g * (o[:, None] - mu_e[None, :]) ** 2).sum(axis=0) for o, g in zip(sub, ref["gammas"]))
    np.testing.assert_allclose(r2.sum_gd, sd, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(r2.sum_gdd, sdd, rtol=1e-8, atol=1e-8)
    for k in (0, 1):
        np.testing.assert_allclose(e2.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
    e2.close()
    eng.close()
    ser.close()
    del obs, s
    torch.cuda.empty_cache()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wide64_segmented_fuzz(seed):
    """Randomised 64-state problems on the segmented plan: ragged lengths (shorter than a group
    of four steps, shorter than the warm-up, not a multiple of the segment length), random
    segment length and warm-up, n = 64 and n < 64 padded to 64 lanes.  The result must equal the
    oracle whether the boundaries verify or the serial plan takes over, and repeat bit-for-bit."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(1000 + seed)
    n = (64, 50, 64)[seed - 1]
    A, pi, mu, sig = _random_model(n, rng, "gaussian")
    lengths = [int(x) for x in rng.integers(1, 40, 4)] + [int(x) for x in rng.integers(300, 3000, 3)] + [1, 2, 5]
    rng.shuffle(lengths)
    obs = [rng.normal(0, 4, T) for T in lengths]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig, want_gamma=True)
    eng = Engine(0)
    eng.set_option("wide_segment_len", int(rng.integers(37, 400)))
    eng.set_option("spec_W", int(rng.integers(120, 260)))
    eng.set_observations("gaussian", obs, n)
    assert eng.get_option("wide_segments") > len(obs)
    res = eng.estep(A, pi, mu, sig, store_gamma=True)
    _check(res, ref)
    k = int(np.argmax(lengths))
    np.testing.assert_allclose(eng.gamma(k), ref["gammas"][k], rtol=1e-8, atol=1e-13)
    r2 = eng.estep(A, pi, mu, sig, store_gamma=True)
    _check(r2, ref)
    r3 = eng.estep(A, pi, mu, sig, store_gamma=True)
    assert np.array_equal(r2.packed, r3.packed)
    eng.close()


def test_wide_probe_sets_warmup_before_first_estep():
    """Default options, 24 states (32 lanes per segment): the forgetting-curve probe measures the
    warm-up length on the data, the first E-step already verifies its segment boundaries, and the
    result is the oracle's."""
    from bench import metastable_matrix, stationary
    from bhmm_amd.engine import Engine
    n, K, T = 24, 6, 30000
    rng = np.random.default_rng(24)
    A = 0.9 * metastable_matrix(n, rng) + 0.1 / n
    pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 1.0, n)
    s = np.repeat(rng.integers(0, n, (K, T // 50)), 50, axis=1)
    obs = [mu[s[k]] + sig[s[k]] * rng.standard_normal(T) for k in range(K)]
    ref = orc.estep("gaussian", obs, A, pi, mu + 0.05, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 3000)
    eng.set_observations("gaussian", obs, n)
    assert eng.get_option("wide_segments") == 60
    res = eng.estep(A, pi, mu + 0.05, sig)
    assert eng.get_option("spec_ok") == 1 and eng.get_option("spec_fail") == 0
    W = eng.get_option("spec_W")
    assert 32 <= W <= 3000
    _check(res, ref)
    r2 = eng.estep(A, pi, mu + 0.05, sig)
    assert np.array_equal(res.packed, r2.packed) and eng.get_option("spec_W") == W
    eng.close()


def test_forward_pass_on_its_own_finer_segments():
    """64 states: the forward kernel fits two wavefronts per SIMD, the backward kernel one, so the
    forward pass runs on a plan with every segment cut in two.  The backward pass then meets a change
    of alpha's rescaling chain in the middle of its segments (it must not reuse the gamma normaliser
    across it): same statistics as the serial plan and as the common plan, odd lengths included."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(12)
    n = 64
    A = rng.random((n, n)) + 8 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    mu, sig = np.linspace(-6, 6, n), rng.uniform(0.4, 1.0, n)
    obs = [rng.normal(0, 4, T) for T in (6001, 2503, 4099)]
    res = {}
    for tag, opts in (("serial", {"wide_segments": 0}), ("common", {"wide_split": 0}), ("split", {"wide_split": 1})):
        eng = Engine(0)
        eng.set_option("tile", 0)      # (the one-segment-per-wavefront kernels: the tile kernels have one plan)
        eng.set_option("spec_W", 32)
        eng.set_option("wide_segment_len", 1000)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_observations("gaussian", obs, n)
        for _ in range(2):
            r = eng.estep(A, pi, mu, sig)
        res[tag] = (r, eng.get_option("wide_segments"), eng.get_option("wide_fwd_segments"), eng.get_option("spec_fail"))
        eng.close()
    assert res["serial"][1] == 0 and res["common"][1] > 3 and res["common"][2] == 0
    assert res["split"][2] == 2 * res["split"][1] and res["split"][3] == 0
    ref = res["serial"][0]
    for tag in ("common", "split"):
        r = res[tag][0]
        np.testing.assert_allclose(r.logL_k, ref.logL_k, rtol=1e-12)
        np.testing.assert_allclose(r.C, ref.C, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r.state_counts, ref.state_counts, rtol=1e-10)
        np.testing.assert_allclose(r.state_counts.sum(), sum(len(o) for o in obs), rtol=1e-12)


def test_row_in_the_denormal_range_in_the_non_lazy_forward_pass():
    """tests/sweeps/stress_small.py seed 2001 case 2860 (saved under tests/golden/cases): 20 narrow states,
    an observation that is all but impossible where the mass sits -- the new alpha row is denormal although
    the emission row is not, the reciprocal of its sum was infinite and the log-likelihood NaN.  The
    reference divides by the denormal sum and stays finite."""
    import os
    from bhmm_amd.engine import Engine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cases", "wide20_denormal_row_2001_2860.npz"),
                allow_pickle=True)
    A, pi, mu, sig, lens = d["A"], d["pi"], d["par0"], d["par1"], d["lens"]
    obs = np.split(d["obs"], np.cumsum(lens)[:-1])
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    assert np.all(np.isfinite(ref["logL"]))
    eng = Engine(0)
    eng.set_observations("gaussian", obs, A.shape[0], chunk=int(d["chunk"]))
    for _ in range(2):
        res = eng.estep(A, pi, mu, sig)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
    eng.close()
