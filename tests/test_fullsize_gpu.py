"""GPU parity at the BASELINE.json sizes that fixture-sized tests cannot reach, plus the entry
points added for them (device-side synthetic workloads, one-byte Viterbi paths, device-resident
Gibbs statistics, partition-independent random stream).

Full-size evidence comes in two forms: ONE trajectory of the batch against the CPU oracle
(seconds of CPU time), and size-independent properties of the whole batch (unit gamma mass per
step, T-1 transitions per trajectory, sub-batch invariance of per-trajectory results)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _engine():
    from bhmm_amd.engine import Engine
    return Engine(0)


def _c3_model():
    from bench import metastable_matrix, stationary
    rng = np.random.default_rng(3000)
    n, M = 8, 64
    A = metastable_matrix(n, rng)
    pi = stationary(A)
    B = rng.dirichlet(np.ones(M), size=n)
    return n, M, A, pi, B, 0.9 * A + 0.1 / n, 0.8 * B + 0.2 / M


# ---- device-side synthetic workloads -----------------------------------------------------
def test_device_generator_equals_host_restatement():
    """bhmm_synth_observations: discrete trajectories and hidden paths bit for bit, Gaussian
    observations to rounding (libm vs device log/cos), ragged K (not a multiple of 64), T not a
    multiple of the 64-step tile."""
    import torch
    from bhmm_amd.engine import synth_observations
    from synth_host import synth_discrete, synth_gaussian
    n, M, A, pi, B, _, _ = _c3_model()
    K, T = 70, 333
    obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
    st = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
    synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=99, states_dev=st.data_ptr())
    o, s = obs.cpu().numpy().reshape(K, T), st.cpu().numpy().reshape(K, T)
    for k in (0, 1, 63, 64, 69):
        ho, hs = synth_discrete(A, pi, B, k, T, 99)
        assert np.array_equal(o[k], ho) and np.array_equal(s[k], hs)
    assert o.min() >= 0 and o.max() < M
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    og = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", og.data_ptr(), A, pi, mu, sig, K, T, seed=99, states_dev=st.data_ptr())
    og, s = og.cpu().numpy().reshape(K, T), st.cpu().numpy().reshape(K, T)
    for k in (0, 69):
        ho, hs = synth_gaussian(A, pi, mu, sig, k, T, 99)
        assert np.array_equal(s[k], hs)
        np.testing.assert_allclose(og[k], ho, rtol=1e-12, atol=1e-12)
    # the draws follow the model: state frequencies near the stationary vector
    big = torch.empty(64 * 50000, dtype=torch.float64, device="cuda:0")
    sb = torch.empty(64 * 50000, dtype=torch.uint8, device="cuda:0")
    synth_observations("gaussian", big.data_ptr(), A, pi, mu, sig, 64, 50000, seed=5, states_dev=sb.data_ptr())
    freq = np.bincount(sb.cpu().numpy(), minlength=n) / sb.numel()
    np.testing.assert_allclose(freq, pi, atol=0.02)
    z = (big.cpu().numpy() - mu[sb.cpu().numpy()]) / sig[sb.cpu().numpy()]
    assert abs(z.mean()) < 0.005 and abs(z.std() - 1.0) < 0.005


# ---- one-byte Viterbi paths ---------------------------------------------------------------
def test_viterbi_u8_equals_int32(golden):
    import torch
    from conftest import split
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=64)
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    ref = np.concatenate(eng.viterbi(*margs))
    total = ref.size
    assert np.array_equal(eng.viterbi_u8(*margs), ref)                     # fresh numpy array
    dev = torch.full((total,), 255, dtype=torch.uint8, device="cuda:0")
    assert eng.viterbi_u8(*margs, out=dev) is dev
    assert np.array_equal(dev.cpu().numpy(), ref)                          # device-resident
    pin = torch.full((total,), 255, dtype=torch.uint8).pin_memory()
    eng.viterbi_u8(*margs, out=pin)
    assert np.array_equal(pin.numpy(), ref)                                # pinned host
    with pytest.raises(ValueError):
        eng.viterbi_u8(*margs, out=np.empty(total - 1, dtype=np.uint8))
    eng.close()
    # 9..64 states (serial back-trace kernel)
    g = golden("g64")
    obs = [g["obs"][:400], g["obs"][400:]]
    eng = _engine()
    eng.set_observations("gaussian", obs, 64)
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    assert np.array_equal(eng.viterbi_u8(*margs), np.concatenate(eng.viterbi(*margs)))
    eng.close()


def test_viterbi_full_size_chunked_and_bit_exact():
    """Batched Viterbi at 256 x 1e5 (configs[1] shape): the chunk-parallel run is accepted
    (viterbi_chunked == 1), and trajectories of it are the oracle's paths bit for bit."""
    import torch
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    K, T = 256, 100000
    m = make_c2_model()
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=21)
    eng = _engine()
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
    paths = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
    eng.viterbi_u8(*margs, out=paths)
    assert eng.get_option("viterbi_chunked") == 1
    p = paths.cpu().numpy().reshape(K, T)
    for k in (0, 131, 255):
        o = obs[k * T:(k + 1) * T].cpu().numpy()
        ref = orc.viterbi(m["A_eval"], orc.pobs_gaussian(o, m["mu_eval"], m["sigma"]), m["pi"])
        assert np.array_equal(p[k], ref)
    # the reference-typed int32 form returns the same paths
    p32 = eng.viterbi(*margs)
    assert p32[7].dtype == np.int32 and np.array_equal(p32[7], p[7])
    eng.close()


def test_chunked_viterbi_with_outliers_infinities_and_odd_sigmas():
    """The condition-free step blocks of k_viterbi_chunks (path_kernels.hpp) on data that exercise what
    they skip: observations 40 sigma from every state (all-zero emission rows -> outputmodel.py:126-130,
    found there through a zero normalising sum), +-inf observations (the reciprocal form of the division
    by sigma gives NaN where the quotient is inf; the clamp of the exponent must turn both into the same
    zero), an observation exactly on a mean, sigmas with all-ones / power-of-two mantissas, fewer than 8
    states (padding lanes) and ragged lengths whose warm-ups end inside a block.  Paths bit for bit."""
    rng = np.random.default_rng(40)
    for n in (8, 5, 2):
        A = rng.random((n, n)) + np.eye(n) * 6.0
        A /= A.sum(axis=1)[:, None]
        pi = np.full(n, 1.0 / n)
        mu = np.linspace(-3.0, 3.0, n)
        sig = rng.uniform(0.4, 1.2, n)
        sig[0] = np.nextafter(1.0, 0.0)             # mantissa all ones
        sig[-1] = 0.5                               # power of two
        lengths = (30011, 20000, 5003, 997)
        obs = []
        for T in lengths:
            s = np.zeros(T, dtype=np.int64)
            for t in range(1, T):
                s[t] = s[t - 1] if rng.random() < 0.97 else rng.integers(0, n)
            o = rng.normal(mu[s], sig[s])
            bad = rng.choice(np.arange(600, T - 5), size=12, replace=False)
            o[bad[:4]] = 80.0                       # density 0 for every state
            o[bad[4:6]] = np.inf
            o[bad[6:8]] = -np.inf
            o[bad[8]] = mu[n // 2]                  # exactly on a mean
            o[bad[9]] = 45.0 * sig.max() + mu[-1]   # in the last state's far tail only
            o[bad[10]:bad[10] + 3] = -90.0          # three outliers in a row
            obs.append(o)
        eng = _engine()
        eng.set_observations("gaussian", obs, n, chunk=640)
        eng.estep(A, pi, mu, sig)                   # (calibrates the warm-up)
        vp = eng.viterbi(A, pi, mu, sig)
        assert eng.get_option("viterbi_chunked") == 1
        for o, p in zip(obs, vp):
            ref = orc.viterbi(A, orc.pobs_gaussian(o, mu, sig), pi)   # (outlier rule included)
            assert np.array_equal(p, ref)
        eng.close()


def test_chunked_viterbi_recovers_from_a_warm_up_that_is_too_short():
    """A warm-up of 32 steps does not bring the survivors of every chunk boundary together at the
    configs[1] model (tools/viterbi_w.py: 96 is still too short, 128 suffices): the boundary check sees
    it, the run is repeated with 64 and 128 steps and accepted -- the same paths, no serial kernel; the
    next call starts from the length that worked."""
    import time
    import torch
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    K, T = 64, 50000
    m = make_c2_model()
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=5)
    margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
    ref = None
    for W in (0, 32):
        eng = _engine()
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8,
                                    chunk=782)
        if W:
            eng.set_option("spec_W", W)
        paths = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
        eng.viterbi_u8(*margs, out=paths)
        assert eng.get_option("viterbi_chunked") == 1
        if ref is None:
            ref = paths.cpu().numpy().copy()
            o = obs[:T].cpu().numpy()
            assert np.array_equal(ref[:T], orc.viterbi(m["A_eval"], orc.pobs_gaussian(o, m["mu_eval"], m["sigma"]),
                                                       m["pi"]))
        else:
            assert np.array_equal(paths.cpu().numpy(), ref)
            t0 = time.perf_counter()
            eng.viterbi_u8(*margs, out=paths)       # second call: no failed attempts any more
            torch.cuda.synchronize()
            assert time.perf_counter() - t0 < 0.02
            assert np.array_equal(paths.cpu().numpy(), ref)
        eng.close()


def test_many_trajectories_fetch_their_log_likelihoods_on_demand():
    """More than 4096 trajectories: the E-step brings only the packed statistics to the host; the
    per-trajectory log-likelihoods are copied when bhmm_estep_fetch is asked for them (before or after a
    statistics-only fetch), and a non-finite one is still found and named."""
    rng = np.random.default_rng(4096)
    n, K = 4, 6000
    A = rng.random((n, n)) + np.eye(n) * 3
    A /= A.sum(axis=1)[:, None]
    pi = np.full(n, 0.25)
    mu, sig = np.array([-2.0, -0.5, 0.5, 2.0]), np.array([0.6, 0.5, 0.5, 0.7])
    obs = [rng.normal(0, 1.5, int(T)) for T in rng.integers(1, 40, K)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = _engine()
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(res.loglik, ref["logL"].sum(), rtol=1e-12)
    eng.estep_launch(A, pi, mu, sig)
    packed = eng.estep_fetch_packed().copy()
    np.testing.assert_allclose(packed[0], ref["logL"].sum(), rtol=1e-12)
    np.testing.assert_allclose(eng.estep_fetch_logL(), ref["logL"], rtol=1e-11, atol=1e-12)
    # an impossible start for one trajectory: its log-likelihood is -inf, the call says which
    pi0 = np.array([1.0, 0.0, 0.0, 0.0])
    A0 = A.copy()
    obs2 = list(obs)
    obs2[4321] = np.array([np.nan, 0.1, 0.2])
    eng.set_observations("gaussian", obs2, n)
    with pytest.raises(AssertionError, match="4321"):
        eng.estep_launch(A0, pi0, mu, sig)
        eng.estep_fetch_packed()
    eng.close()


def test_automatic_plan_halves_its_chunk_count_when_chunks_are_shorter_than_the_warm_up():
    """A batch of 1.2e6 steps: the default plan (32768 chunks of 37 steps) would spend most of its
    time in warm-ups of a few hundred steps; once the warm-up is calibrated (first E-step) the library
    re-plans with 16384 chunks (plan.hpp, tools/chunk_scan.py).  Results against the oracle before and
    after, E-step / Viterbi / Gibbs statistics on the new plan; a caller's own chunk length is kept."""
    import torch
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    K, T = 12, 100000
    m = make_c2_model()
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=12)
    margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
    host = obs.cpu().numpy().reshape(K, T)
    ref = orc.estep("gaussian", list(host), *margs)
    eng = _engine()
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    g0 = eng.num_chunks
    assert g0 > 30000
    for _ in range(2):
        res = eng.estep(*margs)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-9)
    assert 15000 < eng.num_chunks < 17000 and eng.chunk_len > 1.9 * (T * K / g0)
    vp = eng.viterbi(*margs)
    assert eng.get_option("viterbi_chunked") == 1
    assert np.array_equal(vp[5], orc.viterbi(m["A_eval"], orc.pobs_gaussian(host[5], m["mu_eval"], m["sigma"]), m["pi"]))
    paths, C, n0, emis = eng.sample_paths(*margs, seed=3)
    assert C.sum() == K * (T - 1) and n0.sum() == K
    eng.close()
    eng = _engine()
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8, chunk=40)
    g1 = eng.num_chunks
    eng.estep(*margs)
    assert eng.num_chunks == g1
    eng.close()


# ---- E-step at T = 1e6 and at the configs[2] batch -----------------------------------------
def test_one_million_step_discrete_trajectory_vs_oracle():
    """One T = 1e6 discrete trajectory (64-bit offsets inside a long trajectory, 1e6-term
    log-likelihood sums) against the oracle: logL, C, sum gamma, emission counts."""
    import torch
    from bhmm_amd.engine import synth_observations
    n, M, A, pi, B, A_eval, B_eval = _c3_model()
    T = 1000000
    obs = torch.empty(T, dtype=torch.int32, device="cuda:0")
    synth_observations("discrete", obs.data_ptr(), A, pi, B, None, 1, T, seed=77)
    o = obs.cpu().numpy()
    ref = orc.estep("discrete", [o], A_eval, pi, B_eval, want_gamma=True)
    for chunk in (0, 4099):
        eng = _engine()
        eng.set_observations_device("discrete", obs.data_ptr(), np.array([0, T], dtype=np.int64), n,
                                    nsymbols=M, chunk=chunk)
        res = eng.estep(A_eval, pi, B_eval)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-9)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9)
        np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=1e-9, atol=1e-14)
        cnt = np.zeros((n, M))
        orc.update_pout(o, ref["gammas"][0], cnt)
        np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-9)
        eng.close()


def test_configs2_full_batch_properties():
    """BASELINE configs[2] on one GPU: 8-state discrete (M = 64), 1024 x 1e6 -- 8.2e9 alpha
    elements (64-bit indexing), 65 GB workspace.  Unit gamma mass per step, T-1 transitions per
    trajectory, per-trajectory log-likelihoods equal to a 3-trajectory sub-batch (different chunk
    plan) and, for one trajectory, to the oracle."""
    import torch
    from bhmm_amd.engine import synth_observations
    n, M, A, pi, B, A_eval, B_eval = _c3_model()
    K, T = 1024, 1000000
    obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
    synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=3000)
    off = np.arange(K + 1, dtype=np.int64) * T
    eng = _engine()
    eng.set_observations_device("discrete", obs.data_ptr(), off, n, nsymbols=M)
    res = eng.estep(A_eval, pi, B_eval)
    assert eng.get_option("spec_ok") == 1 and eng.get_option("spec_fail") == 0
    np.testing.assert_allclose(res.state_counts.sum(), K * T, rtol=1e-11)
    np.testing.assert_allclose(res.C.sum(), K * (T - 1), rtol=1e-11)
    np.testing.assert_allclose(res.gamma0_sum.sum(), K, rtol=1e-11)
    np.testing.assert_allclose(res.symbol_counts.sum(), K * T, rtol=1e-11)
    # gamma rows sum to one: the weighted count of symbol o over all states is its frequency
    np.testing.assert_allclose(res.symbol_counts.sum(axis=0),
                               torch.bincount(obs, minlength=M).cpu().numpy(), rtol=1e-11)
    assert np.all(np.isfinite(res.logL_k)) and res.logL_k.shape == (K,)
    np.testing.assert_allclose(res.loglik, res.logL_k.sum(), rtol=1e-12)
    full_chunk, full_W = eng.chunk_len, int(eng.get_option("spec_W"))   # the plan the bench's `value` runs on
    full_per_traj = eng.num_chunks // K
    assert eng.num_chunks == K * full_per_traj and full_per_traj > 32
    eng.close()                       # frees the 65 GB workspace
    # The same time decomposition -- chunk length and warm-up of the full batch, forced -- on a 2-trajectory
    # sub-batch, every statistic against the oracle (maximum_likelihood.py:249-282 on _hidden.c:16-183)
    kk = [0, 1023]
    sub2 = torch.cat([obs[k * T:(k + 1) * T] for k in kk])
    eng3 = _engine()
    eng3.set_option("spec_W", full_W)
    eng3.set_observations_device("discrete", sub2.data_ptr(), off[:3], n, nsymbols=M, chunk=full_chunk)
    assert eng3.chunk_len == full_chunk and eng3.num_chunks == 2 * full_per_traj
    res3 = eng3.estep(A_eval, pi, B_eval)
    assert eng3.get_option("spec_ok") == 1 and int(eng3.get_option("spec_W")) == full_W
    np.testing.assert_allclose(res3.logL_k, res.logL_k[kk], rtol=1e-12)
    host2 = [sub2[:T].cpu().numpy(), sub2[T:].cpu().numpy()]
    ref2 = orc.estep("discrete", host2, A_eval, pi, B_eval, want_gamma=True)
    np.testing.assert_allclose(res3.logL_k, ref2["logL"], rtol=1e-11)
    np.testing.assert_allclose(res3.C, ref2["C"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(res3.state_counts, ref2["state_counts"], rtol=1e-9)
    np.testing.assert_allclose(res3.gamma0_sum, ref2["gamma0_sum"], rtol=1e-9, atol=1e-14)
    cnt2 = np.zeros((n, M))
    for o2, g2 in zip(host2, ref2["gammas"]):
        orc.update_pout(o2, g2, cnt2)
    np.testing.assert_allclose(res3.symbol_counts, cnt2, rtol=1e-9, atol=1e-9)
    eng3.close()
    del sub2, ref2
    ks = [0, 511, 1023]               # sub-batch: first, middle, last trajectory
    sub = torch.cat([obs[k * T:(k + 1) * T] for k in ks])
    eng2 = _engine()
    eng2.set_observations_device("discrete", sub.data_ptr(), off[:4], n, nsymbols=M, chunk=7001)
    res2 = eng2.estep(A_eval, pi, B_eval)
    np.testing.assert_allclose(res2.logL_k, res.logL_k[ks], rtol=1e-11)
    eng2.close()
    o = obs[1023 * T:].cpu().numpy()
    ref = orc.estep("discrete", [o], A_eval, pi, B_eval)
    np.testing.assert_allclose(res.logL_k[1023], ref["logL"][0], rtol=1e-11)
    del obs, sub
    torch.cuda.empty_cache()


# ---- Gibbs hidden-path step ---------------------------------------------------------------
def test_gibbs_device_stats_equal_host_stats_and_stream_is_partition_independent(golden):
    """bhmm_sample_paths_dev leaves [C | n0 | emission block] on the device: same numbers as the
    host-returning call.  With global stream offsets a shard of the trajectories samples exactly
    the paths the full batch samples; and the device stream is the numpy restatement's."""
    import torch
    from conftest import split
    from oracle_engine import OracleEngine
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=32)
    paths, C, n0, emis = eng.sample_paths(*margs, seed=4242)
    buf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
    assert eng.path_stats_size == 64 + 8 + 24
    paths2 = eng.sample_paths_dev(*margs, buf.data_ptr(), seed=4242, want_paths=True)
    C2, n02, emis2 = eng.unpack_path_stats(buf.cpu().numpy())
    assert np.array_equal(C, C2) and np.array_equal(n0, n02)
    np.testing.assert_allclose(emis2, emis, rtol=1e-13, atol=1e-13)
    assert all(np.array_equal(a, b) for a, b in zip(paths, paths2))
    # numpy restatement of the stream + oracle forward/sampling == device paths
    ora = OracleEngine()
    ora.set_observations("gaussian", obs, 8)
    opaths, oC, on0, _ = ora.sample_paths(*margs, seed=4242)
    assert all(np.array_equal(a, b) for a, b in zip(paths, opaths))
    assert np.array_equal(C, oC) and np.array_equal(n0, on0)
    # a shard with global stream offsets reproduces its trajectories of the full run
    mine = [1, 3, 4]
    goff = np.concatenate([[0], np.cumsum([len(o) for o in obs])]).astype(np.int64)
    sh = _engine()
    sh.set_observations("gaussian", [obs[k] for k in mine], 8, chunk=50)
    sh.set_stream_offsets(goff[mine])
    sp = sh.sample_paths(*margs, seed=4242)[0]
    assert all(np.array_equal(sp[j], paths[k]) for j, k in enumerate(mine))
    sh.set_stream_offsets(None)       # back to local positions: a different stream
    sp = sh.sample_paths(*margs, seed=4242)[0]
    assert not all(np.array_equal(sp[j], paths[k]) for j, k in enumerate(mine))
    sh.close()
    eng.close()
    # discrete + 9..64 states: packed layout [n][M] / gaussian [3][n]
    g = golden("d8_ragged")
    obs = split(g["obs"], g["lengths"])
    M = g["B"].shape[1]
    eng = _engine()
    eng.set_observations("discrete", obs, 8, nsymbols=M, chunk=40)
    _, C, n0, emis = eng.sample_paths(g["A"], g["pi"], g["B"], seed=9)
    buf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
    eng.sample_paths_dev(g["A"], g["pi"], g["B"], None, buf.data_ptr(), seed=9)
    C2, n02, emis2 = eng.unpack_path_stats(buf.cpu().numpy())
    assert np.array_equal(C, C2) and np.array_equal(n0, n02) and np.array_equal(emis, emis2)
    eng.close()
    g = golden("g64")
    obs = [g["obs"][:400], g["obs"][400:]]
    eng = _engine()
    eng.set_observations("gaussian", obs, 64)
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    _, C, n0, emis = eng.sample_paths(*margs, seed=9)
    buf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
    eng.sample_paths_dev(*margs, buf.data_ptr(), seed=9)
    C2, n02, emis2 = eng.unpack_path_stats(buf.cpu().numpy())
    assert np.array_equal(C, C2) and np.array_equal(n0, n02)
    np.testing.assert_allclose(emis2, emis, rtol=1e-13, atol=1e-13)
    eng.close()


def test_gibbs_full_size_sweep():
    """configs[4] shape: one Gibbs hidden-path sweep over 256 x 1e5 (8 states).  Exact integer
    counts (sum C == K (T-1), sum n0 == K, state occupancies == path histogram), and one
    trajectory's path equal to the oracle's backward sampling given the same uniforms."""
    import torch
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    from oracle_engine import device_uniforms
    K, T = 256, 100000
    m = make_c2_model()
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=31)
    eng = _engine()
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
    paths, C, n0, emis = eng.sample_paths(*margs, seed=123)
    assert C.dtype == np.int64 and C.sum() == K * (T - 1) and n0.sum() == K
    allp = np.concatenate(paths)
    assert np.array_equal(emis[0].astype(np.int64), np.bincount(allp, minlength=8))
    assert np.array_equal(n0, np.bincount([p[0] for p in paths], minlength=8))
    # transition counts are the histogram of consecutive pairs
    pairs = np.zeros((8, 8), dtype=np.int64)
    for p in paths[:16]:
        np.add.at(pairs, (p[:-1], p[1:]), 1)
    assert np.all(pairs <= C)
    for k in (0, 200):
        o = obs[k * T:(k + 1) * T].cpu().numpy()
        _, alpha = orc.forward(m["A_eval"], orc.pobs_gaussian(o, m["mu_eval"], m["sigma"]), m["pi"])
        ref = orc.sample_path(alpha, m["A_eval"], u=device_uniforms(123, k * T, T))
        assert np.array_equal(paths[k], ref)
    eng.close()


# ---- lagged views cut on the device --------------------------------------------------------
@pytest.mark.parametrize("kind", ["gaussian", "discrete", "explicit"])
def test_lagged_views_on_device_equal_host_slices(kind):
    """bhmm_ctx_set_observations_lagged (bhmm/api.py:70-94): the views obs_k[shift::lag] cut on
    the GPU from one upload give the E-step, Viterbi paths and statistics of the same views
    uploaded as separate host arrays -- ragged lengths, a trajectory shorter than the lag (its
    pieces are dropped), stride > 1."""
    import bhmm_amd
    rng = np.random.default_rng(5)
    n, M = 4, 6
    A = rng.random((n, n)) + 2 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    lens = (1000, 333, 7, 2, 4001)
    if kind == "gaussian":
        base = [rng.normal(0, 2, T) for T in lens]
        margs = (A, pi, np.linspace(-2, 2, n), np.full(n, 0.9))
    elif kind == "discrete":
        base = [rng.integers(0, M, T).astype(np.int32) for T in lens]
        margs = (A, pi, rng.dirichlet(np.ones(M), size=n))
    else:
        base = [rng.random((T, n)) + 0.05 for T in lens]
        margs = (A, pi)
    for lag, stride in ((5, 1), (7, 3), (1, 1)):
        lagged = bhmm_amd.lag_observations(base, lag, stride)
        assert all(len(v) > 1 for v in lagged) and len(lagged.views) == len(lagged)
        a, b = _engine(), _engine()
        a.set_observations_lagged(kind, lagged.base, lag, lagged.views, n, nsymbols=M if kind == "discrete" else 0)
        b.set_observations(kind, [np.ascontiguousarray(v) for v in lagged], n,
                           nsymbols=M if kind == "discrete" else 0)
        assert np.array_equal(a.lengths, b.lengths)
        ra, rb = a.estep(*margs), b.estep(*margs)
        assert np.array_equal(ra.packed, rb.packed) and np.array_equal(ra.logL_k, rb.logL_k)
        assert all(np.array_equal(x, y) for x, y in zip(a.viterbi(*margs), b.viterbi(*margs)))
        # ... and both are the oracle's E-step on the numpy views obs_k[shift::lag]
        views = [np.ascontiguousarray(v) for v in lagged]
        if kind == "explicit":
            ll = [orc.forward(A, v, pi)[0] for v in views]
            np.testing.assert_allclose(ra.logL_k, ll, rtol=1e-10)
        else:
            ref = orc.estep(kind, views, *margs)
            np.testing.assert_allclose(ra.logL_k, ref["logL"], rtol=1e-10)
            np.testing.assert_allclose(ra.C, ref["C"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(ra.state_counts, ref["state_counts"], rtol=1e-9)
        a.close()
        b.close()


def test_more_lagged_views_than_a_grid_dimension():
    """100 trajectories at lag 1000 are 1e5 views -- more than HIP's 65 535 limit on gridDim.y,
    which the gather kernel used to map views to."""
    import bhmm_amd
    rng = np.random.default_rng(8)
    n = 3
    A = rng.random((n, n)) + 2 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    mu, sig = np.linspace(-1, 1, n), np.full(n, 0.8)
    base = [rng.normal(0, 1.5, 2500 + 13 * k) for k in range(100)]
    lagged = bhmm_amd.lag_observations(base, 1000)
    assert len(lagged) == 100_000 and len(lagged.views) == len(lagged)
    a, b = _engine(), _engine()
    a.set_observations_lagged("gaussian", lagged.base, 1000, lagged.views, n)
    b.set_observations("gaussian", [np.ascontiguousarray(v) for v in lagged], n)
    ra, rb = a.estep(A, pi, mu, sig), b.estep(A, pi, mu, sig)
    assert np.array_equal(ra.packed, rb.packed) and np.array_equal(ra.logL_k, rb.logL_k)
    ref = orc.estep("gaussian", [np.ascontiguousarray(v) for v in lagged[:500]], A, pi, mu, sig)
    np.testing.assert_allclose(ra.logL_k[:500], ref["logL"], rtol=1e-10)
    a.close()
    b.close()


def test_estimate_hmm_with_lag_uses_device_views():
    """estimate_hmm(observations, n, lag=...) (bhmm/api.py:309-372): same fit whether the lagged
    views are cut on the GPU or uploaded as host copies."""
    import bhmm_amd
    from bhmm_amd.estimators.maximum_likelihood import MaximumLikelihoodEstimator
    rs = np.random.RandomState(1)
    model, O, S = bhmm_amd.testsystems.generate_synthetic_observations(nstates=3, ntrajectories=4,
                                                                       length=5000, rng=rs)
    lagged = bhmm_amd.lag_observations(O, 4)
    init = bhmm_amd.gaussian_hmm(np.full(3, 1 / 3.), np.full((3, 3), 0.1) + 0.7 * np.eye(3),
                                 np.array([-1.5, 0.0, 1.5]), np.ones(3))
    e1 = MaximumLikelihoodEstimator(lagged, 3, initial_model=init, maxit=10, accuracy=1e-9)
    e2 = MaximumLikelihoodEstimator(list(lagged), 3, initial_model=init, maxit=10, accuracy=1e-9)
    h1, h2 = e1.fit(), e2.fit()
    np.testing.assert_array_equal(e1.likelihoods, e2.likelihoods)
    np.testing.assert_array_equal(h1.transition_matrix, h2.transition_matrix)
    assert len(h1.hidden_state_trajectories) == len(lagged) == 16
    assert all(np.array_equal(p, q) for p, q in zip(h1.hidden_state_trajectories,
                                                    h2.hidden_state_trajectories))


# ---- discrete alphabets beyond the LDS --------------------------------------------------------
@pytest.mark.parametrize("n,M", [(8, 5000), (3, 40000)])
def test_large_discrete_alphabet(n, M):
    """bhmm/output_models/discrete.py:130-157 has no limit on the number of symbols (discrete HMMs
    on thousands of microstates are the usual case).  Beyond the LDS capacity (M above ~1200 at
    8 states) the emission table and the count tables live in global memory: E-step (incl. the
    weighted symbol counts of _discrete.c:1-32), Viterbi and the Gibbs step against the oracle."""
    rng = np.random.default_rng(M)
    A = rng.random((n, n)) + 3 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    B = rng.dirichlet(np.full(M, 0.05), size=n) + 1e-9          # sparse-ish rows, no exact zeros
    B /= B.sum(axis=1, keepdims=True)
    obs = [rng.integers(0, M, T).astype(np.int32) for T in (20000, 3001, 1, 777)]
    ref = orc.estep("discrete", obs, A, pi, B, want_gamma=True)
    for chunk in (0, 100):
        eng = _engine()
        eng.set_observations("discrete", obs, n, nsymbols=M, chunk=chunk)
        res = eng.estep(A, pi, B)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-9)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9)
        cnt = np.zeros((n, M))
        for o, g in zip(obs, ref["gammas"]):
            orc.update_pout(o, g, cnt)
        np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
        # gamma export and a second E-step (the global tables are cleared every time)
        res2 = eng.estep(A, pi, B, store_gamma=True)
        np.testing.assert_allclose(res2.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(eng.gamma(1), ref["gammas"][1], rtol=1e-8, atol=1e-13)
        # Viterbi, bit-exact
        pobs = [orc.pobs_discrete(o, B) for o in obs]
        for p, pb in zip(eng.viterbi(A, pi, B), pobs):
            assert np.array_equal(p, orc.viterbi(A, pb, pi))
        # Gibbs step: paths given the uniforms, integer statistics
        u = [rng.random(len(o)) for o in obs]
        sp, C, n0, emis = eng.sample_paths(A, pi, B, u=u)
        refp = [orc.sample_path(orc.forward(A, pb, pi)[1], A, u=uu) for pb, uu in zip(pobs, u)]
        assert all(np.array_equal(a, b) for a, b in zip(sp, refp))
        Cr, n0r = orc.path_counts(refp, n)
        assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
        er = np.zeros((n, M))
        for p, o in zip(refp, obs):
            np.add.at(er, (p, o), 1.0)
        assert np.array_equal(emis, er)
        eng.close()


def test_many_short_trajectories():
    """5000 ragged trajectories of 1..40 steps (the other extreme of the BASELINE shapes: many
    independent short trajectories): per-trajectory log-likelihoods, transition counts, gamma_0 sums
    against the oracle.  More than 64 trajectory blocks: the totals come from k_tail_total."""
    rng = np.random.default_rng(77)
    n = 8
    A = rng.random((n, n)) + 2 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    mu, sig = np.linspace(-4, 4, n), rng.uniform(0.5, 1.5, n)
    lens = rng.integers(1, 41, 5000)
    obs = [rng.normal(0, 3, T) for T in lens]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = _engine()
    eng.set_observations("gaussian", obs, n)
    for _ in range(2):
        res = eng.estep(A, pi, mu, sig)
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-9)
        np.testing.assert_allclose(res.loglik, ref["logL"].sum(), rtol=1e-11)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=1e-9)
        np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=1e-9)
    # a non-finite trajectory is still found and reported (maximum_likelihood.py:385)
    bad = [o.copy() for o in obs]
    bad[4321][0] = np.nan
    eng.set_observations("gaussian", bad, n)
    with pytest.raises(AssertionError):
        eng.estep(A, pi, mu, sig)
    eng.close()


def test_gibbs_step_with_failing_boundary_check(golden):
    """The Gibbs step launches its sampling kernels without waiting for the boundary check of the
    speculative forward pass; a failed check (warm-up far too short) must be noticed afterwards
    and the step repeated on exact alpha rows -- same paths as the oracle."""
    from conftest import split
    g = golden("g8_ragged")
    obs = split(g["obs"], g["lengths"])
    margs = (g["A"], g["pi"], g["mu"], g["sigma"])
    rng = np.random.default_rng(3)
    u = [rng.random(len(o)) for o in obs]
    ref = [orc.sample_path(orc.forward(g["A"], orc.pobs_gaussian(o, g["mu"], g["sigma"]), g["pi"])[1],
                           g["A"], u=uu) for o, uu in zip(obs, u)]
    eng = _engine()
    eng.set_observations("gaussian", obs, 8, chunk=16)
    eng.set_option("spec_W", 2)
    for _ in range(3):
        paths, C, n0, _ = eng.sample_paths(*margs, u=u)
        assert all(np.array_equal(a, b) for a, b in zip(paths, ref))
    assert eng.get_option("spec_fail") >= 1
    Cr, n0r = orc.path_counts(ref, 8)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    eng.close()


def test_discrete_viterbi_one_million_steps():
    """Discrete Viterbi at T = 1e6 (configs[2]'s trajectory length; the chunk-parallel kernel reads
    B directly, no (T, n) emission matrix is materialised): bit-exact against the oracle."""
    import torch
    from bhmm_amd.engine import synth_observations
    n, M, A, pi, B, A_eval, B_eval = _c3_model()
    K, T = 8, 1000000
    obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
    synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=41)
    eng = _engine()
    eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n,
                                nsymbols=M)
    paths = eng.viterbi_u8(A_eval, pi, B_eval).reshape(K, T)
    assert eng.get_option("viterbi_chunked") == 1
    for k in (0, 7):
        o = obs[k * T:(k + 1) * T].cpu().numpy()
        assert np.array_equal(paths[k], orc.viterbi(A_eval, orc.pobs_discrete(o, B_eval), pi))
    eng.close()


def test_wide_family_large_alphabet():
    """16 states (the 9..64-state kernels) with 5000 symbols: E-step, Viterbi and the Gibbs step
    with its symbol-count table beyond the LDS (per-trajectory tables in global memory)."""
    rng = np.random.default_rng(16)
    n, M = 16, 5000
    A = rng.random((n, n)) + 4 * np.eye(n)
    A /= A.sum(axis=1, keepdims=True)
    pi = rng.dirichlet(np.ones(n))
    B = rng.dirichlet(np.full(M, 0.05), size=n) + 1e-9
    B /= B.sum(axis=1, keepdims=True)
    obs = [rng.integers(0, M, T).astype(np.int32) for T in (3000, 777, 1)]
    ref = orc.estep("discrete", obs, A, pi, B, want_gamma=True)
    eng = _engine()
    eng.set_observations("discrete", obs, n, nsymbols=M)
    res = eng.estep(A, pi, B)
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-9)
    np.testing.assert_allclose(res.C, ref["C"], rtol=1e-9, atol=1e-12)
    cnt = np.zeros((n, M))
    for o, g in zip(obs, ref["gammas"]):
        orc.update_pout(o, g, cnt)
    np.testing.assert_allclose(res.symbol_counts, cnt, rtol=1e-9, atol=1e-12)
    pobs = [orc.pobs_discrete(o, B) for o in obs]
    for p, pb in zip(eng.viterbi(A, pi, B), pobs):
        assert np.array_equal(p, orc.viterbi(A, pb, pi))
    u = [rng.random(len(o)) for o in obs]
    sp, C, n0, emis = eng.sample_paths(A, pi, B, u=u)
    refp = [orc.sample_path(orc.forward(A, pb, pi)[1], A, u=uu) for pb, uu in zip(pobs, u)]
    assert all(np.array_equal(a, b) for a, b in zip(sp, refp))
    er = np.zeros((n, M))
    for p, o in zip(refp, obs):
        np.add.at(er, (p, o), 1.0)
    assert np.array_equal(emis, er)
    eng.close()


def test_very_long_chunks_replan_for_a_slowly_forgetting_model():
    """Automatic plans take two or three times the default chunk count when chunks are very long,
    assuming a warm-up of a few hundred steps.  A model that forgets over thousands of steps (dwell
    times of 1e4 steps, strongly overlapping emissions) goes back to the default count at the
    calibration of its first E-step: same log-likelihoods as an explicit coarse plan, unit gamma mass,
    and for one trajectory the oracle's value."""
    import torch
    from bhmm_amd.engine import synth_observations
    n, K, T = 8, 512, 1_190_000                    # 6.1e8 steps: chunks of 18 600 steps at 32 768 chunks
    A = np.full((n, n), 1e-4 / (n - 1))
    np.fill_diagonal(A, 1.0 - 1e-4)
    pi = np.full(n, 1.0 / n)
    mu, sig = np.linspace(-0.7, 0.7, n), np.full(n, 1.0)
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sig, K, T, seed=11)
    off = np.arange(K + 1, dtype=np.int64) * T
    eng = _engine()
    eng.set_observations_device("gaussian", obs.data_ptr(), off, n)
    fine = eng.chunk_len
    assert fine < 12000                            # the plan with the tripled / doubled chunk count
    # gamma rows are stored by the very E-step that re-plans: they must be sized for the NEW plan
    # and be reported as stored
    res = eng.estep(A, pi, mu, sig, store_gamma=True)
    W = eng.get_option("spec_W")
    assert W * 16 > fine and eng.chunk_len > 1.9 * fine, (W, fine, eng.chunk_len)   # re-planned
    np.testing.assert_allclose(res.state_counts.sum(), K * T, rtol=1e-10)
    np.testing.assert_allclose(res.C.sum(), K * (T - 1), rtol=1e-10)
    g_first = eng.gamma(K - 1)
    np.testing.assert_allclose(g_first.sum(axis=1), 1.0, rtol=1e-10)
    res_b = eng.estep(A, pi, mu, sig, store_gamma=True)
    np.testing.assert_allclose(res_b.logL_k, res.logL_k, rtol=1e-13)
    np.testing.assert_allclose(eng.gamma(K - 1), g_first, rtol=0, atol=1e-12)
    del g_first
    coarse = eng.chunk_len
    eng.close()
    sub = obs[:2 * T]
    eng2 = _engine()
    eng2.set_observations_device("gaussian", sub.data_ptr(), off[:3], n, chunk=2 * coarse + 17)
    res2 = eng2.estep(A, pi, mu, sig)
    np.testing.assert_allclose(res2.logL_k, res.logL_k[:2], rtol=1e-11)
    eng2.close()
    o = obs[:2_000_000].cpu().numpy()
    eng3 = _engine()
    eng3.set_observations("gaussian", [o], n)
    r3 = eng3.estep(A, pi, mu, sig)
    ref = orc.estep("gaussian", [o], A, pi, mu, sig)
    np.testing.assert_allclose(r3.logL_k, ref["logL"], rtol=1e-11)
    eng3.close()
    del obs, sub
    torch.cuda.empty_cache()


# ---- BASELINE configs[4] at its full size: the 100-sample chain ---------------------------------
@pytest.mark.parametrize("reversible", [False, True])
def test_configs4_chain_of_100_samples_full_size(reversible):
    """BASELINE configs[4]: BayesianHMM Gibbs sampler, 8 states, 100 posterior samples x 256
    trajectories (x 1e5 steps) through `bayesian_hmm`'s sampler class, native parameter draws.  With
    2.56e7 observations the posterior is narrow: every sample must be a valid model within a few
    posterior standard deviations of the generating one, the chain must MOVE (no frozen sampler),
    and the samples' spread must have the right order of magnitude (~ 1 / sqrt(counts))."""
    import torch
    import bhmm_amd
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    from bhmm_amd.estimators import _tmatrix
    m = make_c2_model()
    n, K, T = 8, 256, 100000
    buf = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", buf.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=2000)
    host = buf.cpu().numpy().reshape(K, T)
    del buf
    obs = [host[k] for k in range(K)]
    A0 = _tmatrix.mle_reversible(m["pi"][:, None] * m["A"], maxerr=1e-14) if reversible else m["A"]
    init = bhmm_amd.gaussian_hmm(m["pi"], A0, m["mu"], m["sigma"])
    smp = bhmm_amd.BayesianHMMSampler(obs, n, initial_model=init, reversible=reversible,
                                      transition_matrix_sampling_steps=30)
    models = smp.sample(100, nburn=5, seed=11)
    assert len(models) == 100
    A = np.array([h.transition_matrix for h in models])
    mu = np.array([h.output_model.means for h in models])
    sg = np.array([h.output_model.sigmas for h in models])
    np.testing.assert_allclose(A.sum(axis=2), 1.0, rtol=1e-12)
    assert np.all(A >= 0) and np.all(sg > 0)
    # near the generating model (the metastable matrix of the bench is itself reversible up to the
    # rescaling of its rows; 1e-2 is hundreds of posterior standard deviations of slack for A)
    assert np.abs(A.mean(axis=0) - m["A"]).max() < 1e-2
    assert np.abs(mu.mean(axis=0) - m["mu"]).max() < 2e-2 and np.abs(sg.mean(axis=0) - m["sigma"]).max() < 2e-2
    # the chain moves, with the spread a posterior of 2.56e7 observations has
    assert np.all(mu.std(axis=0) > 1e-5) and np.all(mu.std(axis=0) < 5e-3)
    assert np.all(A.std(axis=0)[m["A"] > 1e-3] > 1e-6) and A.std(axis=0).max() < 5e-3
    assert len({h.transition_matrix.tobytes() for h in models}) == 100
    if reversible:
        pi_s = _tmatrix.stationary_vector(models[-1].transition_matrix)
        X = pi_s[:, None] * models[-1].transition_matrix
        np.testing.assert_allclose(X, X.T, atol=1e-12)
    smp._engine.close()


def test_gamma_rows_at_full_size_against_the_oracle():
    """gamma export at BASELINE configs[1]'s full size (256 x 1e5): rows of three trajectories against
    the oracle, not only their sums."""
    import torch
    from bench import make_c2_model
    from bhmm_amd.engine import synth_observations
    m = make_c2_model()
    n, K, T = 8, 256, 100000
    buf = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", buf.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=2000)
    eng = _engine()
    eng.set_observations_device("gaussian", buf.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
    res = eng.estep(*margs, store_gamma=True)
    np.testing.assert_allclose(res.state_counts.sum(), K * T, rtol=1e-10)
    for k in (0, 131, 255):
        o = buf[k * T:(k + 1) * T].cpu().numpy()
        ref = orc.estep("gaussian", [o], *margs, want_gamma=True)
        g = eng.gamma(k)
        np.testing.assert_allclose(g, ref["gammas"][0], rtol=1e-8, atol=1e-13)
        np.testing.assert_allclose(g.sum(axis=1), 1.0, rtol=1e-12)
        np.testing.assert_allclose(res.logL_k[k], ref["logL"][0], rtol=1e-12)
    eng.close()
