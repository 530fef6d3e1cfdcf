"""GPU parity of the row-batched matrix-core E-step (csrc/tile_kernels.hpp): 33..64 states (64: BASELINE
configs[3]; eight wavefronts per tile) and 65..128 states (four wavefronts, xi counts by the
time-parallel GEMM), against the CPU oracle of bhmm/hidden/impl_c/_hidden.c:16-183."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _model(n, rng, kind, M=0, dense=True):
    A = rng.random((n, n)) + 0.02
    if not dense:
        A[rng.random((n, n)) < 0.3] = 0.0
    A += np.eye(n) * (4.0 if dense else 0.5)
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-6, 6, n), rng.uniform(0.3, 1.2, n)
    if kind == "discrete":
        return A, pi, rng.dirichlet(np.ones(M), n), None
    return A, pi, None, None


def _observations(kind, rng, lengths, n, M):
    if kind == "gaussian":
        return [rng.normal(0, 4, T) for T in lengths]
    if kind == "discrete":
        return [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    return [rng.random((T, n)) * rng.random((T, 1)) + 1e-3 for T in lengths]     # explicit emission rows


def _reference(kind, obs, A, pi, p0, p1):
    """oracle E-step; explicit emission rows go through the oracle's single-trajectory routines"""
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


def _check(res, ref, rtol=1e-9):
    np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-11)
    np.testing.assert_allclose(res.C, ref["C"], rtol=rtol, atol=1e-11)
    np.testing.assert_allclose(res.gamma0_sum, ref["gamma0_sum"], rtol=rtol, atol=1e-13)
    np.testing.assert_allclose(res.state_counts, ref["state_counts"], rtol=rtol, atol=1e-11)


@pytest.mark.parametrize("n,kind", [(64, "gaussian"), (64, "discrete"), (64, "explicit"),
                                    (33, "gaussian"), (48, "discrete"), (49, "explicit"), (63, "gaussian"),
                                    (65, "gaussian"), (96, "discrete"), (97, "gaussian"),
                                    (128, "gaussian"), (128, "explicit"), (100, "discrete")])
def test_tile_estep_matches_the_oracle(n, kind):
    """Ragged batch cut into time segments (warm-up boundaries verified on the device): statistics,
    stored gamma rows and emission statistics against the oracle; the tile kernels did the work."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(1000 + n)
    M = 40
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (5003, 1, 2900, 2, 4000, 3, 777)
    obs = _observations(kind, rng, lengths, n, M)
    ref = _reference(kind, obs, A, pi, p0, p1)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 600)
    if n <= 64:
        eng.set_option("spec_W", 96)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    res = eng.estep(A, pi, p0, p1, store_gamma=True)
    assert eng.get_option("tile") == 1 and eng.get_option("careful") == 0, eng.get_option("wide_trouble")
    assert eng.get_option("wide_segments") > len(lengths) and eng.get_option("spec_fail") == 0
    _check(res, ref)
    for k in (0, 2, 3, 6):
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


@pytest.mark.parametrize("n", [64, 80, 128])
def test_tile_estep_sparse_model_and_zero_start_probabilities(n):
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(77 + n)
    A, pi, mu, sig = _model(n, rng, "gaussian", dense=False)
    pi[::3] = 0.0
    pi /= pi.sum()
    obs = [rng.normal(0, 4, T) for T in (3001, 2048, 1500)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 500)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    assert eng.get_option("tile") == 1
    _check(res, ref)
    eng.close()


@pytest.mark.parametrize("n", [64, 100])
def test_tile_kernels_leave_their_range_and_the_checked_kernels_take_over(n):
    """An observation 45 sigma from every state: all densities underflow (outlier row,
    outputmodel.py:126-130).  The lazily scaled tile kernels report it; the E-step is repeated with the
    per-step-normalising family and stays there for this data set."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(5 + n)
    A, pi, mu, sig = _model(n, rng, "gaussian")
    sig[:] = 0.5
    obs = [rng.normal(0, 4, T) for T in (4000, 3000)]
    obs[0][1234] = 60.0
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 500)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    assert eng.get_option("tile") == 0 and eng.get_option("wide_trouble") != 0
    _check(res, ref)
    res = eng.estep(A, pi, mu, sig)
    _check(res, ref)
    eng.close()


def test_tile_and_wavefront_kernels_agree_at_64_states():
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(64)
    n = 64
    A, pi, mu, sig = _model(n, rng, "gaussian")
    obs = [rng.normal(0, 4, T) for T in (20000, 15001, 9000, 1)]
    out = {}
    for tile in (0, 1):
        eng = Engine(0)
        eng.set_option("tile", tile)
        eng.set_option("wide_segment_len", 1000)
        eng.set_observations("gaussian", obs, n)
        out[tile] = eng.estep(A, pi, mu, sig)
        assert eng.get_option("tile") == tile
        eng.close()
    np.testing.assert_allclose(out[1].logL_k, out[0].logL_k, rtol=1e-12)
    np.testing.assert_allclose(out[1].packed, out[0].packed, rtol=1e-9, atol=1e-10)


def test_more_than_64_states_warmup_calibrated_by_forward_passes():
    """A slowly forgetting model: the calibration lengthens the warm-up until the boundaries verify, or
    hands the data set to the serial family -- either way the statistics are the oracle's."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(3)
    n = 72
    A = np.eye(n) * 0.97 + 0.03 * rng.dirichlet(np.ones(n), n)
    A /= A.sum(axis=1)[:, None]
    pi = np.full(n, 1.0 / n)
    mu, sig = np.linspace(-2, 2, n), np.full(n, 2.5)          # uninformative observations
    obs = [rng.normal(0, 3, T) for T in (30000, 30000)]
    ref = orc.estep("gaussian", obs, A, pi, mu, sig)
    eng = Engine(0)
    eng.set_option("wide_segment_len", 4000)
    eng.set_observations("gaussian", obs, n)
    res = eng.estep(A, pi, mu, sig)
    _check(res, ref)
    if eng.get_option("tile") == 1:
        assert eng.get_option("spec_W") > 32 and eng.get_option("spec_last_dev") < 1e-11
    res = eng.estep(0.5 * A + 0.5 / n, pi, mu, sig)           # a faster model afterwards
    _check(res, orc.estep("gaussian", obs, 0.5 * A + 0.5 / n, pi, mu, sig))
    eng.close()
