"""The batched backward draw (_hidden.c:283-305,330-378) with uniforms placed 1e-12 either side of a
cumulative-sum edge.  The alpha rows of the time-parallel forward passes equal the serial recursion's only to the
deviation their boundary check measured (<= 1e-11), so such a draw cannot be decided from them: the sampler
kernels record it, k_draw_verify (csrc/draw_verify.hpp) recomputes alpha_t by the serial recursion over a long
window and decides it again with the reference's arithmetic, and a decision that does not stand sends the call
to the exact rows.  Checked state for state against orc.sample_path on the SERIAL alpha (orc.forward), at 8, 64
and 128 states, together with the forced forms of both fallbacks."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _model(n, rng, kind, M):
    A = rng.random((n, n)) + 0.05
    A += np.eye(n) * (2.0 + 0.05 * n)
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-6, 6, n), rng.uniform(0.6, 1.4, n)
    return A, pi, rng.dirichlet(np.ones(M), n), None


def _plant(alpha, A, u, steps, delta=1e-12):
    """u with the uniforms of `steps` = [(t, +1 / -1), ...] moved to delta above / below an interior edge of the
    reference's cumulative sums at that step (given the states the unchanged later steps draw), and the
    reference's path for it.  A change at t only moves steps <= t, so the steps are planted from the top."""
    u = u.copy()
    T, n = alpha.shape
    planted = 0
    for t, sgn in sorted(steps, reverse=True):
        path = orc.sample_path(alpha, A, u=u)
        ps = alpha[t].copy() if t == T - 1 else alpha[t] * A[:, path[t + 1]]      # _hidden.c:349 / :365
        S = 0.0
        for x in ps:
            S += x                                                              # _normalize, ascending
        acc = np.cumsum(ps / S)                                                 # (sequential, like :299-303)
        p = ps / S
        cand = [q for q in range(n - 1) if p[q] > 1e-4 and p[q + 1] > 1e-4 and 1e-3 < acc[q] < 1 - 1e-3]
        if not cand:
            continue
        q = cand[len(cand) // 2]
        u[t] = acc[q] * (1.0 + sgn * delta)
        planted += 1
    return u, orc.sample_path(alpha, A, u=u), planted


def _case(n, kind, lengths, chunk=0):
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(9100 + n)
    M = 23
    A, pi, p0, p1 = _model(n, rng, kind, M)
    if kind == "gaussian":
        obs = [rng.normal(0, 3.5, T) for T in lengths]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    alphas = [orc.forward(A, po, pi)[1] for po in pobs]
    u, ref, planted = [], [], 0
    for al, T in zip(alphas, lengths):
        uu = rng.random(T)
        steps = []
        if T > 2000:
            ts = rng.choice(np.arange(T // 8, T - 3), size=6, replace=False)
            steps = [(int(t), 1 if i % 2 == 0 else -1) for i, t in enumerate(ts)] + [(T - 1, 1)]
        uu, rp, k = _plant(al, A, uu, steps)
        u.append(uu)
        ref.append(rp)
        planted += k
    eng = Engine(0)
    if n > 8:
        eng.set_option("sample_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0, chunk=chunk)
    return eng, (A, pi, p0, p1), u, ref, planted


def _equal(paths, ref):
    return sum(int((p != r).sum()) for p, r in zip(paths, ref)) == 0


@pytest.mark.parametrize("n,kind,lengths,chunk", [
    (8, "gaussian", (200000, 100000, 1, 30000), 4000),
    (6, "discrete", (150000, 2, 70000), 3000),
    (64, "gaussian", (40011, 1, 9000, 20000), 0),
    (20, "discrete", (40011, 1, 9000, 20000), 0),
    (128, "gaussian", (20011, 1, 7000, 2, 300), 0),
    (100, "discrete", (20011, 1, 7000, 2, 300), 0),
])
def test_uniforms_next_to_a_cumulative_sum_edge(n, kind, lengths, chunk):
    eng, model, u, ref, planted = _case(n, kind, lengths, chunk)
    assert planted >= 8
    # (1) the default policy: 64 x the deviation the boundary check measured
    paths, C, n0, emis = eng.sample_paths(*model, u=u)
    assert _equal(paths, ref)
    Cr, n0r = orc.path_counts(ref, n)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    dev = eng.get_option("draw_alpha_dev")
    if n <= 8:
        assert dev > 0.0                                     # the speculative forward pass ran (chunks >> warm-up)
        assert eng.get_option("draw_events") >= planted      # (static watch: 64 x the check's tolerance)
    if 64.0 * dev >= 4e-12:
        assert eng.get_option("draw_checked") >= planted
    # (2) a wide watch: every planted draw is looked at again, all of them stand
    eng.set_option("draw_watch_tol", 1e-9)
    paths = eng.sample_paths(*model, u=u)[0]
    assert _equal(paths, ref)
    if dev > 0.0:
        assert eng.get_option("draw_events") >= planted and eng.get_option("draw_checked") >= planted
        assert eng.get_option("draw_redone") == 0
    # (3) a decision that does not stand (forced): the call is repeated on the exact alpha rows
    eng.set_option("draw_test_redo", 1)
    paths, C, n0, emis = eng.sample_paths(*model, u=u)
    assert _equal(paths, ref)
    assert np.array_equal(C, Cr) and np.array_equal(n0, n0r)
    if dev > 0.0:
        assert eng.get_option("draw_redone") == 1
    eng.set_option("draw_test_redo", 0)
    eng.set_option("draw_watch_tol", 0)
    # (4) the watch off: the contract of round 5 (alpha to the check's tolerance), kept as a switch
    eng.set_option("draw_watch", 0)
    eng.sample_paths(*model, u=u)
    assert eng.get_option("draw_events") == 0
    eng.set_option("draw_watch", 1)
    # (5) the device stream: watched or not, the serial kernels' draw
    seeded = eng.sample_paths(*model, seed=5)[0]
    eng.set_option("spec_enabled", 0)
    serial = eng.sample_paths(*model, seed=5)[0]
    assert eng.get_option("draw_events") == 0
    assert all(np.array_equal(a, b) for a, b in zip(seeded, serial))
    eng.close()
