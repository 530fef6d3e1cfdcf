"""Viterbi over time segments accepted by the margins of the decisions ON the path (k_vit_margin, round 5):
when every boundary of the first pass equals its predecessor's vector to 1e-12 and no decision on the path of
that pass was close, the path is the serial run's without any fix-up round -- checked here byte for byte
against the oracle of bhmm/hidden/impl_c/_hidden.c:186-276, together with the cases in which the rule must
NOT be used (tied decisions on the path, the switch off)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _model(n, rng, kind, M=0):
    A = rng.random((n, n)) + 0.05
    A += np.eye(n) * 8.0
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        return A, pi, np.linspace(-6, 6, n), rng.uniform(0.5, 1.5, n)
    return A, pi, rng.dirichlet(np.ones(M), n), None


def _data(kind, rng, lengths, n, M, p0, p1):
    if kind == "gaussian":
        obs = [rng.normal(0, 4, T) for T in lengths]
        return obs, [orc.pobs_gaussian(o, p0, p1) for o in obs]
    obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
    return obs, [orc.pobs_discrete(o, p0) for o in obs]


@pytest.mark.parametrize("n,kind", [(72, "gaussian"), (96, "discrete"), (128, "gaussian")])
def test_margin_acceptance_65_to_128_states(n, kind):
    """Default policy above 64 states: four E-step warm-ups, margins instead of rounds.  Whatever way a call was
    accepted, the paths are the oracle's; with boundaries that are not bit-identical, none further than 1e-12 and
    no close decision, it was accepted by the margins."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(7100 + n)
    M = 21
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (24001, 1, 9000, 2, 700)
    obs, pobs = _data(kind, rng, lengths, n, M, p0, p1)
    ref = [orc.viterbi(A, po, pi) for po in pobs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    used = 0
    for W in (0, 48, 200):      # the policy's warm-up, one that is too short for some boundaries, one in between
        if W:
            eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, p0, p1)
        assert eng.get_option("viterbi_chunked") == 1 and eng.get_option("viterbi_segments") > 10
        mism, far = eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_far")
        mu, close, rounds = (eng.get_option("viterbi_margin_used"), eng.get_option("viterbi_margin_close"),
                             eng.get_option("viterbi_rounds"))
        if mism > 0 and far == 0 and close == 0:
            assert mu == 1 and rounds == 0
        if mu == 0 and mism > 0:
            assert rounds >= 1
        used += mu
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r), W
    assert used >= 1        # (these seeds: the 48- or the 200-step warm-up leaves rounding-noise boundaries only)
    p8 = eng.viterbi_u8(A, pi, p0, p1)
    assert np.array_equal(p8, np.concatenate(ref).astype(np.uint8))
    eng.set_option("viterbi_margin", 0)          # the switch: fix-up rounds only, the same paths
    paths = eng.viterbi(A, pi, p0, p1)
    assert eng.get_option("viterbi_margin_used") == 0
    for p, r in zip(paths, ref):
        assert np.array_equal(p, r)
    eng.close()


@pytest.mark.parametrize("n,kind", [(64, "gaussian"), (40, "discrete"), (20, "gaussian")])
def test_margin_acceptance_up_to_64_states_when_asked_for(n, kind):
    """Up to 32 states the rule is only tried after a call that needed it (a round is cheap there);
    viterbi_margin = 2 asks for it at once.  33 .. 64 states use it from the first call on."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(7300 + n)
    M = 17
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (30011, 1, 9000, 257)
    obs, pobs = _data(kind, rng, lengths, n, M, p0, p1)
    ref = [orc.viterbi(A, po, pi) for po in pobs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    paths = eng.viterbi(A, pi, p0, p1)
    if n <= 32:                                   # (33 .. 64 states use the rule from the first call on)
        assert eng.get_option("viterbi_margin_used") == 0
    for p, r in zip(paths, ref):
        assert np.array_equal(p, r)
    eng.set_option("viterbi_margin", 2)
    used = 0
    for W in (0, 48, 96):
        if W:
            eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, p0, p1)
        assert eng.get_option("viterbi_chunked") == 1
        mism, far, close = (eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_far"),
                            eng.get_option("viterbi_margin_close"))
        if mism > 0 and far == 0 and close == 0:
            assert eng.get_option("viterbi_margin_used") == 1 and eng.get_option("viterbi_rounds") == 0
        used += eng.get_option("viterbi_margin_used")
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r), W
    if n == 64:
        assert used >= 1
    eng.close()


def test_tied_decisions_on_the_path_are_left_to_the_rounds():
    """Pairs of identical states: v[2k] == v[2k + 1] to the bit at every step, so every decision on the path has a
    runner-up with the same product -- the first-maximum rule (_hidden.c:186-200) decides, and a perturbation
    could decide otherwise.  The margin rule must refuse such a pass (or never be asked), the paths stay the
    oracle's."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(99)
    h = 40
    Ah = rng.random((h, h)) + 0.05 + np.eye(h) * 6.0
    A = np.kron(Ah, np.ones((2, 2)))
    A /= A.sum(axis=1)[:, None]
    n = 2 * h
    pi = np.repeat(rng.dirichlet(np.ones(h)), 2) / 2.0
    mu, sig = np.repeat(np.linspace(-5, 5, h), 2), np.repeat(rng.uniform(0.5, 1.5, h), 2)
    obs = [rng.normal(0, 3.5, T) for T in (20000, 5000)]
    ref = [orc.viterbi(A, orc.pobs_gaussian(o, mu, sig), pi) for o in obs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations("gaussian", obs, n)
    for W in (0, 40):
        if W:
            eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, mu, sig)
        assert eng.get_option("viterbi_margin_used") == 0
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r)
    eng.close()


@pytest.mark.parametrize("n,kind", [(160, "gaussian"), (256, "discrete"), (129, "gaussian"), (200, "discrete")])
def test_row_batched_first_pass_129_to_256_states(n, kind):
    """129 .. 256 states: four segments per workgroup share every pass over A (k_gen_viterbi_rows), accepted when all
    boundaries are bit-identical or by the margins on the path, else the serial kernel decides -- the oracle's paths
    byte for byte either way (int32 and one-byte results), ragged lengths and single steps included."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(7700 + n)
    M = 19
    A, pi, p0, p1 = _model(n, rng, kind, M)
    lengths = (9001, 1, 3000, 2, 650)
    obs, pobs = _data(kind, rng, lengths, n, M, p0, p1)
    ref = [orc.viterbi(A, po, pi) for po in pobs]
    eng = Engine(0)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    accepted = 0
    for W in (0, 160):
        if W:
            eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, p0, p1)
        if eng.get_option("viterbi_chunked") == 1:
            accepted += 1
            assert eng.get_option("viterbi_segments") > len(lengths)
            assert eng.get_option("viterbi_mismatch") == 0 or eng.get_option("viterbi_margin_used") == 1
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r), W
        p8 = eng.viterbi_u8(A, pi, p0, p1)
        assert np.array_equal(p8, np.concatenate(ref).astype(np.uint8))
    assert accepted >= 1
    eng.set_option("viterbi_margin", 0)          # the serial kernel
    paths = eng.viterbi(A, pi, p0, p1)
    assert eng.get_option("viterbi_chunked") == 0
    for p, r in zip(paths, ref):
        assert np.array_equal(p, r)
    eng.close()


def _viterbi_vectors(A, pobs, pi):
    """The max-product vectors of _hidden.c:203-281 (same association; used to place margins, not as a
    reference: 1e-15 is enough for that) and the back-pointers."""
    T, n = pobs.shape
    V = np.empty((T, n))
    ptr = np.zeros((T, n), dtype=np.int64)
    v = pi * pobs[0]
    v = v / v.sum()
    V[0] = v
    cols = np.arange(n)
    for t in range(1, T):
        h = v[:, None] * A
        ih = h.argmax(axis=0)                    # first maximum, like the strict > of :186-200
        vn = (pobs[t] * v[ih]) * A[ih, cols]
        v = vn / vn.sum()
        V[t] = v
        ptr[t] = ih
    return V, ptr


def _path_margins(A, V, path):
    """Relative margin of every decision ON the path: runner-up against winner (k_vit_margin's quantity)."""
    T = len(path)
    out = np.full(T, np.inf)
    for t in range(1, T):
        h = V[t - 1] * A[:, path[t]]
        w = h[path[t - 1]]
        h[path[t - 1]] = -1.0
        out[t] = (w - h.max()) / w
    return out


@pytest.mark.parametrize("n,where", [(72, "final"), (72, "middle"), (64, "final"), (64, "middle")])
def test_decision_with_a_margin_between_delta_and_16_delta_goes_to_the_rounds(n, where):
    """A decision on the path whose margin (3e-10) lies between the bound delta on the first pass's deviation
    (about 1e-10 here) and the 16 delta the rule asks for: the margin rule must refuse the pass (the fix-up rounds
    decide), and the paths are the oracle's.  Placed at the final state (_hidden.c:262-267) and at step T - 2.
    64 states: segments of 4 x 64 = 256 steps on T = 40 * 256 - 1 make the last segment 259 steps long
    (plan_segments rounds the boundaries down to multiples of four), so both places lie above the 256th step of
    their segment -- round 5 sized the margin launch by the nominal segment length and never looked there."""
    from bhmm_amd.engine import Engine
    T = 40 * 256 - 1
    tried = 0
    for seed in range(24):      # (most first passes leave every boundary bit-identical: the rule is then not asked)
        rng = np.random.default_rng(8800 + seed)
        A, pi, mu, sig = _model(n, rng, "gaussian")
        obs = rng.normal(0, 4, T)
        pobs = orc.pobs_gaussian(obs, mu, sig)
        V, ptr = _viterbi_vectors(A, pobs, pi)
        if where == "final":
            a = int(V[T - 1].argmax())
            vb = V[T - 1].copy()
            vb[a] = -1.0
            b = int(vb.argmax())
            pobs[T - 1, b] *= V[T - 1, a] / V[T - 1, b] * (1.0 - 3e-10)
            V, ptr = _viterbi_vectors(A, pobs, pi)
            order = np.sort(V[T - 1])
            planted = (order[-1] - order[-2]) / order[-1]
        else:
            path0 = orc.viterbi(A, pobs, pi)
            t = T - 2
            h = V[t - 1] * A[:, path0[t]]
            w = int(path0[t - 1])
            hb = h.copy()
            hb[w] = -1.0
            i2 = int(hb.argmax())
            pobs[t - 1, i2] *= h[w] / h[i2] * (1.0 - 3e-10)   # v_{t-1}[i2] up to just below the winner's product
            V, ptr = _viterbi_vectors(A, pobs, pi)
        ref = orc.viterbi(A, pobs, pi)
        if where == "middle":
            planted = _path_margins(A, V, ref)[T - 3:].min()
        if not (1.5e-10 < planted < 6e-10):
            continue                                          # (the change moved the path: next seed)
        eng = Engine(0)
        eng.set_option("viterbi_seg_per_simd", 1)
        eng.set_observations("explicit", [pobs], n)
        if n <= 64:
            eng.set_option("viterbi_margin", 2)               # (up to 64 states the rule is only used when asked for)
            eng.set_option("viterbi_seg_warmups", 4)
        eng.set_option("viterbi_W", 64)
        path = eng.viterbi(A, pi)[0]
        assert np.array_equal(path, ref)
        assert eng.get_option("viterbi_chunked") == 1
        assert eng.get_option("viterbi_segments") == (40 if n <= 64 else 160)
        mism, far = eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_far")
        if mism > 0 and far == 0:                             # the first pass was put to the margin rule
            tried += 1
            assert eng.get_option("viterbi_margin_close") >= 1
            assert eng.get_option("viterbi_margin_used") == 0 and eng.get_option("viterbi_rounds") >= 1
        p8 = eng.viterbi_u8(A, pi)
        assert np.array_equal(p8, ref.astype(np.uint8))
        eng.close()
        if tried >= (2 if n > 64 else 1):
            break
    assert tried >= 1


@pytest.mark.parametrize("n,kind", [(64, "gaussian"), (24, "discrete"), (96, "gaussian"), (160, "gaussian")])
def test_mending_round_for_boundaries_further_than_the_tolerance(n, kind):
    """A metastable model (lifetimes of 10 .. 100 steps, overlapping emissions): after a short warm-up some segments
    start further than 1e-12 from their predecessors' vectors.  Those alone are run again up to a kept vector of the
    first pass (k_wide_viterbi_seg / k_gen_viterbi_seg / k_gen_viterbi_rows<.., MEND>, mend_tol), the rest of the pass is accepted by the margins on its path -- the
    oracle's paths byte for byte, as with the mending switched off (fix-up rounds) and with the margins off."""
    from bhmm_amd.engine import Engine
    rng = np.random.default_rng(6600 + n)
    M = 12
    life = np.exp(np.linspace(np.log(10.0), np.log(100.0), n))
    A = rng.random((n, n)) + 0.05
    np.fill_diagonal(A, 0.0)
    A = A / A.sum(axis=1)[:, None] / life[:, None]
    A[np.arange(n), np.arange(n)] = 1.0 - 1.0 / life
    pi = np.full(n, 1.0 / n)
    lengths = (40000, 25000, 3, 30000) if n <= 128 else (14000, 9000, 3)
    if kind == "gaussian":
        p0, p1 = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
        states = [rng.integers(0, n, T) for T in lengths]
        obs = [p0[s] + p1[s] * rng.normal(0, 1, len(s)) for s in states]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        p0, p1 = rng.dirichlet(np.ones(M) * 0.7, n), None
        obs = [rng.integers(0, M, T).astype(np.int32) for T in lengths]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    ref = [orc.viterbi(A, po, pi) for po in pobs]
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
    eng.set_option("viterbi_margin", 2)
    mended = 0
    seen = []
    for W in (256, 128, 64):
        eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, p0, p1)
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r), W
        seen.append((W, eng.get_option("viterbi_chunked"), eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_far"),
                     eng.get_option("viterbi_mended"), eng.get_option("viterbi_margin_used"), eng.get_option("viterbi_rounds")))
        if eng.get_option("viterbi_chunked") == 0:
            break                                 # (too short a warm-up for the rounds as well: the serial kernel from now on)
        if eng.get_option("viterbi_mended") > 0 and eng.get_option("viterbi_margin_used") == 1:
            mended += 1
            assert eng.get_option("viterbi_far") == eng.get_option("viterbi_mended") and eng.get_option("viterbi_rounds") == 0
    assert mended >= 1, seen
    eng.set_option("viterbi_mend", 0)             # the same calls with fix-up rounds instead
    for W in (256, 128):
        eng.set_option("viterbi_W", W)
        paths = eng.viterbi(A, pi, p0, p1)
        assert eng.get_option("viterbi_mended") == 0
        for p, r in zip(paths, ref):
            assert np.array_equal(p, r), W
    p8 = eng.viterbi_u8(A, pi, p0, p1)
    assert np.array_equal(p8, np.concatenate(ref).astype(np.uint8))
    eng.close()
