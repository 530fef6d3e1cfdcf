"""the parts of the reference's initial-model code that are pure numpy, held to
fixtures written from the reference itself (tests/golden/gen_golden_init.py -> init_refs.npz):
coarse graining and the two regularisers of bhmm/init/discrete.py:26-164, and the Gaussian initial
model of bhmm/init/gaussian.py:26-92 from given k-means centres up to the fractional count matrix
(the reference seeds its k-means++ from the clock, so the centres are part of the fixture).  What
stays unpinned in this row is only what lives in msmtools (PCCA+, the transition-matrix estimate)."""
import os

import numpy as np
import pytest

from bhmm_amd.init import discrete as idisc
from bhmm_amd.init import gaussian as igauss

REFS = np.load(os.path.join(os.path.dirname(__file__), "golden", "init_refs.npz"))


def _opt(a):
    a = np.asarray(a)
    return None if (a.size == 1 and a.ravel()[0] == -1) else [int(v) for v in a]


def _eps(a):
    v = float(a)
    return None if np.isnan(v) else v


def test_coarse_grain_transition_matrix_matches_reference():
    for c in range(int(REFS["cg_cases"])):
        got = idisc.coarse_grain_transition_matrix(REFS["cg%d_P" % c], REFS["cg%d_M" % c])
        np.testing.assert_allclose(got, REFS["cg%d_out" % c], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(got.sum(axis=1), 1.0, rtol=1e-13)
        assert got.min() >= 0.0


def test_regularize_hidden_matches_reference():
    for c in range(int(REFS["rh_cases"])):
        p0, P = REFS["rh%d_p0" % c], REFS["rh%d_P" % c]
        p0_in, P_in = p0.copy(), P.copy()
        q0, Q = idisc.regularize_hidden(p0_in, P_in, reversible=False, stationary=False,
                                        eps=_eps(REFS["rh%d_eps" % c]))
        np.testing.assert_allclose(q0, REFS["rh%d_p0_out" % c], rtol=1e-14, atol=0)
        np.testing.assert_allclose(Q, REFS["rh%d_P_out" % c], rtol=1e-14, atol=0)
        assert np.array_equal(p0_in, p0) and np.array_equal(P_in, P)      # inputs untouched


def test_regularize_pobs_matches_reference():
    for c in range(int(REFS["rp_cases"])):
        B = REFS["rp%d_B" % c]
        B_in = B.copy()
        nonempty = _opt(REFS["rp%d_nonempty" % c])
        got = idisc.regularize_pobs(B_in, nonempty=None if nonempty is None else np.array(nonempty),
                                    separate=_opt(REFS["rp%d_separate" % c]), eps=_eps(REFS["rp%d_eps" % c]))
        np.testing.assert_allclose(got, REFS["rp%d_out" % c], rtol=1e-14, atol=0)
        assert np.array_equal(B_in, B)


@pytest.mark.parametrize("case", range(int(REFS["gi_cases"])))
def test_gaussian_initial_model_from_given_centres_matches_reference(case):
    """mixture weights / means / variances of the reference's fit and its fractional counts."""
    pooled = REFS["gi%d_obs" % case]
    lengths = REFS["gi%d_lengths" % case]
    centers = REFS["gi%d_centers" % case]
    obs = np.split(pooled, np.cumsum(lengths)[:-1])
    w, mu, sig = igauss.fit_gmm1d_from_centers(pooled, centers)
    np.testing.assert_allclose(w, REFS["gi%d_weights" % case], rtol=1e-10)
    np.testing.assert_allclose(mu, REFS["gi%d_means" % case], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sig ** 2, REFS["gi%d_covars" % case], rtol=1e-10)
    N = igauss.fractional_counts(obs, mu, sig)
    np.testing.assert_allclose(N, REFS["gi%d_N" % case], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(N.sum(), sum(max(int(T) - 1, 0) for T in lengths), rtol=1e-12)
    # the public entry point with `centers=`: same emission model, a valid transition matrix
    hmm = igauss.init_model_gaussian1d(obs, len(centers), reversible=True, centers=centers)
    np.testing.assert_allclose(hmm.output_model.means, mu, rtol=1e-13)
    np.testing.assert_allclose(hmm.output_model.sigmas, sig, rtol=1e-13)
    np.testing.assert_allclose(hmm.transition_matrix.sum(axis=1), 1.0, rtol=1e-12)
    with pytest.raises(ValueError):
        igauss.init_model_gaussian1d(obs, len(centers) + 1, centers=centers)
