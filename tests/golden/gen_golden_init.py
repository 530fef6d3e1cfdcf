#!/usr/bin/env python
"""Golden fixtures for the initial-model row of SURVEY.md 8(f3), written from the parts of the
reference's initialisers that run here without msmtools.  Build container only (needs
/root/reference); never imported by tests.  Writes tests/golden/init_refs.npz holding inputs and
expected outputs of

  * bhmm/init/discrete.py:26-57   coarse_grain_transition_matrix
  * bhmm/init/discrete.py:60-116  regularize_hidden  (reversible=False: the reversible branch calls
                                  msmtools through enforce_reversible_on_closed)
  * bhmm/init/discrete.py:119-164 regularize_pobs    (with and without `nonempty` / `separate`)
  * bhmm/init/gaussian.py:26-92   init_model_gaussian1d up to the fractional count matrix: the
    vendored mixture fit (bhmm/_external/sklearn/mixture/gmm.py:414-527, pure numpy) and the
    `Nij += outer(pobs[t], pobs[t+1])` loop (:66-78).

What the mixture fit starts from is NOT reproducible in the reference: its k-means++ seeding is a C
extension that calls srand(time(NULL)) (bhmm/_external/clustering/src/kmeans.c:273) and returns
`nstates` of the data points.  The generator therefore hands the fit a fixed choice of data points
through a stand-in for that one function -- a possible outcome of the reference's own seeding -- and
everything after it is the reference's code: EM to its own tolerance, weights / means / variances,
and the count matrix, which is captured where the reference passes it to msmtools
(`msmest.transition_matrix`, :81-85; the call itself cannot run here).

    python tests/golden/gen_golden_init.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)

from gen_golden import import_reference_output_models  # noqa: E402  (msmtools stand-ins)


class _Captured(Exception):
    pass


def main():
    if not hasattr(np, "infty"):            # gmm.py:438 predates numpy 2
        np.infty = np.inf
    import_reference_output_models()
    from bhmm.init import discrete as ref_disc
    from bhmm.init import gaussian as ref_gauss
    out = {}

    # -- coarse graining ----------------------------------------------------------------------
    rng = np.random.default_rng(2657)
    for case, (n, m) in enumerate(((6, 2), (12, 3), (20, 4), (9, 3))):
        P = rng.random((n, n)) ** 2 + np.eye(n) * rng.uniform(1, 4)
        P /= P.sum(axis=1)[:, None]
        M = rng.dirichlet(np.full(m, 0.3), size=n)
        if case == 3:                       # crisp memberships: off-diagonal blocks may come out negative
            M = np.eye(m)[rng.integers(0, m, n)]
            M[:m] = np.eye(m)
        out["cg%d_P" % case] = P
        out["cg%d_M" % case] = M
        out["cg%d_out" % case] = ref_disc.coarse_grain_transition_matrix(P, M)
    out["cg_cases"] = np.array(4)

    # -- regularisation -----------------------------------------------------------------------
    rng = np.random.default_rng(60116)
    for case, (n, eps) in enumerate(((2, None), (3, None), (5, 1e-3), (4, 0.0))):
        P = rng.random((n, n))
        P[rng.random((n, n)) < 0.4] = 0.0
        P += np.eye(n) * 0.5
        P /= P.sum(axis=1)[:, None]
        p0 = rng.random(n)
        p0[rng.integers(0, n)] = 0.0
        p0 /= p0.sum()
        q0, Q = ref_disc.regularize_hidden(p0.copy(), P.copy(), reversible=False, stationary=False, eps=eps)
        out["rh%d_p0" % case] = p0
        out["rh%d_P" % case] = P
        out["rh%d_eps" % case] = np.array(np.nan if eps is None else eps)
        out["rh%d_p0_out" % case] = q0
        out["rh%d_P_out" % case] = Q
    out["rh_cases"] = np.array(4)
    specs = ((3, 8, None, None, None), (2, 6, [0, 1, 3, 4], None, None), (4, 10, None, [2, 7], None),
             (3, 9, [0, 1, 2, 4, 5, 8], [4, 8], 1e-4), (3, 5, None, None, 0.0))
    for case, (n, m, nonempty, separate, eps) in enumerate(specs):
        B = rng.random((n, m))
        B[rng.random((n, m)) < 0.35] = 0.0
        B[:, 0] += 0.1
        B /= B.sum(axis=1)[:, None]
        out["rp%d_B" % case] = B
        out["rp%d_nonempty" % case] = np.array([-1] if nonempty is None else nonempty)
        out["rp%d_separate" % case] = np.array([-1] if separate is None else separate)
        out["rp%d_eps" % case] = np.array(np.nan if eps is None else eps)
        out["rp%d_out" % case] = ref_disc.regularize_pobs(
            B, nonempty=None if nonempty is None else np.array(nonempty), separate=separate, eps=eps)
    out["rp_cases"] = np.array(len(specs))

    # -- Gaussian initial model up to the fractional counts ------------------------------------
    km = sys.modules["bhmm._external.clustering.kmeans_clustering_64"]   # the stand-ins registered above
    msmest = sys.modules["msmtools.estimation"]
    sys.modules["msmtools"].estimation = msmest          # `import msmtools.estimation as msmest` binds this one
    from bhmm._external.sklearn import mixture
    rng = np.random.default_rng(2692)
    cases = (
        (2, (-1.0, 1.0), (0.4, 0.5), (400, 350)),
        (3, (-2.0, 0.0, 2.5), (0.5, 0.4, 0.7), (600, 1, 2, 300)),
        (4, (-3.0, -1.0, 1.0, 3.0), (0.4, 0.4, 0.4, 0.4), (1500,)),
    )
    for case, (n, mus, sigs, lengths) in enumerate(cases):
        # metastable hidden paths, one Gaussian per state
        obs = []
        for T in lengths:
            s = np.empty(T, dtype=np.int64)
            s[0] = rng.integers(0, n)
            for t in range(1, T):
                s[t] = s[t - 1] if rng.random() < 0.93 else rng.integers(0, n)
            obs.append(rng.normal(np.asarray(mus)[s], np.asarray(sigs)[s]))
        pooled = np.concatenate(obs)
        pick = np.sort(rng.choice(pooled.size, n, replace=False))
        centers = pooled[pick][:, None].copy()
        km.init_centers = lambda X, metric, k, c=centers: c.copy()
        got = {}

        def capture(Nij, reversible=True, got=got, **kw):
            got["N"] = np.array(Nij)
            raise _Captured()

        msmest.transition_matrix = capture
        fitted = {}
        orig_fit = mixture.GMM.fit

        def fit(self, X, y=None, fitted=fitted):
            r = orig_fit(self, X, y)
            fitted.update(weights=np.array(self.weights_), means=np.array(self.means_[:, 0]),
                          covars=np.array(self.covars_[:, 0]), converged=bool(self.converged_))
            return r

        mixture.GMM.fit = fit
        try:
            ref_gauss.init_model_gaussian1d([o.copy() for o in obs], n, reversible=True)
        except _Captured:
            pass
        finally:
            mixture.GMM.fit = orig_fit
        assert fitted["converged"], "reference mixture fit did not converge"
        out["gi%d_obs" % case] = pooled
        out["gi%d_lengths" % case] = np.array(lengths)
        out["gi%d_centers" % case] = centers[:, 0]
        out["gi%d_weights" % case] = fitted["weights"]
        out["gi%d_means" % case] = fitted["means"]
        out["gi%d_covars" % case] = fitted["covars"]
        out["gi%d_N" % case] = got["N"]
    out["gi_cases"] = np.array(len(cases))

    path = os.path.join(HERE, "init_refs.npz")
    np.savez_compressed(path, **out)
    print("init_refs.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024.0, len(out)))


if __name__ == "__main__":
    main()
