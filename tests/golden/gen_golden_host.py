#!/usr/bin/env python
"""Golden fixtures for the HOST-side rows of SURVEY.md 8(f), written from the parts of the
reference that run without msmtools (pure numpy).  Build container only (needs /root/reference);
never imported by tests.  Writes tests/golden/host_refs.npz holding inputs and expected outputs of

  * bhmm/estimators/_tmatrix_disconnected.py:126-190  transition_matrix_partial_rev
    (random count matrices with a reversible set S that has outgoing counts) and :59-65 nonempty_set;
  * bhmm/output_models/gaussian.py:274-320  GaussianOutputModel.sample  under np.random.seed;
  * bhmm/output_models/discrete.py:217-251  DiscreteOutputModel.sample  under np.random.seed;
  * bhmm/util/statistics.py:34-151  confidence_interval / confidence_interval_arr.

The functions of that module which call msmtools (connected_sets, estimate_P's reversible
branches, stationary_distribution, is_reversible, rdl_decomposition) cannot run here and stay
"parity unpinned" (DESIGN.md).

    python tests/golden/gen_golden_host.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)

from gen_golden import import_reference_output_models  # noqa: E402  (msmtools stand-ins)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    GaussianOutputModel, DiscreteOutputModel = import_reference_output_models()
    tmd = load("bhmm/estimators/_tmatrix_disconnected.py", "ref_tmatrix_disconnected")
    from bhmm.util import statistics as ref_stat
    out = {}

    # -- partially reversible estimator ---------------------------------------------------
    rng = np.random.default_rng(126190)
    ncase = 0
    for n, ns in ((3, 2), (4, 2), (5, 3), (6, 4), (8, 5), (8, 7), (7, 1)):
        for density in (1.0, 0.6):
            C = rng.random((n, n)) * rng.choice([5.0, 200.0, 3e4])
            C[rng.random((n, n)) > density] = 0.0
            S = np.zeros(n, dtype=bool)
            S[rng.permutation(n)[:ns]] = True
            # the set must be connected among itself and have outgoing counts
            idx = np.where(S)[0]
            for a, b in zip(idx, np.roll(idx, 1)):
                C[a, b] += 1.0 + rng.random()
            C[idx[0], np.where(~S)[0][0]] += 0.5 + rng.random()
            P = np.eye(n)
            tmd.transition_matrix_partial_rev(C, P, S, maxiter=1000000, maxerr=1e-12)
            out["prev%d_C" % ncase] = C
            out["prev%d_S" % ncase] = S
            out["prev%d_P" % ncase] = P
            ncase += 1
    out["prev_cases"] = np.array(ncase)
    Cn = np.array([[0.0, 2.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.5, 0.0, 0.0, 1.0]])
    out["nonempty_C"] = Cn
    out["nonempty_0"] = np.asarray(tmd.nonempty_set(Cn))
    out["nonempty_1"] = np.asarray(tmd.nonempty_set(Cn, mincount_connectivity=1.0))

    # -- Gibbs emission draws under np.random.seed ------------------------------------------
    rng = np.random.default_rng(274320)
    for case, (n, sizes) in enumerate(((3, (50, 1, 0)), (4, (1000, 2, 37, 5)), (2, (7, 20000)))):
        mu = rng.normal(0, 2, n)
        sig = rng.random(n) + 0.5
        obs = [rng.normal(mu[i], sig[i], sizes[i]) for i in range(n)]
        gm = GaussianOutputModel(n, means=mu.copy(), sigmas=sig.copy())
        np.random.seed(1000 + case)
        gm.sample([o.copy() for o in obs])
        out["gs%d_mu" % case] = mu
        out["gs%d_sigma" % case] = sig
        out["gs%d_sizes" % case] = np.array(sizes)
        out["gs%d_obs" % case] = np.concatenate(obs)
        out["gs%d_seed" % case] = np.array(1000 + case)
        out["gs%d_mu_new" % case] = np.array(gm.means)
        out["gs%d_sigma_new" % case] = np.array(gm.sigmas)
    out["gs_cases"] = np.array(3)
    for case, (n, M) in enumerate(((2, 3), (3, 10), (4, 6))):
        B = rng.dirichlet(np.ones(M), size=n)
        if case == 2:
            B[:, 0] = 0.0                       # a symbol that never occurs: its entry stays put
            B /= B.sum(axis=1)[:, None]
        obs_by_state = [rng.choice(M, size=rng.integers(0 if i else 40, 200), p=B[i]).astype(np.int64)
                        for i in range(n)]
        dm = DiscreteOutputModel(B.copy())
        prior = getattr(dm, "prior", None)
        np.random.seed(2000 + case)
        dm.sample([o.copy() for o in obs_by_state])
        out["ds%d_B" % case] = B
        out["ds%d_sizes" % case] = np.array([len(o) for o in obs_by_state])
        out["ds%d_obs" % case] = np.concatenate(obs_by_state).astype(np.int32)
        out["ds%d_seed" % case] = np.array(2000 + case)
        out["ds%d_prior" % case] = np.asarray(prior, dtype=np.float64)
        out["ds%d_B_new" % case] = np.array(dm.output_probabilities)
    out["ds_cases"] = np.array(3)

    # -- sample statistics --------------------------------------------------------------------
    rng = np.random.default_rng(34151)
    ci_in, ci_out = [], []
    for size, alpha in ((1, 0.95), (2, 0.5), (10, 0.95), (101, 0.68), (1000, 0.95), (1000, 0.0), (50, 1.0)):
        d = rng.standard_gamma(2.0, size)
        ci_in.append(np.concatenate([[alpha, size], d]))
        ci_out.append(ref_stat.confidence_interval(d, alpha))
    out["ci_n"] = np.array(len(ci_in))
    for i, (a, b) in enumerate(zip(ci_in, ci_out)):
        out["ci%d_in" % i] = a
        out["ci%d_out" % i] = np.array(b, dtype=np.float64)
    data2 = rng.normal(0, 1, (200, 5))
    data3 = rng.normal(0, 1, (64, 3, 4)) ** 2
    lo2, up2 = ref_stat.confidence_interval_arr(data2, conf=0.9)
    lo3, up3 = ref_stat.confidence_interval_arr(data3)
    lol, upl = ref_stat.confidence_interval_arr([data2[i] for i in range(30)], conf=0.5)
    out.update(cia2=data2, cia2_lo=lo2, cia2_up=up2, cia3=data3, cia3_lo=lo3, cia3_up=up3,
               cial_lo=lol, cial_up=upl)

    path = os.path.join(HERE, "host_refs.npz")
    np.savez_compressed(path, **out)
    print("host_refs.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024.0, len(out)))


if __name__ == "__main__":
    main()
