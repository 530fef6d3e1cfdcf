#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ from the REFERENCE implementation.

Runs only in a container that has /root/reference (it cannot run on the GPU box and is
never imported by tests).  Expected outputs come from

  * the reference C kernels (bhmm/hidden/impl_c/_hidden.c, output_models/impl_c/_gaussian.c,
    _discrete.c) compiled in place into oracle/_ref/libbhmm_ref.so (oracle/Makefile) -- the
    `config.kernel = 'c'` path, which is the contract (bhmm/util/config.py:24);
  * the reference Python kernels (bhmm/hidden/impl_python/hidden.py), loaded straight from
    the reference tree with importlib (needs only numpy);
  * the reference output-model classes (bhmm/output_models/{gaussian,discrete}.py) for the
    emission M-step `estimate`, imported with inert stand-in modules for msmtools and the
    un-built Cython extensions (neither is executed on the paths used here: the models are
    switched to their 'python' implementation).

Only inputs and expected outputs are written -- no reference source text.

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402  (ctypes access to oracle/_ref only)


# ---------------------------------------------------------------------------------------
# reference loaders
# ---------------------------------------------------------------------------------------

def load_ref_python_hidden():
    spec = importlib.util.spec_from_file_location(
        "ref_impl_python_hidden", os.path.join(REF, "bhmm/hidden/impl_python/hidden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Inert(types.ModuleType):
    """Stand-in module whose attributes are further inert objects (never executed here)."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert(self.__name__ + "." + name)

    def __call__(self, *a, **k):
        raise RuntimeError("inert stand-in %s was called" % self.__name__)


def import_reference_output_models():
    for name in ["msmtools", "msmtools.estimation", "msmtools.analysis", "msmtools.util",
                 "msmtools.dtraj", "msmtools.analysis.dense", "msmtools.analysis.dense.pcca",
                 "msmtools.estimation.dense", "msmtools.estimation.dense.transition_matrix",
                 "bhmm.hidden.impl_c.hidden", "bhmm.output_models.impl_c.gaussian",
                 "bhmm.output_models.impl_c.discrete", "bhmm._external.clustering.kmeans_clustering_64",
                 "bhmm._external.clustering.kmeans_clustering_32"]:
        sys.modules.setdefault(name, _Inert(name))
    sys.path.insert(0, REF)
    import bhmm  # noqa: F401
    from bhmm.output_models.gaussian import GaussianOutputModel
    from bhmm.output_models.discrete import DiscreteOutputModel
    return GaussianOutputModel, DiscreteOutputModel


# ---------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------

def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def metastable_T(n, rng, lifetime_min=10.0, lifetime_max=100.0):
    """Recipe of bhmm/util/testsystems.py:26-65 restated with an explicit generator."""
    lt = np.linspace(np.log(lifetime_min), np.log(lifetime_max), n)
    diag = 1.0 - 1.0 / np.exp(lt)
    X = rng.random((n, n))
    X = X + X.T
    T = X / X.sum(axis=1)[:, None]
    for i in range(n):
        T[i, i] = 0
        T[i, :] *= (1.0 - diag[i]) / T[i, :].sum()
        T[i, i] = 1.0 - T[i, :].sum()
    return T


def stationary(T):
    w, v = np.linalg.eig(T.T)
    p = np.real(v[:, np.argmax(np.real(w))])
    return p / p.sum()


def sample_hidden(T, pi, n, rng):
    s = np.empty(n, dtype=np.int64)
    s[0] = rng.choice(len(pi), p=pi)
    cdf = np.cumsum(T, axis=1)
    u = rng.random(n)
    for t in range(1, n):
        s[t] = min(np.searchsorted(cdf[s[t - 1]], u[t]), len(pi) - 1)
    return s


def ref_c_all(A, pi, pobs):
    """Everything the reference C path yields for one trajectory with explicit pobs."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    pobs = np.ascontiguousarray(pobs, dtype=np.float64)
    logL, alpha = orc.ref_forward(A, pobs, pi)
    beta = orc.ref_backward(A, pobs)
    gamma = orc.ref_gamma(alpha, beta)
    C = orc.ref_transition_counts(alpha, beta, A, pobs)
    vit = orc.ref_viterbi(A, pobs, pi)
    return dict(logL=logL, alpha=alpha, beta=beta, gamma=gamma, C=C, viterbi=vit)


def ref_pobs_gaussian(obs, mu, sig, ignore_outliers=True):
    """C p_obs (gaussian.pyx:87-105 -> _gaussian.c:45-70) + outputmodel.py:119-131."""
    p = orc.ref_pobs_gaussian(np.asarray(obs, dtype=np.float64), mu, sig)
    if ignore_outliers:
        out = np.where(p.sum(axis=1) == 0)[0]
        if out.size > 0:
            p[out, :] = 1.0
    return p


def check_python_twin(hp, A, pi, pobs, c):
    """The reference's own test (test_hidden.py:244-256): python vs C kernels, allclose."""
    lp, al = hp.forward(A, pobs, pi, dtype=np.float64)
    be = hp.backward(A, pobs, dtype=np.float64)
    Cp = hp.transition_counts(al, be, A, pobs, dtype=np.float64)
    assert np.allclose(lp, c['logL']) and np.allclose(al, c['alpha'])
    assert np.allclose(be, c['beta']) and np.allclose(Cp, c['C'])
    vp = hp.viterbi(A, pobs, pi, dtype=np.float64)
    assert np.array_equal(vp, c['viterbi']), "python/C viterbi disagree"
    return dict(py_logL=lp, py_viterbi_sha1=sha1(vp.astype(np.int32)))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def sub_rows(T, n=64):
    """Deterministic sample of row indices incl. both ends."""
    if T <= n:
        return np.arange(T)
    idx = np.unique(np.concatenate([np.arange(8), np.arange(T - 8, T),
                                    np.linspace(0, T - 1, n).astype(int)]))
    return idx


# ---------------------------------------------------------------------------------------
# fixtures
# ---------------------------------------------------------------------------------------

def main():
    assert orc.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    hp = load_ref_python_hidden()
    GaussianOutputModel, DiscreteOutputModel = import_reference_output_models()

    # -- KAT1: deterministic toy of bhmm/tests/test_hidden.py:58-71 -------------------
    A = np.array([[0.9, 0.1], [0.1, 0.9]])
    pi = np.array([0.5, 0.5])
    pobs = np.array([[0.1, 0.9]] * 4 + [[0.5, 0.5]] + [[0.9, 0.1]] * 5)
    c = ref_c_all(A, pi, pobs)
    py = check_python_twin(hp, A, pi, pobs, c)
    u = orc.libc_uniforms(10, 42)
    sp = orc.ref_sample_path(c['alpha'], A, pobs, seed=42)
    save("kat1_toy", A=A, pi=pi, pobs=pobs, sample_u=u, sample_path_seed42=sp, **c, **py)

    # -- KAT2: model of test_hidden.py:73-84 with a fixed legacy seed -------------------
    A = np.array([[0.97, 0.02, 0.01], [0.1, 0.8, 0.1], [0.01, 0.02, 0.97]])
    pi = np.array([0.45, 0.1, 0.45])
    mu = np.array([-1.0, 0.0, 1.0])
    sig = np.array([0.5, 0.5, 0.5])
    obs = np.random.RandomState(20260101).randint(3, size=10000).astype(np.float64)
    pobs = ref_pobs_gaussian(obs, mu, sig)
    c = ref_c_all(A, pi, pobs)
    py = check_python_twin(hp, A, pi, pobs, c)
    rows = sub_rows(10000)
    save("kat2_gauss3", A=A, pi=pi, mu=mu, sigma=sig, obs=obs.astype(np.int8), rows=rows,
         logL=c['logL'], C=c['C'], state_counts=c['gamma'].sum(axis=0),
         alpha_rows=c['alpha'][rows], beta_rows=c['beta'][rows], gamma_rows=c['gamma'][rows],
         gamma0=c['gamma'][0], viterbi=c['viterbi'].astype(np.int8),
         viterbi_sha1=sha1(c['viterbi'].astype(np.int32)), **py)

    # -- G8: 8-state Gaussian, ragged batch (dalton recipe, testsystems.py:159-160) -------
    rng = np.random.default_rng(8001)
    N = 8
    Tm = metastable_T(N, rng)
    mu = np.linspace(-5, 5, N)
    sig = np.linspace(0.5, 2.0, N)
    pis = stationary(Tm)
    lengths = [3000, 1, 2, 4097, 257, 64]
    obs_list, per = [], []
    A_eval = 0.9 * Tm + 0.1 / N
    mu_eval = mu + 0.1
    pi_eval = np.full(N, 1.0 / N)
    for k, T in enumerate(lengths):
        s = sample_hidden(Tm, pis, T, rng)
        o = rng.normal(mu[s], sig[s])
        obs_list.append(o)
        pobs = ref_pobs_gaussian(o, mu_eval, sig)
        c = ref_c_all(A_eval, pi_eval, pobs)
        check_python_twin(hp, A_eval, pi_eval, pobs, c)
        per.append(c)
    gm = GaussianOutputModel(N, means=mu_eval.copy(), sigmas=sig.copy())
    gm.set_implementation('python')
    gm.estimate(obs_list, [c['gamma'] for c in per])
    arrs = dict(A=A_eval, pi=pi_eval, mu=mu_eval, sigma=sig, lengths=np.array(lengths),
                obs=np.concatenate(obs_list), logL=np.array([c['logL'] for c in per]),
                C=np.array([c['C'] for c in per]), gamma0=np.array([c['gamma'][0] for c in per]),
                state_counts=np.array([c['gamma'].sum(axis=0) for c in per]),
                mu_new=gm.means, sigma_new=gm.sigmas,
                viterbi=np.concatenate([c['viterbi'] for c in per]).astype(np.int8))
    for k in (0, 3):
        rows = sub_rows(lengths[k])
        arrs['rows%d' % k] = rows
        arrs['alpha_rows%d' % k] = per[k]['alpha'][rows]
        arrs['beta_rows%d' % k] = per[k]['beta'][rows]
        arrs['gamma_rows%d' % k] = per[k]['gamma'][rows]
    arrs['gamma4'] = per[4]['gamma']          # one full gamma (257 x 8)
    # path sampling of trajectory 0 with the libc stream after srand(7)
    arrs['sample_u0'] = orc.libc_uniforms(lengths[0], 7)
    arrs['sample_path0_seed7'] = orc.ref_sample_path(
        per[0]['alpha'], A_eval, np.zeros((lengths[0], N)), seed=7).astype(np.int8)
    save("g8_ragged", **arrs)

    # -- outliers: an observation so far out that every state underflows -----------------
    o = obs_list[4].copy()
    o[[0, 100, 256]] = [1.0e3, -4.0e2, 7.5e2]
    pobs = ref_pobs_gaussian(o, mu_eval, sig)
    assert np.all(pobs[[0, 100, 256]] == 1.0)
    c = ref_c_all(A_eval, pi_eval, pobs)
    save("g8_outliers", A=A_eval, pi=pi_eval, mu=mu_eval, sigma=sig, obs=o, logL=c['logL'],
         C=c['C'], gamma=c['gamma'], viterbi=c['viterbi'].astype(np.int8))

    # -- D8: 8-state discrete, M=64, ragged ------------------------------------------------
    rng = np.random.default_rng(8002)
    M = 64
    B = rng.dirichlet(np.ones(M), size=N)
    Tm = metastable_T(N, rng)
    pis = stationary(Tm)
    lengths = [5000, 300, 1, 1025]
    B_eval = 0.8 * B + 0.2 / M
    A_eval = 0.9 * Tm + 0.1 / N
    obs_list, per = [], []
    for T in lengths:
        s = sample_hidden(Tm, pis, T, rng)
        cdf = np.cumsum(B, axis=1)
        o = np.minimum([np.searchsorted(cdf[si], ui) for si, ui in zip(s, rng.random(T))], M - 1)
        o = np.asarray(o, dtype=np.int32)
        obs_list.append(o)
        dm = DiscreteOutputModel(B_eval.copy())
        pobs = np.ascontiguousarray(dm.p_obs(o))
        c = ref_c_all(A_eval, pis, pobs)
        check_python_twin(hp, A_eval, pis, pobs, c)
        per.append(c)
    dm = DiscreteOutputModel(B_eval.copy())
    dm.set_implementation('python')
    dm.estimate(obs_list, [c['gamma'] for c in per])
    B_c = np.zeros((N, M))
    for o, c in zip(obs_list, per):
        orc.ref_update_pout(o, c['gamma'], B_c)
    B_c /= B_c.sum(axis=1)[:, None]
    assert np.allclose(B_c, dm.output_probabilities)
    rows = sub_rows(lengths[0])
    save("d8_ragged", A=A_eval, pi=pis, B=B_eval, lengths=np.array(lengths),
         obs=np.concatenate(obs_list).astype(np.int8), logL=np.array([c['logL'] for c in per]),
         C=np.array([c['C'] for c in per]), gamma0=np.array([c['gamma'][0] for c in per]),
         state_counts=np.array([c['gamma'].sum(axis=0) for c in per]), B_new=B_c,
         rows0=rows, alpha_rows0=per[0]['alpha'][rows], beta_rows0=per[0]['beta'][rows],
         gamma_rows0=per[0]['gamma'][rows], gamma1=per[1]['gamma'],
         viterbi=np.concatenate([c['viterbi'] for c in per]).astype(np.int8))

    # -- zeros in A and B (alpha entries exactly 0, c != 0) --------------------------------
    A = np.array([[0.5, 0.5, 0.0], [0.0, 0.7, 0.3], [0.2, 0.0, 0.8]])
    B = np.array([[0.6, 0.4, 0.0, 0.0], [0.0, 0.5, 0.5, 0.0], [0.1, 0.0, 0.4, 0.5]])
    pi = np.array([1.0, 0.0, 0.0])
    o = np.array([0, 1, 1, 2, 3, 2, 0, 0, 1, 2, 2, 3, 3, 0, 1], dtype=np.int32)
    pobs = np.ascontiguousarray(DiscreteOutputModel(B.copy()).p_obs(o))
    c = ref_c_all(A, pi, pobs)
    check_python_twin(hp, A, pi, pobs, c)
    save("d3_zeros", A=A, pi=pi, B=B, obs=o, **c)

    # -- the reference's own double-well test trajectory (bhmm/tests/data) ----------------
    o = np.loadtxt(os.path.join(REF, "bhmm/tests/data/2well_traj_100K.dat"), dtype=int)
    M = int(o.max()) + 1
    x = np.arange(M)
    B = np.vstack([np.exp(-0.5 * ((x - 35.0) / 6.0) ** 2), np.exp(-0.5 * ((x - 62.0) / 7.0) ** 2)])
    B = (B + 1e-6) / (B + 1e-6).sum(axis=1)[:, None]
    A = np.array([[0.998, 0.002], [0.003, 0.997]])
    pi = np.array([0.6, 0.4])
    pobs = np.ascontiguousarray(DiscreteOutputModel(B.copy()).p_obs(o))
    c = ref_c_all(A, pi, pobs)
    check_python_twin(hp, A, pi, pobs, c)
    rows = sub_rows(len(o), 128)
    save("d2_doublewell", A=A, pi=pi, B=B, obs=o.astype(np.uint8), logL=c['logL'], C=c['C'],
         state_counts=c['gamma'].sum(axis=0), gamma0=c['gamma'][0], rows=rows,
         alpha_rows=c['alpha'][rows], beta_rows=c['beta'][rows], gamma_rows=c['gamma'][rows],
         viterbi_bits=np.packbits(c['viterbi'].astype(np.uint8)),
         viterbi_sha1=sha1(c['viterbi'].astype(np.int32)))

    # -- N=64 Gaussian (configs[3] shape, short) --------------------------------------------
    rng = np.random.default_rng(8064)
    N = 64
    Tm = metastable_T(N, rng)
    mu = np.linspace(-5, 5, N)
    sig = np.linspace(0.5, 2.0, N)
    pis = stationary(Tm)
    T = 700
    s = sample_hidden(Tm, pis, T, rng)
    o = rng.normal(mu[s], sig[s])
    A_eval = 0.9 * Tm + 0.1 / N
    pobs = ref_pobs_gaussian(o, mu + 0.05, sig)
    c = ref_c_all(A_eval, pis, pobs)
    check_python_twin(hp, A_eval, pis, pobs, c)
    rows = sub_rows(T, 16)
    save("g64", A=A_eval, pi=pis, mu=mu + 0.05, sigma=sig, obs=o, logL=c['logL'], C=c['C'],
         state_counts=c['gamma'].sum(axis=0), gamma0=c['gamma'][0], rows=rows,
         alpha_rows=c['alpha'][rows], beta_rows=c['beta'][rows], gamma_rows=c['gamma'][rows],
         viterbi=c['viterbi'].astype(np.int8))

    # -- emission pdf anchors of bhmm/tests/test_output_gaussian.py:31-35 -------------------
    rng = np.random.default_rng(31)
    o = rng.standard_normal(2000)
    mu = np.array([-1.0, 0.0, 1.0])
    sig = np.array([0.5, 1.0, 2.0])
    gm = GaussianOutputModel(3, means=mu.copy(), sigmas=sig.copy())
    gm.set_implementation('python')
    p_py = gm.p_obs(o)
    p_c = ref_pobs_gaussian(o, mu, sig)
    assert np.allclose(p_py, p_c)
    save("pobs_gauss3", obs=o, mu=mu, sigma=sig, pobs=p_c)


if __name__ == "__main__":
    main()
