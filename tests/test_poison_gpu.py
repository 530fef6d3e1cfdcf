"""The debugging aid BHMM_AMD_POISON=1 (fresh device allocations and the LDS of every compute unit filled with
0xFF bytes = NaN doubles / -1 integers) as a regression test: a kernel that reads what nothing wrote fails
here every time instead of once in a few hundred fresh contexts (the backward tile
kernel's read beyond the LDS tile at 65..96 states).  The variable is read once per process, hence the child."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from bhmm_amd.engine import Engine
from oracle import oracle as orc
rng = np.random.default_rng(1)
for n, kind in ((8, "gaussian"), (20, "gaussian"), (48, "gaussian"), (64, "gaussian"), (65, "gaussian"), (96, "discrete"), (128, "gaussian"), (200, "gaussian"), (260, "discrete")):
    M = 12
    A = rng.random((n, n)) + np.eye(n) * 3.0
    A /= A.sum(axis=1)[:, None]
    pi = rng.dirichlet(np.ones(n))
    if kind == "gaussian":
        p0, p1 = np.linspace(-0.1 * n, 0.1 * n, n), rng.uniform(0.5, 1.5, n)
        # (beyond 128 states a narrower spread: at 0.12 n observations 25 sigma from every state make stretches of
        # p o beta exactly zero, and the REFERENCE's own counts come out 0 / 0 there)
        obs = [rng.normal(0, (0.12 if n <= 128 else 0.04) * n, T) for T in (6000, 1, 2500)]
        pobs = [orc.pobs_gaussian(o, p0, p1) for o in obs]
    else:
        p0, p1 = rng.dirichlet(np.ones(M), n), None
        obs = [rng.integers(0, M, T).astype(np.int32) for T in (6000, 1, 2500)]
        pobs = [orc.pobs_discrete(o, p0) for o in obs]
    u = [rng.random(len(o)) for o in obs]
    ref = orc.estep(kind, obs, A, pi, p0, p1)
    for fresh in range(2):
        eng = Engine(0)
        eng.set_option("wide_segment_len", 400)
        eng.set_observations(kind, obs, n, nsymbols=M if kind == "discrete" else 0)
        res = eng.estep(A, pi, p0, p1)
        if n != 128:                  # (at 128 states these data leave the lazily scaled kernels' range: the
            # self-checks fire with and without the poison, and the order-faithful family takes over)
            assert eng.get_option("wide_trouble") == 0, (n, eng.get_option("wide_trouble"))
        if n in (48, 64, 65, 96, 200, 260):
            assert eng.get_option("tile") == 1, (n, eng.get_option("tile_reason"))
        np.testing.assert_allclose(res.logL_k, ref["logL"], rtol=1e-10)
        np.testing.assert_allclose(res.C, ref["C"], rtol=1e-8, atol=1e-10)
        for p, po in zip(eng.viterbi(A, pi, p0, p1), pobs):
            assert np.array_equal(p, orc.viterbi(A, po, pi)), n
        for p, po, uu in zip(eng.sample_paths(A, pi, p0, p1, u=u)[0], pobs, u):
            assert np.array_equal(p, orc.sample_path(orc.forward(A, po, pi)[1], A, u=uu)), n
        eng.close()
print("poisoned run ok")
'''


def test_every_kernel_family_with_poisoned_allocations_and_lds(launcher):
    """(Started through the pre-GPU launcher of tests/conftest.py: this pytest process has initialised the
    GPU by now and must not start programs itself.)"""
    env = dict(os.environ, BHMM_AMD_POISON="1")
    r = launcher.run([[sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tests"))]], timeout=600, env=env)[0]
    assert r["rc"] == 0 and "poisoned run ok" in r["out"], r["out"][-6000:]
