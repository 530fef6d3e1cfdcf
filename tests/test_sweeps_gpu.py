"""A fixed slice of the random parity sweeps of tests/sweeps/ (stress_small, stress_hidden, stress_reuse,
stress_em, stress_gibbs, stress_carry, stress_many_states: 9..200 states) as GPU tests: each tool compares the HIP engine with the oracle over random
models / data (1..40 states, both emission kinds, sparse matrices, outliers and very narrow states,
densities in the denormal range, ragged trajectories, odd chunk lengths) and exits non-zero on any
mismatch.  The tools run as fresh processes through the launcher of conftest.py."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seed,cases", [("stress_small.py", 7, 150), ("stress_small.py", 32, 150),
                                             ("stress_hidden.py", 9, 100), ("stress_reuse.py", 2, 100),
                                             ("stress_em.py", 1, 25), ("stress_gibbs.py", 1, 15),
                                             ("stress_carry.py", 3, 30), ("stress_many_states.py", 11, 120)])
def test_random_parity_sweep(launcher, tool, seed, cases):
    r = launcher.run([[sys.executable, os.path.join(ROOT, "tests", "sweeps", tool), str(seed), str(cases)]],
                     timeout=900)[0]
    assert r["rc"] == 0, r["out"][-3000:]
    assert "0 failures" in r["out"], r["out"][-3000:]
