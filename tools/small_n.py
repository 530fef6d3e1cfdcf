"""Gaussian E-step on 256 x 1e5 steps for 2..8 states (padded to 2 / 4 / 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
K, T = 256, 100000
for n in (2, 3, 4, 5, 8):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    obs = torch.randn(K * T, dtype=torch.float64, device="cuda") * 3
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    for _ in range(3):
        eng.estep(*args)
    t0 = time.perf_counter()
    for _ in range(20):
        eng.estep(*args)
    dt = (time.perf_counter() - t0) / 20
    print("n=%d: %.3f ms per E-step, %.3e steps/s, sweep %.3f ms, W %g, chunks %d, ok/fail %g/%g" % (
        n, dt * 1e3, K * T / dt, eng.kernel_ms(2), eng.get_option("spec_W"), eng.num_chunks,
        eng.get_option("spec_ok"), eng.get_option("spec_fail")))
    eng.close()
