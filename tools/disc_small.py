"""8-state discrete (M = 64) E-step on 256 x 1e5 steps: a small stand-in for configs[2]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
rng = np.random.default_rng(3000)
n, M, K, T = 8, 64, 256, 100000
A = metastable_matrix(n, rng); pi = stationary(A)
B = rng.dirichlet(np.ones(M), size=n)
obs = torch.randint(0, M, (K * T,), dtype=torch.int32, device="cuda")
eng = Engine(0)
eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, nsymbols=M)
args = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
for _ in range(3):
    eng.estep(*args)
t0 = time.perf_counter()
for _ in range(20):
    eng.estep(*args)
dt = (time.perf_counter() - t0) / 20
print("discrete 256 x 1e5: %.3f ms per E-step, %.3e steps/s, sweep %.3f ms, W %g" % (dt * 1e3, K * T / dt, eng.kernel_ms(2), eng.get_option("spec_W")))
