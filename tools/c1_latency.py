"""configs[0] (N=3, K=1, T=1e4, gaussian): latency of one E-step on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
rng = np.random.default_rng(1)
n, T = 3, 10000
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.array([-2.0, 0.0, 2.0]), np.array([0.7, 1.0, 0.8])
s = np.zeros(T, dtype=int)
for t in range(1, T):
    s[t] = rng.choice(n, p=A[s[t - 1]])
obs = [mu[s] + sig[s] * rng.standard_normal(T)]
eng = Engine(0)
eng.set_observations("gaussian", obs, n)
for _ in range(5):
    eng.estep(A, pi, mu, sig)
t0 = time.perf_counter()
for _ in range(200):
    r = eng.estep(A, pi, mu, sig)
dt = (time.perf_counter() - t0) / 200
print("configs[0] E-step: %.1f us (%.2e steps/s), chunks %d x %d, W %g, ok/fail %g/%g" % (
    dt * 1e6, T / dt, eng.num_chunks, eng.chunk_len, eng.get_option("spec_W"), eng.get_option("spec_ok"), eng.get_option("spec_fail")))
