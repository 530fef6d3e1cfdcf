import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, synth_gaussian
from bhmm_amd.engine import Engine
K, T = 256, 100000
model = make_c2_model()
obs = torch.from_numpy(synth_gaussian(model, K, T, seed=2000).reshape(-1)).cuda()
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
A, pi, mu, sg = model["A_eval"], model["pi"], model["mu_eval"], model["sigma"]
for _ in range(5):
    eng.estep(A, pi, mu, sg)
n = 200
tl = tf = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    eng.estep_launch(A, pi, mu, sg)
    b = time.perf_counter()
    r = eng.estep_fetch()
    c = time.perf_counter()
    tl += b - a
    tf += c - b
tot = time.perf_counter() - t0
print("per step: total %.1f us, launch call (includes the verdict sync) %.1f us, fetch %.1f us, kernel %.1f us (estep_total event %.1f)" % (
    1e6 * tot / n, 1e6 * tl / n, 1e6 * tf / n, 1e3 * eng.kernel_ms(2), 1e3 * eng.kernel_ms(4)))
