import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations
dev = torch.device("cuda", 0)
rng = np.random.default_rng(64)
n, K, T = 64, 128, 100000
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sig, K, T, seed=6400, device=0)
mu2 = torch.tensor(mu + 0.05, device=dev); sg = torch.tensor(sig, device=dev)
p = torch.empty(K * T, n, dtype=torch.float64, device=dev)
B = 1 << 20
for a in range(0, K * T, B):
    x = (obs[a:a + B, None] - mu2[None, :]) / sg[None, :]
    p[a:a + B] = torch.exp(-0.5 * x * x) / (sg[None, :] * 2.5066282746310002)
eng = Engine(0)
eng.set_observations_device("explicit", p.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
margs = (0.9 * A + 0.1 / n, pi)
for _ in range(4):
    r = eng.estep(*margs)
ts = [1e3 * timeit(lambda: eng.estep(*margs), 5, eng.sync) for _ in range(4)]
print("explicit: E-step ms", " ".join("%.3f" % t for t in ts), "loglik %.10e" % r.loglik, "tile", eng.get_option("tile"), "W", eng.get_option("spec_W"))
