"""configs[3] E-step (64-state Gaussian, 128 x 1e5, data drawn from the model as in bench.py): ms per E-step, repeated --
for same-box comparisons of library variants (BHMM_AMD_LIB=...).   python tools/c3_estep_time.py [reps]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
for _ in range(4):
    eng.estep(*margs)
ts = [1e3 * timeit(lambda: eng.estep(*margs), 5, eng.sync) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]
r = eng.estep(*margs)
print("%s: E-step ms %s | loglik %.10e | W %d tile %d" % (os.environ.get("BHMM_AMD_LIB", "default").split("/")[-2] if os.environ.get("BHMM_AMD_LIB") else "default",
      " ".join("%.3f" % t for t in ts), r.loglik, eng.get_option("spec_W"), eng.get_option("tile")))
