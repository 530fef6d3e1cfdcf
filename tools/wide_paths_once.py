"""64-state Gaussian model of tools/gen_time.py, 128 x 1e4: ten Viterbi passes and ten Gibbs path steps
over time segments, for a per-kernel profile (tools/profile_r04.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
n, K, T = 64, 128, 10000
rng = np.random.default_rng(n)
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
pdev = torch.empty(K * T, dtype=torch.uint8, device=dev)
eng.estep(*margs)
for _ in range(10):
    eng.viterbi_u8(*margs, out=pdev)
for _ in range(10):
    eng.sample_paths(*margs, seed=1, want_paths=False)
eng.sync()
print("viterbi segmented", eng.get_option("viterbi_chunked"), "W", eng.get_option("viterbi_W"), "rounds", eng.get_option("viterbi_rounds"),
      "| draw segmented", eng.get_option("sample_segmented"), "W", eng.get_option("sample_W"), "rounds", eng.get_option("sample_rounds"))
eng.close()
