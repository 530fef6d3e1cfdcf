"""configs[2] on one GPU (8-state discrete, M = 64, 1024 x 1e6) a few E-steps: for kernel traces / PMC."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine, synth_observations
rng = np.random.default_rng(3000)
n, M = 8, 64
K, T = int(os.environ.get("C3_K", "1024")), int(os.environ.get("C3_T", "1000000"))
A = metastable_matrix(n, rng)
pi = stationary(A)
B = rng.dirichlet(np.ones(M), size=n)
obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=3000)
eng = Engine(0)
eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, nsymbols=M, chunk=int(os.environ.get("C3_CHUNK", "0")))
args = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
for _ in range(4):
    r = eng.estep(*args)
print("ms", eng.kernel_ms(4), eng.kernel_ms(2), eng.chunk_len, eng.get_option("spec_W"))
