"""configs[3] (64-state Gaussian, 128 x 1e5) E-step a few times: for rocprofv3 kernel traces.
   python tools/c4_once.py [spec_W [segment_len]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import hidden_paths_gpu, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
rng = np.random.default_rng(64)
n, K, T = 64, 128, 100000
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
s = hidden_paths_gpu(A, pi, K, T, 12).long()
obs = (torch.tensor(mu, device=dev)[s] + torch.tensor(sig, device=dev)[s] * torch.randn((K, T), device=dev, dtype=torch.float64)).reshape(-1)
args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
eng = Engine(0)
if os.environ.get("C4_SPLIT") is not None:
    eng.set_option("wide_split", int(os.environ["C4_SPLIT"]))
if len(sys.argv) > 1:
    eng.set_option("spec_W", int(sys.argv[1]))
if len(sys.argv) > 2:
    eng.set_option("wide_segment_len", int(sys.argv[2]))
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
for _ in range(4):
    eng.estep(*args)
dt = timeit(lambda: eng.estep(*args), 3)
print("ms %.2f" % (dt * 1e3), "segs", eng.get_option("wide_segments"), "W", eng.get_option("spec_W"),
      "ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"), "dev", eng.get_option("spec_last_dev"))
