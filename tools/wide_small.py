"""E-step timing of the wide family below 64 states (not BASELINE configs): N = 16 and 32,
K = 256 x 1e5 Gaussian."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n in (16, 32):
    rng = np.random.default_rng(n)
    K, T = 256, 100000
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 1.0, n)
    g = torch.Generator(device=dev); g.manual_seed(n)
    s = torch.randint(0, n, (K, T // 50), device=dev, generator=g).repeat_interleave(50, dim=1)
    obs = (torch.tensor(mu, device=dev)[s] + torch.tensor(sig, device=dev)[s]
           * torch.randn((K, T), device=dev, dtype=torch.float64, generator=g)).reshape(-1)
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    for _ in range(4):
        eng.estep(*args)
    dt = timeit(lambda: eng.estep(*args), 3)
    print("N", n, "ms %.2f" % (dt * 1e3), "steps/s %.3g" % (K * T / dt), "segs", eng.get_option("wide_segments"),
          "W", eng.get_option("spec_W"), "ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"))
    eng.close()
