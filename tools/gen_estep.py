"""65..128 states, 128 x 1e4, seeded data (the shapes of bench.py's more_than_64_states entries): E-step
time and the two kernels' share, for before / after comparisons of the tile kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n in (65, 80, 96, 97, 128):
    K, T = 128, 10000
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    g = torch.Generator(device=dev); g.manual_seed(n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev, generator=g) * 3.0
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    for _ in range(3):
        eng.estep(*margs)
    dt = timeit(lambda: eng.estep(*margs), 5, eng.sync)
    print("n=%d: E-step %.2f ms  forward %.2f  backward + statistics %.2f  W %d segments %d tile %d"
          % (n, dt * 1e3, eng.kernel_ms(0), eng.kernel_ms(2), eng.get_option("spec_W"), eng.get_option("wide_segments"), eng.get_option("tile")), flush=True)
    eng.close()
