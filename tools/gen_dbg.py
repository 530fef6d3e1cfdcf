"""why does a >64-state context leave the tile kernels?  python tools/gen_dbg.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [65, 128]:
    K, T = 128, 10000
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    for gen in (True, False):
        g = torch.Generator(device=dev); g.manual_seed(n)
        obs = torch.randn(K * T, dtype=torch.float64, device=dev, generator=g if gen else None) * 3.0
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
        for i in range(3):
            eng.estep(*margs)
            print(n, gen, i, {k: eng.get_option(k) for k in ("tile", "wide_trouble", "careful", "wide_segments", "spec_W", "spec_ok", "spec_fail", "spec_last_dev")}, flush=True)
        eng.close()
