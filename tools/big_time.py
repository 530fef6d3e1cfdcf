"""More than 128 states (big_kernels.hpp: A streamed from L2): E-step time with fixed data, per kernel.
   python tools/big_time.py [n ...]        BIG_W=<warm-up> fixes the warm-up"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
shapes = {129: (128, 4000), 192: (128, 4000), 256: (128, 4000), 300: (128, 4000), 384: (64, 4000), 512: (64, 2000)}
for n in [int(a) for a in sys.argv[1:]] or [192, 256, 384, 512]:
    K, T = shapes.get(n, (128, 10000 if n <= 128 else 4000))
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    g = torch.Generator(device=dev); g.manual_seed(n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev, generator=g) * 3.0
    eng = Engine(0)
    if os.environ.get("BIG_W"):
        eng.set_option("spec_W", int(os.environ["BIG_W"]))
    if os.environ.get("BIG_TILE") == "0":
        eng.set_option("tile", 0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    r = eng.estep(*margs); eng.estep(*margs)
    dt = timeit(lambda: eng.estep(*margs), 3, eng.sync)
    flops = 6.0 * n * n * K * T
    print("n=%d K=%d T=%d: E-step %.2f ms (%.2f TFLOP/s = %.3f of 78.6) fwd %.2f bwd+xi %.2f | tile %d reason %d trouble %d W %d segs %d dev %.1e fail %d logL %.6f"
          % (n, K, T, 1e3 * dt, flops / dt / 1e12, flops / dt / 1e12 / 78.6, eng.kernel_ms(0), eng.kernel_ms(2),
             eng.get_option("tile"), eng.get_option("tile_reason"), eng.get_option("wide_trouble"), eng.get_option("spec_W"),
             eng.get_option("wide_segments"), eng.get_option("spec_last_dev"), eng.get_option("spec_fail"), r.loglik), flush=True)
    eng.close()
