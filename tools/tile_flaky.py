"""fresh contexts over and over: does the tile path ever leave (self-check, calibration) on data it handles?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for n, K, T in ((65, 128, 10000), (128, 128, 10000), (64, 128, 20000)):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    ref = None
    for rep in range(reps):
        obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0      # new data every time
        ref = None
        junk = torch.full((int(1e8) + 1000 * rep,), float("nan"), dtype=torch.float64, device=dev)
        del junk
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        r = eng.estep(*margs)
        r = eng.estep(*margs)
        st = {k: eng.get_option(k) for k in ("tile", "wide_trouble", "careful", "wide_segments", "spec_W", "spec_ok", "spec_fail", "spec_last_dev")}
        if ref is None:
            ref = r
        dl = abs(r.loglik - ref.loglik) / abs(ref.loglik)
        dc = float(np.max(np.abs(r.C - ref.C)))
        if st["tile"] != 1 or dl > 1e-12 or dc > 1e-6:
            bad += 1
            print("n", n, "rep", rep, st, "dlogL", dl, "dC", dc, flush=True)
        eng.close()
    print("n", n, "done", reps, "reps; last", st, flush=True)
print("bad:", bad)
