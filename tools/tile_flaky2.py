"""the sequence of tools/gen_time.py (64 states incl. Viterbi and path sampling, then 65 states), repeated"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for n, K, T in ((64, 128, 10000), (65, 128, 10000)):
        rng = np.random.default_rng(n)
        A = metastable_matrix(n, rng); pi = stationary(A)
        mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
        obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
        eng.estep(*margs); 
        st0 = {k: eng.get_option(k) for k in ("tile", "wide_trouble", "careful", "wide_segments", "spec_W", "spec_ok", "spec_fail", "spec_last_dev")}
        eng.estep(*margs)
        st = {k: eng.get_option(k) for k in ("tile", "wide_trouble", "careful", "wide_segments", "spec_W", "spec_ok", "spec_fail", "spec_last_dev")}
        if "nov" not in sys.argv:
            eng.viterbi(*margs)
            eng.sample_paths(*margs, seed=1, want_paths=False)
        print(rep, n, "first", st0 if st0["tile"] != 1 else "ok", "second", st if st["tile"] != 1 else "ok", flush=True)
        eng.close()
