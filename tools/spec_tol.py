"""BASELINE configs[1] (8-state Gaussian, 256 x 1e5): E-step time and calibrated warm-up against the
option spec_tol (tolerance of the boundary check, default 1e-11); statistics compared with the default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_c2_model, timeit
from bhmm_amd.engine import Engine, synth_observations

dev = torch.device("cuda", 0)
model = make_c2_model()
K, T = 256, 100000
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), model["A"], model["pi"], model["mu"], model["sigma"], K, T, seed=11, device=0)
args = (model["A_eval"], model["pi"], model["mu_eval"], model["sigma"])
ref = None
for tol in (1e-11, 1e-10, 1e-9, 1e-8):
    eng = Engine(0)
    eng.set_option("spec_tol", tol)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
    for _ in range(5):
        r = eng.estep(*args)
    dt = timeit(lambda: eng.estep(*args), 20, eng.sync)
    r = eng.estep(*args)
    ref = ref or r
    print("spec_tol %.0e: E-step %.3f ms  W %d  boundary dev %.2e  ok/fail %d/%d  logL rel %.2e  C rel %.2e"
          % (tol, dt * 1e3, eng.get_option("spec_W"), eng.get_option("spec_last_dev"), eng.get_option("spec_ok"),
             eng.get_option("spec_fail"), abs(r.loglik - ref.loglik) / abs(ref.loglik),
             float(np.max(np.abs(r.C - ref.C) / np.maximum(ref.C, 1e-3)))), flush=True)
    eng.close()
