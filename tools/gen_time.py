"""Timing of the any-N family (more than 64 states): E-step / Viterbi / Gibbs path step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations
dev = torch.device("cuda", 0)
for n, K, T in ((64, 128, 10000), (65, 128, 10000), (100, 128, 10000), (128, 128, 10000), (256, 128, 4000), (512, 64, 2000)):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0      # (any data times the same)
    eng = Engine(0)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    eng.estep(*margs); eng.estep(*margs)
    dt = timeit(lambda: eng.estep(*margs), 2, eng.sync)
    tv = timeit(lambda: eng.viterbi(*margs), 1, eng.sync)
    ts = timeit(lambda: eng.sample_paths(*margs, seed=1, want_paths=False), 1, eng.sync)
    print("n=%d K=%d T=%d: E-step %.1f ms (%.3g steps/s, %.2f TFLOP/s on 6 n^2), Viterbi %.1f ms, Gibbs path step %.1f ms, kernel ms %s"
          % (n, K, T, 1e3 * dt, K * T / dt, 6.0 * n * n * K * T / dt / 1e12, 1e3 * tv, 1e3 * ts,
             [round(eng.kernel_ms(i), 2) for i in range(5)]),
          "| tile", eng.get_option("tile"), "reason", eng.get_option("tile_reason"), "self-checks", eng.get_option("wide_trouble"),
          "W", eng.get_option("spec_W"), "dev", eng.get_option("spec_last_dev"), "fail", eng.get_option("spec_fail"))
    eng.close()
