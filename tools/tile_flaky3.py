"""Hunting the transient self-check of the 65-state tile E-step: the sequence of
tools/gen_time.py -- a 64-state context worked and closed, then a fresh 65-state context -- many times;
prints what the first E-steps of the 65-state context report."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
K, T = 128, 10000
mods = {}
for n in (64, 65):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mods[n] = (0.9 * A + 0.1 / n, pi, np.linspace(-5, 5, n) + 0.05, np.linspace(0.5, 2.0, n))
hits = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rep in range(reps):
    obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
    if rep % 2 == 0:
        e = Engine(0)
        e.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 64)
        e.estep(*mods[64]); e.estep(*mods[64]); e.viterbi(*mods[64]); e.sample_paths(*mods[64], seed=1, want_paths=False)
        e.close()
    obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
    e = Engine(0)
    e.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 65)
    e.estep(*mods[65])
    info = (e.get_option("tile"), e.get_option("tile_reason"), e.get_option("wide_trouble"), e.get_option("tile_retries"), e.get_option("spec_W"))
    e.estep(*mods[65])
    if info[3] or info[2] or not info[0]:
        hits += 1
        print("rep", rep, "first E-step: tile %d reason %d self-checks %d retries %d W %d" % tuple(int(x) for x in info), flush=True)
    e.close()
print("tile_flaky3: %d contexts, %d with a self-check / retry / fallback" % (reps, hits))
