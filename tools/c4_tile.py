"""configs[3] (64-state Gaussian, 128 x 1e5): the row-batched matrix-core kernels (tile_kernels.hpp)
against the one-segment-per-wavefront kernels, same data -- statistics compared, both timed.
   python tools/c4_tile.py [tile_per_cu ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import timeit
from bhmm_amd.engine import Engine, synth_observations

dev = torch.device("cuda", 0)
rng = np.random.default_rng(64)
n, K, T = int(os.environ.get("C4_N", 64)), 128, 100000      # (C4_N=48: the same comparison for 33..63 states)
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sig, K, T, seed=6400, device=0)
args = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)


def run(tile, per_cu=1, W=None):
    eng = Engine(0)
    eng.set_option("tile", tile)
    eng.set_option("tile_per_cu", per_cu)
    if W:
        eng.set_option("spec_W", W)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    for _ in range(12):
        r = eng.estep(*args)
    dt = timeit(lambda: eng.estep(*args), 3)
    r = eng.estep(*args)
    eng.sync()
    km = eng.kernel_ms_all().copy()
    print("tile %d per_cu %d: %.2f ms  kernels(ms) %s  segs %d W %d ok/fail %d/%d dev %.2e tile_used %d careful %d trouble %d"
          % (tile, per_cu, dt * 1e3, np.round(km, 3), eng.get_option("wide_segments"), eng.get_option("spec_W"),
             eng.get_option("spec_ok"), eng.get_option("spec_fail"), eng.get_option("spec_last_dev"),
             eng.get_option("tile"), eng.get_option("careful"), eng.get_option("wide_trouble")), flush=True)
    eng.close()
    return r


ref = run(0)
for pc in [int(a) for a in sys.argv[1:]] or [1, 2]:
    r = run(1, pc)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    print("   logL rel %.2e  logL_k %.2e  C %.2e (abs %.2e)  counts %.2e  gd %.2e  gdd %.2e  gamma0 %.2e"
          % (abs(r.loglik - ref.loglik) / abs(ref.loglik), rel(r.logL_k, ref.logL_k),
             float(np.max(np.abs(r.C - ref.C) / np.maximum(ref.C, 1e-6))), float(np.max(np.abs(r.C - ref.C))),
             rel(r.state_counts, ref.state_counts), rel(r.sum_gd, ref.sum_gd), rel(r.sum_gdd, ref.sum_gdd),
             float(np.max(np.abs(r.gamma0_sum - ref.gamma0_sum)))), flush=True)
