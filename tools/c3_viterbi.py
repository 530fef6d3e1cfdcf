"""Viterbi at the configs[2] shape (8-state discrete, M = 64, 1024 x 1e6) on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations
rng = np.random.default_rng(3000)
n, M, K, T = 8, 64, int(os.environ.get("C3_K", "1024")), 1000000
A = metastable_matrix(n, rng); pi = stationary(A); B = rng.dirichlet(np.ones(M), size=n)
obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=3000)
eng = Engine(0)
eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, nsymbols=M)
args = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
dt = timeit(lambda: eng.viterbi_u8(*args, out=pdev), 2, eng.sync)
print("Viterbi %d x %d discrete: %.1f ms (%.2e steps/s), chunked %d" % (K, T, dt * 1e3, K * T / dt, eng.get_option("viterbi_chunked")))
sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
dt = timeit(lambda: eng.sample_paths_dev(*args, None, sbuf.data_ptr(), seed=1), 2, eng.sync)
C, n0, _ = eng.unpack_path_stats(sbuf.cpu().numpy())
print("Gibbs sweep: %.1f ms (%.2e steps/s), counts ok %s" % (dt * 1e3, K * T / dt, C.sum() == K * (T - 1)))
