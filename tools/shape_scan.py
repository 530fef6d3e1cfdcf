"""E-step / Gibbs path step / Viterbi time for other state counts and batch shapes than the BASELINE
configs (Gaussian emissions, metastable model of n states): steps/s and the plan that was used.
python tools/shape_scan.py n K T [chunk [M]]   (M > 0: discrete emissions with M symbols)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations
n, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(n)
A = metastable_matrix(n, rng); pi = stationary(A)
eng = Engine(0)
if M == 0:
    mu = np.linspace(-(n - 1), n - 1, n); sg = np.full(n, 0.9)
    obs = torch.empty(K * T, dtype=torch.float64, device="cuda:0")
    synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sg, K, T, seed=n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.1, sg)
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, chunk=chunk)
else:
    B = rng.dirichlet(np.ones(M), size=n)
    obs = torch.empty(K * T, dtype=torch.int32, device="cuda:0")
    synth_observations("discrete", obs.data_ptr(), A, pi, B, None, K, T, seed=n)
    margs = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M, None)
    eng.set_observations_device("discrete", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n,
                                nsymbols=M, chunk=chunk)
def em_like():
    eng.estep_launch(*margs)
    eng.estep_fetch_packed()
for _ in range(5):
    em_like()
dt = timeit(em_like, 5, eng.sync)
sbuf = torch.zeros(eng.path_stats_size, dtype=torch.float64, device="cuda:0")
dg = timeit(lambda: eng.sample_paths_dev(*margs, sbuf.data_ptr(), seed=1), 5, eng.sync)
pdev = torch.empty(K * T, dtype=torch.uint8, device="cuda:0")
dv = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 3, eng.sync)
print(("M=%d " % M if M else "") + "n=%d K=%d T=%d: %d chunks x %d, W %d | E-step %.3f ms (%.2e steps/s, sweep %.3f) | Gibbs %.3f ms | Viterbi %.3f ms (chunked %d)" % (
    n, K, T, eng.num_chunks, eng.chunk_len, eng.get_option("spec_W"), dt * 1e3, K * T / dt, eng.kernel_ms(2),
    dg * 1e3, dv * 1e3, eng.get_option("viterbi_chunked")))
