import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n in (65, 128):
    K, T = 128, 10000
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    for ps in (1, 2, 4, 8):
        eng = Engine(0)
        eng.set_option("viterbi_seg_per_simd", ps)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        pdev = torch.empty(K * T, dtype=torch.uint8, device=dev)
        for i in range(5):
            dt = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 1, eng.sync)
            print("n=%d per_simd %d call %d: %.2f ms segs %d W %d mismatch %d rounds %d chunked %d" % (n, ps, i, dt * 1e3,
                  eng.get_option("viterbi_segments"), eng.get_option("viterbi_W"), eng.get_option("viterbi_mismatch"),
                  eng.get_option("viterbi_rounds"), eng.get_option("viterbi_chunked")), flush=True)
        eng.close()
