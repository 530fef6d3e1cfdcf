"""Segment-parallel Viterbi for 9..64 states: time against the warm-up length (spec_W fixed by the caller)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations

dev = torch.device("cuda", 0)
for kind, n, K, T in (("gaussian", 64, 128, 10000), ("gaussian", 32, 128, 10000), ("gaussian", 64, 128, 100000)):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    p0, p1 = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    obs = torch.empty(K * T, dtype=torch.float64, device=dev)
    synth_observations(kind, obs.data_ptr(), A, pi, p0, p1, K, T, seed=n, device=0)
    margs = (0.9 * A + 0.1 / n, pi, p0 + 0.05, p1)
    for ps in (2,):
        for W in (64, 128, 192, 256, 384, 512, 768, 1024):
            eng = Engine(0)
            eng.set_option("viterbi_seg_per_simd", ps)
            eng.set_option("spec_W", W)
            eng.set_observations_device(kind, obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
            pdev = torch.empty(K * T, dtype=torch.uint8, device=dev)
            for _ in range(2):
                eng.viterbi_u8(*margs, out=pdev)
            dt = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 3, eng.sync)
            print("n=%d %d x %d per_simd %d W %d: %.2f ms  segmented %d  segments %d  W %d  mismatch %d rounds %d"
                  % (n, K, T, ps, W, dt * 1e3, eng.get_option("viterbi_chunked"), eng.get_option("viterbi_segments"),
                     eng.get_option("viterbi_W"), eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_rounds")), flush=True)
            eng.close()
