"""configs[3] data of bench.py (drawn from the model): Viterbi over time segments against the warm-up length --
boundaries not bit-identical / further than 1e-12 after the first pass, rounds, margin acceptance, time.
   python tools/c3_vit_scan.py [W ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations
dev = torch.device("cuda", 0)
rng = np.random.default_rng(64)
n, K, T = 64, 128, 100000
A = metastable_matrix(n, rng); pi = stationary(A)
mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), A, pi, mu, sig, K, T, seed=6400, device=0)
margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
out = torch.empty(K * T, dtype=torch.uint8, device=dev)
ref = None
for margin in (2, 0):
    for W in [int(a) for a in sys.argv[1:]] or [128, 192, 256, 384, 512, 640, 768, 904]:
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        eng.set_option("viterbi_margin", margin)
        eng.set_option("viterbi_W", W)
        run = lambda: (eng.set_option("viterbi_W", W), eng.viterbi_u8(*margs, out=out))
        run(); run()
        dt = timeit(run, 3, eng.sync)
        g = eng.get_option
        chk = int(out.to(torch.int64).sum().item())
        ref = chk if ref is None else ref
        print("margin %d W %4d: %.2f ms | mismatch %d far %d rounds %d margin used %d close %d | same paths %s"
              % (margin, W, 1e3 * dt, g("viterbi_mismatch"), g("viterbi_far"), g("viterbi_rounds"), g("viterbi_margin_used"),
                 g("viterbi_margin_close"), chk == ref), flush=True)
        eng.close()
