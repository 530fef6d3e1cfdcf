"""Viterbi over time segments at the bench's shapes: time, fix-up rounds, path-margin acceptance.
   python tools/vit_once.py [n ...]      VIT_MARGIN=0 switches the margin acceptance off"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    K, T = (128, 100000) if n == 64 else ((128, 4000) if n > 128 else (128, 10000))
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    g = torch.Generator(device=dev); g.manual_seed(n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev, generator=g) * 3.0
    eng = Engine(0)
    if os.environ.get("VIT_MARGIN") is not None:
        eng.set_option("viterbi_margin", int(os.environ["VIT_MARGIN"]))
    Ws = [int(w) for w in os.environ.get("VIT_W", "0").split(",")]
    eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    eng.estep(*margs)
    out = torch.empty(K * T, dtype=torch.uint8, device=dev)
    run = lambda: eng.viterbi_u8(*margs, out=out)
    for W in Ws:
      if W:
        eng.set_option("viterbi_W", W)
      run(); run()
      dt = timeit(run, 3, eng.sync)
      print("n=%d K=%d T=%d: Viterbi %.2f ms | segments %d W %d mismatch %d far %d rounds %d margin used %d close %d | checksum %d"
            % (n, K, T, 1e3 * dt, eng.get_option("viterbi_segments"), eng.get_option("viterbi_W"), eng.get_option("viterbi_mismatch"), eng.get_option("viterbi_far"),
             eng.get_option("viterbi_rounds"), eng.get_option("viterbi_margin_used"), eng.get_option("viterbi_margin_close"),
             int(out.to(torch.int64).sum().item())), flush=True)
    eng.close()
