"""Debugging aid: E-step / Viterbi / Gibbs path step of every kernel family with BHMM_AMD_POISON=1 (fresh
device allocations filled with NaN / -1): a kernel that reads what nothing wrote shows up as a fallback,
a self-check or a wrong result.   BHMM_AMD_POISON=1 python tools/poison_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary
from bhmm_amd.engine import Engine
dev = torch.device("cuda", 0)
for n, K, T in ((8, 64, 20000), (16, 64, 5000), (33, 64, 5000), (64, 128, 10000), (65, 128, 10000), (66, 128, 4000), (80, 64, 6000),
                (97, 64, 4000), (100, 128, 10000), (127, 64, 4000), (128, 128, 10000), (140, 16, 1000)):
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    mu, sig = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
    obs = torch.randn(K * T, dtype=torch.float64, device=dev) * 3.0
    margs = (0.9 * A + 0.1 / n, pi, mu + 0.05, sig)
    out = []
    for rep in range(3):
        eng = Engine(0)
        eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n)
        r = eng.estep(*margs)
        r = eng.estep(*margs)
        v = eng.viterbi_u8(*margs)
        s = eng.sample_paths(*margs, seed=3, want_paths=False)
        out.append((r.loglik, np.asarray(r.C).sum(), int(v.astype(np.int64).sum()), np.asarray(s[1]).trace()))
        diag = (eng.get_option("tile"), eng.get_option("tile_reason"), eng.get_option("wide_trouble"), eng.get_option("careful"),
                eng.get_option("spec_fail"), eng.get_option("viterbi_chunked"), eng.get_option("sample_segmented"))
        eng.close()
    same = all(o == out[0] for o in out)
    if not same:
        print("   differing:", out)
    print("n=%d: tile %d reason %d self-checks %d careful %d spec_fail %d | viterbi segmented %d draw segmented %d | three fresh contexts agree: %s  logL %.6f"
          % ((n,) + tuple(int(x) for x in diag) + (same, out[0][0])), flush=True)
