"""9..64 states: the Gibbs path step (forward filter + backward draw + path statistics) over time
segments against the serial draw, same seed -- statistics compared, both timed.
   python tools/wide_sample.py [per_simd ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import metastable_matrix, stationary, timeit
from bhmm_amd.engine import Engine, synth_observations

dev = torch.device("cuda", 0)
per_simd = [int(a) for a in sys.argv[1:] if not a.startswith("c")] or [1, 2, 4]
only = [int(a[1:]) for a in sys.argv[1:] if a.startswith("c")]
for ci, (kind, n, K, T) in enumerate((("gaussian", 64, 128, 10000), ("gaussian", 32, 128, 10000), ("gaussian", 16, 128, 10000),
                      ("discrete", 64, 128, 10000), ("gaussian", 64, 128, 100000), ("gaussian", 20, 7, 50001))):
    if only and ci not in only:
        continue
    rng = np.random.default_rng(n)
    A = metastable_matrix(n, rng); pi = stationary(A)
    M = 32
    if kind == "gaussian":
        p0, p1 = np.linspace(-5, 5, n), np.linspace(0.5, 2.0, n)
        obs = torch.empty(K * T, dtype=torch.float64, device=dev)
    else:
        p0, p1 = rng.dirichlet(np.ones(M) * 0.3, n), None
        obs = torch.empty(K * T, dtype=torch.int32, device=dev)
    synth_observations(kind, obs.data_ptr(), A, pi, p0, p1, K, T, seed=n, device=0)
    margs = (0.9 * A + 0.1 / n, pi, p0 + 0.05 if kind == "gaussian" else p0, p1)
    out = {}
    for ps in [0] + per_simd:
        eng = Engine(0)
        if ps == 0:
            eng.set_option("spec_enabled", 0)
        else:
            eng.set_option("sample_seg_per_simd", ps)
        eng.set_observations_device(kind, obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n,
                                    **({"nsymbols": M} if kind == "discrete" else {}))
        for _ in range(4):
            eng.sample_paths(*margs, seed=1, want_paths=False)
        dt = timeit(lambda: eng.sample_paths(*margs, seed=1, want_paths=False), 3, eng.sync)
        r = eng.sample_paths(*margs, seed=1, want_paths=False)
        out[ps] = r
        print("%s n=%d %d x %d per_simd %d: %.2f ms  segmented %d  segments %d  W %d  mismatch %d rounds %d  fwd segmented %d  same counts %s"
              % (kind, n, K, T, ps, dt * 1e3, eng.get_option("sample_segmented"), eng.get_option("sample_segments"),
                 eng.get_option("sample_W"), eng.get_option("sample_mismatch"), eng.get_option("sample_rounds"), eng.get_option("sample_forward_segmented"),
                 np.array_equal(np.asarray(r[1]), np.asarray(out[0][1]))), flush=True)
        eng.close()
