"""Diagnostics for tests/test_viterbi_margin_gpu.py::test_decision_with_a_margin_...: per seed, what the first
pass left at the boundaries and whether the planted decision was seen."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from oracle import oracle as orc
from bhmm_amd.engine import Engine
import test_viterbi_margin_gpu as tv

n, T = 64, 40 * 256 - 1
for seed in range(16):
    rng = np.random.default_rng(8800 + seed)
    A, pi, mu, sig = tv._model(n, rng, "gaussian")
    obs = rng.normal(0, 4, T)
    pobs = orc.pobs_gaussian(obs, mu, sig)
    V, ptr = tv._viterbi_vectors(A, pobs, pi)
    path0 = orc.viterbi(A, pobs, pi)
    t = T - 2
    h = V[t - 1] * A[:, path0[t]]; w = int(path0[t - 1]); hb = h.copy(); hb[w] = -1.0; i2 = int(hb.argmax())
    pobs[t - 1, i2] *= h[w] / h[i2] * (1.0 - 3e-10)
    V, ptr = tv._viterbi_vectors(A, pobs, pi)
    ref = orc.viterbi(A, pobs, pi)
    marg = tv._path_margins(A, V, ref)
    planted = marg[T - 3:].min()
    eng = Engine(0)
    eng.set_option("viterbi_seg_per_simd", 1)
    eng.set_observations("explicit", [pobs], n)
    eng.set_option("viterbi_margin", 2)
    eng.set_option("viterbi_seg_warmups", 4)
    eng.set_option("viterbi_W", 64)
    path = eng.viterbi(A, pi)[0]
    g = eng.get_option
    print(seed, "planted %.3g" % planted, "equal", np.array_equal(path, ref),
          "segs", g("viterbi_segments"), "mism", g("viterbi_mismatch"), "far", g("viterbi_far"), "close", g("viterbi_margin_close"),
          "used", g("viterbi_margin_used"), "rounds", g("viterbi_rounds"), flush=True)
    eng.close()
