"""Viterbi at the configs[1] shape (and a discrete variant): scale-free kernel (k_viterbi_fast) against the
order-faithful one (BHMM_AMD_NO_VITERBI_FAST=1 in a second process), kernel times via rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_c2_model, timeit
from bhmm_amd.engine import Engine, synth_observations
K, T = 256, 100000
m = make_c2_model()
dev = torch.device("cuda", 0)
obs = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", obs.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=2000)
eng = Engine(0)
eng.set_observations_device("gaussian", obs.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, 8)
margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
eng.estep(*margs)
pdev = torch.empty(K * T, dtype=torch.uint8, device=dev)
dt = timeit(lambda: eng.viterbi_u8(*margs, out=pdev), 3, eng.sync, batches=7)
print("gaussian 256 x 1e5: %.3f ms  fast=%d chunked=%d close=%d" % (1e3 * dt, eng.get_option("viterbi_fast"),
      eng.get_option("viterbi_chunked"), eng.get_option("viterbi_close")), "sha", hash(pdev.cpu().numpy().tobytes()) & 0xffffffff)
# different models: how often does the fast kernel decide?
rng = np.random.default_rng(0)
nfast = 0
for trial in range(20):
    A = m["A_eval"] * (1 + 0.1 * rng.random((8, 8)))
    A /= A.sum(axis=1)[:, None]
    eng.viterbi_u8(A, m["pi"], m["mu_eval"] + 0.05 * rng.normal(size=8), m["sigma"], out=pdev)
    nfast += int(eng.get_option("viterbi_fast"))
print("fast kernel decided %d of 20 random models" % nfast)
eng.close()
# discrete, 8 states, M = 64, 256 x 1e5
rng = np.random.default_rng(3000)
from bench import metastable_matrix, stationary
n, M = 8, 64
A = metastable_matrix(n, rng); pi = stationary(A); B = rng.dirichlet(np.ones(M), size=n)
o = torch.empty(K * T, dtype=torch.int32, device=dev)
synth_observations("discrete", o.data_ptr(), A, pi, B, None, K, T, seed=3000)
e2 = Engine(0)
e2.set_observations_device("discrete", o.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, nsymbols=M)
ma = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
e2.estep(*ma)
dt = timeit(lambda: e2.viterbi_u8(*ma, out=pdev), 3, e2.sync, batches=7)
print("discrete 256 x 1e5: %.3f ms  fast=%d close=%d" % (1e3 * dt, e2.get_option("viterbi_fast"), e2.get_option("viterbi_close")),
      "sha", hash(pdev.cpu().numpy().tobytes()) & 0xffffffff)
