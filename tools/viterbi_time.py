"""Batched Viterbi at the configs[1] shape (8-state Gaussian, 256 x 1e5) and a discrete variant
(M = 64): time per call with the paths left on the device, whether the chunk-parallel run was
accepted, and a hash of the paths (bit-identity across builds).  Kernel times: run under
`rocprofv3 --kernel-trace --stats`."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_c2_model, timeit, metastable_matrix, stationary
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
print("gaussian 256 x 1e5: %.3f ms  chunked=%d close=%d sha1 %s" % (
    1e3 * dt, eng.get_option("viterbi_chunked"), eng.get_option("viterbi_close"),
    hashlib.sha1(pdev.cpu().numpy().tobytes()).hexdigest()[:16]))
eng.close()
rng = np.random.default_rng(3000)
n, M = 8, 64
A = metastable_matrix(n, rng); pi = stationary(A); B = rng.dirichlet(np.ones(M), size=n)
o = torch.empty(K * T, dtype=torch.int32, device=dev)
synth_observations("discrete", o.data_ptr(), A, pi, B, None, K, T, seed=3000)
e2 = Engine(0)
e2.set_observations_device("discrete", o.data_ptr(), np.arange(K + 1, dtype=np.int64) * T, n, nsymbols=M)
ma = (0.9 * A + 0.1 / n, pi, 0.8 * B + 0.2 / M)
e2.estep(*ma)
dt = timeit(lambda: e2.viterbi_u8(*ma, out=pdev), 3, e2.sync, batches=7)
print("discrete 256 x 1e5: %.3f ms  chunked=%d close=%d sha1 %s" % (
    1e3 * dt, e2.get_option("viterbi_chunked"), e2.get_option("viterbi_close"),
    hashlib.sha1(pdev.cpu().numpy().tobytes()).hexdigest()[:16]))
e2.close()
