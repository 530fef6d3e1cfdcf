"""EM sequence at the configs[1] shape with and without the carried boundary vectors (option "carry"):
per-iteration sweep time by HIP events, the warm-up length used, and the two likelihood sequences."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bhmm_amd
from bench import make_c2_model
from bhmm_amd.engine import synth_observations
from bhmm_amd.estimators import _tmatrix
K, T, n = 256, 100000, 8
m = make_c2_model()
dev = torch.device("cuda", 0)
buf = torch.empty(K * T, dtype=torch.float64, device=dev)
synth_observations("gaussian", buf.data_ptr(), m["A"], m["pi"], m["mu"], m["sigma"], K, T, seed=2000)
host = buf.cpu().numpy().reshape(K, T)
obs = [host[k] for k in range(K)]
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = {}
for carry in (1, 0, 1):
    init = bhmm_amd.gaussian_hmm(m["pi"], m["A_eval"], m["mu_eval"], m["sigma"])
    est = bhmm_amd.MaximumLikelihoodEstimator(obs, n, initial_model=init, reversible=False)
    eng = est._engine
    eng.set_option("carry", carry)
    lls, ms, ws, wall = [], [], [], []
    for it in range(NIT):
        t0 = time.perf_counter()
        lls.append(est.em_step())
        wall.append(time.perf_counter() - t0)
        ms.append(eng.kernel_ms(2))
        ws.append(int(eng.get_option("carry_W")))
    print("carry", carry, "spec_W", eng.get_option("spec_W"), "ok/fail", eng.get_option("carry_ok"), eng.get_option("carry_fail"),
          "spec ok/fail", eng.get_option("spec_ok"), eng.get_option("spec_fail"), "kappa %.3g" % eng.get_option("carry_kappa"))
    print("  sweep ms (events) by tens:", " ".join("%.3f" % np.mean(ms[i:i + 10]) for i in range(0, NIT, 10)))
    print("  wall ms by tens:          ", " ".join("%.3f" % (1e3 * np.mean(wall[i:i + 10])) for i in range(0, NIT, 10)))
    print("  carried warm-up W:", ws[:12], "...", ws[-6:])
    out[carry] = np.array(lls)
    eng.close()
d = np.abs(out[1] - out[0]) / np.abs(out[0])
print("max rel diff of the log-likelihood sequences: %.3g" % d.max())
# diagnostics: a plain E-step loop on a fresh engine, constant model
from bhmm_amd.engine import Engine
e = Engine(0)
e.set_observations("gaussian", obs, n)
margs = (m["A_eval"], m["pi"], m["mu_eval"], m["sigma"])
for _ in range(30):
    e.estep(*margs)
print("plain engine: kernel ms", [round(e.kernel_ms(i), 3) for i in range(5)], "chunks", e.num_chunks, e.chunk_len,
      "careful", e.get_option("careful"), "W", e.get_option("spec_W"))
init = bhmm_amd.gaussian_hmm(m["pi"], m["A_eval"], m["mu_eval"], m["sigma"])
est = bhmm_amd.MaximumLikelihoodEstimator(obs, n, initial_model=init, reversible=False)
for _ in range(5):
    est.em_step()
g = est._engine
print("estimator engine: kernel ms", [round(g.kernel_ms(i), 3) for i in range(5)], "chunks", g.num_chunks, g.chunk_len,
      "careful", g.get_option("careful"), "W", g.get_option("spec_W"), "sigmas", est.hmm.output_model.sigmas)
